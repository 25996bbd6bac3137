// ORACLE — TEST INFRASTRUCTURE ONLY (see gl.hpp header).
//
// The Rescue Prime permutations of the reference's other two algebraic configurations (air/src/config.rs:224-245:
// rpo_config / rpx_config = the SAME AlgLmcs / AlgChallenger as Poseidon2 with another permutation), restated from
//   crates/crypto/src/hash/algebraic_sponge/rescue/rpo/mod.rs:183-207   RPO: 7 x (MDS, +ARK1, x^7, MDS, +ARK2, x^(1/7))
//   .../rescue/rpx/mod.rs:183-268                                        RPX: FB, E, FB, E, FB, E, M  (FB = an RPO round,
//        E = +ARK1 then x^7 in the cubic extension F_p[phi]/(phi^3 - phi - 1) on the four triples, M = MDS, +ARK1)
//   .../rescue/rpx/mod.rs:279-330 (cubic_ext::mul / square / power7), .../rescue/mod.rs:27-100 (ALPHA = 7, INV_ALPHA),
//   .../rescue/mds/mod.rs:43-.. (circulant MDS), constants in rescue_constants.inc (tools/gen_rescue_constants.py).
// Pinned: RPO by the reference's 19 hash_elements vectors (rpo/tests.rs:241-267 -> tests/golden/kat.json); RPX shares every
// piece with it except the E round, which has no literal vector in tree (its tests compare implementations with each other).
#pragma once
#include "gl.hpp"

namespace oracle {

#include "rescue_constants.inc"

static const uint64_t RESCUE_INV_ALPHA = 10540996611094048183ULL;  // 7 * INV_ALPHA = 1 mod (p - 1)

static inline void rescue_mds(uint64_t s[12]) {
  uint64_t r[12];
  for (int i = 0; i < 12; i++) {
    uint64_t acc = 0;
    for (int j = 0; j < 12; j++) acc = fadd(acc, fmul(RESCUE_MDS_ROW0[(j - i + 12) % 12], s[j]));  // row i = row 0 rotated right by i
    r[i] = acc;
  }
  for (int i = 0; i < 12; i++) s[i] = r[i];
}
static inline void rescue_add(uint64_t s[12], const unsigned long long* ark) {
  for (int i = 0; i < 12; i++) s[i] = fadd(s[i], ark[i]);
}
static inline uint64_t pow7(uint64_t x) {
  uint64_t x2 = fmul(x, x), x4 = fmul(x2, x2);
  return fmul(fmul(x4, x2), x);
}
static inline void rescue_fb_round(uint64_t s[12], int r) {
  rescue_mds(s);
  rescue_add(s, RESCUE_ARK1 + 12 * r);
  for (int i = 0; i < 12; i++) s[i] = pow7(s[i]);
  rescue_mds(s);
  rescue_add(s, RESCUE_ARK2 + 12 * r);
  for (int i = 0; i < 12; i++) s[i] = fpow(s[i], RESCUE_INV_ALPHA);
}
static inline void rpo_permute(uint64_t s[12]) {
  for (int r = 0; r < 7; r++) rescue_fb_round(s, r);
}

// a0 + a1 phi + a2 phi^2,  phi^3 = phi + 1
struct C3 {
  uint64_t c[3];
};
static inline C3 c3_mul(const C3& a, const C3& b) {
  // schoolbook: d0..d4, then phi^3 = phi + 1, phi^4 = phi^2 + phi
  uint64_t d0 = fmul(a.c[0], b.c[0]);
  uint64_t d1 = fadd(fmul(a.c[0], b.c[1]), fmul(a.c[1], b.c[0]));
  uint64_t d2 = fadd(fadd(fmul(a.c[0], b.c[2]), fmul(a.c[1], b.c[1])), fmul(a.c[2], b.c[0]));
  uint64_t d3 = fadd(fmul(a.c[1], b.c[2]), fmul(a.c[2], b.c[1]));
  uint64_t d4 = fmul(a.c[2], b.c[2]);
  C3 r;
  r.c[0] = fadd(d0, d3);
  r.c[1] = fadd(fadd(d1, d3), d4);
  r.c[2] = fadd(d2, d4);
  return r;
}
static inline C3 c3_pow7(const C3& x) {
  C3 x2 = c3_mul(x, x), x4 = c3_mul(x2, x2);
  return c3_mul(c3_mul(x4, x2), x);
}
static inline void rpx_permute(uint64_t s[12]) {
  for (int r = 0; r < 6; r += 2) {
    rescue_fb_round(s, r);
    rescue_add(s, RESCUE_ARK1 + 12 * (r + 1));  // (E) round r + 1
    for (int k = 0; k < 4; k++) {
      C3 v = c3_pow7(C3{{s[3 * k], s[3 * k + 1], s[3 * k + 2]}});
      s[3 * k] = v.c[0];
      s[3 * k + 1] = v.c[1];
      s[3 * k + 2] = v.c[2];
    }
  }
  rescue_mds(s);  // (M) round 6
  rescue_add(s, RESCUE_ARK1 + 12 * 6);
}

}  // namespace oracle
