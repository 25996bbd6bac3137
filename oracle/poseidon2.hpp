// ORACLE — TEST INFRASTRUCTURE ONLY (see gl.hpp header).
//
// Poseidon2 permutation over Goldilocks (width 12) and the four sponge conventions the hot path
// uses.  The permutation body lives in the external crate p3-goldilocks 0.6.2
// (`Poseidon2Goldilocks<12>`, called at crates/crypto/src/hash/algebraic_sponge/poseidon2/mod.rs:22-37);
// this restates the reference's own in-tree, un-optimised formulation of the same function:
//   * linear layers / s-box ........ crates/crypto/src/hash/algebraic_sponge/poseidon2/mod.rs:234-319
//   * round schedule ............... core/src/chiplets/hasher.rs:89-115
//   * constants .................... .../poseidon2/constants.rs:18-211 (-> p2_constants.inc)
// Pinned by the KAT at .../poseidon2/test.rs:7-39 (tests/test_oracle_kat.py).
#pragma once
#include "gl.hpp"
#include "rescue.hpp"

namespace oracle {
#include "p2_constants.inc"

// mod.rs:262-281 (matmul_m4): multiply each 4-chunk by [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]
static inline void p2_matmul_m4(uint64_t s[12]) {
  for (int i = 0; i < 3; i++) {
    uint64_t* x = s + 4 * i;
    uint64_t t01 = fadd(x[0], x[1]);
    uint64_t t23 = fadd(x[2], x[3]);
    uint64_t t0123 = fadd(t01, t23);
    uint64_t t01123 = fadd(t0123, x[1]);
    uint64_t t01233 = fadd(t0123, x[3]);
    uint64_t x0 = x[0], x2 = x[2];
    x[3] = fadd(t01233, fadd(x0, x0));
    x[1] = fadd(t01123, fadd(x2, x2));
    x[0] = fadd(t01123, t01);
    x[2] = fadd(t01233, t23);
  }
}
// mod.rs:233-251 (apply_matmul_external)
static inline void p2_matmul_external(uint64_t s[12]) {
  p2_matmul_m4(s);
  uint64_t stored[4] = {0, 0, 0, 0};
  for (int j = 0; j < 3; j++)
    for (int l = 0; l < 4; l++) stored[l] = fadd(stored[l], s[4 * j + l]);
  for (int i = 0; i < 12; i++) s[i] = fadd(s[i], stored[i % 4]);
}
// mod.rs:288-298 (matmul_internal): s[i] = s[i]*diag[i] + sum(s)
static inline void p2_matmul_internal(uint64_t s[12]) {
  uint64_t sum = 0;
  for (int i = 0; i < 12; i++) sum = fadd(sum, s[i]);
  for (int i = 0; i < 12; i++) s[i] = fadd(fmul(s[i], P2_MAT_DIAG[i]), sum);
}
static inline uint64_t p2_sbox(uint64_t x) {
  uint64_t x2 = fmul(x, x), x3 = fmul(x2, x), x4 = fmul(x2, x2);
  return fmul(x3, x4);
}
// hasher.rs:89-115: M_E; 4x(+rc, x^7, M_E); 22x(s0+=rc, s0^7, M_I); 4x(+rc, x^7, M_E)
static inline void p2_permute(uint64_t s[12]) {
  p2_matmul_external(s);
  for (int r = 0; r < 4; r++) {
    for (int i = 0; i < 12; i++) s[i] = p2_sbox(fadd(s[i], P2_ARK_EXT_INITIAL[12 * r + i]));
    p2_matmul_external(s);
  }
  for (int r = 0; r < 22; r++) {
    s[0] = p2_sbox(fadd(s[0], P2_ARK_INT[r]));
    p2_matmul_internal(s);
  }
  for (int r = 0; r < 4; r++) {
    for (int i = 0; i < 12; i++) s[i] = p2_sbox(fadd(s[i], P2_ARK_EXT_TERMINAL[12 * r + i]));
    p2_matmul_external(s);
  }
}

// The algebraic configurations (air/src/config.rs:212-273) differ only in this permutation: 0 = Poseidon2, 3 = RPO, 4 = RPX
// (the values of LMCS_* in lmcs.hpp; orc_set_lmcs).  Sponge, compression, hash_elements and the duplex challenger call it.
inline int g_alg_perm = 0;
static inline void alg_permute(uint64_t s[12]) {
  if (g_alg_perm == 3) rpo_permute(s);
  else if (g_alg_perm == 4) rpx_permute(s);
  else p2_permute(s);
}

// (i) LMCS leaf sponge: crates/stateful-hasher/src/field_sponge.rs:41-59 (StatefulSponge::absorb_into,
//     WIDTH 12, RATE 8): overwrite rate chunk-wise; permute per full chunk; trailing partial chunk
//     zero-filled then permuted; empty input = no-op.
static inline void sponge_absorb(uint64_t state[12], const uint64_t* in, size_t n) {
  size_t pos = 0;
  while (true) {
    for (int i = 0; i < 8; i++) {
      if (pos < n) {
        state[i] = in[pos++];
      } else {
        if (i != 0) {
          for (int k = i; k < 8; k++) state[k] = 0;
          alg_permute(state);
        }
        return;
      }
    }
    alg_permute(state);
  }
}
// (ii) Merkle 2-to-1 compression: TruncatedPermutation<_,2,4,12> (air/src/config.rs:213-220)
//      = perm([L | R | 0000])[0..4] = Poseidon2::merge (algebraic_sponge/mod.rs:153-165)
static inline void compress(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
  uint64_t s[12];
  for (int i = 0; i < 4; i++) {
    s[i] = l[i];
    s[4 + i] = r[i];
    s[8 + i] = 0;
  }
  alg_permute(s);
  for (int i = 0; i < 4; i++) out[i] = s[i];
}
// (iii) Poseidon2::hash_elements (algebraic_sponge/mod.rs:215-265): state[8] = len mod 8 first.
static inline void hash_elements(const uint64_t* in, size_t n, uint64_t out[4]) {
  uint64_t s[12] = {0};
  s[8] = n % 8;
  size_t i = 0;
  for (size_t k = 0; k < n; k++) {
    s[i++] = in[k];
    if (i == 8) {
      alg_permute(s);
      i = 0;
    }
  }
  if (i > 0) {
    while (i != 8) s[i++] = 0;
    alg_permute(s);
  }
  for (int k = 0; k < 4; k++) out[k] = s[k];
}

}  // namespace oracle
