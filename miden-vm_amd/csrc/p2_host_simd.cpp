// Eight Poseidon2 2-to-1 compressions at once on the HOST (AVX-512: one node per 64-bit lane).
//
// Used for the top levels of every Merkle tree (lmcs.hip: lmcs_compress_layers finishes them on the host because a level of a few
// dozen nodes costs the GPU a lone-wave permutation each).  The scalar host permutation (poseidon2.cuh p2_permute) needs ~2 us per
// node on the bench box, one call here ~3.4 us for eight nodes; a level of 16 nodes is two of these calls.  Same function as p2_permute -- reference
// crates/crypto/src/hash/algebraic_sponge/poseidon2/mod.rs:22-37, 233-319 (M_E via M4, M_I = diag + 1 1^T, x^7), schedule
// core/src/chiplets/hasher.rs:89-115 -- values canonical between operations; checked against the CPU checker in
// tests/test_host_compress_simd.py (host, no GPU) and through every tree root of the GPU parity tests.
// Compiled for the host only (no -x hip); selected at run time when the CPU has AVX-512 F + DQ.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>

namespace p2c_host {
#include "p2_constants.inc"
}

typedef uint64_t u64;
#define P2V_TARGET __attribute__((target("avx512f,avx512dq")))
typedef __m512i v8;

namespace {
const u64 GLP = 0xFFFFFFFF00000001ULL, GLEPS = 0xFFFFFFFFULL;

P2V_TARGET inline v8 vset(u64 x) { return _mm512_set1_epi64((long long)x); }
// canonical a, b -> canonical a + b
P2V_TARGET inline v8 vadd(v8 a, v8 b) {
  const v8 s = _mm512_add_epi64(a, b);
  const __mmask8 m = _mm512_cmplt_epu64_mask(s, a) | _mm512_cmpge_epu64_mask(s, vset(GLP));
  return _mm512_mask_add_epi64(s, m, s, vset(GLEPS));  // - p = + eps (mod 2^64)
}
P2V_TARGET inline v8 vsub(v8 a, v8 b) {
  const v8 d = _mm512_sub_epi64(a, b);
  const __mmask8 m = _mm512_cmplt_epu64_mask(a, b);
  return _mm512_mask_sub_epi64(d, m, d, vset(GLEPS));  // + p = - eps (mod 2^64)
}
P2V_TARGET inline v8 vdbl(v8 a) { return vadd(a, a); }
P2V_TARGET inline v8 vhalve(v8 a) {  // a / 2: (a + p) / 2 for odd a
  const v8 t = _mm512_srli_epi64(a, 1);
  const __mmask8 odd = _mm512_test_epi64_mask(a, vset(1));
  return _mm512_mask_add_epi64(t, odd, t, vset(0x7FFFFFFF80000001ULL));
}
// (hi, lo) mod p, canonical (gl.cuh gl_reduce128)
P2V_TARGET inline v8 vreduce128(v8 hi, v8 lo) {
  const v8 eps = vset(GLEPS);
  const v8 hi_hi = _mm512_srli_epi64(hi, 32), hi_lo = _mm512_and_si512(hi, eps);
  v8 t0 = _mm512_sub_epi64(lo, hi_hi);
  t0 = _mm512_mask_sub_epi64(t0, _mm512_cmplt_epu64_mask(lo, hi_hi), t0, eps);
  const v8 t1 = _mm512_sub_epi64(_mm512_slli_epi64(hi_lo, 32), hi_lo);  // hi_lo * eps
  v8 r = _mm512_add_epi64(t0, t1);
  r = _mm512_mask_add_epi64(r, _mm512_cmplt_epu64_mask(r, t1), r, eps);
  return _mm512_mask_sub_epi64(r, _mm512_cmpge_epu64_mask(r, vset(GLP)), r, vset(GLP));
}
P2V_TARGET inline v8 vmul(v8 a, v8 b) {
  const v8 eps = vset(GLEPS);
  const v8 ah = _mm512_srli_epi64(a, 32), bh = _mm512_srli_epi64(b, 32);
  const v8 ll = _mm512_mul_epu32(a, b), hl = _mm512_mul_epu32(ah, b), lh = _mm512_mul_epu32(a, bh), hh = _mm512_mul_epu32(ah, bh);
  const v8 mid = _mm512_add_epi64(hl, _mm512_srli_epi64(ll, 32));                     // < 2^64: (2^32-1)^2 + 2^32 - 1
  const v8 mid2 = _mm512_add_epi64(lh, _mm512_and_si512(mid, eps));
  const v8 lo = _mm512_or_si512(_mm512_slli_epi64(mid2, 32), _mm512_and_si512(ll, eps));
  const v8 hi = _mm512_add_epi64(_mm512_add_epi64(hh, _mm512_srli_epi64(mid, 32)), _mm512_srli_epi64(mid2, 32));
  return vreduce128(hi, lo);
}
P2V_TARGET inline v8 vsbox(v8 x) {
  const v8 x2 = vmul(x, x), x3 = vmul(x2, x), x4 = vmul(x2, x2);
  return vmul(x3, x4);
}
// state <- circ(2 M4, M4, M4) * state,  M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]  (mod.rs:233-281)
P2V_TARGET inline void vexternal(v8 s[12]) {
  for (int i = 0; i < 12; i += 4) {
    const v8 t01 = vadd(s[i], s[i + 1]), t23 = vadd(s[i + 2], s[i + 3]);
    const v8 t0123 = vadd(t01, t23);
    const v8 t01123 = vadd(t0123, s[i + 1]), t01233 = vadd(t0123, s[i + 3]);
    const v8 x0 = s[i], x2 = s[i + 2];
    s[i + 3] = vadd(t01233, vdbl(x0));
    s[i + 1] = vadd(t01123, vdbl(x2));
    s[i] = vadd(t01123, t01);
    s[i + 2] = vadd(t01233, t23);
  }
  v8 st[4];
  for (int l = 0; l < 4; l++) st[l] = vadd(vadd(s[l], s[4 + l]), s[8 + l]);
  for (int i = 0; i < 12; i++) s[i] = vadd(s[i], st[i & 3]);
}
// diag [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 1/4, -1/4, 1/8] + the all-ones matrix  (mod.rs:288-298)
P2V_TARGET inline void vinternal(v8 s[12]) {
  v8 sum = s[0];
  for (int i = 1; i < 12; i++) sum = vadd(sum, s[i]);
  v8 d;
  s[0] = vsub(sum, vdbl(s[0]));
  s[1] = vadd(sum, s[1]);
  s[2] = vadd(sum, vdbl(s[2]));
  s[3] = vadd(sum, vhalve(s[3]));
  d = vdbl(s[4]);
  s[4] = vadd(sum, vadd(d, s[4]));
  s[5] = vadd(sum, vdbl(vdbl(s[5])));
  s[6] = vsub(sum, vhalve(s[6]));
  d = vdbl(s[7]);
  s[7] = vsub(sum, vadd(d, s[7]));
  s[8] = vsub(sum, vdbl(vdbl(s[8])));
  s[9] = vadd(sum, vhalve(vhalve(s[9])));
  s[10] = vsub(sum, vhalve(vhalve(s[10])));
  s[11] = vadd(sum, vhalve(vhalve(vhalve(s[11]))));
}
P2V_TARGET void permute8(v8 s[12]) {
  vexternal(s);
  for (int r = 0; r < 4; r++) {
    for (int i = 0; i < 12; i++) s[i] = vsbox(vadd(s[i], vset(p2c_host::P2_ARK_EXT_INITIAL[12 * r + i])));
    vexternal(s);
  }
  for (int r = 0; r < 22; r++) {
    s[0] = vsbox(vadd(s[0], vset(p2c_host::P2_ARK_INT[r])));
    vinternal(s);
  }
  for (int r = 0; r < 4; r++) {
    for (int i = 0; i < 12; i++) s[i] = vsbox(vadd(s[i], vset(p2c_host::P2_ARK_EXT_TERMINAL[12 * r + i])));
    vexternal(s);
  }
}
}  // namespace

bool p2_host_simd_available() {
  // MH_HOST_SIMD=0: the scalar host functions (what a CPU without AVX-512 runs), for tests and comparisons
  static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && !(getenv("MH_HOST_SIMD") && atoi(getenv("MH_HOST_SIMD")) == 0);
  return ok;
}

// Eight full permutations: states[j][0..12) canonical in, canonical out.  Call only when p2_host_simd_available().
P2V_TARGET void p2_host_permute8(u64* states) {
  alignas(64) u64 lanes[12][8];
  for (int j = 0; j < 8; j++)
    for (int k = 0; k < 12; k++) lanes[k][j] = states[12 * j + k];
  v8 s[12];
  for (int k = 0; k < 12; k++) s[k] = _mm512_load_si512(lanes[k]);
  permute8(s);
  for (int k = 0; k < 12; k++) _mm512_store_si512(lanes[k], s[k]);
  for (int j = 0; j < 8; j++)
    for (int k = 0; k < 12; k++) states[12 * j + k] = lanes[k][j];
}

// out[j] = compress(pairs[j]) for j < n <= 8: pairs = n x (left || right) canonical words, out = n x 4 words.
// Call only when p2_host_simd_available().
P2V_TARGET void p2_host_compress8(const u64* pairs, int n, u64* out) {
  alignas(64) u64 lanes[12][8];
  std::memset(lanes, 0, sizeof lanes);
  for (int j = 0; j < n; j++)
    for (int k = 0; k < 8; k++) lanes[k][j] = pairs[8 * j + k];
  v8 s[12];
  for (int k = 0; k < 12; k++) s[k] = _mm512_load_si512(lanes[k]);
  permute8(s);
  for (int k = 0; k < 4; k++) _mm512_store_si512(lanes[k], s[k]);
  for (int j = 0; j < n; j++)
    for (int k = 0; k < 4; k++) out[4 * j + k] = lanes[k][j];
}
