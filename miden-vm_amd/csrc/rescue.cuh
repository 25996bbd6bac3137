// The Rescue Prime permutations of the reference's RPO and RPX configurations (air/src/config.rs:224-245: the same algebraic
// LMCS and duplex challenger as Poseidon2, another permutation), host + device:
//   crates/crypto/src/hash/algebraic_sponge/rescue/rpo/mod.rs:183-207   RPO: 7 x (MDS, +ARK1, x^7, MDS, +ARK2, x^(1/7))
//   .../rescue/rpx/mod.rs:183-268, 279-330                               RPX: FB, E, FB, E, FB, E, M; E = +ARK1, then x^7 in
//        F_p[phi]/(phi^3 - phi - 1) on the four triples of the state; M = MDS, +ARK1
// Plain canonical arithmetic (gl.cuh): these configurations are supported for completeness of ProvingOptions, not tuned --
// the inverse S-box alone is 72 multiplications per element and round (the addition chain of rescue/mod.rs:40-85).
#pragma once
#include "poseidon2.cuh"

namespace rescue {
#include "rescue_constants.inc"

GL_HD u64 c_mds(int i) {
  constexpr u64 R[12] = {7, 23, 8, 26, 13, 10, 9, 7, 6, 22, 21, 8};  // = RESCUE_MDS_ROW0 (checked by tests against the .inc)
  return R[i];
}
// row i of the circulant matrix = row 0 rotated right by i; entries < 2^5: accumulate the 32-bit halves in 64 bits
GL_HD void mds(u64 s[12]) {
  u64 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    lo[i] = 0;
    hi[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < 12; i++)
#pragma unroll
    for (int j = 0; j < 12; j++) {
      const u64 m = c_mds((j - i + 12) % 12);
      lo[i] += m * (s[j] & 0xFFFFFFFFULL);
      hi[i] += m * (s[j] >> 32);
    }
#pragma unroll
  for (int i = 0; i < 12; i++) {
    // lo + 2^32 * hi, both < 2^41:  2^32 * hi = (hi_lo32 << 32) + hi_hi * 2^64,  2^64 = 2^32 - 1 (mod p)
    const u64 h_lo = hi[i] & 0xFFFFFFFFULL, h_hi = hi[i] >> 32;
    u64 r = gl_add(gl_canon(lo[i]), gl_canon(h_lo << 32));
    r = gl_add(r, gl_canon(h_hi * 0xFFFFFFFFULL));
    s[i] = r;
  }
}
GL_HD u64 pow7(u64 x) {
  const u64 x2 = gl_sqr(x), x4 = gl_sqr(x2);
  return gl_mul(gl_mul(x4, x2), x);
}
GL_HD u64 sqr_n(u64 x, int n) {
  for (int i = 0; i < n; i++) x = gl_sqr(x);
  return x;
}
// x^(1/7) = x^10540996611094048183, the addition chain of rescue/mod.rs:40-85
GL_HD u64 inv_pow7(u64 x) {
  const u64 t1 = gl_sqr(x);                         // 10
  const u64 t2 = gl_sqr(t1);                        // 100
  const u64 t3 = gl_mul(sqr_n(t2, 3), t2);          // 100100
  const u64 t4 = gl_mul(sqr_n(t3, 6), t3);          // 100100100100
  const u64 t5 = gl_mul(sqr_n(t4, 12), t4);         // 24 bits
  const u64 t6 = gl_mul(sqr_n(t5, 6), t3);          // 30 bits
  const u64 t7 = gl_mul(sqr_n(t6, 31), t6);         // 61 bits
  const u64 a = sqr_n(gl_mul(gl_sqr(t7), t6), 2);
  const u64 b = gl_mul(gl_mul(t1, t2), x);
  return gl_mul(a, b);
}
GL_HD void fb_round(u64 s[12], int r) {
  mds(s);
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = pow7(gl_add(s[i], RESCUE_ARK1[12 * r + i]));
  mds(s);
#pragma unroll 1
  for (int i = 0; i < 12; i++) s[i] = inv_pow7(gl_add(s[i], RESCUE_ARK2[12 * r + i]));
}
GL_HD void rpo_permute(u64 s[12]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int r = 0; r < 7; r++) fb_round(s, r);
}
GL_HD void c3_mul(const u64 a[3], const u64 b[3], u64 o[3]) {
  const u64 d0 = gl_mul(a[0], b[0]);
  const u64 d1 = gl_add(gl_mul(a[0], b[1]), gl_mul(a[1], b[0]));
  const u64 d2 = gl_add(gl_add(gl_mul(a[0], b[2]), gl_mul(a[1], b[1])), gl_mul(a[2], b[0]));
  const u64 d3 = gl_add(gl_mul(a[1], b[2]), gl_mul(a[2], b[1]));
  const u64 d4 = gl_mul(a[2], b[2]);
  o[0] = gl_add(d0, d3);                 // phi^3 = phi + 1, phi^4 = phi^2 + phi
  o[1] = gl_add(gl_add(d1, d3), d4);
  o[2] = gl_add(d2, d4);
}
GL_HD void ext_round(u64 s[12], int r) {
#pragma unroll 1
  for (int k = 0; k < 4; k++) {
    u64 x[3], x2[3], x4[3], x6[3], x7[3];
    for (int i = 0; i < 3; i++) x[i] = gl_add(s[3 * k + i], RESCUE_ARK1[12 * r + 3 * k + i]);
    c3_mul(x, x, x2);
    c3_mul(x2, x2, x4);
    c3_mul(x4, x2, x6);
    c3_mul(x6, x, x7);
    for (int i = 0; i < 3; i++) s[3 * k + i] = x7[i];
  }
}
GL_HD void rpx_permute(u64 s[12]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int r = 0; r < 6; r += 2) {
    fb_round(s, r);
    ext_round(s, r + 1);
  }
  mds(s);
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], RESCUE_ARK1[72 + i]);
}

}  // namespace rescue

#if defined(__HIP_DEVICE_COMPILE__) && defined(RESCUE_FAST)
// Device fast path (the hash kernels of lmcs.hip / fri.hip define RESCUE_FAST after including poseidon2_fast.cuh): the S-boxes of
// four state elements at a time through the 13-instruction product with SGPR carry chains (p2f_mulN<4>: stage-interleaved, no
// wait-state padding).  Values between the layers are any 64-bit representative; the MDS layer works on 32-bit halves of
// whatever it is given, constants are added with p2f_add_canon, the state is canonicalised once at the end.
namespace rescue {
__device__ __forceinline__ void mul4(u64 (&r)[4], const u64 (&a)[4], const u64 (&b)[4]) { p2f_mulN<4>(r, a, b); }
__device__ __forceinline__ void sqr4_n(u64 (&x)[4], int n) {
#pragma unroll 1
  for (int i = 0; i < n; i++) mul4(x, x, x);
}
__device__ __forceinline__ void pow7_x4(u64 (&x)[4]) {
  u64 x2[4], x4[4], x6[4];
  mul4(x2, x, x);
  mul4(x4, x2, x2);
  mul4(x6, x4, x2);
  mul4(x, x6, x);
}
__device__ __forceinline__ void inv_pow7_x4(u64 (&x)[4]) {  // the chain of inv_pow7 above on four elements
  u64 t1[4], t2[4], t3[4], t4[4], t5[4], t6[4], t7[4], a[4], b[4];
  mul4(t1, x, x);
  mul4(t2, t1, t1);
#pragma unroll
  for (int i = 0; i < 4; i++) t3[i] = t2[i];
  sqr4_n(t3, 3); mul4(t3, t3, t2);
#pragma unroll
  for (int i = 0; i < 4; i++) t4[i] = t3[i];
  sqr4_n(t4, 6); mul4(t4, t4, t3);
#pragma unroll
  for (int i = 0; i < 4; i++) t5[i] = t4[i];
  sqr4_n(t5, 12); mul4(t5, t5, t4);
#pragma unroll
  for (int i = 0; i < 4; i++) t6[i] = t5[i];
  sqr4_n(t6, 6); mul4(t6, t6, t3);
#pragma unroll
  for (int i = 0; i < 4; i++) t7[i] = t6[i];
  sqr4_n(t7, 31); mul4(t7, t7, t6);
  mul4(a, t7, t7); mul4(a, a, t6); sqr4_n(a, 2);
  mul4(b, t1, t2); mul4(b, b, x);
  mul4(x, a, b);
}
__device__ __forceinline__ void fb_round_fast(u64 s[12], int r) {
  mds(s);  // canonical out
#pragma unroll
  for (int g = 0; g < 12; g += 4) {
    u64 x[4];
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = gl_add(s[g + i], RESCUE_ARK1[12 * r + g + i]);
    pow7_x4(x);
#pragma unroll
    for (int i = 0; i < 4; i++) s[g + i] = x[i];
  }
  mds(s);
#pragma unroll 1
  for (int g = 0; g < 12; g += 4) {
    u64 x[4];
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = gl_add(s[g + i], RESCUE_ARK2[12 * r + g + i]);
    inv_pow7_x4(x);
#pragma unroll
    for (int i = 0; i < 4; i++) s[g + i] = x[i];
  }
}
__device__ __forceinline__ void canon12(u64 s[12]) {
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
}
__device__ __forceinline__ void rpo_permute_fast(u64 s[12]) {
#pragma unroll 1
  for (int r = 0; r < 7; r++) fb_round_fast(s, r);
  canon12(s);
}
__device__ __forceinline__ void rpx_permute_fast(u64 s[12]) {
#pragma unroll 1
  for (int r = 0; r < 6; r += 2) {
    fb_round_fast(s, r);
    canon12(s);  // the E round runs on canonical values (plain extension arithmetic)
    ext_round(s, r + 1);
  }
  mds(s);
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], RESCUE_ARK1[72 + i]);
}
}  // namespace rescue
#endif

// The permutation of an algebraic configuration by its MH_LMCS_* id (0 Poseidon2, 3 RPO, 4 RPX)
GL_HD void alg_permute(int lmcs, u64 s[12]) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(RESCUE_FAST)
  if (lmcs == 3) rescue::rpo_permute_fast(s);
  else if (lmcs == 4) rescue::rpx_permute_fast(s);
  else p2_permute(s);
#else
  if (lmcs == 3) rescue::rpo_permute(s);
  else if (lmcs == 4) rescue::rpx_permute(s);
  else p2_permute(s);
#endif
}
