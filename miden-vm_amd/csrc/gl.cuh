// Goldilocks field (p = 2^64 - 2^32 + 1) and its quadratic extension F_p[x]/(x^2 - 7) for
// gfx950.  All values are CANONICAL (< p) on entry and exit of every function here.
//
// Replaces (semantically) p3-goldilocks 0.6.2 as used by the reference through
// `Felt` (crates/field/src/native/mod.rs:56-58) and `QuadFelt` (x^2 = 7,
// processor/src/execution/operations/field_ops/mod.rs:227-241).
//
// MI355X note: there is no 64-bit integer multiplier; a 64x64->128 product is four
// v_mad_u64_u32 (quarter rate).  The reduction below uses 2^64 = 2^32 - 1 and 2^96 = -1 (mod p)
// so it costs only adds/subs/compares.  MFMA is not applicable (no dense contraction).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

#define GL_HD __host__ __device__ __forceinline__

typedef uint64_t u64;
typedef uint32_t u32;

static constexpr u64 GL_P = 0xFFFFFFFF00000001ULL;
static constexpr u64 GL_EPS = 0xFFFFFFFFULL;  // 2^64 mod p

GL_HD u64 gl_add(u64 a, u64 b) {
  u64 s = a + b;
  u64 t = s + GL_EPS;  // s - p (mod 2^64)
  return (s < a || s >= GL_P) ? t : s;
}
GL_HD u64 gl_sub(u64 a, u64 b) {
  u64 d = a - b;
  return (a < b) ? d - GL_EPS : d;  // d + p (mod 2^64)
}
GL_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
GL_HD u64 gl_dbl(u64 a) { return gl_add(a, a); }

// (hi:lo) mod p, canonical.
GL_HD u64 gl_reduce128(u64 hi, u64 lo) {
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= GL_EPS;          // borrow: 2^64 = eps
  u64 t1 = (hi_lo << 32) - hi_lo;        // hi_lo * eps
  u64 r = t0 + t1;
  if (r < t1) r += GL_EPS;               // carry
  return r >= GL_P ? r - GL_P : r;
}
GL_HD u64 gl_mul(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return gl_reduce128(__umul64hi(a, b), a * b);
#else
  unsigned __int128 p = (unsigned __int128)a * b;
  return gl_reduce128((u64)(p >> 64), (u64)p);
#endif
}
GL_HD u64 gl_sqr(u64 a) { return gl_mul(a, a); }
GL_HD u64 gl_pow(u64 a, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, a);
    a = gl_sqr(a);
    e >>= 1;
  }
  return r;
}
GL_HD u64 gl_exp_pow2(u64 a, int k) {
  for (int i = 0; i < k; i++) a = gl_sqr(a);
  return a;
}
// a^(p - 2), p - 2 = 0xFFFFFFFE_FFFFFFFF = (2^32 - 2) * 2^32 + (2^32 - 1): an addition chain through a^(2^k - 1), k = 2, 3, 6,
// 12, 15, 30, 31 -- 63 squarings + 9 multiplications instead of the 63 + 62 of square-and-multiply (inverse of 0 is 0).
GL_HD u64 gl_inv(u64 a) {
  const u64 x2 = gl_mul(gl_sqr(a), a);
  const u64 x3 = gl_mul(gl_sqr(x2), a);
  const u64 x6 = gl_mul(gl_exp_pow2(x3, 3), x3);
  const u64 x12 = gl_mul(gl_exp_pow2(x6, 6), x6);
  const u64 x15 = gl_mul(gl_exp_pow2(x12, 3), x3);
  const u64 x30 = gl_mul(gl_exp_pow2(x15, 15), x15);
  const u64 x31 = gl_mul(gl_sqr(x30), a);
  const u64 hi = gl_sqr(x31);       // a^(2^32 - 2)
  const u64 lo = gl_mul(hi, a);     // a^(2^32 - 1)
  return gl_mul(gl_exp_pow2(hi, 32), lo);
}
GL_HD u64 gl_canon(u64 a) { return a >= GL_P ? a - GL_P : a; }

static constexpr u64 GL_GENERATOR = 7;                       // domain.rs:358-361
static constexpr u64 GL_ROOT_2_32 = 1753635133440165772ULL;  // asm/stark/constants.masm:5
GL_HD u64 gl_two_adic_generator(int k) { return gl_exp_pow2(GL_ROOT_2_32, 32 - k); }
GL_HD u64 gl_lde_shift(int log_lde) { return gl_exp_pow2(GL_GENERATOR, 32 - log_lde); }

GL_HD u32 bitrev32(u32 x, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
  u32 r = 0;
  for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
#endif
}

// ---- quadratic extension ---------------------------------------------------------------------
struct e2 {
  u64 c0, c1;
};
GL_HD e2 e2_make(u64 a, u64 b = 0) { return e2{a, b}; }
GL_HD e2 e2_add(e2 a, e2 b) { return {gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)}; }
GL_HD e2 e2_sub(e2 a, e2 b) { return {gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)}; }
GL_HD e2 e2_neg(e2 a) { return {gl_neg(a.c0), gl_neg(a.c1)}; }
GL_HD u64 gl_mul7(u64 a) {  // 7a = 8a - a
  u64 a2 = gl_dbl(a), a4 = gl_dbl(a2), a8 = gl_dbl(a4);
  return gl_sub(a8, a);
}
GL_HD e2 e2_mul(e2 a, e2 b) {
  u64 a0b0 = gl_mul(a.c0, b.c0), a1b1 = gl_mul(a.c1, b.c1);
  u64 cross = gl_mul(gl_add(a.c0, a.c1), gl_add(b.c0, b.c1));
  return {gl_add(a0b0, gl_mul7(a1b1)), gl_sub(gl_sub(cross, a0b0), a1b1)};
}
GL_HD e2 e2_mulf(e2 a, u64 b) { return {gl_mul(a.c0, b), gl_mul(a.c1, b)}; }
GL_HD e2 e2_sqr(e2 a) { return e2_mul(a, a); }
GL_HD e2 e2_inv(e2 a) {
  u64 n = gl_sub(gl_sqr(a.c0), gl_mul7(gl_sqr(a.c1)));
  u64 ni = gl_inv(n);
  return {gl_mul(a.c0, ni), gl_mul(gl_neg(a.c1), ni)};
}
GL_HD e2 e2_pow(e2 a, u64 e) {
  e2 r = e2_make(1);
  while (e) {
    if (e & 1) r = e2_mul(r, a);
    a = e2_sqr(a);
    e >>= 1;
  }
  return r;
}
GL_HD e2 e2_exp_pow2(e2 a, int k) {
  for (int i = 0; i < k; i++) a = e2_sqr(a);
  return a;
}
GL_HD bool e2_eq(e2 a, e2 b) { return a.c0 == b.c0 && a.c1 == b.c1; }
GL_HD bool e2_is_zero(e2 a) { return (a.c0 | a.c1) == 0; }
