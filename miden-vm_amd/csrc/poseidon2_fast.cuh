// Poseidon2 (Goldilocks, t = 12, x^7, 4+22+4 rounds) restructured for the gfx950 VALU.
//
// Same function as p2_permute (poseidon2.cuh; reference:
// crates/crypto/src/hash/algebraic_sponge/poseidon2/mod.rs:22-37, schedule core/src/chiplets/
// hasher.rs:89-115) -- parity is checked bit-for-bit by the GPU tests.  What differs is the
// arithmetic schedule.  Measured on MI355X (tools/instbench): every VALU instruction, including
// v_mad_u64_u32 (32x32+64) and the 64-bit v_lshl_add_u64, costs about the same issue slot, and a
// canonical modular add costs ~6 of them.  So:
//   * S-box multiplications keep any representative < 2^64 (no canonicalisation between rounds);
//   * linear layers run on "wide" values V = L + H*2^32 held as two 64-bit integers, built with
//     v_mad_u64_u32 straight from 32-bit halves: additions are single 64-bit adds with no carry or
//     modular fix-up, small-constant products are free (mad / shift-add);
//   * one fold back to 64 bits per element per S-box (2^64 = 2^32 - 1 mod p);
//   * the 22 internal rounds keep elements 1..11 wide across rounds; the fractional diagonal
//     [-2,1,2,1/2,3,4,-1/2,-3,-4,1/4,-1/4,1/8] is cleared by carrying the whole state scaled by 8^r
//     (integer diagonal [-16,8,16,4,24,32,-4,-24,-32,2,-2,1]); the S-box of round r then costs one
//     extra multiplication by the constant 8^(-6r), and wide values are refolded every 4 rounds.
#pragma once
#include "poseidon2.cuh"

namespace p2c {
#include "p2_fast_constants.inc"
}

#if defined(__HIP_DEVICE_COMPILE__)

// acc + x * K as ONE v_mad_u64_u32 (32x32+64).  Inline asm because LLVM re-associates the C form
// (x*1 + y*1 -> (x+y)*1, zero-extensions through v_mov) and loses the free 64-bit accumulate.
// The carry-out SGPR pair is a dummy.
template <u32 K>
__device__ __forceinline__ u64 p2f_mad(u64 acc, u32 x) {
  u64 d, cy;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(x), "n"(K), "v"(acc));
  return d;
}
template <u32 K>
__device__ __forceinline__ u64 p2f_zmul(u32 x) {  // zero-extend (K = 1) / small multiple
  u64 d, cy;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(cy) : "v"(x), "n"(K));
  return d;
}
__device__ __forceinline__ u32 lo32(u64 x) { return (u32)x; }
__device__ __forceinline__ u32 hi32(u64 x) { return (u32)(x >> 32); }

// a*b mod p for ANY a, b < 2^64; result is some representative < 2^64 (not canonical).
__device__ __forceinline__ u64 p2f_mul(u64 a, u64 b) {
  const u32 a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
  const u64 p00 = (u64)a0 * b0;
  const u64 m = (u64)a0 * b1 + (p00 >> 32);
  const u64 m2 = (u64)a1 * b0 + (u32)m;
  const u64 hi = (u64)a1 * b1 + (m >> 32) + (m2 >> 32);
  const u64 lo = (m2 << 32) | (u32)p00;
  const u32 x2 = lo32(hi), x3 = hi32(hi);
  u64 t = (u64)x2 * 0xFFFFFFFFu + lo;  // + x2 * (2^64 mod p)
  t += (t < lo) ? GL_EPS : 0;
  u64 r = t - x3;                       // - x3 (2^96 = -1)
  r -= (t < x3) ? GL_EPS : 0;
  return r;
}
// (A hand-scheduled x^7 -- VCC carry chains, even-aligned register pairs with standing zeros, 17 VALU
// per multiplication instead of 25, two multiplications interleaved to cover the VCC wait states --
// was measured at 2.86-2.92 G perm/s vs 2.93 for this compiler-scheduled form: the saved instructions
// were 4-byte v_mov's, the added carry ops are 8-byte VOP3 encodings that issue ~1.5x slower, so it
// was dropped.)
__device__ __forceinline__ u64 p2f_sbox(u64 x) {
  const u64 x2 = p2f_mul(x, x);
  const u64 x3 = p2f_mul(x2, x);
  const u64 x4 = p2f_mul(x2, x2);
  return p2f_mul(x3, x4);
}
// x + c for any x < 2^64 and canonical c
__device__ __forceinline__ u64 p2f_add_canon(u64 x, u64 c) {
  u64 s = x + c;
  s += (s < c) ? GL_EPS : 0;
  return s;
}
// Fold a non-negative wide value (L, H < 2^62) to a 64-bit representative.
__device__ __forceinline__ u64 p2f_fold(u64 L, u64 H) {
  const u64 m = (u64)hi32(H) * 0xFFFFFFFFu + L;  // H_hi * 2^64 -> * eps; no overflow by the bounds
  const u32 hl = lo32(H);
  const u32 s = hi32(m) + hl;
  u64 r = ((u64)s << 32) | lo32(m);
  r += (s < hl) ? GL_EPS : 0;
  return r;
}
// Signed wide value (|L|, |H| < 2^61): add a multiple of p that makes both parts positive, then fold.
// bias = m*p split as (m + w*2^32, m*eps - w) with m = w = 2^30.
__device__ __forceinline__ u64 p2f_fold_signed(u64 L, u64 H) {
  const u64 BL = ((u64)1 << 30) + ((u64)1 << 62);
  const u64 BH = ((u64)1 << 30) * 0xFFFFFFFFULL - ((u64)1 << 30);
  return p2f_fold(L + BL, H + BH);
}

// out = circ(2*M4, M4, M4) * s (+ rc), s given as 64-bit values; result folded back to 64 bits.
// M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]  (poseidon2/mod.rs:233-281)
__device__ __forceinline__ void p2f_external(u64 s[12], const unsigned long long* rc) {
  u64 oL[12], oH[12];
#pragma unroll
  for (int i = 0; i < 12; i += 4) {
#pragma unroll
    for (int part = 0; part < 2; part++) {
      const u32 a = part ? hi32(s[i]) : lo32(s[i]), b = part ? hi32(s[i + 1]) : lo32(s[i + 1]);
      const u32 c = part ? hi32(s[i + 2]) : lo32(s[i + 2]), d = part ? hi32(s[i + 3]) : lo32(s[i + 3]);
      const u64 t01 = p2f_mad<1>(p2f_zmul<1>(a), b);
      const u64 t23 = p2f_mad<1>(p2f_zmul<1>(c), d);
      const u64 t0123 = t01 + t23;
      const u64 t01123 = p2f_mad<1>(t0123, b);
      const u64 t01233 = p2f_mad<1>(t0123, d);
      u64* o = part ? oH : oL;
      o[i + 3] = p2f_mad<2>(t01233, a);
      o[i + 1] = p2f_mad<2>(t01123, c);
      o[i] = t01123 + t01;
      o[i + 2] = t01233 + t23;
    }
  }
#pragma unroll
  for (int l = 0; l < 4; l++) {
    const u64 stL = oL[l] + oL[4 + l] + oL[8 + l], stH = oH[l] + oH[4 + l] + oH[8 + l];
#pragma unroll
    for (int i = l; i < 12; i += 4) {
      u64 L = oL[i] + stL, H = oH[i] + stH;
      if (rc) {
        L += rc[i] & 0xFFFFFFFFULL;
        H += rc[i] >> 32;
      }
      s[i] = p2f_fold(L, H);
    }
  }
}

__device__ __forceinline__ void p2f_permute(u64 s[12]) {
  // initial linear layer + round constants of external round 0
  p2f_external(s, p2c::P2_ARK_EXT_INITIAL);
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = p2f_sbox(s[i]);
    // linear layer, then the NEXT round's constants (the last one is internal round 0: element 0 only)
    if (r < 3) {
      p2f_external(s, p2c::P2_ARK_EXT_INITIAL + 12 * (r + 1));
    } else {
      p2f_external(s, nullptr);
    }
  }
  // ---- internal rounds in the 8^r-scaled domain ----
  u64 t0 = p2f_add_canon(s[0], p2c::P2F_ARK_INT_SCALED[0]);
  u64 L[12], H[12];  // [0] unused
#pragma unroll
  for (int i = 1; i < 12; i++) {
    L[i] = p2f_zmul<1>(lo32(s[i]));
    H[i] = p2f_zmul<1>(hi32(s[i]));
  }
#pragma unroll 1
  for (int r = 0; r < 22; r++) {
    // y = 8^r * (s0 + rc)^7
    const u64 y = p2f_mul(p2f_sbox(t0), p2c::P2F_INT_K[r]);
    u64 sL = p2f_mad<1>(L[1], lo32(y)), sH = p2f_mad<1>(H[1], hi32(y));
#pragma unroll
    for (int i = 2; i < 12; i++) {
      sL += L[i];
      sH += H[i];
    }
    const u64 s8L = sL << 3, s8H = sH << 3;
    // element 0: -16*y + 8*sum (+ the next round's scaled constant), folded for the next S-box
    {
      u64 nL = s8L - p2f_zmul<16>(lo32(y)), nH = s8H - p2f_zmul<16>(hi32(y));
      if (r < 21) {
        const u64 rc = p2c::P2F_ARK_INT_SCALED[r + 1];
        nL += rc & 0xFFFFFFFFULL;
        nH += rc >> 32;
      }
      t0 = p2f_fold_signed(nL, nH);
    }
#define P2F_UPD(i, EXPR_L, EXPR_H) \
  {                                \
    const u64 xl = L[i], xh = H[i]; \
    L[i] = EXPR_L;                 \
    H[i] = EXPR_H;                 \
  }
    P2F_UPD(1, (xl << 3) + s8L, (xh << 3) + s8H)                                  //   8
    P2F_UPD(2, (xl << 4) + s8L, (xh << 4) + s8H)                                  //  16
    P2F_UPD(3, (xl << 2) + s8L, (xh << 2) + s8H)                                  //   4
    P2F_UPD(4, (((xl << 1) + xl) << 3) + s8L, (((xh << 1) + xh) << 3) + s8H)      //  24
    P2F_UPD(5, ((xl << 1) << 4) + s8L, ((xh << 1) << 4) + s8H)                    //  32
    P2F_UPD(6, s8L - (xl << 2), s8H - (xh << 2))                                  //  -4
    P2F_UPD(7, s8L - (((xl << 1) + xl) << 3), s8H - (((xh << 1) + xh) << 3))      // -24
    P2F_UPD(8, s8L - (xl << 5), s8H - (xh << 5))                                  // -32
    P2F_UPD(9, (xl << 1) + s8L, (xh << 1) + s8H)                                  //   2
    P2F_UPD(10, s8L - (xl << 1), s8H - (xh << 1))                                 //  -2
    P2F_UPD(11, xl + s8L, xh + s8H)                                               //   1
#undef P2F_UPD
    if ((r & 3) == 3) {  // parts have grown by <= 7 bits per round from < 2^32: refold before 2^61
#pragma unroll
      for (int i = 1; i < 12; i++) {
        const u64 v = p2f_fold_signed(L[i], H[i]);
        L[i] = p2f_zmul<1>(lo32(v));
        H[i] = p2f_zmul<1>(hi32(v));
      }
    }
  }
  // leave the scaled domain (factor 8^22) and add the first terminal round constants
  s[0] = p2f_add_canon(p2f_mul(t0, p2c::P2F_DESCALE), p2c::P2_ARK_EXT_TERMINAL[0]);
#pragma unroll
  for (int i = 1; i < 12; i++)
    s[i] = p2f_add_canon(p2f_mul(p2f_fold_signed(L[i], H[i]), p2c::P2F_DESCALE), p2c::P2_ARK_EXT_TERMINAL[i]);
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = p2f_sbox(s[i]);
    if (r < 3) {
      p2f_external(s, p2c::P2_ARK_EXT_TERMINAL + 12 * (r + 1));
    } else {
      p2f_external(s, nullptr);
    }
  }
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
}

#else
// host pass: kernels are only parsed, never code-generated
__device__ void p2f_permute(u64 s[12]);
__device__ u64 p2f_mul(u64 a, u64 b);
__device__ u32 lo32(u64 x);
__device__ u32 hi32(u64 x);
#endif  // __HIP_DEVICE_COMPILE__
