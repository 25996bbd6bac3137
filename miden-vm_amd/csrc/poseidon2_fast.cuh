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
//     [-2,1,2,1/2,3,4,-1/2,-3,-4,1/4,-1/4,1/8] is cleared by carrying the whole state scaled by c_r = 2 * 8^r
//     (integer diagonal [-16,8,16,4,24,32,-4,-24,-32,2,-2,1]); the S-box of round r then costs one
//     extra multiplication by the constant c_r^(-6), and wide values are refolded every 4 rounds;
//   * the diagonal pairs (+k, -k) -- elements (3,6), (4,7), (5,8), (9,10) -- are carried as h = (x_i + x_j)/2, whose
//     two-step recurrence h(r+1) = k^2 h(r-1) + 8*sum is ONE shift-add per part and round (no 64-bit subtraction, which is
//     two instructions plus the shift), and the sum reads one value per pair; tests/test_p2_fast_schedule.py checks the algebra
//     on exact integers, tools/p2_pair_bounds.py the magnitudes.  10 953 -> 10 120 VALU instructions per permutation.
#pragma once
#include "poseidon2.cuh"

namespace p2c {
#include "p2_fast_constants.inc"
}

#ifndef P2F_ASM
#define P2F_ASM 1  // 1: multiplication with SGPR carry chains in inline asm; 0: plain C (see p2f_mul)
#endif
#ifndef P2F_GROUP
#define P2F_GROUP 4  // S-boxes of an external round interleaved per asm block
#endif

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
//
// Two forms (P2F_ASM, default 1):
//  * P2F_ASM = 0: plain C.  hipcc turns it into ~23 VALU instructions: every 32 -> 64-bit zero extension is a
//    v_mov into an even-aligned pair, every carry a v_cmp_lt_u64 + v_cndmask + 64-bit add.
//  * P2F_ASM = 1: 13 VALU + 2 SALU.  Carries never become vector values: v_mad_u64_u32's own carry-out and
//    v_add_co / v_addc_co / v_subb_co chains hand them over in SGPR pairs, the two flag combinations run on the
//    scalar unit, and the partial sums are written straight into the halves of the register pair the next
//    v_mad_u64_u32 accumulates (no moves):
//        p00 = a0*b0;  m = a0*b1;  m += a1*b0 (carry cm);  w1 = p00.hi + m.lo (k1);  acc.lo = m.hi + k1 (k2);
//        acc.hi = cm|k2;  hi = a1*b1 + acc;  lo = {p00.lo, w1}
//        t = hi.lo*(2^32-1) + lo (c1);   r = t - hi.hi + (c1 - borrow)*(2^32-1)   [2^64 = 2^32-1, 2^96 = -1 mod p]
//    gfx950 needs 2 wait states between a VALU that writes an SGPR and a VALU that reads it; the statements are
//    `asm volatile` (kept in program order), N independent products are interleaved stage by stage so that for
//    N >= 3 the distance is there by construction, and N = 1, 2 insert the missing s_nop themselves.
__device__ __forceinline__ u64 p2f_mul_c(u64 a, u64 b) {
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
#define P2F_A asm volatile
// wait states still missing between stage k of product i and stage k+1 of the same product (N-1 instructions lie between)
#ifndef P2F_GAP_NOPS
#define P2F_GAP_NOPS 1  // 0: experiment only (tools/permbench_nogap): rely on the compiler's own s_nop between asm statements
#endif
#define P2F_GAP()                                          \
  do {                                                     \
    if (P2F_GAP_NOPS && N == 1) P2F_A("s_nop 1");          \
    else if (P2F_GAP_NOPS && N == 2) P2F_A("s_nop 0");     \
  } while (0)
// v_mad_u64_u32 always writes a carry-out pair.  Where it is not needed it goes to one of eight fixed scratch pairs,
// rotating between neighbouring statements: hipcc separates two inline-asm statements that touch a common
// register by an s_nop (its dst-forwarding rule cannot see inside the asm), and it would hand every dead carry the
// same pair.
#define P2F_SCR(i, PRE, POST, ...)                                                  \
  do {                                                                              \
    switch ((i) & 7) {                                                              \
      case 0: P2F_A(PRE "s[84:85]" POST : __VA_ARGS__ : "s84", "s85"); break;        \
      case 1: P2F_A(PRE "s[86:87]" POST : __VA_ARGS__ : "s86", "s87"); break;        \
      case 2: P2F_A(PRE "s[88:89]" POST : __VA_ARGS__ : "s88", "s89"); break;        \
      case 3: P2F_A(PRE "s[90:91]" POST : __VA_ARGS__ : "s90", "s91"); break;        \
      case 4: P2F_A(PRE "s[92:93]" POST : __VA_ARGS__ : "s92", "s93"); break;        \
      case 5: P2F_A(PRE "s[94:95]" POST : __VA_ARGS__ : "s94", "s95"); break;        \
      case 6: P2F_A(PRE "s[96:97]" POST : __VA_ARGS__ : "s96", "s97"); break;        \
      default: P2F_A(PRE "s[98:99]" POST : __VA_ARGS__ : "s98", "s99"); break;       \
    }                                                                               \
  } while (0)
#define P2F_MAD0(i, D, X, Y) P2F_SCR(i, "v_mad_u64_u32 %0, ", ", %1, %2, 0", "=v"(D) : "v"(X), "v"(Y))
#define P2F_MADA(i, D, X, Y, ACC) P2F_SCR(i, "v_mad_u64_u32 %0, ", ", %1, %2, %3", "=v"(D) : "v"(X), "v"(Y), "v"(ACC))
#define P2F_CARRY_IN(i, OP, D, X, C) P2F_SCR(i, OP " %0, ", ", %1, 0, %2", "=v"(D) : "v"(X), "s"(C))
// the same with the second factor in scalar registers (a wave-uniform constant: no v_mov of its halves into VGPRs)
#define P2F_MAD0S(i, D, X, Y) P2F_SCR(i, "v_mad_u64_u32 %0, ", ", %1, %2, 0", "=v"(D) : "v"(X), "s"(Y))
#define P2F_MADAS(i, D, X, Y, ACC) P2F_SCR(i, "v_mad_u64_u32 %0, ", ", %1, %2, %3", "=v"(D) : "v"(X), "s"(Y), "v"(ACC))
// BS: every b[i] is wave-uniform (a table constant); it is read from SGPRs
template <int N, bool BS = false>
__device__ __forceinline__ void p2f_mulN(u64 (&r)[N], const u64 (&a)[N], const u64 (&b)[N]) {
  u64 p00[N], m[N], hi[N], t[N];
  u64 cm[N], k1[N], k2[N], k3[N], c1[N], bb[N], bw[N], c3[N];  // SGPR pairs: lane masks of carries
  u32 w1[N], accl[N], acch[N], rl[N], rh[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    if constexpr (BS) P2F_MAD0S(i, p00[i], lo32(a[i]), lo32(b[i]));
    else P2F_MAD0(i, p00[i], lo32(a[i]), lo32(b[i]));
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    if constexpr (BS) P2F_MAD0S(i, m[i], lo32(a[i]), hi32(b[i]));
    else P2F_MAD0(i, m[i], lo32(a[i]), hi32(b[i]));
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    if constexpr (BS) P2F_A("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(m[i]), "=s"(cm[i]) : "v"(hi32(a[i])), "s"(lo32(b[i])), "0"(m[i]));
    else P2F_A("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(m[i]), "=s"(cm[i]) : "v"(hi32(a[i])), "v"(lo32(b[i])), "0"(m[i]));
  }
#pragma unroll
  for (int i = 0; i < N; i++) P2F_A("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(w1[i]), "=s"(k1[i]) : "v"(hi32(p00[i])), "v"(lo32(m[i])));
  P2F_GAP();
#pragma unroll
  for (int i = 0; i < N; i++) P2F_A("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(accl[i]), "=s"(k2[i]) : "v"(hi32(m[i])), "s"(k1[i]));
#pragma unroll
  for (int i = 0; i < N; i++)  // cm and k2 exclude each other (a carried m leaves m.hi <= 2^32 - 5)
    k3[i] = cm[i] | k2[i];  // scalar unit
#pragma unroll
  for (int i = 0; i < N; i++) {
    const u32 zero = 0;
    P2F_CARRY_IN(i, "v_addc_co_u32_e64", acch[i], zero, k3[i]);
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    const u64 acc = ((u64)acch[i] << 32) | accl[i];
    if constexpr (BS) P2F_MADAS(i, hi[i], hi32(a[i]), hi32(b[i]), acc);
    else P2F_MADA(i, hi[i], hi32(a[i]), hi32(b[i]), acc);
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    const u64 lo = ((u64)w1[i] << 32) | lo32(p00[i]);
    P2F_A("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(t[i]), "=s"(c1[i]) : "v"(lo32(hi[i])), "v"(lo));
  }
  P2F_GAP();
  // V = t + c1*(2^32 - 1) - x3 (mod 2^64): low word t.lo - x3 - c1, high word t.hi + c1 - borrow
#pragma unroll
  for (int i = 0; i < N; i++) P2F_A("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(rl[i]), "=s"(bb[i]) : "v"(lo32(t[i])), "v"(hi32(hi[i])), "s"(c1[i]));
#pragma unroll
  for (int i = 0; i < N; i++) P2F_CARRY_IN(i, "v_addc_co_u32_e64", rh[i], hi32(t[i]), c1[i]);
  if (P2F_GAP_NOPS && N == 1) P2F_A("s_nop 0");
#pragma unroll
  for (int i = 0; i < N; i++) P2F_A("v_subb_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rh[i]), "=s"(bw[i]) : "0"(rh[i]), "s"(bb[i]));
  P2F_GAP();
  // V < 0 (bw): the wrapped value is >= 2^64 - 2^32; subtract 2^32 - 1 once more: lo += 1 (carry c3), hi -= 1 - c3
#pragma unroll
  for (int i = 0; i < N; i++) P2F_A("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rl[i]), "=s"(c3[i]) : "0"(rl[i]), "s"(bw[i]));
#pragma unroll
  for (int i = 0; i < N; i++) k3[i] = bw[i] & ~c3[i];  // scalar unit
#pragma unroll
  for (int i = 0; i < N; i++) {
    P2F_CARRY_IN(i, "v_subb_co_u32_e64", rh[i], rh[i], k3[i]);
    r[i] = ((u64)rh[i] << 32) | rl[i];
  }
}
#undef P2F_GAP
#undef P2F_SCR
#undef P2F_MAD0
#undef P2F_MADA
#undef P2F_MAD0S
#undef P2F_MADAS
#undef P2F_CARRY_IN
// The same 13-instruction product with NON-volatile statements, each carry consumer carrying its own 2 wait states: for
// code whose schedule must stay free (loads hoisted above the products, independent products interleaved by the compiler:
// the NTT passes).  Costs an s_nop 1 in front of five instructions; hidden when several waves share the SIMD.
__device__ __forceinline__ u64 p2f_mul_nv(u64 a, u64 b) {
  u64 p00, m, hi, t, d0, d1, d2, d3, d4, d5, cm, k1, k2, c1, bb, bw, c3;
  u32 w1, accl, acch, rl, rh;
  const u32 zero = 0;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p00), "=s"(d0) : "v"(lo32(a)), "v"(lo32(b)));
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(m), "=s"(d1) : "v"(lo32(a)), "v"(hi32(b)));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(m), "=s"(cm) : "v"(hi32(a)), "v"(lo32(b)), "0"(m));
  asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(w1), "=s"(k1) : "v"(hi32(p00)), "v"(lo32(m)));
  asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(accl), "=s"(k2) : "v"(hi32(m)), "s"(k1));
  const u64 k3 = cm | k2;
  asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(acch), "=s"(d2) : "v"(zero), "s"(k3));
  const u64 acc = ((u64)acch << 32) | accl;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(hi), "=s"(d3) : "v"(hi32(a)), "v"(hi32(b)), "v"(acc));
  const u64 lo = ((u64)w1 << 32) | lo32(p00);
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(t), "=s"(c1) : "v"(lo32(hi)), "v"(lo));
  asm("s_nop 1\n\tv_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(rl), "=s"(bb) : "v"(lo32(t)), "v"(hi32(hi)), "s"(c1));
  asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rh), "=s"(d4) : "v"(hi32(t)), "s"(c1));
  asm("s_nop 1\n\tv_subb_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rh), "=s"(bw) : "0"(rh), "s"(bb));
  asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rl), "=s"(c3) : "0"(rl), "s"(bw));
  const u64 mk = bw & ~c3;
  asm("v_subb_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rh), "=s"(d5) : "0"(rh), "s"(mk));
  return ((u64)rh << 32) | rl;
}
__device__ __forceinline__ u64 p2f_mul(u64 a, u64 b) {
#if P2F_ASM
  u64 r[1];
  const u64 x[1] = {a}, y[1] = {b};
  p2f_mulN<1>(r, x, y);
  return r[0];
#else
  return p2f_mul_c(a, b);
#endif
}
// a * k for a wave-uniform k
__device__ __forceinline__ u64 p2f_mul_k(u64 a, u64 k) {
#if P2F_ASM
  u64 r[1];
  const u64 x[1] = {a}, y[1] = {k};
  p2f_mulN<1, true>(r, x, y);
  return r[0];
#else
  return p2f_mul_c(a, k);
#endif
}
__device__ __forceinline__ u64 p2f_sbox(u64 x) {
#if P2F_ASM
  const u64 x2 = p2f_mul(x, x);
  u64 q[2];
  const u64 qa[2] = {x2, x2}, qb[2] = {x, x2};
  p2f_mulN<2>(q, qa, qb);  // x^3, x^4
  return p2f_mul(q[0], q[1]);
#else
  const u64 x2 = p2f_mul(x, x);
  const u64 x3 = p2f_mul(x2, x);
  const u64 x4 = p2f_mul(x2, x2);
  return p2f_mul(x3, x4);
#endif
}
// S-box layer of an external round: the 12 S-boxes are independent; P2F_GROUP of them run interleaved.
__device__ __forceinline__ void p2f_sbox12(u64 s[12]) {
#if P2F_ASM
#pragma unroll
  for (int g = 0; g < 12; g += P2F_GROUP) {
    u64 x[P2F_GROUP], x2[P2F_GROUP], x3[P2F_GROUP];
#pragma unroll
    for (int i = 0; i < P2F_GROUP; i++) x[i] = s[g + i];
    p2f_mulN<P2F_GROUP>(x2, x, x);
    p2f_mulN<P2F_GROUP>(x3, x2, x);
    p2f_mulN<P2F_GROUP>(x, x2, x2);
    p2f_mulN<P2F_GROUP>(x2, x3, x);
#pragma unroll
    for (int i = 0; i < P2F_GROUP; i++) s[g + i] = x2[i];
  }
#else
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = p2f_sbox(s[i]);
#endif
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
template <bool RC>
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
      if (RC) {  // a template flag: as a run-time test of the pointer it became four v_cndmask per element
        L += rc[i] & 0xFFFFFFFFULL;
        H += rc[i] >> 32;
      }
      s[i] = p2f_fold(L, H);
    }
  }
}

__device__ __forceinline__ void p2f_permute(u64 s[12]) {
  // initial linear layer + round constants of external round 0
  p2f_external<true>(s, p2c::P2_ARK_EXT_INITIAL);
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
    p2f_sbox12(s);
    // linear layer, then the NEXT round's constants (the last one is internal round 0: element 0 only)
    if (r < 3) {
      p2f_external<true>(s, p2c::P2_ARK_EXT_INITIAL + 12 * (r + 1));
    } else {
      p2f_external<false>(s, nullptr);
    }
  }
  // ---- internal rounds, all values scaled by c_r = 2 * 8^r ----
  // X' = 8 * (diag * X + sum), integer diagonal [-16, 8, 16, 4, 24, 32, -4, -24, -32, 2, -2, 1].  Elements (3,6), (4,7), (5,8),
  // (9,10) have diagonals (+k, -k): with h = (X_i + X_j) / 2 and b = X_i - X_j one round is h' = (k/2) b + s8, b' = 2 k h, so
  //     h(r+1) = k^2 * h(r-1) + s8(r)          (s8 = 8 * sum; the sum needs only 2 h per pair)
  // is all a round does for a pair -- one shift-add per part instead of two shifts, an addition and a subtraction -- and after the
  // last round X_i = h(22) + k h(21), X_j = h(22) - k h(21).  The factor 2 in c_r makes h(0) = x_i + x_j an integer sum.
  // Round 0 takes the TRUE state (its constants: P2G_ARK[0] = ark_0, P2G_K[0] = c_0 = 2).
  u64 t0 = p2f_add_canon(s[0], p2c::P2G_ARK[0]);
  u64 X1L, X1H, X2L, X2H, X11L, X11H;
  u64 AL[4], AH[4], BL[4], BH[4];  // pair q: A = h(even round), B = h(odd round)
  {
    const u64 y = p2f_mul_k(p2f_sbox(t0), p2c::P2G_K[0]);
    u64 RL = p2f_zmul<1>(lo32(s[1])), RH = p2f_zmul<1>(hi32(s[1]));
#pragma unroll
    for (int i = 2; i < 12; i++) {
      RL = p2f_mad<1>(RL, lo32(s[i]));
      RH = p2f_mad<1>(RH, hi32(s[i]));
    }
    // sum = y + 2 * (x_1 + ... + x_11);  element 0: -16 y + 8 sum = 8 * (2 R - y)
    const u64 s8L = p2f_mad<1>(RL << 1, lo32(y)) << 3, s8H = p2f_mad<1>(RH << 1, hi32(y)) << 3;
    {
      const u64 rc = p2c::P2G_ARK[1];
      t0 = p2f_fold_signed((((RL << 1) - (u64)lo32(y)) << 3) + (rc & 0xFFFFFFFFULL), (((RH << 1) - (u64)hi32(y)) << 3) + (rc >> 32));
    }
    X1L = p2f_mad<16>(s8L, lo32(s[1])), X1H = p2f_mad<16>(s8H, hi32(s[1]));     // 8 * (2 x)
    X2L = p2f_mad<32>(s8L, lo32(s[2])), X2H = p2f_mad<32>(s8H, hi32(s[2]));     // 16 * (2 x)
    X11L = p2f_mad<2>(s8L, lo32(s[11])), X11H = p2f_mad<2>(s8H, hi32(s[11]));   // 1 * (2 x)
#define P2G_PAIR0(q, i, j, K)                                                         \
  AL[q] = p2f_mad<1>(p2f_zmul<1>(lo32(s[i])), lo32(s[j]));                            \
  AH[q] = p2f_mad<1>(p2f_zmul<1>(hi32(s[i])), hi32(s[j]));                            \
  BL[q] = p2f_mad<K>(s8L, lo32(s[i])) - p2f_zmul<K>(lo32(s[j]));                      \
  BH[q] = p2f_mad<K>(s8H, hi32(s[i])) - p2f_zmul<K>(hi32(s[j]));
    P2G_PAIR0(0, 3, 6, 4)
    P2G_PAIR0(1, 4, 7, 24)
    P2G_PAIR0(2, 5, 8, 32)
    P2G_PAIR0(3, 9, 10, 2)
#undef P2G_PAIR0
  }
  // round r: C = h(r), P = h(r - 1) -> P = h(r + 1)
#define P2G_ROUND(r, CL, CH, PL, PH, HAS_RC)                                                             \
  {                                                                                                       \
    const u64 y = p2f_mul_k(p2f_sbox(t0), p2c::P2G_K[r]);                                                   \
    const u64 RL = ((CL[0] + CL[1] + CL[2] + CL[3]) << 1) + X1L + X2L + X11L;                             \
    const u64 RH = ((CH[0] + CH[1] + CH[2] + CH[3]) << 1) + X1H + X2H + X11H;                             \
    const u64 s8L = p2f_mad<1>(RL, lo32(y)) << 3, s8H = p2f_mad<1>(RH, hi32(y)) << 3;                     \
    {                                                                                                     \
      u64 nL = (RL - (u64)lo32(y)) << 3, nH = (RH - (u64)hi32(y)) << 3; /* -16 y + 8 (R + y) */           \
      if (HAS_RC) {                                                                                       \
        const u64 rc = p2c::P2G_ARK[(r) + 1];                                                             \
        nL += rc & 0xFFFFFFFFULL;                                                                         \
        nH += rc >> 32;                                                                                   \
      }                                                                                                   \
      t0 = p2f_fold_signed(nL, nH);                                                                       \
    }                                                                                                     \
    X1L = (X1L << 3) + s8L, X1H = (X1H << 3) + s8H;                                                       \
    X2L = (X2L << 4) + s8L, X2H = (X2H << 4) + s8H;                                                       \
    X11L += s8L, X11H += s8H;                                                                             \
    PL[0] = (PL[0] << 4) + s8L, PH[0] = (PH[0] << 4) + s8H;                               /* 4^2 */       \
    {                                                                           /* 24^2 = 9 * 64 */       \
      u64 uL, uH; /* 9 h as ONE shift-add: from C, LLVM makes it a 64 x 32-bit product (two mads and two moves per part) */ \
      asm("v_lshl_add_u64 %0, %1, 3, %1" : "=v"(uL) : "v"(PL[1]));                                        \
      asm("v_lshl_add_u64 %0, %1, 3, %1" : "=v"(uH) : "v"(PH[1]));                                        \
      PL[1] = ((uL << 2) << 4) + s8L, PH[1] = ((uH << 2) << 4) + s8H;                                     \
    }                                                                                                     \
    PL[2] = ((PL[2] << 6) << 4) + s8L, PH[2] = ((PH[2] << 6) << 4) + s8H;                 /* 32^2 */      \
    PL[3] = (PL[3] << 2) + s8L, PH[3] = (PH[3] << 2) + s8H;                               /* 2^2 */       \
  }
#define P2G_REFOLD(L_, H_)                        \
  {                                               \
    const u64 v = p2f_fold_signed(L_, H_);        \
    L_ = p2f_zmul<1>(lo32(v));                    \
    H_ = p2f_zmul<1>(hi32(v));                    \
  }
#pragma unroll 1
  for (int r = 1; r < 22; r += 2) {
    P2G_ROUND(r, BL, BH, AL, AH, r < 21)
    if ((r & 3) == 3) {  // parts < 2^32 grow to < 2^60.5 in four rounds (tools/p2_pair_bounds.py): refold before 2^61
      P2G_REFOLD(X1L, X1H)
      P2G_REFOLD(X2L, X2H)
      P2G_REFOLD(X11L, X11H)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        P2G_REFOLD(AL[q], AH[q])
        P2G_REFOLD(BL[q], BH[q])
      }
    }
    if (r + 1 < 22) P2G_ROUND(r + 1, AL, AH, BL, BH, true)
  }
#undef P2G_ROUND
#undef P2G_REFOLD
  // A = h(22), B = h(21): back to the elements, out of the scaled domain (factor c_22), then the first terminal round constants
  s[0] = t0;
  s[1] = p2f_fold_signed(X1L, X1H);
  s[2] = p2f_fold_signed(X2L, X2H);
  s[11] = p2f_fold_signed(X11L, X11H);
#define P2G_UNPAIR(q, i, j, KL, KH)                         \
  {                                                         \
    const u64 kL = KL, kH = KH;                             \
    s[i] = p2f_fold_signed(AL[q] + kL, AH[q] + kH);         \
    s[j] = p2f_fold_signed(AL[q] - kL, AH[q] - kH);         \
  }
  P2G_UNPAIR(0, 3, 6, BL[0] << 2, BH[0] << 2)
  P2G_UNPAIR(1, 4, 7, ((BL[1] << 1) + BL[1]) << 3, ((BH[1] << 1) + BH[1]) << 3)
  P2G_UNPAIR(2, 5, 8, BL[2] << 5, BH[2] << 5)
  P2G_UNPAIR(3, 9, 10, BL[3] << 1, BH[3] << 1)
#undef P2G_UNPAIR
#if P2F_ASM
#pragma unroll
  for (int g = 0; g < 12; g += 4) {
    u64 x[4], k[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { x[i] = s[g + i]; k[i] = p2c::P2G_DESCALE; }
    p2f_mulN<4>(x, x, k);
#pragma unroll
    for (int i = 0; i < 4; i++) s[g + i] = x[i];
  }
#else
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = p2f_mul(s[i], p2c::P2G_DESCALE);
#endif
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = p2f_add_canon(s[i], p2c::P2_ARK_EXT_TERMINAL[i]);
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
    p2f_sbox12(s);
    if (r < 3) {
      p2f_external<true>(s, p2c::P2_ARK_EXT_TERMINAL + 12 * (r + 1));
    } else {
      p2f_external<false>(s, nullptr);
    }
  }
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
}

#else
// host pass: kernels are only parsed, never code-generated
__device__ void p2f_permute(u64 s[12]);
__device__ u64 p2f_mul(u64 a, u64 b);
__device__ u64 p2f_mul_nv(u64 a, u64 b);
__device__ u64 p2f_mul_k(u64 a, u64 k);
__device__ u64 p2f_sbox(u64 x);
__device__ void p2f_sbox12(u64 s[12]);
__device__ u32 lo32(u64 x);
__device__ u32 hi32(u64 x);
#endif  // __HIP_DEVICE_COMPILE__
