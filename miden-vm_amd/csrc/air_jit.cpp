// DAG -> HIP source -> hiprtc -> gfx950 code objects (see air_jit.hpp).
#include "air_jit.hpp"
#include "gl.cuh"
#include <hip/hiprtc.h>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <sys/stat.h>
#include <unistd.h>

namespace {

// Self-contained device prelude: canonical Goldilocks arithmetic identical to gl.cuh (so that compiled and
// interpreted evaluation agree bit for bit) and the argument block of air_jit.hpp.
const char* JIT_PRELUDE = R"SRC(
typedef unsigned long long u64;
typedef unsigned int u32;
#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL
#define FI static __device__ inline __attribute__((always_inline))
FI u64 gl_add(u64 a, u64 b) { u64 s = a + b; u64 t = s + GL_EPS; return (s < a || s >= GL_P) ? t : s; }
FI u64 gl_sub(u64 a, u64 b) { u64 d = a - b; return (a < b) ? d - GL_EPS : d; }
FI u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
FI u64 gl_reduce128(u64 hi, u64 lo) {
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= GL_EPS;
  u64 t1 = (hi_lo << 32) - hi_lo;
  u64 r = t0 + t1;
  if (r < t1) r += GL_EPS;
  return r >= GL_P ? r - GL_P : r;
}
typedef unsigned __int128 u128;
FI u64 gl_reduce128_lazy(u64 hi, u64 lo) {  // gl_reduce128 without its last line: any representative < 2^64
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= GL_EPS;
  u64 t1 = (hi_lo << 32) - hi_lo;
  u64 r = t0 + t1;
  if (r < t1) r += GL_EPS;
  return r;
}
FI u64 gl_mul_c(u64 a, u64 b) { const u128 p = (u128)a * b; return gl_reduce128((u64)(p >> 64), (u64)p); }  // one 64 x 64 -> 128 product (four v_mad_u64_u32), not __umul64hi + a second low product
#ifndef MH_JIT_ASM_MUL
// 0: plain C products, 1: the asm product everywhere (the default since round 6), 2: for base-field gates only (rounds 4-5)
#define MH_JIT_ASM_MUL 1
#endif
#ifndef MH_JIT_NOPS
#define MH_JIT_NOPS 1  // 0: TIMING PROBE ONLY (no wait states between a carry's writer and its reader: gfx950 documents 2)
#endif
#if MH_JIT_NOPS
#define JNOP "s_nop 1\n\t"
#else
#define JNOP ""
#endif
#if MH_JIT_ASM_MUL
// the 13-instruction SGPR-carry-chain product of poseidon2_fast.cuh (p2f_mul_nv: non-volatile statements carrying their own
// wait states), canonicalised on exit -- an experiment switch (-DMH_JIT_ASM_MUL=1 through $MH_JIT_FLAGS), see DESIGN.md section 3
FI u32 jlo(u64 x) { return (u32)x; }
FI u32 jhi(u64 x) { return (u32)(x >> 32); }
FI u64 gl_mul(u64 a, u64 b) {
  u64 p00, m, hi, t, d0, d1, d2, d3, d4, d5, cm, k1, k2, c1, bb, bw, c3;
  u32 w1, accl, acch, rl, rh;
  const u32 zero = 0;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p00), "=s"(d0) : "v"(jlo(a)), "v"(jlo(b)));
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(m), "=s"(d1) : "v"(jlo(a)), "v"(jhi(b)));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(m), "=s"(cm) : "v"(jhi(a)), "v"(jlo(b)), "0"(m));
  asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(w1), "=s"(k1) : "v"(jhi(p00)), "v"(jlo(m)));
  asm(JNOP "v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(accl), "=s"(k2) : "v"(jhi(m)), "s"(k1));
  const u64 k3 = cm | k2;
  asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(acch), "=s"(d2) : "v"(zero), "s"(k3));
  const u64 acc = ((u64)acch << 32) | accl;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(hi), "=s"(d3) : "v"(jhi(a)), "v"(jhi(b)), "v"(acc));
  const u64 lo = ((u64)w1 << 32) | jlo(p00);
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(t), "=s"(c1) : "v"(jlo(hi)), "v"(lo));
  asm(JNOP "v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(rl), "=s"(bb) : "v"(jlo(t)), "v"(jhi(hi)), "s"(c1));
  asm(JNOP "v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rh), "=s"(d4) : "v"(jhi(t)), "s"(c1));
  asm(JNOP "v_subb_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rh), "=s"(bw) : "0"(rh), "s"(bb));
  asm(JNOP "v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rl), "=s"(c3) : "0"(rl), "s"(bw));
  const u64 mk = bw & ~c3;
  asm("v_subb_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rh), "=s"(d5) : "0"(rh), "s"(mk));
  const u64 r = ((u64)rh << 32) | rl;
  const u64 u = r + GL_EPS;  // r >= p  <=>  r + eps carries out of 64 bits
  return u < r ? u : r;
}
#else
FI u64 gl_mul(u64 a, u64 b) { return gl_mul_c(a, b); }
#endif
struct e2 { u64 c0, c1; };
FI u64 gl_mul7(u64 a) { u64 a2 = gl_add(a, a), a4 = gl_add(a2, a2), a8 = gl_add(a4, a4); return gl_sub(a8, a); }
FI e2 e2_add(e2 a, e2 b) { return {gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)}; }
FI e2 e2_sub(e2 a, e2 b) { return {gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)}; }
FI e2 e2_neg(e2 a) { return {gl_neg(a.c0), gl_neg(a.c1)}; }
FI e2 e2_addf(e2 a, u64 b) { return {gl_add(a.c0, b), a.c1}; }
FI e2 e2_subf(e2 a, u64 b) { return {gl_sub(a.c0, b), a.c1}; }
FI e2 e2_fsub(u64 a, e2 b) { return {gl_sub(a, b.c0), gl_neg(b.c1)}; }
// MH_JIT_ASM_MUL == 2: the asm product for base-field gates only, EF products stay in C (their chunks lose with the asm form)
#if MH_JIT_ASM_MUL == 2
#define gl_mul_ef gl_mul_c
#else
#define gl_mul_ef gl_mul
#endif
FI e2 e2_mul(e2 a, e2 b) {
  u64 a0b0 = gl_mul_ef(a.c0, b.c0), a1b1 = gl_mul_ef(a.c1, b.c1);
  u64 cross = gl_mul_ef(gl_add(a.c0, a.c1), gl_add(b.c0, b.c1));
  return {gl_add(a0b0, gl_mul7(a1b1)), gl_sub(gl_sub(cross, a0b0), a1b1)};
}
FI e2 e2_mulf(e2 a, u64 b) { return {gl_mul_ef(a.c0, b), gl_mul_ef(a.c1, b)}; }
// The alpha fold with the modular reduction delayed to the end of the chunk: alpha^k is uniform, so it is cut into three 22-bit limbs
// (scalar unit), a constraint value into 32-bit halves, and the 54-bit partial products are summed by weight in six plain 64-bit
// accumulators -- 6 v_mad_u64_u32 per (base value, alpha component) instead of a modular multiplication and a modular addition;
// < 2^9 folds per chunk keep every accumulator below 2^64.  MH_JIT_FOLD=0 (flag) selects the former per-constraint form.
#ifndef MH_JIT_FOLD
#define MH_JIT_FOLD 1
#endif
#ifndef MH_JIT_FOLDV
#define MH_JIT_FOLDV 1  // 0: the limb-by-limb recombination of round 4
#endif
struct fold_acc { u64 w0, w1, w2, w3, w4, w5; };
// acc + a * x as ONE v_mad_u64_u32 with the uniform limb in an SGPR; inline asm because LLVM reassociates the C form of these sums
// (every product of the chunk then stays live to the end: 512 VGPRs and scratch)
// The carry-out nobody reads goes to one of six fixed scratch SGPR pairs, a different one for each of the six products of fold_limbs:
// hipcc separates two asm statements that touch a common register by an s_nop (its dst-forwarding rule cannot look inside) and would
// hand every dead carry the same pair -- 12 s_nop per base constraint (ISA of the core AIR's chunks, round 6).
template <int I> FI void fold_mad(u64& acc, u32 a, u32 x) {
  u64 d;
  if (I == 0) asm("v_mad_u64_u32 %0, s[88:89], %1, %2, %3" : "=v"(d) : "s"(a), "v"(x), "v"(acc) : "s88", "s89");
  else if (I == 1) asm("v_mad_u64_u32 %0, s[90:91], %1, %2, %3" : "=v"(d) : "s"(a), "v"(x), "v"(acc) : "s90", "s91");
  else if (I == 2) asm("v_mad_u64_u32 %0, s[92:93], %1, %2, %3" : "=v"(d) : "s"(a), "v"(x), "v"(acc) : "s92", "s93");
  else if (I == 3) asm("v_mad_u64_u32 %0, s[94:95], %1, %2, %3" : "=v"(d) : "s"(a), "v"(x), "v"(acc) : "s94", "s95");
  else if (I == 4) asm("v_mad_u64_u32 %0, s[96:97], %1, %2, %3" : "=v"(d) : "s"(a), "v"(x), "v"(acc) : "s96", "s97");
  else asm("v_mad_u64_u32 %0, s[98:99], %1, %2, %3" : "=v"(d) : "s"(a), "v"(x), "v"(acc) : "s98", "s99");
  acc = d;
}
FI void fold_limbs(fold_acc& f, u64 alpha, u64 x) {  // three 22-bit limbs of alpha x two 32-bit halves of x: 54-bit products
  const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
  const u32 a0 = (u32)alpha & 0x3fffffu, a1 = (u32)(alpha >> 22) & 0x3fffffu, a2 = (u32)(alpha >> 44);
  fold_mad<0>(f.w0, a0, x0); fold_mad<1>(f.w1, a1, x0); fold_mad<2>(f.w2, a2, x0);
  fold_mad<3>(f.w3, a0, x1); fold_mad<4>(f.w4, a1, x1); fold_mad<5>(f.w5, a2, x1);
}
FI u64 fold_value(const fold_acc& f) {  // w0 + 2^22 w1 + 2^44 w2 + 2^32 (w3 + 2^22 w4 + 2^44 w5) mod p, canonical
#if MH_JIT_FOLDV == 0
  u64 r = gl_reduce128(0, f.w0);
  r = gl_add(r, gl_reduce128(f.w1 >> 42, f.w1 << 22));
  r = gl_add(r, gl_reduce128(f.w2 >> 20, f.w2 << 44));
  r = gl_add(r, gl_reduce128(f.w3 >> 32, f.w3 << 32));
  r = gl_add(r, gl_reduce128(f.w4 >> 10, f.w4 << 54));
  r = gl_add(r, gl_mul_c(gl_reduce128(0, f.w5), GL_EPS << 12));  // 2^76 = 2^12 (2^64 mod p)
  return r;
#else
  // two 108-bit sums in 128-bit arithmetic, the upper one reduced and shifted into the lower, ONE canonical reduction (the six
  // reductions and five modular additions of the form above were 139 VALU instructions per call)
  const u128 lo = (u128)f.w0 + ((u128)f.w1 << 22) + ((u128)f.w2 << 44);
  const u128 up = (u128)f.w3 + ((u128)f.w4 << 22) + ((u128)f.w5 << 44);
  const u64 u = gl_reduce128_lazy((u64)(up >> 64), (u64)up);
  const u128 v = lo + ((u128)u << 32);   // < 2^108 + 2^96
  return gl_reduce128((u64)(v >> 64), (u64)v);
#endif
}
// ---- any-representative ("lazy") arithmetic: MH_JIT_LAZYVAL (generator switch, default on) ----
// Between gates a value is ANY u64 congruent to it mod p; only what leaves the chunk as a field element (the fold's accumulator, the
// outputs of a lookup program) is canonical.  A canonicalising addition costs two compares and two selects more than the carry fix-up
// needs, a canonicalising product a compare, a subtraction and two selects.  `_c` forms take one operand known (at code-generation
// time) to be canonical -- a trace cell, a constant, a uniform value -- for which a single fix-up is exact:
//   a + c, c <= p - 1:  a wrapped sum is <= 2^64 - 2^32 - 1, so + eps cannot wrap again;
//   a - c, c <= p - 1:  a wrapped difference is >= 2^32, so - eps cannot wrap again.
// The `_g` forms (both operands arbitrary) apply the fix-up twice.
FI u64 lz_canon(u64 a) { return a >= GL_P ? a - GL_P : a; }
FI u64 lz_add_c(u64 a, u64 c) { u64 s = a + c; return s < a ? s + GL_EPS : s; }
FI u64 lz_add_g(u64 a, u64 b) { u64 s = a + b; if (s < a) { s += GL_EPS; if (s < GL_EPS) s += GL_EPS; } return s; }
FI u64 lz_sub_c(u64 a, u64 c) { u64 d = a - c; return a < c ? d - GL_EPS : d; }
FI u64 lz_sub_g(u64 a, u64 b) { u64 d = a - b; if (a < b) { const u64 e = d - GL_EPS; d = d < GL_EPS ? e - GL_EPS : e; } return d; }
#define lz_reduce128 gl_reduce128_lazy
FI u64 lz_mul_c(u64 a, u64 b) { const u128 p = (u128)a * b; return lz_reduce128((u64)(p >> 64), (u64)p); }
#if MH_JIT_ASM_MUL
FI u64 lz_mul_asm(u64 a, u64 b) {  // the 13 instructions of gl_mul above, without the canonicalisation behind them
  u64 p00, m, hi, t, d0, d1, d2, d3, d4, d5, cm, k1, k2, c1, bb, bw, c3;
  u32 w1, accl, acch, rl, rh;
  const u32 zero = 0;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p00), "=s"(d0) : "v"(jlo(a)), "v"(jlo(b)));
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(m), "=s"(d1) : "v"(jlo(a)), "v"(jhi(b)));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(m), "=s"(cm) : "v"(jhi(a)), "v"(jlo(b)), "0"(m));
  asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(w1), "=s"(k1) : "v"(jhi(p00)), "v"(jlo(m)));
  asm(JNOP "v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(accl), "=s"(k2) : "v"(jhi(m)), "s"(k1));
  const u64 k3 = cm | k2;
  asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(acch), "=s"(d2) : "v"(zero), "s"(k3));
  const u64 acc = ((u64)acch << 32) | accl;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(hi), "=s"(d3) : "v"(jhi(a)), "v"(jhi(b)), "v"(acc));
  const u64 lo = ((u64)w1 << 32) | jlo(p00);
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(t), "=s"(c1) : "v"(jlo(hi)), "v"(lo));
  asm(JNOP "v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(rl), "=s"(bb) : "v"(jlo(t)), "v"(jhi(hi)), "s"(c1));
  asm(JNOP "v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rh), "=s"(d4) : "v"(jhi(t)), "s"(c1));
  asm(JNOP "v_subb_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rh), "=s"(bw) : "0"(rh), "s"(bb));
  asm(JNOP "v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rl), "=s"(c3) : "0"(rl), "s"(bw));
  const u64 mk = bw & ~c3;
  asm("v_subb_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rh), "=s"(d5) : "0"(rh), "s"(mk));
  return ((u64)rh << 32) | rl;
}
#endif
#if MH_JIT_ASM_MUL
#define lz_mul lz_mul_asm
#else
#define lz_mul lz_mul_c
#endif
// N independent products, stage by stage (the interleaving of poseidon2_fast.cuh's p2f_mulN): stage k of every product, then stage k + 1.
// A carry's reader then sits N - 1 instructions behind its writer, so for N >= 3 the two wait states gfx950 wants between a VALU that writes
// an SGPR and a VALU that reads it exist by construction and N = 2 needs one `s_nop 0` per link -- against lz_mul_asm's five `s_nop 1`
// per product PLUS the `s_nop 0` hipcc puts between any two asm statements that share a register (its dst-forwarding rule cannot look
// inside): ~20 wait states per 13-instruction product in the chunks of the core AIR (ISA counts: DESIGN.md section 3c, round 6).
// ONE asm statement per stage (N instructions; the stages S9 / S10, which read the same carry, share one): the order inside a statement is
// fixed, the order of the statements follows from their data dependences, so they need not be `volatile` -- a volatile statement without
// a memory clobber still counts as a memory access for hipcc's uniform-load analysis, which then reads alpha^k and the uniform table
// through vector loads and cuts the fold's limbs on the VALU (measured on the first form of this function: +13 % VALU, 262 VGPRs in one
// chunk).  Outputs are early-clobber: a later instruction of the statement must not find its input overwritten.  The generator groups
// products whose operands are ready (MH_JIT_MULGROUP); results are any representative, as lz_mul's.  Text generated by
// tools/gen_jit_mulN.py.
template <int N> FI void lz_mulN(u64 (&r)[N], const u64 (&a)[N], const u64 (&b)[N]) {
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = lz_mul(a[i], b[i]);
}
#if MH_JIT_ASM_MUL
template <> __device__ inline __attribute__((always_inline)) void lz_mulN<2>(u64 (&r)[2], const u64 (&a)[2], const u64 (&b)[2]) {
  u64 p00[2], m[2], hi[2], t[2], lo[2], acc[2], cm[2], k1[2], k2[2], k3[2], c1[2], bb[2], bw[2], c3[2], mk[2], d[4];
  u32 w1[2], accl[2], acch[2], rl[2], rh[2];
  const u32 zero = 0;
  asm("v_mad_u64_u32 %0, %1, %8, %9, 0\n\tv_mad_u64_u32 %2, %3, %10, %11, 0\n\tv_mad_u64_u32 %4, %5, %12, %13, 0\n\tv_mad_u64_u32 %6, %7, %14, %15, 0"
      : "=&v"(p00[0]), "=&s"(d[0]), "=&v"(m[0]), "=&s"(d[1]), "=&v"(p00[1]), "=&s"(d[2]), "=&v"(m[1]), "=&s"(d[3])
      : "v"(jlo(a[0])), "v"(jlo(b[0])), "v"(jlo(a[0])), "v"(jhi(b[0])), "v"(jlo(a[1])), "v"(jlo(b[1])), "v"(jlo(a[1])), "v"(jhi(b[1])));
  asm("v_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %2, %3, %7, %8, %2"
      : "=&v"(m[0]), "=&s"(cm[0]), "=&v"(m[1]), "=&s"(cm[1])
      : "v"(jhi(a[0])), "v"(jlo(b[0])), "0"(m[0]), "v"(jhi(a[1])), "v"(jlo(b[1])), "2"(m[1]));
  asm("v_add_co_u32_e64 %0, %1, %4, %5\n\tv_add_co_u32_e64 %2, %3, %6, %7"
      : "=&v"(w1[0]), "=&s"(k1[0]), "=&v"(w1[1]), "=&s"(k1[1])
      : "v"(jhi(p00[0])), "v"(jlo(m[0])), "v"(jhi(p00[1])), "v"(jlo(m[1])));
  asm("s_nop 0\n\t" "v_addc_co_u32_e64 %0, %1, %4, 0, %5\n\tv_addc_co_u32_e64 %2, %3, %6, 0, %7"
      : "=&v"(accl[0]), "=&s"(k2[0]), "=&v"(accl[1]), "=&s"(k2[1])
      : "v"(jhi(m[0])), "s"(k1[0]), "v"(jhi(m[1])), "s"(k1[1]));
  for (int i = 0; i < 2; i++) k3[i] = cm[i] | k2[i];  // scalar unit; cm and k2 exclude each other (a carried m leaves m.hi <= 2^32 - 5)
  asm("v_addc_co_u32_e64 %0, %1, %4, 0, %5\n\tv_addc_co_u32_e64 %2, %3, %6, 0, %7"
      : "=&v"(acch[0]), "=&s"(d[0]), "=&v"(acch[1]), "=&s"(d[1])
      : "v"(zero), "s"(k3[0]), "v"(zero), "s"(k3[1]));
  for (int i = 0; i < 2; i++) { acc[i] = ((u64)acch[i] << 32) | accl[i]; lo[i] = ((u64)w1[i] << 32) | jlo(p00[i]); }
  asm("v_mad_u64_u32 %0, %1, %4, %5, %6\n\tv_mad_u64_u32 %2, %3, %7, %8, %9"
      : "=&v"(hi[0]), "=&s"(d[0]), "=&v"(hi[1]), "=&s"(d[1])
      : "v"(jhi(a[0])), "v"(jhi(b[0])), "v"(acc[0]), "v"(jhi(a[1])), "v"(jhi(b[1])), "v"(acc[1]));
  asm("v_mad_u64_u32 %0, %1, %4, -1, %5\n\tv_mad_u64_u32 %2, %3, %6, -1, %7"
      : "=&v"(t[0]), "=&s"(c1[0]), "=&v"(t[1]), "=&s"(c1[1])
      : "v"(jlo(hi[0])), "v"(lo[0]), "v"(jlo(hi[1])), "v"(lo[1]));
  asm("s_nop 0\n\t" "v_subb_co_u32_e64 %0, %1, %8, %9, %10\n\tv_subb_co_u32_e64 %2, %3, %11, %12, %13\n\tv_addc_co_u32_e64 %4, %5, %14, 0, %15\n\tv_addc_co_u32_e64 %6, %7, %16, 0, %17"
      : "=&v"(rl[0]), "=&s"(bb[0]), "=&v"(rl[1]), "=&s"(bb[1]), "=&v"(rh[0]), "=&s"(d[0]), "=&v"(rh[1]), "=&s"(d[1])
      : "v"(jlo(t[0])), "v"(jhi(hi[0])), "s"(c1[0]), "v"(jlo(t[1])), "v"(jhi(hi[1])), "s"(c1[1]), "v"(jhi(t[0])), "s"(c1[0]), "v"(jhi(t[1])), "s"(c1[1]));
  asm("v_subb_co_u32_e64 %0, %1, %0, 0, %5\n\tv_subb_co_u32_e64 %2, %3, %2, 0, %7"
      : "=&v"(rh[0]), "=&s"(bw[0]), "=&v"(rh[1]), "=&s"(bw[1])
      : "0"(rh[0]), "s"(bb[0]), "2"(rh[1]), "s"(bb[1]));
  asm("s_nop 0\n\t" "v_addc_co_u32_e64 %0, %1, %0, 0, %5\n\tv_addc_co_u32_e64 %2, %3, %2, 0, %7"
      : "=&v"(rl[0]), "=&s"(c3[0]), "=&v"(rl[1]), "=&s"(c3[1])
      : "0"(rl[0]), "s"(bw[0]), "2"(rl[1]), "s"(bw[1]));
  for (int i = 0; i < 2; i++) mk[i] = bw[i] & ~c3[i];  // scalar unit
  asm("v_subb_co_u32_e64 %0, %1, %0, 0, %5\n\tv_subb_co_u32_e64 %2, %3, %2, 0, %7"
      : "=&v"(rh[0]), "=&s"(d[0]), "=&v"(rh[1]), "=&s"(d[1])
      : "0"(rh[0]), "s"(mk[0]), "2"(rh[1]), "s"(mk[1]));
  for (int i = 0; i < 2; i++) r[i] = ((u64)rh[i] << 32) | rl[i];
}
template <> __device__ inline __attribute__((always_inline)) void lz_mulN<3>(u64 (&r)[3], const u64 (&a)[3], const u64 (&b)[3]) {
  u64 p00[3], m[3], hi[3], t[3], lo[3], acc[3], cm[3], k1[3], k2[3], k3[3], c1[3], bb[3], bw[3], c3[3], mk[3], d[6];
  u32 w1[3], accl[3], acch[3], rl[3], rh[3];
  const u32 zero = 0;
  asm("v_mad_u64_u32 %0, %1, %12, %13, 0\n\tv_mad_u64_u32 %2, %3, %14, %15, 0\n\tv_mad_u64_u32 %4, %5, %16, %17, 0\n\tv_mad_u64_u32 %6, %7, %18, %19, 0\n\tv_mad_u64_u32 %8, %9, %20, %21, 0\n\tv_mad_u64_u32 %10, %11, %22, %23, 0"
      : "=&v"(p00[0]), "=&s"(d[0]), "=&v"(m[0]), "=&s"(d[1]), "=&v"(p00[1]), "=&s"(d[2]), "=&v"(m[1]), "=&s"(d[3]), "=&v"(p00[2]), "=&s"(d[4]), "=&v"(m[2]), "=&s"(d[5])
      : "v"(jlo(a[0])), "v"(jlo(b[0])), "v"(jlo(a[0])), "v"(jhi(b[0])), "v"(jlo(a[1])), "v"(jlo(b[1])), "v"(jlo(a[1])), "v"(jhi(b[1])), "v"(jlo(a[2])), "v"(jlo(b[2])), "v"(jlo(a[2])), "v"(jhi(b[2])));
  asm("v_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %2, %3, %9, %10, %2\n\tv_mad_u64_u32 %4, %5, %12, %13, %4"
      : "=&v"(m[0]), "=&s"(cm[0]), "=&v"(m[1]), "=&s"(cm[1]), "=&v"(m[2]), "=&s"(cm[2])
      : "v"(jhi(a[0])), "v"(jlo(b[0])), "0"(m[0]), "v"(jhi(a[1])), "v"(jlo(b[1])), "2"(m[1]), "v"(jhi(a[2])), "v"(jlo(b[2])), "4"(m[2]));
  asm("v_add_co_u32_e64 %0, %1, %6, %7\n\tv_add_co_u32_e64 %2, %3, %8, %9\n\tv_add_co_u32_e64 %4, %5, %10, %11"
      : "=&v"(w1[0]), "=&s"(k1[0]), "=&v"(w1[1]), "=&s"(k1[1]), "=&v"(w1[2]), "=&s"(k1[2])
      : "v"(jhi(p00[0])), "v"(jlo(m[0])), "v"(jhi(p00[1])), "v"(jlo(m[1])), "v"(jhi(p00[2])), "v"(jlo(m[2])));
  asm("v_addc_co_u32_e64 %0, %1, %6, 0, %7\n\tv_addc_co_u32_e64 %2, %3, %8, 0, %9\n\tv_addc_co_u32_e64 %4, %5, %10, 0, %11"
      : "=&v"(accl[0]), "=&s"(k2[0]), "=&v"(accl[1]), "=&s"(k2[1]), "=&v"(accl[2]), "=&s"(k2[2])
      : "v"(jhi(m[0])), "s"(k1[0]), "v"(jhi(m[1])), "s"(k1[1]), "v"(jhi(m[2])), "s"(k1[2]));
  for (int i = 0; i < 3; i++) k3[i] = cm[i] | k2[i];  // scalar unit; cm and k2 exclude each other (a carried m leaves m.hi <= 2^32 - 5)
  asm("v_addc_co_u32_e64 %0, %1, %6, 0, %7\n\tv_addc_co_u32_e64 %2, %3, %8, 0, %9\n\tv_addc_co_u32_e64 %4, %5, %10, 0, %11"
      : "=&v"(acch[0]), "=&s"(d[0]), "=&v"(acch[1]), "=&s"(d[1]), "=&v"(acch[2]), "=&s"(d[2])
      : "v"(zero), "s"(k3[0]), "v"(zero), "s"(k3[1]), "v"(zero), "s"(k3[2]));
  for (int i = 0; i < 3; i++) { acc[i] = ((u64)acch[i] << 32) | accl[i]; lo[i] = ((u64)w1[i] << 32) | jlo(p00[i]); }
  asm("v_mad_u64_u32 %0, %1, %6, %7, %8\n\tv_mad_u64_u32 %2, %3, %9, %10, %11\n\tv_mad_u64_u32 %4, %5, %12, %13, %14"
      : "=&v"(hi[0]), "=&s"(d[0]), "=&v"(hi[1]), "=&s"(d[1]), "=&v"(hi[2]), "=&s"(d[2])
      : "v"(jhi(a[0])), "v"(jhi(b[0])), "v"(acc[0]), "v"(jhi(a[1])), "v"(jhi(b[1])), "v"(acc[1]), "v"(jhi(a[2])), "v"(jhi(b[2])), "v"(acc[2]));
  asm("v_mad_u64_u32 %0, %1, %6, -1, %7\n\tv_mad_u64_u32 %2, %3, %8, -1, %9\n\tv_mad_u64_u32 %4, %5, %10, -1, %11"
      : "=&v"(t[0]), "=&s"(c1[0]), "=&v"(t[1]), "=&s"(c1[1]), "=&v"(t[2]), "=&s"(c1[2])
      : "v"(jlo(hi[0])), "v"(lo[0]), "v"(jlo(hi[1])), "v"(lo[1]), "v"(jlo(hi[2])), "v"(lo[2]));
  asm("v_subb_co_u32_e64 %0, %1, %12, %13, %14\n\tv_subb_co_u32_e64 %2, %3, %15, %16, %17\n\tv_subb_co_u32_e64 %4, %5, %18, %19, %20\n\tv_addc_co_u32_e64 %6, %7, %21, 0, %22\n\tv_addc_co_u32_e64 %8, %9, %23, 0, %24\n\tv_addc_co_u32_e64 %10, %11, %25, 0, %26"
      : "=&v"(rl[0]), "=&s"(bb[0]), "=&v"(rl[1]), "=&s"(bb[1]), "=&v"(rl[2]), "=&s"(bb[2]), "=&v"(rh[0]), "=&s"(d[0]), "=&v"(rh[1]), "=&s"(d[1]), "=&v"(rh[2]), "=&s"(d[2])
      : "v"(jlo(t[0])), "v"(jhi(hi[0])), "s"(c1[0]), "v"(jlo(t[1])), "v"(jhi(hi[1])), "s"(c1[1]), "v"(jlo(t[2])), "v"(jhi(hi[2])), "s"(c1[2]), "v"(jhi(t[0])), "s"(c1[0]), "v"(jhi(t[1])), "s"(c1[1]), "v"(jhi(t[2])), "s"(c1[2]));
  asm("v_subb_co_u32_e64 %0, %1, %0, 0, %7\n\tv_subb_co_u32_e64 %2, %3, %2, 0, %9\n\tv_subb_co_u32_e64 %4, %5, %4, 0, %11"
      : "=&v"(rh[0]), "=&s"(bw[0]), "=&v"(rh[1]), "=&s"(bw[1]), "=&v"(rh[2]), "=&s"(bw[2])
      : "0"(rh[0]), "s"(bb[0]), "2"(rh[1]), "s"(bb[1]), "4"(rh[2]), "s"(bb[2]));
  asm("v_addc_co_u32_e64 %0, %1, %0, 0, %7\n\tv_addc_co_u32_e64 %2, %3, %2, 0, %9\n\tv_addc_co_u32_e64 %4, %5, %4, 0, %11"
      : "=&v"(rl[0]), "=&s"(c3[0]), "=&v"(rl[1]), "=&s"(c3[1]), "=&v"(rl[2]), "=&s"(c3[2])
      : "0"(rl[0]), "s"(bw[0]), "2"(rl[1]), "s"(bw[1]), "4"(rl[2]), "s"(bw[2]));
  for (int i = 0; i < 3; i++) mk[i] = bw[i] & ~c3[i];  // scalar unit
  asm("v_subb_co_u32_e64 %0, %1, %0, 0, %7\n\tv_subb_co_u32_e64 %2, %3, %2, 0, %9\n\tv_subb_co_u32_e64 %4, %5, %4, 0, %11"
      : "=&v"(rh[0]), "=&s"(d[0]), "=&v"(rh[1]), "=&s"(d[1]), "=&v"(rh[2]), "=&s"(d[2])
      : "0"(rh[0]), "s"(mk[0]), "2"(rh[1]), "s"(mk[1]), "4"(rh[2]), "s"(mk[2]));
  for (int i = 0; i < 3; i++) r[i] = ((u64)rh[i] << 32) | rl[i];
}
template <> __device__ inline __attribute__((always_inline)) void lz_mulN<4>(u64 (&r)[4], const u64 (&a)[4], const u64 (&b)[4]) {
  u64 p00[4], m[4], hi[4], t[4], lo[4], acc[4], cm[4], k1[4], k2[4], k3[4], c1[4], bb[4], bw[4], c3[4], mk[4], d[8];
  u32 w1[4], accl[4], acch[4], rl[4], rh[4];
  const u32 zero = 0;
  asm("v_mad_u64_u32 %0, %1, %16, %17, 0\n\tv_mad_u64_u32 %2, %3, %18, %19, 0\n\tv_mad_u64_u32 %4, %5, %20, %21, 0\n\tv_mad_u64_u32 %6, %7, %22, %23, 0\n\tv_mad_u64_u32 %8, %9, %24, %25, 0\n\tv_mad_u64_u32 %10, %11, %26, %27, 0\n\tv_mad_u64_u32 %12, %13, %28, %29, 0\n\tv_mad_u64_u32 %14, %15, %30, %31, 0"
      : "=&v"(p00[0]), "=&s"(d[0]), "=&v"(m[0]), "=&s"(d[1]), "=&v"(p00[1]), "=&s"(d[2]), "=&v"(m[1]), "=&s"(d[3]), "=&v"(p00[2]), "=&s"(d[4]), "=&v"(m[2]), "=&s"(d[5]), "=&v"(p00[3]), "=&s"(d[6]), "=&v"(m[3]), "=&s"(d[7])
      : "v"(jlo(a[0])), "v"(jlo(b[0])), "v"(jlo(a[0])), "v"(jhi(b[0])), "v"(jlo(a[1])), "v"(jlo(b[1])), "v"(jlo(a[1])), "v"(jhi(b[1])), "v"(jlo(a[2])), "v"(jlo(b[2])), "v"(jlo(a[2])), "v"(jhi(b[2])), "v"(jlo(a[3])), "v"(jlo(b[3])), "v"(jlo(a[3])), "v"(jhi(b[3])));
  asm("v_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %2, %3, %11, %12, %2\n\tv_mad_u64_u32 %4, %5, %14, %15, %4\n\tv_mad_u64_u32 %6, %7, %17, %18, %6"
      : "=&v"(m[0]), "=&s"(cm[0]), "=&v"(m[1]), "=&s"(cm[1]), "=&v"(m[2]), "=&s"(cm[2]), "=&v"(m[3]), "=&s"(cm[3])
      : "v"(jhi(a[0])), "v"(jlo(b[0])), "0"(m[0]), "v"(jhi(a[1])), "v"(jlo(b[1])), "2"(m[1]), "v"(jhi(a[2])), "v"(jlo(b[2])), "4"(m[2]), "v"(jhi(a[3])), "v"(jlo(b[3])), "6"(m[3]));
  asm("v_add_co_u32_e64 %0, %1, %8, %9\n\tv_add_co_u32_e64 %2, %3, %10, %11\n\tv_add_co_u32_e64 %4, %5, %12, %13\n\tv_add_co_u32_e64 %6, %7, %14, %15"
      : "=&v"(w1[0]), "=&s"(k1[0]), "=&v"(w1[1]), "=&s"(k1[1]), "=&v"(w1[2]), "=&s"(k1[2]), "=&v"(w1[3]), "=&s"(k1[3])
      : "v"(jhi(p00[0])), "v"(jlo(m[0])), "v"(jhi(p00[1])), "v"(jlo(m[1])), "v"(jhi(p00[2])), "v"(jlo(m[2])), "v"(jhi(p00[3])), "v"(jlo(m[3])));
  asm("v_addc_co_u32_e64 %0, %1, %8, 0, %9\n\tv_addc_co_u32_e64 %2, %3, %10, 0, %11\n\tv_addc_co_u32_e64 %4, %5, %12, 0, %13\n\tv_addc_co_u32_e64 %6, %7, %14, 0, %15"
      : "=&v"(accl[0]), "=&s"(k2[0]), "=&v"(accl[1]), "=&s"(k2[1]), "=&v"(accl[2]), "=&s"(k2[2]), "=&v"(accl[3]), "=&s"(k2[3])
      : "v"(jhi(m[0])), "s"(k1[0]), "v"(jhi(m[1])), "s"(k1[1]), "v"(jhi(m[2])), "s"(k1[2]), "v"(jhi(m[3])), "s"(k1[3]));
  for (int i = 0; i < 4; i++) k3[i] = cm[i] | k2[i];  // scalar unit; cm and k2 exclude each other (a carried m leaves m.hi <= 2^32 - 5)
  asm("v_addc_co_u32_e64 %0, %1, %8, 0, %9\n\tv_addc_co_u32_e64 %2, %3, %10, 0, %11\n\tv_addc_co_u32_e64 %4, %5, %12, 0, %13\n\tv_addc_co_u32_e64 %6, %7, %14, 0, %15"
      : "=&v"(acch[0]), "=&s"(d[0]), "=&v"(acch[1]), "=&s"(d[1]), "=&v"(acch[2]), "=&s"(d[2]), "=&v"(acch[3]), "=&s"(d[3])
      : "v"(zero), "s"(k3[0]), "v"(zero), "s"(k3[1]), "v"(zero), "s"(k3[2]), "v"(zero), "s"(k3[3]));
  for (int i = 0; i < 4; i++) { acc[i] = ((u64)acch[i] << 32) | accl[i]; lo[i] = ((u64)w1[i] << 32) | jlo(p00[i]); }
  asm("v_mad_u64_u32 %0, %1, %8, %9, %10\n\tv_mad_u64_u32 %2, %3, %11, %12, %13\n\tv_mad_u64_u32 %4, %5, %14, %15, %16\n\tv_mad_u64_u32 %6, %7, %17, %18, %19"
      : "=&v"(hi[0]), "=&s"(d[0]), "=&v"(hi[1]), "=&s"(d[1]), "=&v"(hi[2]), "=&s"(d[2]), "=&v"(hi[3]), "=&s"(d[3])
      : "v"(jhi(a[0])), "v"(jhi(b[0])), "v"(acc[0]), "v"(jhi(a[1])), "v"(jhi(b[1])), "v"(acc[1]), "v"(jhi(a[2])), "v"(jhi(b[2])), "v"(acc[2]), "v"(jhi(a[3])), "v"(jhi(b[3])), "v"(acc[3]));
  asm("v_mad_u64_u32 %0, %1, %8, -1, %9\n\tv_mad_u64_u32 %2, %3, %10, -1, %11\n\tv_mad_u64_u32 %4, %5, %12, -1, %13\n\tv_mad_u64_u32 %6, %7, %14, -1, %15"
      : "=&v"(t[0]), "=&s"(c1[0]), "=&v"(t[1]), "=&s"(c1[1]), "=&v"(t[2]), "=&s"(c1[2]), "=&v"(t[3]), "=&s"(c1[3])
      : "v"(jlo(hi[0])), "v"(lo[0]), "v"(jlo(hi[1])), "v"(lo[1]), "v"(jlo(hi[2])), "v"(lo[2]), "v"(jlo(hi[3])), "v"(lo[3]));
  asm("v_subb_co_u32_e64 %0, %1, %16, %17, %18\n\tv_subb_co_u32_e64 %2, %3, %19, %20, %21\n\tv_subb_co_u32_e64 %4, %5, %22, %23, %24\n\tv_subb_co_u32_e64 %6, %7, %25, %26, %27\n\tv_addc_co_u32_e64 %8, %9, %28, 0, %29\n\tv_addc_co_u32_e64 %10, %11, %30, 0, %31\n\tv_addc_co_u32_e64 %12, %13, %32, 0, %33\n\tv_addc_co_u32_e64 %14, %15, %34, 0, %35"
      : "=&v"(rl[0]), "=&s"(bb[0]), "=&v"(rl[1]), "=&s"(bb[1]), "=&v"(rl[2]), "=&s"(bb[2]), "=&v"(rl[3]), "=&s"(bb[3]), "=&v"(rh[0]), "=&s"(d[0]), "=&v"(rh[1]), "=&s"(d[1]), "=&v"(rh[2]), "=&s"(d[2]), "=&v"(rh[3]), "=&s"(d[3])
      : "v"(jlo(t[0])), "v"(jhi(hi[0])), "s"(c1[0]), "v"(jlo(t[1])), "v"(jhi(hi[1])), "s"(c1[1]), "v"(jlo(t[2])), "v"(jhi(hi[2])), "s"(c1[2]), "v"(jlo(t[3])), "v"(jhi(hi[3])), "s"(c1[3]), "v"(jhi(t[0])), "s"(c1[0]), "v"(jhi(t[1])), "s"(c1[1]), "v"(jhi(t[2])), "s"(c1[2]), "v"(jhi(t[3])), "s"(c1[3]));
  asm("v_subb_co_u32_e64 %0, %1, %0, 0, %9\n\tv_subb_co_u32_e64 %2, %3, %2, 0, %11\n\tv_subb_co_u32_e64 %4, %5, %4, 0, %13\n\tv_subb_co_u32_e64 %6, %7, %6, 0, %15"
      : "=&v"(rh[0]), "=&s"(bw[0]), "=&v"(rh[1]), "=&s"(bw[1]), "=&v"(rh[2]), "=&s"(bw[2]), "=&v"(rh[3]), "=&s"(bw[3])
      : "0"(rh[0]), "s"(bb[0]), "2"(rh[1]), "s"(bb[1]), "4"(rh[2]), "s"(bb[2]), "6"(rh[3]), "s"(bb[3]));
  asm("v_addc_co_u32_e64 %0, %1, %0, 0, %9\n\tv_addc_co_u32_e64 %2, %3, %2, 0, %11\n\tv_addc_co_u32_e64 %4, %5, %4, 0, %13\n\tv_addc_co_u32_e64 %6, %7, %6, 0, %15"
      : "=&v"(rl[0]), "=&s"(c3[0]), "=&v"(rl[1]), "=&s"(c3[1]), "=&v"(rl[2]), "=&s"(c3[2]), "=&v"(rl[3]), "=&s"(c3[3])
      : "0"(rl[0]), "s"(bw[0]), "2"(rl[1]), "s"(bw[1]), "4"(rl[2]), "s"(bw[2]), "6"(rl[3]), "s"(bw[3]));
  for (int i = 0; i < 4; i++) mk[i] = bw[i] & ~c3[i];  // scalar unit
  asm("v_subb_co_u32_e64 %0, %1, %0, 0, %9\n\tv_subb_co_u32_e64 %2, %3, %2, 0, %11\n\tv_subb_co_u32_e64 %4, %5, %4, 0, %13\n\tv_subb_co_u32_e64 %6, %7, %6, 0, %15"
      : "=&v"(rh[0]), "=&s"(d[0]), "=&v"(rh[1]), "=&s"(d[1]), "=&v"(rh[2]), "=&s"(d[2]), "=&v"(rh[3]), "=&s"(d[3])
      : "0"(rh[0]), "s"(mk[0]), "2"(rh[1]), "s"(mk[1]), "4"(rh[2]), "s"(mk[2]), "6"(rh[3]), "s"(mk[3]));
  for (int i = 0; i < 4; i++) r[i] = ((u64)rh[i] << 32) | rl[i];
}
#endif
// Extension-field products: the asm product too since round 6 (MH_JIT_ASM_MUL=1).  Round 5 kept them in C (25 VALU per base product
// against 13: 618 of them per point in the core AIR, 30 % of its instructions) because the asm form's SGPR carries spilled into VGPR
// lanes in the EF chunks; with the any-representative arithmetic, the uniform table and the 200-register chunk budget in place that no
// longer happens (0 v_writelane in all ten chunks) and the core AIR's quotient goes 12.98 -> 12.80 ms for 10 % fewer VALU
// instructions -- the chunks are bound by dependent-issue and load latency at 2-3 waves per SIMD, not by the instruction count alone.
// Tried on top and dropped (round 6, profiles/r06_jit_core.txt): the carry chains of a product MERGED into two multi-instruction
// statements (k1 -> k2 -> k3 and bb -> bw -> c3 -> mk through vcc, two SGPR pairs per product instead of eleven outputs): correct in
// isolation (tools/jit_mulcheck: 6.3 M products incl. 48 k on the borrow path), wrong digests / a memory fault inside the chunks --
// the statements contain s_or_b64 / s_andn2_b64, which write SCC, and hipcc keeps SCC alive across an asm statement that does not
// declare it (the carry of its own s_add_u32 / s_addc_u32 address arithmetic) -- and 1-7 % SLOWER once fenced: the scheduler can no
// longer interleave the links of several products.
#if MH_JIT_ASM_MUL == 1
#define lz_mul_ef lz_mul_asm
#else
#define lz_mul_ef lz_mul_c
#endif
FI u64 lz_mul7(u64 a) {  // 7 a: a 67-bit integer, top limb < 7
  const u128 p = (u128)a * 7u;
  const u64 t1 = (u64)(p >> 64) * GL_EPS, r = (u64)p + t1;
  return r < t1 ? r + GL_EPS : r;
}
template <int CA, int CB> FI u64 lz_add(u64 a, u64 b) { return CB ? lz_add_c(a, b) : CA ? lz_add_c(b, a) : lz_add_g(a, b); }
template <int CB> FI u64 lz_sub(u64 a, u64 b) { return CB ? lz_sub_c(a, b) : lz_sub_g(a, b); }
template <int CA> FI u64 lz_neg(u64 a) { return CA ? GL_P - a : lz_sub_g(0, a); }   // p - 0 = p: a representative of 0
template <int CA, int CB> FI e2 lz_e2_add(e2 a, e2 b) { return {lz_add<CA, CB>(a.c0, b.c0), lz_add<CA, CB>(a.c1, b.c1)}; }
template <int CA, int CB> FI e2 lz_e2_sub(e2 a, e2 b) { return {lz_sub<CB>(a.c0, b.c0), lz_sub<CB>(a.c1, b.c1)}; }
template <int CA> FI e2 lz_e2_neg(e2 a) { return {lz_neg<CA>(a.c0), lz_neg<CA>(a.c1)}; }
template <int CA, int CB> FI e2 lz_e2_addf(e2 a, u64 b) { return {lz_add<CA, CB>(a.c0, b), a.c1}; }
template <int CA, int CB> FI e2 lz_e2_subf(e2 a, u64 b) { return {lz_sub<CB>(a.c0, b), a.c1}; }
template <int CA, int CB> FI e2 lz_e2_fsub(u64 a, e2 b) { return {lz_sub<CB>(a, b.c0), lz_neg<CB>(b.c1)}; }
FI e2 lz_e2_mul(e2 a, e2 b) {  // schoolbook: four products, two general additions, one times-seven
  return {lz_add_g(lz_mul_ef(a.c0, b.c0), lz_mul7(lz_mul_ef(a.c1, b.c1))), lz_add_g(lz_mul_ef(a.c0, b.c1), lz_mul_ef(a.c1, b.c0))};
}
FI e2 lz_e2_mulf(e2 a, u64 b) { return {lz_mul_ef(a.c0, b), lz_mul_ef(a.c1, b)}; }
struct JitArgs {
  const u64* main_lde; const u64* aux_lde; const u64* prep_lde; u64* spill; u64* acc; const u64* tw; const u64* coset_tab;
  const u64* inv_first; const u64* inv_last; const u64* periodic; const u64* publics; const u64* randomness;
  const u64* aux_values; const u64* alpha_pows; const u64* uni; const u64* acc_in;
  u64 wh_inv, q0, q_count, spill_stride, beta0, beta1;
  int log_n, log_cosets, log_d, log_dl, jc_shift, log_n_prev;
  u32 t0, periodic_rows;
};
static_assert(sizeof(JitArgs) == 208, "JitArgs layout");
// one kernel for the whole DAG (MH_JIT_FUSE): nothing moves across a region boundary -- neither by the scheduler nor as a value the
// compiler remembers from an earlier load (a cell read again in a later region is LOADED again: its register was given back)
// The branch on an opaque uniform value (never taken: one s_cmp + s_cbranch) makes every region a basic block of its own: the
// compiler's per-block passes (CodeGenPrepare, instruction selection, machine CSE, the scheduler) are quadratic in the block size --
// the core AIR's 50 k instructions as ONE block took 130 s to compile, as eight blocks they take what the eight chunk kernels took.
#define MH_REGION_FENCE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" : "+s"(mh_live) : : "memory"); if (mh_live == 0) return; } while (0)
)SRC";

struct Ev {
  uint32_t node;
  int32_t fold_k;  // -1: compute the node, >= 0: fold constraint k (= this node) into the accumulator
};

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

struct Chunk {
  size_t ev_lo, ev_hi;
  std::string src;
  std::vector<char> code;
  std::string log;
  bool from_cache = false, no_cache = false;
};

// ---- on-disk cache of compiled chunks: an AIR is fixed per application, its kernels are compiled once ----
// $MH_JIT_CACHE_DIR, else $XDG_CACHE_HOME/midenhip, else $HOME/.cache/midenhip; "off" disables it.
std::string cache_dir() {
  const char* d = getenv("MH_JIT_CACHE_DIR");
  std::string dir;
  if (d && *d) {
    if (std::string(d) == "off") return "";
    dir = d;
  } else if (const char* x = getenv("XDG_CACHE_HOME")) {
    dir = std::string(x) + "/midenhip";
  } else if (const char* h = getenv("HOME")) {
    dir = std::string(h) + "/.cache/midenhip";
  } else {
    return "";
  }
  std::string cmd_path;
  for (size_t i = 1; i <= dir.size(); i++)  // mkdir -p
    if (i == dir.size() || dir[i] == '/') {
      cmd_path = dir.substr(0, i);
      (void)mkdir(cmd_path.c_str(), 0700);  // code objects are loaded from here: private to the user
    }
  return dir;
}
std::string cache_key(const std::string& src_only) {
  u64 h1 = 0xcbf29ce484222325ULL, h2 = 0x9ae16a3b2f90404fULL;  // two FNV-1a style streams
  const char* extra = getenv("MH_JIT_FLAGS");
  const std::string src = src_only + (extra ? std::string("\n//flags:") + extra : std::string());
  for (unsigned char ch : src) {
    h1 = (h1 ^ ch) * 0x100000001b3ULL;
    h2 = (h2 ^ ch) * 0x9e3779b97f4a7c15ULL + 0x7f4a7c15ULL;
  }
  int ver_major = 0, ver_minor = 0;
  (void)hiprtcVersion(&ver_major, &ver_minor);
  char buf[96];
  snprintf(buf, sizeof buf, "gfx950-rtc%d.%d-%016llx%016llx.co", ver_major, ver_minor, (unsigned long long)h1, (unsigned long long)h2);
  return buf;
}
// Entry = "MHJC0001" | u64 payload length | u64 FNV-1a of the payload | payload.  A truncated or damaged entry (crash,
// full disk) fails the check and is treated as absent.
u64 cache_sum(const std::vector<char>& code) {
  u64 h = 0xcbf29ce484222325ULL;
  for (unsigned char ch : code) h = (h ^ ch) * 0x100000001b3ULL;
  return h;
}
bool cache_load(const std::string& path, std::vector<char>& code) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  u64 len = 0, sum = 0;
  if (raw.size() < 24 + 64 || memcmp(raw.data(), "MHJC0001", 8) != 0) return false;
  memcpy(&len, raw.data() + 8, 8);
  memcpy(&sum, raw.data() + 16, 8);
  if (len != raw.size() - 24) return false;
  code.assign(raw.begin() + 24, raw.end());
  if (cache_sum(code) != sum) {
    code.clear();
    return false;
  }
  return true;
}
void cache_store(const std::string& path, const std::vector<char>& code) {
  const std::string tmp = path + ".tmp" + std::to_string((unsigned long long)getpid());
  {
    std::ofstream f(tmp, std::ios::binary);
    if (!f) return;
    const u64 len = code.size(), sum = cache_sum(code);
    f.write("MHJC0001", 8);
    f.write((const char*)&len, 8);
    f.write((const char*)&sum, 8);
    f.write(code.data(), (std::streamsize)code.size());
    if (!f) return;
  }
  if (rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
}

// What the compiler did with a chunk, read from its code object (ELF64): the kernel descriptor `mh_jit_chunk.kd` (64 bytes in
// .rodata; llvm AMDGPUUsage "Kernel Descriptor") holds the private-segment (scratch) size at offset 4 and the granulated VGPR count
// in compute_pgm_rsrc1[5:0] (units of 8 on gfx90a+).  No GPU needed: the same answer at build time (mh_jit_precompile) and at load.
// -> false if the object cannot be parsed (then nothing is concluded from it).
bool code_object_info(const std::vector<char>& co, unsigned* scratch_bytes, unsigned* vgprs) {
  auto rd = [&](size_t off, int bytes) -> u64 {
    u64 v = 0;
    if (off + (size_t)bytes > co.size()) return 0;
    memcpy(&v, co.data() + off, (size_t)bytes);
    return v;
  };
  if (co.size() < 64 || memcmp(co.data(), "\177ELF", 4) != 0 || co[4] != 2) return false;
  const u64 shoff = rd(0x28, 8), shentsize = rd(0x3A, 2), shnum = rd(0x3C, 2);
  if (!shoff || shentsize < 64 || shoff + shnum * shentsize > co.size()) return false;
  struct Sec { u64 type, addr, off, size, link, entsize; };
  std::vector<Sec> secs(shnum);
  for (u64 i = 0; i < shnum; i++) {
    const size_t b = shoff + i * shentsize;
    secs[i] = {rd(b + 4, 4), rd(b + 0x10, 8), rd(b + 0x18, 8), rd(b + 0x20, 8), rd(b + 0x28, 4), rd(b + 0x38, 8)};
  }
  for (const Sec& st : secs) {
    if ((st.type != 2 && st.type != 11) || st.entsize < 24 || st.link >= shnum) continue;  // SHT_SYMTAB / SHT_DYNSYM
    const Sec& str = secs[st.link];
    for (u64 k = 0; k < st.size / st.entsize; k++) {
      const size_t b = st.off + k * st.entsize;
      const u64 name = rd(b, 4), value = rd(b + 8, 8);
      if (str.off + name + 16 > co.size() || strncmp(co.data() + str.off + name, "mh_jit_chunk.kd", 16) != 0) continue;
      for (const Sec& sc : secs)
        if (sc.type == 1 && value >= sc.addr && value + 64 <= sc.addr + sc.size) {  // SHT_PROGBITS holding the descriptor
          const size_t kd = sc.off + (value - sc.addr);
          if (kd + 64 > co.size()) return false;
          *scratch_bytes = (unsigned)rd(kd + 4, 4);
          *vgprs = (unsigned)(((rd(kd + 48, 4) & 63) + 1) * 8);
          return true;
        }
    }
  }
  return false;
}

void hiprtc_check(hiprtcResult r, const char* what) {
  if (r != HIPRTC_SUCCESS) throw MhError(MH_ERR_INTERNAL, std::string("hiprtc: ") + what + ": " + hiprtcGetErrorString(r));
}

}  // namespace

struct JitProgram {
  mh_ctx* ctx = nullptr;
  std::vector<hipModule_t> modules;
  std::vector<hipFunction_t> fns;
  hipFunction_t fn_uni = nullptr;  // fills JitArgs::uni (null: the DAG has no uniform gate)
  size_t n_spill = 0;  // u64 slots per point crossing chunk boundaries (HBM planes)
  size_t n_uni = 0;    // u64 entries of the uniform table
  bool fused = false;  // one kernel for the whole DAG (MH_JIT_FUSE): regions instead of chunk kernels, k_quot_finish inside
  size_t n_regions = 0;
};

void jit_program_free(JitProgram* p) {
  if (!p) return;
  for (hipModule_t m : p->modules) (void)hipModuleUnload(m);
  delete p;
}
size_t jit_program_chunks(const JitProgram* p) { return !p ? 0 : p->fused ? p->n_regions : p->fns.size(); }
bool jit_program_fused(const JitProgram* p) { return p && p->fused; }
// Largest VGPR count over the compiled chunks (what bounds their occupancy: 512 / VGPRs waves per SIMD on gfx950); 0 = none / unknown.
int jit_program_max_vgprs(const JitProgram* p) {
  int mx = 0;
  if (p)
    for (hipFunction_t f : p->fns) {
      int v = 0;
      if (hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_NUM_REGS, f) == hipSuccess && v > mx) mx = v;
    }
  return mx;
}

thread_local bool g_jit_compile_only = false;
thread_local int g_jit_last_chunks = 0;

static JitProgram* jit_program_build_cuts(mh_ctx* ctx, const DagIR& ir, const std::vector<size_t>* forced_cuts, int depth, int press_max);
JitProgram* jit_program_build(mh_ctx* ctx, const DagIR& ir) { return jit_program_build_cuts(ctx, ir, nullptr, 0, 0); }
// forced_cuts: the chunk end positions (ascending, last = the number of events) of a retry after a chunk came back from the compiler
// with scratch memory -- see the end of the compile step
// press_max: fused programs only -- the largest number of 64-bit words a region may hold alive at once (0: $MH_JIT_FUSE_PRESS)
static JitProgram* jit_program_build_cuts(mh_ctx* ctx, const DagIR& ir, const std::vector<size_t>* forced_cuts, int depth, int press_max) {
  const int mode = ir.outputs ? 1 : env_int("MH_JIT", -1);  // 0: never, 1: always, default: large DAGs only
  if (mode == 0) return nullptr;
  const std::vector<DagNode>& nodes = ir.nodes;
  // ---- uniform gates: no trace cell, periodic value or selector in their cone (powers of the LogUp challenges, bus prefixes, sums of
  // public values) -- the same for every point.  They are evaluated ONCE per call by the program's uniform kernel into a table
  // (JitArgs::uni) and are leaves for everything below: never computed per point, never spilled.  MH_JIT_UNI=0: per point, as before.
  std::vector<char> uniform(nodes.size(), 0);
  for (size_t i = 0; i < nodes.size(); i++) {
    const DagNode& nd = nodes[i];
    if (nd.op == DOP_CONST || nd.op == DOP_PUBLIC || nd.op == DOP_RANDOMNESS || nd.op == DOP_AUX_VALUE) uniform[i] = 1;
    else if (dag_is_gate(nd.op)) uniform[i] = uniform[nd.a] && (nd.op == DOP_NEG || uniform[nd.b]);
  }
  const bool use_uni = env_int("MH_JIT_UNI", 1) != 0;
  const bool lazy_vals = env_int("MH_JIT_LAZYVAL", 1) != 0;  // any-representative arithmetic between gates (prelude: lz_*)
  std::vector<int32_t> uni_slot(nodes.size(), -1);
  size_t n_uni = 0;
  if (use_uni)
    for (size_t i = 0; i < nodes.size(); i++)
      if (ir.live[i] && dag_is_gate(nodes[i].op) && uniform[i]) {
        uni_slot[i] = (int32_t)n_uni;
        n_uni += nodes[i].ext ? 2 : 1;
      }
  auto interior = [&](uint32_t id) { return dag_is_gate(nodes[id].op) && uni_slot[id] < 0; };
  size_t n_gates = 0;
  for (size_t i = 0; i < nodes.size(); i++) n_gates += ir.live[i] && dag_is_gate(nodes[i].op);
  if (mode != 1 && n_gates < (size_t)env_int("MH_JIT_MIN_GATES", 400)) return nullptr;

  // ---- sums of (uniform EF coefficient) x (base value): LogUp message encodings `prefix + sum_i beta^i f_i` ----
  // The coefficients depend on challenges, constants and public values only -- the same for every point --, so such a sum is the
  // alpha fold's pattern: the coefficient is cut into 22-bit limbs on the scalar unit, the value into 32-bit halves, 6 v_mad_u64_u32
  // per (term, component) into weight accumulators and ONE reduction per sum, instead of two modular products and two modular
  // additions per term (72 VALU instructions -> 16, + ~130 per sum).  An EF ADD tree (through single-use ADD nodes) with at least
  // $MH_JIT_DOT (default 3, 0 = off) such terms becomes one "dot" gate: its operands are the terms' (coefficient, value) pairs and the
  // other addends; the ADD and MUL nodes inside are absorbed (never emitted).  Exact: the same field value, canonical on exit.
  std::vector<char> absorbed(nodes.size(), 0);
  struct Dot {
    std::vector<std::pair<uint32_t, uint32_t>> terms;  // (uniform EF coefficient, base value)
    std::vector<uint32_t> others;
  };
  std::vector<int32_t> dot_of(nodes.size(), -1);
  std::vector<Dot> dots;
  {
    const int min_terms = env_int("MH_JIT_DOT", 3);
    std::vector<uint32_t> uses(nodes.size(), 0);
    for (size_t i = 0; i < nodes.size(); i++)
      if (ir.live[i] && interior((uint32_t)i)) {
        uses[nodes[i].a]++;
        if (nodes[i].op != DOP_NEG) uses[nodes[i].b]++;
      }
    for (uint32_t c : ir.cons) uses[c]++;
    auto is_term = [&](uint32_t id, uint32_t& u, uint32_t& v) {
      const DagNode& nd = nodes[id];
      if (nd.op != DOP_MUL || uses[id] != 1) return false;
      if (nodes[nd.a].ext && uniform[nd.a] && !nodes[nd.b].ext) { u = nd.a; v = nd.b; return true; }
      if (nodes[nd.b].ext && uniform[nd.b] && !nodes[nd.a].ext) { u = nd.b; v = nd.a; return true; }
      return false;
    };
    if (min_terms > 0 && !ir.outputs)
      for (size_t h = nodes.size(); h-- > 0;) {
        if (!ir.live[h] || absorbed[h] || nodes[h].op != DOP_ADD || !nodes[h].ext || uniform[h]) continue;
        Dot d;
        std::vector<uint32_t> inner, st{(uint32_t)h};
        while (!st.empty()) {
          const uint32_t x = st.back();
          st.pop_back();
          for (uint32_t c : {nodes[x].a, nodes[x].b}) {
            uint32_t u, v;
            if (nodes[c].op == DOP_ADD && nodes[c].ext && uses[c] == 1 && !uniform[c]) { inner.push_back(c); st.push_back(c); }
            else if (is_term(c, u, v)) { inner.push_back(c); d.terms.push_back({u, v}); }
            else d.others.push_back(c);
          }
        }
        if ((int)d.terms.size() < min_terms || d.terms.size() > 400) continue;  // <= 2^9 products of < 2^54 per accumulator
        for (uint32_t x : inner) absorbed[x] = 1;
        dot_of[h] = (int32_t)dots.size();
        dots.push_back(std::move(d));
      }
  }
  // the operands of a gate as the emitter sees them
  auto ops = [&](uint32_t id, std::vector<uint32_t>& out) {
    out.clear();
    if (dot_of[id] >= 0) {
      const Dot& d = dots[dot_of[id]];
      for (uint32_t o : d.others) out.push_back(o);
      for (auto& t : d.terms) { out.push_back(t.first); out.push_back(t.second); }
      return;
    }
    out.push_back(nodes[id].a);
    if (nodes[id].op != DOP_NEG) out.push_back(nodes[id].b);
  };
  // ---- emission order: depth-first from each constraint in turn (a value is computed when first needed,
  // which keeps the cut sets between chunks small), the fold right after the constraint's node ----
  std::vector<Ev> seq;
  {
    std::vector<char> done(nodes.size(), 0);
    std::vector<std::pair<uint32_t, size_t>> st;
    std::vector<uint32_t> o;
    for (size_t k = 0; k < ir.cons.size(); k++) {
      const uint32_t root = ir.cons[k];
      if (interior(root) && !done[root]) st.push_back({root, 0});
      while (!st.empty()) {
        const uint32_t id = st.back().first;
        if (done[id]) { st.pop_back(); continue; }
        ops(id, o);
        bool pushed = false;
        while (st.back().second < o.size()) {
          const uint32_t c = o[st.back().second++];
          if (interior(c) && !done[c]) { st.push_back({c, 0}); pushed = true; break; }
        }
        if (pushed) continue;
        done[id] = 1;
        seq.push_back({id, -1});
        st.pop_back();
      }
      seq.push_back({root, (int32_t)k});
    }
  }
  // ---- cut into chunks of roughly equal multiplication count ----
  auto cost = [&](const Ev& e) -> int {
    const DagNode& nd = nodes[e.node];
    if (e.fold_k >= 0) return nd.ext ? 3 : 2;
    if (dot_of[e.node] >= 0) return 4 + (int)dots[dot_of[e.node]].terms.size();
    if (nd.op != DOP_MUL) return 1;
    const bool ea = nodes[nd.a].ext, eb = nodes[nd.b].ext;
    return ea && eb ? 4 : (ea || eb ? 2 : 1);
  };
  const int budget = std::max(16, env_int("MH_JIT_CHUNK", 320));
  std::vector<Chunk> chunks;
  std::vector<long> crossing(seq.size() + 1, 0);
  {
    // crossing[i] = u64 words alive across a cut after seq[i]: computed at or before i, used after i.  A cut is placed where this is
    // smallest inside a window around the budget ($MH_JIT_CUTWIN percent, 0 = cut at the budget): every crossing value is either
    // recomputed or goes through a spill plane (16 bytes of traffic per point and word).
    {
      std::vector<long> def_pos(nodes.size(), -1), last_use(nodes.size(), -1);
      std::vector<uint32_t> o;
      for (size_t i = 0; i < seq.size(); i++) {
        const uint32_t id = seq[i].node;
        if (seq[i].fold_k >= 0) { if (interior(id)) last_use[id] = (long)i; continue; }
        def_pos[id] = (long)i;
        ops(id, o);
        for (uint32_t c : o)
          if (interior(c)) last_use[c] = (long)i;
      }
      std::vector<long> diff(seq.size() + 2, 0);
      for (size_t id = 0; id < nodes.size(); id++)
        if (def_pos[id] >= 0 && last_use[id] > def_pos[id]) {
          const long w = nodes[id].ext ? 2 : 1;
          diff[def_pos[id]] += w;       // alive across cuts after positions def .. last_use - 1
          diff[last_use[id]] -= w;
        }
      long run = 0;
      for (size_t i = 0; i < seq.size(); i++) { run += diff[i]; crossing[i] = run; }
    }
    const int win = std::max(0, env_int("MH_JIT_CUTWIN", 25));
    // $MH_JIT_CUTK = K > 0: chunk sizes are free between 0.4 and 1.6 budgets and the programme minimises
    //   sum over cuts (crossing words + P)  +  K * sum over chunks (cost / budget)^2
    // -- a constraint system whose values rarely cross (Poseidon2 permutation rows, chiplet sections) is cut into MORE, smaller
    // kernels (fewer VGPRs, more waves per SIMD, and the cuts are free), one with 40-100 words alive everywhere (the core AIR's
    // operation flags) keeps large chunks.  K = 0: the fixed window above.
    const long cut_k = std::max(0, env_int("MH_JIT_CUTK", 0)), cut_p = std::max(0, env_int("MH_JIT_CUTP", 10));
    const long lo_b = cut_k ? (long)budget * 2 / 5 : (long)budget * (100 - win) / 100,
               hi_b = cut_k ? (long)budget * 8 / 5 : (long)budget * (100 + win) / 100;
    // dynamic programme over the cut positions: best[i] = smallest sum of crossings with a cut after seq[i - 1], every chunk's cost
    // inside [lo_b, hi_b] (the last one from lo_b / 2); fewest crossings overall, not greedily chunk by chunk
    const size_t S = seq.size();
    std::vector<long> pre(S + 1, 0);
    for (size_t i = 0; i < S; i++) pre[i + 1] = pre[i] + cost(seq[i]);
    const long INF = 1L << 60;
    std::vector<long> best(S + 1, INF);
    std::vector<size_t> from(S + 1, 0);
    best[0] = 0;
    for (size_t i = 1; i <= S; i++) {
      const bool fold_next = i < S && seq[i].fold_k >= 0;  // never separate a node from the fold that consumes it
      if (fold_next) continue;
      for (size_t j = i; j-- > 0;) {
        const long c = pre[i] - pre[j];
        if (c > hi_b && j + 1 < i) break;   // a single over-budget event still forms a chunk
        if (best[j] >= INF) continue;
        if (c < (i != S ? lo_b : lo_b / 2) && !(i == S && j == 0)) continue;  // the last chunk may be half a window short, not a stub
        long v = best[j] + (i < S ? crossing[i - 1] + (cut_k ? cut_p : 0) : 0);
        if (cut_k) v += cut_k * c * c / ((long)budget * budget);
        if (v < best[i]) { best[i] = v; from[i] = j; }
      }
    }
    if (forced_cuts) {
      size_t lo = 0;
      for (size_t e : *forced_cuts) { chunks.push_back({lo, e, {}, {}, {}}); lo = e; }
    } else if (best[S] >= INF) {  // no partition inside the window (degenerate budgets): cut at the budget
      size_t lo = 0;
      long acc = 0;
      for (size_t i = 0; i < S; i++) {
        acc += cost(seq[i]);
        const bool fold_next = i + 1 < S && seq[i + 1].fold_k >= 0;
        if (acc >= budget && !fold_next) { chunks.push_back({lo, i + 1, {}, {}, {}}); lo = i + 1; acc = 0; }
      }
      if (lo < S) chunks.push_back({lo, S, {}, {}, {}});
    } else {
      std::vector<size_t> cuts;
      for (size_t i = S; i > 0; i = from[i]) cuts.push_back(i);
      size_t lo = 0;
      for (size_t k = cuts.size(); k-- > 0;) { chunks.push_back({lo, cuts[k], {}, {}, {}}); lo = cuts[k]; }
    }
    if (env_int("MH_JIT_STATS", 0))
      for (const Chunk& ch : chunks) {
        long peak = 0;
        for (size_t i = ch.ev_lo; i < ch.ev_hi; i++) peak = std::max(peak, crossing[i]);
        fprintf(stderr, "[mh jit]   chunk [%zu, %zu): cost %ld, peak live words %ld, crossing at its end %ld\n", ch.ev_lo, ch.ev_hi,
                pre[ch.ev_hi] - pre[ch.ev_lo], peak, ch.ev_hi < S ? crossing[ch.ev_hi - 1] : 0L);
      }
  }
  const size_t n_chunks = chunks.size();
  // ---- what each chunk evaluates, and which values cross chunk boundaries ----
  // A value defined in an earlier chunk is either RECOMPUTED in the chunk that needs it (its cone down to cells / constants / values
  // this chunk already holds is cheap: operation flags, small sums -- a spilled value costs an 8-byte store and an 8-byte load per
  // point and reader, the constraint kernels of the real AIRs are bound by exactly that traffic) or LOADED from an HBM spill plane
  // (then, and only then, its defining chunk stores it).  $MH_JIT_RECOMP = the largest cone, in estimated VALU instructions, that is
  // recomputed (0: spill everything, the former behaviour).
  std::vector<int32_t> def_chunk(nodes.size(), -1);
  for (size_t ci = 0; ci < n_chunks; ci++)
    for (size_t i = chunks[ci].ev_lo; i < chunks[ci].ev_hi; i++)
      if (seq[i].fold_k < 0) def_chunk[seq[i].node] = (int32_t)ci;
  const int recomp_max = env_int("MH_JIT_RECOMP", 250);  // round-5 sweep with the lazy generator, core AIR: 160 / 200 / 250 / 300 / 400 -> 15.6 / 15.6 / 14.5 / 15.1 / 15.3 ms
  auto valu_cost = [&](uint32_t id) -> int {  // rough VALU instructions of one gate
    const DagNode& nd = nodes[id];
    if (dot_of[id] >= 0) return 130 + 16 * (int)dots[dot_of[id]].terms.size() + 12 * (int)dots[dot_of[id]].others.size();
    const bool ea = nodes[nd.a].ext, eb = nd.op != DOP_NEG && nodes[nd.b].ext;
    if (nd.op == DOP_MUL) return ea && eb ? 80 : (ea || eb ? 44 : 22);
    return (ea || eb) ? 12 : 6;
  };
  struct Item {
    uint32_t node;
    int32_t fold_k;  // -1: compute the node, >= 0: fold / output k, -2: load the node from its spill plane
  };
  std::vector<std::vector<Item>> items(n_chunks);
  std::vector<char> spilled(nodes.size(), 0);
  std::vector<int32_t> last_load(nodes.size(), -1);
  size_t n_recomputed = 0;
  {
    std::vector<int32_t> mat(nodes.size(), -1);   // chunk in which the value was last materialised
    std::vector<int32_t> seen(nodes.size(), -1);  // scratch of the cone walks (stamp)
    int32_t stamp = 0;
    std::vector<uint32_t> cone;
    for (size_t ci = 0; ci < n_chunks; ci++) {
      auto have = [&](uint32_t u) { return !interior(u) || mat[u] == (int32_t)ci; };
      auto load = [&](uint32_t u) {
        spilled[u] = 1;
        last_load[u] = (int32_t)ci;
        mat[u] = (int32_t)ci;
        items[ci].push_back({u, -2});
      };
      // make `u` (defined in an earlier chunk) available here
      auto ensure = [&](uint32_t u) {
        if (have(u)) return;
        if (spilled[u] || recomp_max <= 0) { load(u); return; }
        // the cone of u down to what this chunk holds or can load: post-order, cost
        stamp++;
        cone.clear();
        long cst = 0;
        std::vector<std::pair<uint32_t, size_t>> dfs;
        std::vector<uint32_t> o;
        dfs.push_back({u, 0});
        while (!dfs.empty()) {
          const uint32_t id = dfs.back().first;
          if (seen[id] == stamp) { dfs.pop_back(); continue; }
          ops(id, o);
          bool pushed = false;
          while (dfs.back().second < o.size()) {
            const uint32_t c = o[dfs.back().second++];
            if (interior(c) && mat[c] != (int32_t)ci && !spilled[c] && seen[c] != stamp) { dfs.push_back({c, 0}); pushed = true; break; }
          }
          if (pushed) continue;
          seen[id] = stamp;
          cone.push_back(id);
          cst += valu_cost(id);
          dfs.pop_back();
          if (cst > recomp_max) break;
        }
        if (cst > recomp_max) { load(u); return; }
        for (uint32_t id : cone) {
          ops(id, o);
          for (uint32_t c : o)
            if (interior(c) && mat[c] != (int32_t)ci) load(c);  // a spilled value inside the cone
          mat[id] = (int32_t)ci;
          items[ci].push_back({id, -1});
          n_recomputed++;
        }
      };
      std::vector<uint32_t> opv;
      for (size_t i = chunks[ci].ev_lo; i < chunks[ci].ev_hi; i++) {
        const uint32_t id = seq[i].node;
        if (seq[i].fold_k >= 0) {
          ensure(id);
          items[ci].push_back({id, seq[i].fold_k});
          continue;
        }
        ops(id, opv);
        for (uint32_t c : opv) ensure(c);
        mat[id] = (int32_t)ci;
        items[ci].push_back({id, -1});
      }
    }
  }
  // ---- fused programs: a region holds at most `press_max` words alive ----
  // The register file gives a lane 128 64-bit words at two waves per SIMD; the limb accumulators of the fold, the selectors and the
  // addresses take ~25 of them for the whole kernel, a product in flight ~8.  What a region keeps alive is known here exactly: walk
  // its items, a value (computed, recomputed, loaded from its slot, or a cell read where first used) lives from its definition to its
  // last use INSIDE the region.  A region above the bound is cut in two where the fewest words cross, and everything is planned
  // again -- without compiling: the compiler is asked once, and only if it still reports scratch memory the bound is lowered.
  const bool fuse_mode = !ir.outputs && env_int("MH_JIT_FUSE", 0) != 0;
  if (fuse_mode && !press_max) press_max = std::max(16, env_int("MH_JIT_FUSE_PRESS", 84));
  if (fuse_mode) {
    std::vector<long> press(n_chunks, 0);
    std::vector<uint32_t> o;
    for (size_t ci = 0; ci < n_chunks; ci++) {
      std::map<uint64_t, std::pair<long, long>> span;  // key -> (definition, last use) in item positions; key = node id (cells share their node)
      auto use = [&](uint32_t id, long i) {
        const DagNode& nd = nodes[id];
        const bool cell = nd.op == DOP_MAIN || nd.op == DOP_AUX || nd.op == DOP_PREP || nd.op == DOP_PERIODIC;
        if (!cell && !interior(id)) return;
        auto it = span.find(id);
        if (it == span.end()) span[id] = {i, i};  // a cell: read where first used
        else it->second.second = i;
      };
      long i = 0;
      for (const Item& it : items[ci]) {
        if (it.fold_k >= 0) use(it.node, i);
        else if (it.fold_k == -2) span[it.node] = {i, i};
        else {
          ops(it.node, o);
          for (uint32_t c : o) use(c, i);
          span[it.node] = {i, i};
        }
        i++;
      }
      std::vector<long> diff((size_t)i + 2, 0);
      for (auto& kv : span) {
        const long w = nodes[kv.first].ext ? 2 : 1;
        diff[kv.second.first] += w;
        diff[kv.second.second + 1] -= w;
      }
      long run = 0;
      for (long k = 0; k <= i; k++) { run += diff[k]; press[ci] = std::max(press[ci], run); }
    }
    if (env_int("MH_JIT_STATS", 0)) {
      fprintf(stderr, "[mh jit] fused: words alive per region (bound %d):", press_max);
      for (long v : press) fprintf(stderr, " %ld", v);
      fprintf(stderr, "\n");
    }
    if (depth < 12 && env_int("MH_JIT_SPLIT", 1)) {
      std::vector<size_t> cuts;
      bool any = false;
      for (size_t ci = 0; ci < n_chunks; ci++) {
        const size_t lo = chunks[ci].ev_lo, hi = chunks[ci].ev_hi;
        if (press[ci] > press_max && hi - lo >= 8) {
          size_t best_m = 0;
          long best_x = -1;
          for (size_t m = lo + (hi - lo) / 3; m <= lo + 2 * (hi - lo) / 3; m++) {
            if (m <= lo || m >= hi || seq[m].fold_k >= 0) continue;
            if (best_x < 0 || crossing[m - 1] < best_x) { best_x = crossing[m - 1]; best_m = m; }
          }
          if (best_m) { cuts.push_back(best_m); any = true; }
        }
        cuts.push_back(hi);
      }
      if (any) return jit_program_build_cuts(ctx, ir, &cuts, depth + 1, press_max);
    }
  }
  // spill planes (one u64 plane per base value, two per EF value), reused once the last reader has run
  std::vector<int32_t> slot(nodes.size(), -1);
  size_t n_spill = 0;
  {
    std::vector<uint32_t> free_slots;
    std::vector<std::vector<uint32_t>> dies(n_chunks);
    for (size_t ci = 0; ci < n_chunks; ci++) {
      for (size_t i = chunks[ci].ev_lo; i < chunks[ci].ev_hi; i++) {
        if (seq[i].fold_k >= 0) continue;
        const uint32_t id = seq[i].node;
        if (!spilled[id]) continue;
        const int need = nodes[id].ext ? 2 : 1;
        uint32_t s;
        if (need == 1 && !free_slots.empty()) {
          s = free_slots.back();
          free_slots.pop_back();
        } else {
          s = (uint32_t)n_spill;
          n_spill += need;
        }
        slot[id] = (int32_t)s;
        dies[last_load[id]].push_back(id);
      }
      for (uint32_t id : dies[ci]) {
        free_slots.push_back((uint32_t)slot[id]);
        if (nodes[id].ext) free_slots.push_back((uint32_t)slot[id] + 1);
      }
    }
  }
  if (env_int("MH_JIT_STATS", 0)) {
    size_t n_loads = 0, n_stores = 0;
    for (size_t ci = 0; ci < n_chunks; ci++)
      for (const Item& it : items[ci]) n_loads += it.fold_k == -2 ? (nodes[it.node].ext ? 2 : 1) : 0;
    for (size_t i = 0; i < nodes.size(); i++) n_stores += spilled[i] ? (nodes[i].ext ? 2 : 1) : 0;
    size_t n_terms = 0;
    for (const Dot& d : dots) n_terms += d.terms.size();
    fprintf(stderr, "[mh jit] %zu chunks, %zu gates, %zu recomputed, spill planes %zu, per point: %zu spill loads, %zu spill stores; %zu dot gates with %zu terms\n",
            n_chunks, seq.size(), n_recomputed, n_spill, n_loads, n_stores, dots.size(), n_terms);
  }
  // ---- which values are canonical (< p) by construction: every leaf and table entry; with lazy arithmetic no gate result is,
  // except a dot gate without further addends (fold_value ends canonical) ----
  std::vector<char> canon(nodes.size(), 1);
  if (lazy_vals)
    for (size_t i = 0; i < nodes.size(); i++)
      if (interior((uint32_t)i)) canon[i] = dot_of[i] >= 0 && dots[dot_of[i]].others.empty();
  // ---- chains of one-sided lazy additions (`lz_add_c` / `lz_sub_c`: s = a + c, s < a ? s + eps : s) ----
  // Each link is a carry-select whose two arms depend on the previous link; LLVM's DAG combiner walks both arms of every select when it
  // reasons about the next one, without memoising: a chain of 23 links (the running sum over the 24 aux columns of the Keccak sponge)
  // took 370 s in "DAG Combining 2" of ONE chunk, every other chunk of the same AIR 1-3 s.  A value that ends a run of MH_JIT_LZCHAIN
  // links (default 16) goes through an empty asm statement: the combiner sees an opaque value and the walk starts over.  No
  // instruction is emitted for it; AIRs whose chains are shorter get the same source as before.
  const int lz_chain_max = env_int("MH_JIT_LZCHAIN", 16);
  std::vector<uint16_t> lz_chain(nodes.size(), 0);
  // ---- ONE kernel for the whole DAG (MH_JIT_FUSE, constraint programs): the chunks become REGIONS of one straight-line kernel ----
  // What the separate chunk kernels paid for being separate: every value crossing a cut went through an HBM plane (24 + 24 words per
  // point in the core AIR), the partial alpha-folds through a read-modify-write of the quotient buffer per chunk (+ a reduction of the
  // limb accumulators per chunk), every chunk re-read its cells from HBM (the core AIR's 51 columns are read 283 times by 8 chunks),
  // and every launch ended in a tail.  Fused: a workgroup = 256 consecutive rows of one coset; crossing values live in LDS slots
  // `L[slot][tid]`; the limb accumulators stay in registers from the first constraint to the last and k_quot_finish is applied in the
  // same kernel (ONE 16-byte store per point); the columns most regions read are staged once per workgroup in LDS tiles
  // `T[col][257]` -- the next-row cell of lane i is the current-row cell of lane i + 1, entry 256 is the first row after the tile --
  // and a region fence (MH_REGION_FENCE) keeps the compiler from holding anything across a boundary that the plan gave back.
  const bool fuse = fuse_mode;
  const size_t n_regions = n_chunks;
  std::vector<size_t> region_ends;
  for (const Chunk& ch : chunks) region_ends.push_back(ch.ev_hi);
  size_t lds_words = 0, n_spill_hbm = 0;
  std::vector<int32_t> tile_main, tile_aux, tile_prep;  // LDS word offset of a staged column (aux: per base plane), -1: read from HBM
  std::vector<int32_t> slot_lds, slot_hbm;              // per spill slot: its LDS word offset, or its HBM plane
  if (fuse) {
    size_t w_main = 0, w_aux = 0, w_prep = 0;
    for (const DagNode& nd : nodes) {
      if (nd.op == DOP_MAIN) w_main = std::max<size_t>(w_main, nd.a + 1);
      if (nd.op == DOP_AUX) w_aux = std::max<size_t>(w_aux, nd.a + 1);
      if (nd.op == DOP_PREP) w_prep = std::max<size_t>(w_prep, nd.a + 1);
    }
    tile_main.assign(w_main, -1); tile_aux.assign(2 * w_aux, -1); tile_prep.assign(w_prep, -1);
    const size_t budget_words = (size_t)std::max(0, env_int("MH_JIT_LDS_KB", 64)) * 1024 / 8;  // per workgroup of 256 lanes
    // Candidates for the LDS, by the HBM accesses (8 bytes per point each) they save: a spill slot saves the store and the loads of
    // every value that passes through it (2-3 for most), a column read by k regions saves k - 1 reads (up to 7 in the core AIR).
    struct Cand { int kind; uint32_t idx; long saves; size_t words; };  // kind 0 / 1 / 2: main / aux / preprocessed column, 3: spill slot
    std::vector<Cand> cands;
    {
      std::vector<std::set<size_t>> um(w_main), ua(w_aux), up(w_prep);
      std::vector<long> slot_saves(n_spill, 0);
      std::vector<uint32_t> o;
      auto leaf = [&](uint32_t id, size_t ci) {
        const DagNode& nd = nodes[id];
        if (nd.op == DOP_MAIN) um[nd.a].insert(ci);
        else if (nd.op == DOP_AUX) ua[nd.a].insert(ci);
        else if (nd.op == DOP_PREP) up[nd.a].insert(ci);
      };
      for (size_t ci = 0; ci < n_chunks; ci++)
        for (const Item& it : items[ci]) {
          if (it.fold_k == -2) {
            for (int k = 0; k < (nodes[it.node].ext ? 2 : 1); k++) slot_saves[slot[it.node] + k]++;
            continue;
          }
          if (it.fold_k >= 0) { leaf(it.node, ci); continue; }
          ops(it.node, o);
          for (uint32_t c : o) leaf(c, ci);
        }
      for (size_t id = 0; id < nodes.size(); id++)
        if (spilled[id])
          for (int k = 0; k < (nodes[id].ext ? 2 : 1); k++) slot_saves[slot[id] + k]++;
      for (size_t i = 0; i < w_main; i++) if (um[i].size() >= 2) cands.push_back({0, (uint32_t)i, (long)um[i].size() - 1, 257});
      for (size_t i = 0; i < w_aux; i++) if (ua[i].size() >= 2) cands.push_back({1, (uint32_t)i, 2 * ((long)ua[i].size() - 1), 514});
      for (size_t i = 0; i < w_prep; i++) if (up[i].size() >= 2) cands.push_back({2, (uint32_t)i, (long)up[i].size() - 1, 257});
      for (size_t i = 0; i < n_spill; i++) cands.push_back({3, (uint32_t)i, slot_saves[i], 256});
    }
    std::stable_sort(cands.begin(), cands.end(), [](const Cand& x, const Cand& y) { return x.saves * (long)y.words > y.saves * (long)x.words; });
    slot_lds.assign(n_spill, -1);
    long saved = 0, all = 0;
    size_t n_cols = 0, n_slots_lds = 0;
    for (const Cand& cd : cands) {
      all += cd.saves;
      if (lds_words + cd.words > budget_words) continue;
      if (cd.kind == 0) tile_main[cd.idx] = (int32_t)lds_words;
      else if (cd.kind == 2) tile_prep[cd.idx] = (int32_t)lds_words;
      else if (cd.kind == 1) { tile_aux[2 * cd.idx] = (int32_t)lds_words; tile_aux[2 * cd.idx + 1] = (int32_t)lds_words + 257; }
      else slot_lds[cd.idx] = (int32_t)lds_words;
      (cd.kind == 3 ? n_slots_lds : n_cols)++;
      lds_words += cd.words;
      saved += cd.saves;
    }
    slot_hbm.assign(n_spill, -1);
    for (size_t i = 0; i < n_spill; i++)
      if (slot_lds[i] < 0) slot_hbm[i] = (int32_t)n_spill_hbm++;
    if (env_int("MH_JIT_STATS", 0))
      fprintf(stderr, "[mh jit] fused: %zu regions, %zu of %zu spill slots and %zu columns in LDS (%zu KB per workgroup): %ld of %ld avoidable 8-byte HBM accesses per point avoided\n",
              n_regions, n_slots_lds, n_spill, n_cols, lds_words * 8 / 1024, saved, all);
  }
  // a spilled word: its LDS slot `L[off + tid]` (fused, as planned above) or its HBM plane
  auto spill_ref = [&](int32_t s) -> std::string {
    char b[96];
    if (fuse && slot_lds[s] >= 0) snprintf(b, sizeof b, "L[%d + tid]", slot_lds[s]);
    else snprintf(b, sizeof b, "a.spill[%dull * a.spill_stride + qb]", fuse ? slot_hbm[s] : s);
    return b;
  };
  std::ostringstream fused_regions;
  bool fuse_need_x = false, fuse_need_fl = false, fuse_any_fold = false;
  const bool fuse_fold_persist = env_int("MH_JIT_FUSE_FOLDREG", 1) != 0;
  int fold_terms = 0;  // folds since the limb accumulators were last reduced (per chunk; fused: across regions)
  // ---- source per chunk ----
  const bool lazy_loads = env_int("MH_JIT_LAZY", 1) != 0;
  char buf[256];
  for (size_t ci = 0; ci < n_chunks; ci++) {
    Chunk& ch = chunks[ci];
    std::ostringstream decl, body;
    // cells, periodic values and spilled values are read where they are first used (the live ranges start there: ~90 VGPRs and five
    // waves per SIMD instead of ~230 and two, for the chunks of the core AIR), or all at the top of the kernel (MH_JIT_LAZY=0)
    std::ostringstream& ld = lazy_loads ? body : decl;
    std::set<std::string> declared;
    bool need_x = false, need_fl = false;
    auto ref = [&](uint32_t id) -> std::string {
      const DagNode& nd = nodes[id];
      std::string name;
      switch (nd.op) {
        case DOP_CONST:
          snprintf(buf, sizeof buf, "0x%llxULL", (unsigned long long)nd.c);
          return buf;
        case DOP_MAIN:
          snprintf(buf, sizeof buf, "m%u_%u", nd.a, nd.b);
          name = buf;
          if (declared.insert(name).second) {
            if (fuse && tile_main[nd.a] >= 0)
              snprintf(buf, sizeof buf, "  const u64 %s = L[%d + tid];\n", name.c_str(), tile_main[nd.a] + (nd.b ? 1 : 0));
            else
              snprintf(buf, sizeof buf, "  const u64 %s = a.main_lde[((%uull * B + jc) << a.log_n) + %s];\n", name.c_str(), nd.a,
                       nd.b ? "rn" : "r");
            ld << buf;
          }
          return name;
        case DOP_AUX:
          snprintf(buf, sizeof buf, "x%u_%u", nd.a, nd.b);
          name = buf;
          if (declared.insert(name).second) {
            const char* rr = nd.b ? "rn" : "r";
            if (fuse && tile_aux[2 * nd.a] >= 0) {
              snprintf(buf, sizeof buf, "  const e2 %s = {L[%d + tid], L[%d + tid]};\n", name.c_str(), tile_aux[2 * nd.a] + (nd.b ? 1 : 0),
                       tile_aux[2 * nd.a + 1] + (nd.b ? 1 : 0));
              ld << buf;
            } else {
              snprintf(buf, sizeof buf, "  const e2 %s = {a.aux_lde[((%uull * B + jc) << a.log_n) + %s], ", name.c_str(), 2 * nd.a, rr);
              ld << buf;
              snprintf(buf, sizeof buf, "a.aux_lde[((%uull * B + jc) << a.log_n) + %s]};\n", 2 * nd.a + 1, rr);
              ld << buf;
            }
          }
          return name;
        case DOP_PREP:
          snprintf(buf, sizeof buf, "p%u_%u", nd.a, nd.b);
          name = buf;
          if (declared.insert(name).second) {
            if (fuse && tile_prep[nd.a] >= 0)
              snprintf(buf, sizeof buf, "  const u64 %s = L[%d + tid];\n", name.c_str(), tile_prep[nd.a] + (nd.b ? 1 : 0));
            else
              snprintf(buf, sizeof buf, "  const u64 %s = a.prep_lde[((%uull * B + jc) << a.log_n) + %s];\n", name.c_str(), nd.a,
                       nd.b ? "rn" : "r");
            ld << buf;
          }
          return name;
        case DOP_PUBLIC:
          snprintf(buf, sizeof buf, "a.publics[%u]", nd.a);
          return buf;
        case DOP_PERIODIC:
          snprintf(buf, sizeof buf, "per%u", nd.a);
          name = buf;
          if (declared.insert(name).second) {
            snprintf(buf, sizeof buf, "  const u64 %s = a.periodic[%uull * a.periodic_rows + ((r * D + a.t0 + t) %% a.periodic_rows)];\n",
                     name.c_str(), nd.a);
            ld << buf;
          }
          return name;
        case DOP_IS_FIRST: need_fl = true; return "sel_first";
        case DOP_IS_LAST: need_fl = true; return "sel_last";
        case DOP_IS_TRANSITION: need_x = true; return "sel_trans";
        case DOP_RANDOMNESS:
          snprintf(buf, sizeof buf, "e2{a.randomness[%u], a.randomness[%u]}", 2 * nd.a, 2 * nd.a + 1);
          return buf;
        case DOP_AUX_VALUE:
          snprintf(buf, sizeof buf, "e2{a.aux_values[%u], a.aux_values[%u]}", 2 * nd.a, 2 * nd.a + 1);
          return buf;
        default: break;
      }
      if (uni_slot[id] >= 0) {  // a uniform gate: from the table the uniform kernel filled
        if (nd.ext) snprintf(buf, sizeof buf, "e2{a.uni[%d], a.uni[%d]}", uni_slot[id], uni_slot[id] + 1);
        else snprintf(buf, sizeof buf, "a.uni[%d]", uni_slot[id]);
        return buf;
      }
      snprintf(buf, sizeof buf, "v%u", id);
      return buf;  // computed, recomputed or loaded earlier in this chunk (items)
    };
    bool any_fold = false;
    if (!fuse) fold_terms = 0;
    // $MH_JIT_PREFETCH = K > 0: a load (cell, periodic value, spilled value) is issued K items AHEAD of its first use instead of at it
    // -- between "all at the top" (MH_JIT_LAZY=0: every cell of the chunk alive from its first instruction) and "at first use" (the
    // wave waits out the full HBM latency unless the compiler hoists the load itself).
    std::vector<Item> order;
    {
      const long K = std::max(0, env_int("MH_JIT_PREFETCH", 0));
      const std::vector<Item>& its = items[ci];
      if (K == 0 || !lazy_loads) {
        order = its;
      } else {
        std::vector<char> emitted(its.size(), 0);
        std::set<uint32_t> seen_leaf;
        std::vector<uint32_t> o;
        size_t ahead = 0;  // items below this index have been scanned for loads
        auto scan = [&](size_t j) {
          const Item& x = its[j];
          if (x.fold_k == -2) { order.push_back(x); emitted[j] = 1; return; }
          auto leaf = [&](uint32_t c) {
            const int op = nodes[c].op;
            if ((op == DOP_MAIN || op == DOP_AUX || op == DOP_PREP || op == DOP_PERIODIC) && seen_leaf.insert(c).second) order.push_back({c, -3});
          };
          if (x.fold_k >= 0) { leaf(x.node); return; }
          ops(x.node, o);
          for (uint32_t c : o) leaf(c);
        };
        for (size_t i = 0; i < its.size(); i++) {
          for (; ahead < its.size() && ahead <= i + (size_t)K; ahead++) scan(ahead);
          if (!emitted[i]) order.push_back(its[i]);
        }
      }
    }
    // $MH_JIT_MULGROUP = G > 1 (default 0 = off): products are emitted in stage-interleaved groups of up to G (prelude: lz_mulN).  A MUL
    // gate contributes 1 (base x base), 2 (EF x base) or 4 (EF x EF, schoolbook) products; a group is filled with the MUL gates among the
    // next $MH_JIT_MULWIN items whose operands are defined already (cells, uniform values, gates emitted above) -- they are hoisted to
    // the group's position, everything else keeps its place.  Round 6, core AIR, ISA of the nine chunks: 38.5 k -> 11.6 k wait states and
    // 89 k -> 63 k issue slots per wave (VALU + 2 %), every digest unchanged -- and SLOWER on the device: 12.81 ms -> 14.10 (G = 4, window
    // 32), 12.97 (window 0: a gate's own products only), 13.25 (G = 4 with MH_JIT_PREFETCH=64): the stage statements are rigid blocks
    // that need all their operands at once (cell loads exposed at two waves per SIMD), hoisting costs registers (two more chunks after
    // the re-split), and what the 13 free-floating statements of lz_mul_asm let hipcc's scheduler interleave is worth more than the wait
    // states they carry -- those overlap with the other wave's VALU issue.  Kept as a switch (bit-exactness and offline-compile tests).
    const long mul_group = lazy_vals ? std::max(0, env_int("MH_JIT_MULGROUP", 0)) : 0;
    const long mul_win = std::max(0, env_int("MH_JIT_MULWIN", 32));
    std::vector<char> taken(order.size(), 0);
    std::vector<char> defd(nodes.size(), 0);  // gates whose value has a name in this chunk so far
    auto n_products = [&](uint32_t g) { const bool ea = nodes[nodes[g].a].ext, eb = nodes[nodes[g].b].ext; return ea && eb ? 4 : (ea || eb ? 2 : 1); };
    auto groupable = [&](const Item& x) { return x.fold_k == -1 && nodes[x.node].op == DOP_MUL && dot_of[x.node] < 0; };
    auto ready = [&](uint32_t c) { return !interior(c) || defd[c]; };
    for (size_t oi = 0; oi < order.size(); oi++) {
      if (taken[oi]) continue;
      const Item& it = order[oi];
      const uint32_t id = it.node;
      const DagNode& nd = nodes[id];
      if (it.fold_k == -1 || it.fold_k == -2) defd[id] = 1;  // (set before the emission below: nothing reads it in between)
      if (it.fold_k == -3) {  // prefetch: the leaf's load goes here
        (void)ref(id);
        continue;
      }
      if (it.fold_k == -2) {  // produced by an earlier chunk: from its spill plane(s)
        if (nd.ext)
          ld << "  const e2 v" << id << " = {" << spill_ref(slot[id]) << ", " << spill_ref(slot[id] + 1) << "};\n";
        else
          ld << "  const u64 v" << id << " = " << spill_ref(slot[id]) << ";\n";
        continue;
      }
      if (it.fold_k >= 0) {
        any_fold = true;
        const std::string x = ref(id);
        if (ir.outputs) {  // output k -> planes 2k (c0) and 2k+1 (c1, EF outputs only), rows r (single coset)
          const int k = it.fold_k;
          const char* cf = canon[id] ? "" : "lz_canon";  // what leaves the program is a field element
          if (nd.ext)
            body << "  { const e2 o = " << x << "; a.acc[(" << 2 * k << "ull << a.log_n) + r] = " << cf << "(o.c0); a.acc[(" << 2 * k + 1
                 << "ull << a.log_n) + r] = " << cf << "(o.c1); }\n";
          else
            body << "  a.acc[(" << 2 * k << "ull << a.log_n) + r] = " << cf << "(" << x << ");\n";
          continue;
        }
        snprintf(buf, sizeof buf, "e2{a.alpha_pows[%d], a.alpha_pows[%d]}", 2 * it.fold_k, 2 * it.fold_k + 1);
        body << "#if MH_JIT_FOLD\n";
        if (nd.ext)
          body << "  { const u64 al0 = a.alpha_pows[" << 2 * it.fold_k << "], al1 = a.alpha_pows[" << 2 * it.fold_k + 1 << "], al7 = gl_mul7(al1);\n"
               << "    fold_limbs(f0, al0, " << x << ".c0); fold_limbs(f0, al7, " << x << ".c1); fold_limbs(f1, al0, " << x
               << ".c1); fold_limbs(f1, al1, " << x << ".c0); }\n";
        else
          body << "  fold_limbs(f0, a.alpha_pows[" << 2 * it.fold_k << "], " << x << "); fold_limbs(f1, a.alpha_pows[" << 2 * it.fold_k + 1
               << "], " << x << ");\n";
        fold_terms += nd.ext ? 2 : 1;
        if (fold_terms >= 400) {  // keep every accumulator below 2^64: <= 2^9 products of < 2^54 between two reductions
          body << "  acc = e2_add(acc, e2{fold_value(f0), fold_value(f1)}); f0 = {0, 0, 0, 0, 0, 0}; f1 = {0, 0, 0, 0, 0, 0};\n";
          fold_terms = 0;
        }
        body << "#else\n";
        body << "  acc = e2_add(acc, " << (nd.ext ? "e2_mul(" : "e2_mulf(") << buf << ", " << x << "));\n";
        body << "#endif\n";
        continue;
      }
      if (dot_of[id] >= 0) {
        const Dot& d = dots[dot_of[id]];
        body << "  fold_acc d" << id << "a = {0, 0, 0, 0, 0, 0}, d" << id << "b = {0, 0, 0, 0, 0, 0};\n";
        for (auto& t : d.terms) {
          const std::string U = ref(t.first), Bv = ref(t.second);
          body << "  { const e2 cu = " << U << "; const u64 cb = " << Bv << "; fold_limbs(d" << id << "a, cu.c0, cb); fold_limbs(d" << id
               << "b, cu.c1, cb); }\n";
        }
        std::string rhs = "e2{fold_value(d" + std::to_string(id) + "a), fold_value(d" + std::to_string(id) + "b)}";
        bool rc = true;  // the running sum is canonical so far
        int links = 0, pieces = 0;
        for (uint32_t o : d.others) {
          const std::string O = ref(o);
          if (lazy_vals) {
            const std::string tp = std::string("<") + (rc ? "1" : "0") + ", " + (canon[o] ? "1" : "0") + ">(";
            rhs = (nodes[o].ext ? "lz_e2_add" : "lz_e2_addf") + tp + rhs + ", " + O + ")";
            rc = false;
            if (lz_chain_max > 0 && ++links >= lz_chain_max) {  // see lz_chain above
              body << "  e2 d" << id << "p" << pieces << " = " << rhs << "; asm(\"\" : \"+v\"(d" << id << "p" << pieces << ".c0), \"+v\"(d" << id << "p"
                   << pieces << ".c1));\n";
              rhs = "d" + std::to_string(id) + "p" + std::to_string(pieces++);
              links = 0;
            }
          } else {
            rhs = (nodes[o].ext ? "e2_add(" : "e2_addf(") + rhs + ", " + O + ")";
          }
        }
        body << "  const e2 v" << id << " = " << rhs << ";\n";
        if (spilled[id] && def_chunk[id] == (int32_t)ci)
          body << "  " << spill_ref(slot[id]) << " = v" << id << ".c0; " << spill_ref(slot[id] + 1) << " = v" << id << ".c1;\n";
        continue;
      }
      if (mul_group > 1 && groupable(it)) {
        std::vector<size_t> unit{oi};
        long np = n_products(id);
        defd[id] = 0;  // not an operand of its own group
        for (size_t oj = oi + 1; oj < order.size() && oj <= oi + (size_t)mul_win && np < mul_group; oj++) {
          if (taken[oj] || !groupable(order[oj])) continue;
          const uint32_t g = order[oj].node;
          if (!ready(nodes[g].a) || !ready(nodes[g].b) || np + n_products(g) > std::max(mul_group, 4L)) continue;
          unit.push_back(oj);
          np += n_products(g);
        }
        defd[id] = 1;
        if (np >= 2) {
          std::vector<std::string> pa, pb;
          for (size_t u : unit) {
            const DagNode& g = nodes[order[u].node];
            const std::string A = ref(g.a), Bv = ref(g.b);
            const bool ea = nodes[g.a].ext, eb = nodes[g.b].ext;
            if (ea && eb) {  // schoolbook: a0 b0, a1 b1, a0 b1, a1 b0
              pa.insert(pa.end(), {A + ".c0", A + ".c1", A + ".c0", A + ".c1"});
              pb.insert(pb.end(), {Bv + ".c0", Bv + ".c1", Bv + ".c1", Bv + ".c0"});
            } else if (ea || eb) {
              const std::string& E = ea ? A : Bv;
              const std::string& F = ea ? Bv : A;
              pa.insert(pa.end(), {E + ".c0", E + ".c1"});
              pb.insert(pb.end(), {F, F});
            } else {
              pa.push_back(A);
              pb.push_back(Bv);
            }
          }
          body << "  u64 g" << id << "r[" << np << "]; { const u64 ga[" << np << "] = {";
          for (size_t k = 0; k < pa.size(); k++) body << (k ? ", " : "") << pa[k];
          body << "}, gb[" << np << "] = {";
          for (size_t k = 0; k < pb.size(); k++) body << (k ? ", " : "") << pb[k];
          body << "}; lz_mulN<" << np << ">(g" << id << "r, ga, gb); }\n";
          long k = 0;
          for (size_t u : unit) {
            const uint32_t gid = order[u].node;
            const DagNode& g = nodes[gid];
            const bool ea = nodes[g.a].ext, eb = nodes[g.b].ext;
            const std::string R = "g" + std::to_string(id) + "r[";
            if (ea && eb) {
              body << "  const e2 v" << gid << " = {lz_add_g(" << R << k << "], lz_mul7(" << R << k + 1 << "])), lz_add_g(" << R << k + 2 << "], " << R
                   << k + 3 << "])};\n";
              k += 4;
            } else if (ea || eb) {
              body << "  const e2 v" << gid << " = {" << R << k << "], " << R << k + 1 << "]};\n";
              k += 2;
            } else {
              body << "  const u64 v" << gid << " = " << R << k << "];\n";
              k += 1;
            }
            if (spilled[gid] && def_chunk[gid] == (int32_t)ci) {
              if (g.ext)
                body << "  " << spill_ref(slot[gid]) << " = v" << gid << ".c0; " << spill_ref(slot[gid] + 1) << " = v" << gid << ".c1;\n";
              else
                body << "  " << spill_ref(slot[gid]) << " = v" << gid << ";\n";
            }
            taken[u] = 1;
            defd[gid] = 1;
          }
          continue;
        }
      }
      const std::string A = ref(nd.a);
      const bool ea = nodes[nd.a].ext;
      std::string rhs;
      bool opaque = false;
      if (lazy_vals) {
        const std::string ca = canon[nd.a] ? "1" : "0";
        if (nd.op == DOP_NEG) {
          rhs = (ea ? "lz_e2_neg<" : "lz_neg<") + ca + ">(" + A + ")";
        } else {
          const std::string Bv = ref(nd.b);
          const bool eb = nodes[nd.b].ext;
          const std::string cb = canon[nd.b] ? "1" : "0", tab = "<" + ca + ", " + cb + ">(", tba = "<" + cb + ", " + ca + ">(";
          // a one-sided lazy addition / subtraction (exactly the `_c` forms of the prelude) extends the chain of its lazy operand
          const bool one_sided = (nd.op == DOP_ADD && (canon[nd.a] != 0) != (canon[nd.b] != 0)) || (nd.op == DOP_SUB && canon[nd.b] && !canon[nd.a]);
          if (one_sided && lz_chain_max > 0) {
            const uint32_t lazy_op = canon[nd.a] ? nd.b : nd.a;
            lz_chain[id] = (uint16_t)(lz_chain[lazy_op] + 1);
            if (lz_chain[id] >= lz_chain_max) opaque = true, lz_chain[id] = 0;
          }
          if (nd.op == DOP_ADD)
            rhs = ea && eb ? "lz_e2_add" + tab + A + ", " + Bv + ")" : ea ? "lz_e2_addf" + tab + A + ", " + Bv + ")"
                  : eb     ? "lz_e2_addf" + tba + Bv + ", " + A + ")" : "lz_add" + tab + A + ", " + Bv + ")";
          else if (nd.op == DOP_SUB)
            rhs = ea && eb ? "lz_e2_sub" + tab + A + ", " + Bv + ")" : ea ? "lz_e2_subf" + tab + A + ", " + Bv + ")"
                  : eb     ? "lz_e2_fsub" + tab + A + ", " + Bv + ")" : "lz_sub<" + cb + ">(" + A + ", " + Bv + ")";
          else
            rhs = ea && eb ? "lz_e2_mul(" + A + ", " + Bv + ")" : ea ? "lz_e2_mulf(" + A + ", " + Bv + ")"
                  : eb     ? "lz_e2_mulf(" + Bv + ", " + A + ")" : "lz_mul(" + A + ", " + Bv + ")";
        }
      } else if (nd.op == DOP_NEG) {
        rhs = (ea ? "e2_neg(" : "gl_neg(") + A + ")";
      } else {
        const std::string Bv = ref(nd.b);
        const bool eb = nodes[nd.b].ext;
        if (nd.op == DOP_ADD)
          rhs = ea && eb ? "e2_add(" + A + ", " + Bv + ")" : ea ? "e2_addf(" + A + ", " + Bv + ")" : eb ? "e2_addf(" + Bv + ", " + A + ")"
                                                                                                     : "gl_add(" + A + ", " + Bv + ")";
        else if (nd.op == DOP_SUB)
          rhs = ea && eb ? "e2_sub(" + A + ", " + Bv + ")" : ea ? "e2_subf(" + A + ", " + Bv + ")" : eb ? "e2_fsub(" + A + ", " + Bv + ")"
                                                                                                     : "gl_sub(" + A + ", " + Bv + ")";
        else
          rhs = ea && eb ? "e2_mul(" + A + ", " + Bv + ")" : ea ? "e2_mulf(" + A + ", " + Bv + ")" : eb ? "e2_mulf(" + Bv + ", " + A + ")"
                                                                                                     : "gl_mul(" + A + ", " + Bv + ")";
      }
      if (opaque && nd.ext)
        body << "  e2 v" << id << " = " << rhs << "; asm(\"\" : \"+v\"(v" << id << ".c0), \"+v\"(v" << id << ".c1));\n";
      else if (opaque)
        body << "  u64 v" << id << " = " << rhs << "; asm(\"\" : \"+v\"(v" << id << "));\n";
      else
        body << "  const " << (nd.ext ? "e2" : "u64") << " v" << id << " = " << rhs << ";\n";
      if (spilled[id] && def_chunk[id] == (int32_t)ci) {
        if (nd.ext)
          body << "  " << spill_ref(slot[id]) << " = v" << id << ".c0; " << spill_ref(slot[id] + 1) << " = v" << id << ".c1;\n";
        else
          body << "  " << spill_ref(slot[id]) << " = v" << id << ";\n";
      }
    }
    if (fuse) {  // a region of the fused kernel: its own scope (names may repeat), fenced from the next
      fuse_need_x |= need_x;
      fuse_need_fl |= need_fl;
      fuse_any_fold |= any_fold;
      fused_regions << "  {  // region " << ci << "\n" << decl.str() << body.str();
      // MH_JIT_FUSE_FOLDREG=0: the limb accumulators (24 registers) are reduced into `acc` (4) at the end of every region that folds
      if (!fuse_fold_persist && any_fold) {
        fused_regions << "#if MH_JIT_FOLD\n  acc = e2_add(acc, e2{fold_value(f0), fold_value(f1)}); f0 = {0, 0, 0, 0, 0, 0}; f1 = {0, 0, 0, 0, 0, 0};\n#endif\n";
        fold_terms = 0;
      }
      fused_regions << "  }\n";
      if (ci + 1 < n_chunks) fused_regions << "  MH_REGION_FENCE();\n";
      continue;
    }
    std::ostringstream src;
    src << JIT_PRELUDE;
    src << "#ifdef MH_JIT_WAVES\n__attribute__((amdgpu_waves_per_eu(MH_JIT_WAVES, MH_JIT_WAVES)))\n#endif\n"
           "extern \"C\" __global__ __launch_bounds__(256) void mh_jit_chunk(JitArgs a) {\n"
           "  const u64 qb = blockIdx.x * 256ull + threadIdx.x;\n"
           "  if (qb >= a.q_count) return;\n"
           "  const u64 q = a.q0 + qb;\n"
           "  const u64 n = 1ull << a.log_n, D = 1ull << a.log_d, Dl = 1ull << a.log_dl, B = 1ull << a.log_cosets;\n"
           "  const u64 t = q >> a.log_n, r = q & (n - 1), rn = (r + 1) & (n - 1);\n"
           "  const u64 jc = t << a.jc_shift;\n"
           "  (void)D; (void)Dl; (void)B; (void)rn; (void)jc;\n";
    if (need_x || need_fl)
      src << "  const u64 half = n >> 1;\n"
             "  const u64 w = (r < half || half == 0) ? a.tw[half ? r : 0] : gl_neg(a.tw[r - half]);\n"
             "  const u64 x = gl_mul(a.coset_tab[t], w);\n"
             "  const u64 sel_trans = gl_sub(x, a.wh_inv); (void)sel_trans;\n";
    if (need_fl)
      src << "  const u64 sel_first = gl_mul(a.coset_tab[Dl + t], a.inv_first[q]);\n"
             "  const u64 sel_last = gl_mul(a.coset_tab[Dl + t], a.inv_last[q]);\n";
    src << "  e2 acc = {0, 0}; (void)acc;\n  fold_acc f0 = {0, 0, 0, 0, 0, 0}, f1 = {0, 0, 0, 0, 0, 0}; (void)f0; (void)f1;\n" << decl.str() << body.str();
    if (!ir.outputs && any_fold) src << "#if MH_JIT_FOLD\n  acc = e2_add(acc, e2{fold_value(f0), fold_value(f1)});\n#endif\n";
    if (!ir.outputs && (any_fold || ci == 0)) {
      src << "  u64* p0 = a.acc + (((2 * t) << a.log_n) + r);\n  u64* p1 = a.acc + (((2 * t + 1) << a.log_n) + r);\n";
      if (ci == 0) src << "  *p0 = acc.c0; *p1 = acc.c1;\n";
      else src << "  *p0 = gl_add(*p0, acc.c0); *p1 = gl_add(*p1, acc.c1);\n";
    }
    src << "}\n";
    ch.src = src.str();
  }
  if (fuse) {
    std::ostringstream src;
    src << JIT_PRELUDE;
    // two waves per SIMD: 256 registers per lane (the unified file holds 512), two workgroups of 256 lanes and <= 80 KB of LDS per CU
    src << "#ifndef MH_JIT_WAVES\n#define MH_JIT_WAVES 2\n#endif\n"
           "__attribute__((amdgpu_waves_per_eu(MH_JIT_WAVES, MH_JIT_WAVES)))\n"
           "extern \"C\" __global__ __launch_bounds__(256) void mh_jit_chunk(JitArgs a) {\n";
    if (lds_words) src << "  __shared__ u64 L[" << lds_words << "];\n";
    src << "  const u32 tid = threadIdx.x; (void)tid;\n"
           "  u32 mh_live = 1;\n"
           "  const u64 qb = blockIdx.x * 256ull + tid;\n"   // the host launches whole workgroups only (n >= 256 rows, blocks of 2^k >= 256 points)
           "  const u64 q = a.q0 + qb;\n"
           "  const u64 n = 1ull << a.log_n, D = 1ull << a.log_d, Dl = 1ull << a.log_dl, B = 1ull << a.log_cosets;\n"
           "  const u64 t = q >> a.log_n, r = q & (n - 1), rn = (r + 1) & (n - 1);\n"
           "  const u64 jc = t << a.jc_shift;\n"
           "  (void)D; (void)Dl; (void)B; (void)rn; (void)jc;\n";
    // stage the tiles: every lane its own row, the last lane also the first row after the tile (rn wraps at the end of the coset)
    bool any_tile = false;
    auto stage = [&](const char* mat, size_t plane, int32_t off) {
      if (off < 0) return;
      any_tile = true;
      src << "  { const u64* cp = a." << mat << " + ((" << plane << "ull * B + jc) << a.log_n); L[" << off << " + tid] = cp[r]; if (tid == 255) L[" << off + 256
          << "] = cp[rn]; }\n";
    };
    for (size_t i = 0; i < tile_main.size(); i++) stage("main_lde", i, tile_main[i]);
    for (size_t i = 0; i < tile_aux.size(); i++) stage("aux_lde", i, tile_aux[i]);
    for (size_t i = 0; i < tile_prep.size(); i++) stage("prep_lde", i, tile_prep[i]);
    if (any_tile) src << "  __syncthreads();\n";
    if (fuse_need_x || fuse_need_fl)
      src << "  const u64 half = n >> 1;\n"
             "  const u64 w = (r < half || half == 0) ? a.tw[half ? r : 0] : gl_neg(a.tw[r - half]);\n"
             "  const u64 x = gl_mul(a.coset_tab[t], w);\n"
             "  const u64 sel_trans = gl_sub(x, a.wh_inv); (void)sel_trans;\n";
    if (fuse_need_fl)
      src << "  const u64 sel_first = gl_mul(a.coset_tab[Dl + t], a.inv_first[q]);\n"
             "  const u64 sel_last = gl_mul(a.coset_tab[Dl + t], a.inv_last[q]);\n";
    src << "  e2 acc = {0, 0};\n  fold_acc f0 = {0, 0, 0, 0, 0, 0}, f1 = {0, 0, 0, 0, 0, 0}; (void)f0; (void)f1;\n";
    src << "  MH_REGION_FENCE();\n" << fused_regions.str();
    if (fuse_any_fold) src << "#if MH_JIT_FOLD\n  acc = e2_add(acc, e2{fold_value(f0), fold_value(f1)});\n#endif\n";
    // k_quot_finish (quotient.hip) in the same kernel: * 1/Z_H of the coset, + beta * the previous AIRs' accumulation
    src << "  e2 qv = e2_mulf(acc, a.coset_tab[2 * Dl + t]);\n"
           "  if (a.acc_in) {\n"
           "    const u64 rp = r & ((1ull << a.log_n_prev) - 1);\n"
           "    const e2 old = {a.acc_in[((2 * t) << a.log_n_prev) + rp], a.acc_in[((2 * t + 1) << a.log_n_prev) + rp]};\n"
           "    qv = e2_add(e2_mul(old, e2{a.beta0, a.beta1}), qv);\n"
           "  }\n"
           "  a.acc[((2 * t) << a.log_n) + r] = qv.c0;\n  a.acc[((2 * t + 1) << a.log_n) + r] = qv.c1;\n}\n";
    chunks.clear();
    chunks.push_back({0, seq.size(), src.str(), {}, {}});
  }
  const size_t n_main_kernels = chunks.size();  // chunk kernels, or the one fused kernel
  // ---- the uniform kernel: every live uniform gate, in node order (operands precede their gate), canonical arithmetic, one lane ----
  if (n_uni) {
    std::ostringstream src;
    src << JIT_PRELUDE << "extern \"C\" __global__ void mh_jit_chunk(JitArgs a) {\n  if (blockIdx.x | threadIdx.x) return;\n  u64* U = (u64*)a.uni;\n";
    auto uref = [&](uint32_t id) -> std::string {
      const DagNode& nd = nodes[id];
      switch (nd.op) {
        case DOP_CONST: snprintf(buf, sizeof buf, "0x%llxULL", (unsigned long long)nd.c); return buf;
        case DOP_PUBLIC: snprintf(buf, sizeof buf, "a.publics[%u]", nd.a); return buf;
        case DOP_RANDOMNESS: snprintf(buf, sizeof buf, "e2{a.randomness[%u], a.randomness[%u]}", 2 * nd.a, 2 * nd.a + 1); return buf;
        case DOP_AUX_VALUE: snprintf(buf, sizeof buf, "e2{a.aux_values[%u], a.aux_values[%u]}", 2 * nd.a, 2 * nd.a + 1); return buf;
        default: break;
      }
      snprintf(buf, sizeof buf, "un%u", id);
      return buf;
    };
    for (size_t i = 0; i < nodes.size(); i++) {
      if (uni_slot[i] < 0) continue;
      const DagNode& nd = nodes[i];
      const std::string A = uref(nd.a);
      const bool ea = nodes[nd.a].ext;
      std::string rhs;
      if (nd.op == DOP_NEG) {
        rhs = (ea ? "e2_neg(" : "gl_neg(") + A + ")";
      } else {
        const std::string Bv = uref(nd.b);
        const bool eb = nodes[nd.b].ext;
        const char* f2 = nd.op == DOP_ADD ? "add" : nd.op == DOP_SUB ? "sub" : "mul";
        if (ea && eb) rhs = std::string("e2_") + f2 + "(" + A + ", " + Bv + ")";
        else if (!ea && !eb) rhs = std::string("gl_") + f2 + (nd.op == DOP_MUL ? "_c(" : "(") + A + ", " + Bv + ")";
        else if (nd.op == DOP_SUB) rhs = ea ? "e2_subf(" + A + ", " + Bv + ")" : "e2_fsub(" + A + ", " + Bv + ")";
        else rhs = std::string("e2_") + f2 + "f(" + (ea ? A + ", " + Bv : Bv + ", " + A) + ")";
      }
      // "un<id>", not "u<id>": node 32 / 64 / 128 would shadow the prelude's u32 / u64 / u128 for the rest of the kernel (found by the
      // randomised parity test, tests/test_gpu_fuzz_parity.py, on its first statement)
      src << "  const " << (nd.ext ? "e2" : "u64") << " un" << i << " = " << rhs << ";\n";
      if (nd.ext) src << "  U[" << uni_slot[i] << "] = un" << i << ".c0; U[" << uni_slot[i] + 1 << "] = un" << i << ".c1;\n";
      else src << "  U[" << uni_slot[i] << "] = un" << i << ";\n";
    }
    src << "}\n";
    chunks.push_back({0, 0, src.str(), {}, {}});  // compiled and cached with the chunks; launched once per call, ahead of them
  }
  const size_t n_kernels = chunks.size();
  if (const char* dir = getenv("MH_JIT_DUMP")) {
    for (size_t ci = 0; ci < n_kernels; ci++) {
      std::ofstream f(std::string(dir) + "/chunk" + std::to_string(ci) + ".hip");
      f << chunks[ci].src;
    }
  }
  if (env_int("MH_JIT_NO_COMPILE", 0)) return nullptr;  // source generation only (tools, no GPU)
  const bool compile_only = g_jit_compile_only;           // mh_jit_precompile: fill the cache, load nothing (no GPU needed)

  // ---- compile the chunks in parallel (cached on disk by source hash) ----
  const std::string cdir = cache_dir();
  const char* ro_env = getenv("MH_JIT_CACHE_RO_DIR");
  const std::string rodir = ro_env && *ro_env && cdir != ro_env && !compile_only ? ro_env : "";  // mh_jit_precompile fills ITS directory
  std::atomic<size_t> next{0};
  size_t n_limit = n_kernels;  // workers take chunk indices below this bound
  std::atomic<bool> failed{false};
  std::string first_error;
  std::mutex err_mu;
  auto worker = [&]() {
    for (;;) {
      const size_t ci = next.fetch_add(1);
      if (ci >= n_limit || failed.load()) return;
      Chunk& ch = chunks[ci];
      try {
        const std::string key = cache_key(ch.src);
        const std::string cpath = cdir.empty() ? "" : cdir + "/" + key;
        // $MH_JIT_CACHE_RO_DIR: a read-only cache consulted first and never written (the kernels shipped with the package: a box whose
        // hiprtc differs simply misses there and compiles into the writable cache, not into the package directory)
        if (!ch.no_cache && ((!rodir.empty() && cache_load(rodir + "/" + key, ch.code)) || (!cpath.empty() && cache_load(cpath, ch.code)))) {
          ch.from_cache = true;
          continue;
        }
        ch.from_cache = false;
        hiprtcProgram prog;
        hiprtc_check(hiprtcCreateProgram(&prog, ch.src.c_str(), "mh_jit_chunk.hip", 0, nullptr, nullptr), "create");
        // $MH_JIT_FLAGS: one extra compiler option for experiments (e.g. -DMH_JIT_ASM_MUL=1); it is part of the cache key
        const char* extra = getenv("MH_JIT_FLAGS");
        const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", extra && *extra ? extra : "-DMH_JIT_DEFAULT"};
        const hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
        if (r != HIPRTC_SUCCESS) {
          size_t ls = 0;
          hiprtcGetProgramLogSize(prog, &ls);
          std::string lg(ls, 0);
          if (ls) hiprtcGetProgramLog(prog, &lg[0]);
          hiprtcDestroyProgram(&prog);
          throw MhError(MH_ERR_INTERNAL, "hiprtc compile of constraint chunk " + std::to_string(ci) + " failed: " + lg.substr(0, 800));
        }
        size_t cs = 0;
        hiprtc_check(hiprtcGetCodeSize(prog, &cs), "code size");
        ch.code.resize(cs);
        hiprtc_check(hiprtcGetCode(prog, ch.code.data()), "get code");
        hiprtcDestroyProgram(&prog);
        if (!cpath.empty()) cache_store(cpath, ch.code);
      } catch (const std::exception& e) {
        std::lock_guard<std::mutex> g(err_mu);
        if (!failed.exchange(true)) first_error = e.what();
      }
    }
  };
  {
    unsigned nt = std::max(1u, std::min<unsigned>((unsigned)n_kernels, std::min(64u, std::thread::hardware_concurrency())));
    nt = (unsigned)std::max(1, env_int("MH_JIT_THREADS", (int)nt));
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nt; i++) th.emplace_back(worker);
    for (auto& t : th) t.join();
  }
  if (failed.load()) throw MhError(MH_ERR_INTERNAL, first_error);
  // ---- a chunk the register allocator could not fit into 256 VGPRs (it then parks values in accumulation registers: 264 "VGPRs" =
  // ONE wave per SIMD; or in scratch memory) runs several times slower than two chunks of half its size (core AIR, round 5: two
  // 264-register chunks took the quotient from 13.4 to 18.7 ms).  Such a chunk is cut in two at the
  // position of its middle third with the fewest crossing values, and everything is generated again (at most three times;
  // $MH_JIT_SPLIT=0: keep what the first cut gave). ----
  if (fuse) {
    // The fused kernel must fit 256 registers without scratch memory (two waves per SIMD).  When it does not, the bound on the words a
    // region holds alive is lowered by an eighth and the regions above it are cut again.
    unsigned scratch = 0, vg = 0;
    const bool known = code_object_info(chunks[0].code, &scratch, &vg);
    if (env_int("MH_JIT_STATS", 0) && known) fprintf(stderr, "[mh jit]   fused kernel: %u VGPRs, %u bytes of scratch, %zu regions\n", vg, scratch, n_regions);
    // (a few dwords of scratch are tolerated: the scheduler fills the 256 registers it is given and the allocator then misses by a
    // handful of values -- 18 spilled dwords in the core AIR's 50 k instructions; cutting regions to remove them costs recomputation)
    if (known && ((int)scratch > env_int("MH_JIT_FUSE_SCRATCH", 128) || (int)vg > env_int("MH_JIT_FUSE_MAXREGS", 256)) && press_max * 7 / 8 >= env_int("MH_JIT_FUSE_PRESS_MIN", 56) && env_int("MH_JIT_SPLIT", 1))
      return jit_program_build_cuts(ctx, ir, &region_ends, 0, press_max * 7 / 8);  // the same regions, those above the lower bound cut again
  } else if (depth < 3 && env_int("MH_JIT_SPLIT", 1)) {
    // the unified VGPR + AGPR budget of a chunk.  256 would still be two waves per SIMD, but a chunk AT the limit is fragile (one box of
    // the pool ran the core AIR's 256-register chunk five times slower than the others: 3.5 ms instead of 0.65 per 2^22 points) and
    // cutting it costs nothing: core AIR 256 / 248 / 200 / 168 -> 13.46 / 13.27 / 13.17 / 13.97 ms (round 5)
    const int max_regs = env_int("MH_JIT_MAXREGS", 200);
    // ... and a chunk between 168 (three waves per SIMD below it) and that budget is cut again only where the cut is cheap -- at most
    // $MH_JIT_SOFTCROSS words alive across it: chiplets 4.42 -> 4.28 ms, Poseidon2 2.76 -> 2.53 ms with every chunk below 168, while the
    // core AIR, whose cuts cross 40-100 words of operation flags, loses (13.17 -> 13.97 ms) and keeps its 176-184-register chunks
    const int soft_regs = env_int("MH_JIT_SOFTREGS", 168), soft_cross = env_int("MH_JIT_SOFTCROSS", 32);
    std::vector<size_t> cuts;
    bool any = false;
    for (size_t ci = 0; ci < n_main_kernels; ci++) {
      unsigned scratch = 0, vg = 0;
      const size_t lo = chunks[ci].ev_lo, hi = chunks[ci].ev_hi;
      if (code_object_info(chunks[ci].code, &scratch, &vg) && (scratch > 0 || (int)vg > std::min(soft_regs, max_regs)) && hi - lo >= 8) {
        size_t best_m = 0;
        long best_x = -1;
        for (size_t m = lo + (hi - lo) / 3; m <= lo + 2 * (hi - lo) / 3; m++) {
          if (m <= lo || m >= hi || seq[m].fold_k >= 0) continue;  // never separate a node from the fold that consumes it
          if (best_x < 0 || crossing[m - 1] < best_x) { best_x = crossing[m - 1]; best_m = m; }
        }
        const bool must = scratch > 0 || (int)vg > max_regs;
        if (best_m && (must || best_x <= soft_cross)) {
          if (env_int("MH_JIT_STATS", 0))
            fprintf(stderr, "[mh jit] chunk %zu [%zu, %zu) is above the register budget (%u bytes of scratch, %u registers): cut again at %zu\n", ci, lo, hi, scratch, vg, best_m);
          cuts.push_back(best_m);
          any = true;
        }
      }
      cuts.push_back(hi);
    }
    if (any) return jit_program_build_cuts(ctx, ir, &cuts, depth + 1, press_max);
  }
  if (env_int("MH_JIT_STATS", 0))
    for (size_t ci = 0; ci < n_kernels; ci++) {
      unsigned scratch = 0, vg = 0;
      if (code_object_info(chunks[ci].code, &scratch, &vg)) fprintf(stderr, "[mh jit]   kernel %zu: %u VGPRs, %u bytes of scratch\n", ci, vg, scratch);
    }
  if (compile_only) {
    g_jit_last_chunks = (int)n_kernels;
    return nullptr;
  }
  std::unique_ptr<JitProgram> prog(new JitProgram());
  prog->ctx = ctx;
  prog->n_spill = fuse ? n_spill_hbm : n_spill;
  prog->n_uni = n_uni;
  prog->fused = fuse;
  prog->n_regions = n_regions;
  try {
    for (size_t ci = 0; ci < n_kernels; ci++) {
      hipModule_t m = nullptr;
      hipFunction_t f = nullptr;
      hipError_t e = hipModuleLoadData(&m, chunks[ci].code.data());
      if (e == hipSuccess) e = hipModuleGetFunction(&f, m, "mh_jit_chunk");
      if (e != hipSuccess && chunks[ci].from_cache) {
        // a cached code object the runtime refuses (stale toolchain, damaged beyond the checksum): drop the entry,
        // compile this chunk again and load the fresh code
        (void)hipGetLastError();
        if (m) (void)hipModuleUnload(m);
        m = nullptr;
        if (!cdir.empty()) (void)remove((cdir + "/" + cache_key(chunks[ci].src)).c_str());
        chunks[ci].no_cache = true;
        next.store(ci);
        n_limit = ci + 1;
        worker();
        if (failed.load()) throw MhError(MH_ERR_INTERNAL, first_error);
        HIP_CHECK(hipModuleLoadData(&m, chunks[ci].code.data()));
        e = hipModuleGetFunction(&f, m, "mh_jit_chunk");
      }
      if (e != hipSuccess) {
        if (m) (void)hipModuleUnload(m);
        HIP_CHECK(e);
      }
      prog->modules.push_back(m);
      if (ci < n_main_kernels) prog->fns.push_back(f);
      else prog->fn_uni = f;
    }
  } catch (...) {
    jit_program_free(prog.release());
    throw;
  }
  return prog.release();
}

void jit_quotient_run(mh_ctx* c, const JitProgram* p, JitArgs a, size_t total) {
  // Points are swept in blocks so that the spill planes stay bounded (n_spill * 32 MB) for any trace height.  Cache-sized blocks do
  // not pay: the cells and planes of a 2^17 / 2^18-point block fit the 256 MB Infinity Cache, yet the core AIR's quotient takes
  // 19.5 / 16.8 ms against 15.6 / 14.5 / 14.3 / 14.2 ms with 2^19 / 2^21 / 2^22 / 2^23-point blocks (round 5; tails of half-filled
  // launches cost more than the re-reads save).  A fused program whose crossing values all sit in LDS has no planes: one launch.
  MH_REQUIRE(!p->fused || (a.log_n >= 8 && total % 256 == 0), "internal: fused constraint kernel on a trace below 2^8 rows");
  const size_t block = p->fused && p->n_spill == 0 ? total : std::min(total, (size_t)1 << std::max(10, env_int("MH_JIT_BLOCK_LOG", 22)));
  DevBuf spill(std::max<size_t>(1, p->n_spill) * (p->n_spill ? block : 1) * 8);
  a.spill = spill.u();
  a.spill_stride = block;
  DevBuf uni(std::max<size_t>(1, p->n_uni) * 8);
  a.uni = uni.u();
  if (p->fn_uni) {
    void* params[] = {&a};
    HIP_CHECK(hipModuleLaunchKernel(p->fn_uni, 1, 1, 1, 64, 1, 1, 0, c->stream, params, nullptr));
  }
  for (size_t q0 = 0; q0 < total; q0 += block) {
    a.q0 = q0;
    a.q_count = std::min(block, total - q0);
    void* params[] = {&a};
    for (hipFunction_t f : p->fns)
      HIP_CHECK(hipModuleLaunchKernel(f, (unsigned)((a.q_count + 255) / 256), 1, 1, 256, 1, 1, 0, c->stream, params, nullptr));
  }
  // no host wait: the spill area and the uniform table are pool buffers, their reuse is ordered on the stream
}
