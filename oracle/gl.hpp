// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked, imported or executed by the product path
// (miden-vm_amd/): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// Goldilocks field F_p, p = 2^64 - 2^32 + 1, and its quadratic extension F_p[x]/(x^2 - 7).
//
// CPU restatement of arithmetic that the reference delegates to the external crate
// `p3-goldilocks 0.6.2` (Cargo.lock:2950-3366), anchored on the reference's own call sites:
//   * Felt = repr(transparent) u64 over Goldilocks ........ crates/field/src/native/mod.rs:56-58
//   * QuadFelt multiplication uses x^2 = 7 ................. processor/src/execution/operations/field_ops/mod.rs:227-241
//   * 2^32-th root of unity = 1753635133440165772 .......... crates/lib/core/asm/stark/constants.masm:5
//   * multiplicative generator 7 (canonical LDE shift) ..... crates/lifted-stark/src/domain.rs:358-361
//
// Deliberately the "obviously correct" formulation (unsigned __int128 + %), NOT the fast
// reduction the HIP kernels use: the two must agree bit-for-bit on canonical outputs.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>

namespace oracle {

typedef unsigned __int128 u128;
static const uint64_t P = 0xFFFFFFFF00000001ULL;
static const uint64_t GENERATOR = 7;
static const uint64_t ROOT_2_32 = 1753635133440165772ULL;
static const int TWO_ADICITY = 32;

#ifdef ORACLE_FAST
// (ORACLE_FAST, see fmul below) canonical operands: one conditional correction instead of a 128-bit division per addition
static inline uint64_t fadd(uint64_t a, uint64_t b) { uint64_t s = a + b; return (s < a || s >= P) ? s - P : s; }
static inline uint64_t fsub(uint64_t a, uint64_t b) { uint64_t d = a - b; return a < b ? d + P : d; }
#else
static inline uint64_t fadd(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a + b) % P); }
static inline uint64_t fsub(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a + P - b) % P); }
#endif
static inline uint64_t fneg(uint64_t a) { return a == 0 ? 0 : P - a; }
#ifdef ORACLE_FAST
// Build variant for bench.py's cpu_baseline leg only (liboracle_fast.so): same results as the
// obviously-correct `% P` form below (tests/test_oracle_kat.py cross-checks the two libraries), but
// without a 128-bit division per multiplication, so the CPU baseline is not handicapped.
static inline uint64_t fmul(uint64_t a, uint64_t b) {
  u128 p = (u128)a * b;
  uint64_t lo = (uint64_t)p, hi = (uint64_t)(p >> 64);
  uint64_t hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
  uint64_t t0 = lo - hh;
  if (lo < hh) t0 -= 0xFFFFFFFFULL;
  uint64_t t1 = (hl << 32) - hl;
  uint64_t r = t0 + t1;
  if (r < t1) r += 0xFFFFFFFFULL;
  return r >= P ? r - P : r;
}
#else
static inline uint64_t fmul(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % P); }
#endif
static inline uint64_t fpow(uint64_t a, uint64_t e) {
  uint64_t r = 1;
  while (e) {
    if (e & 1) r = fmul(r, a);
    a = fmul(a, a);
    e >>= 1;
  }
  return r;
}
static inline uint64_t finv(uint64_t a) { return fpow(a, P - 2); }
static inline uint64_t fexp_pow2(uint64_t a, int k) {
  for (int i = 0; i < k; i++) a = fmul(a, a);
  return a;
}
// omega_{2^k} = omega_{2^32}^(2^(32-k))   (p3 two_adic_generator)
static inline uint64_t two_adic_generator(int k) { return fexp_pow2(ROOT_2_32, TWO_ADICITY - k); }
// canonical LDE shift g^(2^(32 - log_lde))   (domain.rs:358-361)
static inline uint64_t canonical_lde_shift(int log_lde) { return fexp_pow2(GENERATOR, TWO_ADICITY - log_lde); }

static inline uint32_t bitrev(uint32_t x, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
}
static inline int log2_strict(size_t n) {
  int k = 0;
  while (((size_t)1 << k) < n) k++;
  return k;
}

// ---- quadratic extension, element = c0 + c1*x, x^2 = 7 ---------------------------------------
struct E2 {
  uint64_t c0, c1;
};
static inline E2 e2(uint64_t a, uint64_t b = 0) { return E2{a, b}; }
static inline E2 eadd(E2 a, E2 b) { return {fadd(a.c0, b.c0), fadd(a.c1, b.c1)}; }
static inline E2 esub(E2 a, E2 b) { return {fsub(a.c0, b.c0), fsub(a.c1, b.c1)}; }
static inline E2 eneg(E2 a) { return {fneg(a.c0), fneg(a.c1)}; }
static inline E2 emul(E2 a, E2 b) {
  return {fadd(fmul(a.c0, b.c0), fmul(7, fmul(a.c1, b.c1))), fadd(fmul(a.c0, b.c1), fmul(a.c1, b.c0))};
}
static inline E2 emulf(E2 a, uint64_t b) { return {fmul(a.c0, b), fmul(a.c1, b)}; }
static inline E2 einv(E2 a) {
  // 1/(c0 + c1 x) = (c0 - c1 x) / (c0^2 - 7 c1^2)
  uint64_t n = fsub(fmul(a.c0, a.c0), fmul(7, fmul(a.c1, a.c1)));
  uint64_t ni = finv(n);
  return {fmul(a.c0, ni), fmul(fneg(a.c1), ni)};
}
static inline E2 epow(E2 a, uint64_t e) {
  E2 r = e2(1);
  while (e) {
    if (e & 1) r = emul(r, a);
    a = emul(a, a);
    e >>= 1;
  }
  return r;
}
static inline E2 eexp_pow2(E2 a, int k) {
  for (int i = 0; i < k; i++) a = emul(a, a);
  return a;
}
static inline bool eeq(E2 a, E2 b) { return a.c0 == b.c0 && a.c1 == b.c1; }
static inline bool eiszero(E2 a) { return a.c0 == 0 && a.c1 == 0; }

}  // namespace oracle
