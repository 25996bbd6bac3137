// The products the generated constraint kernels are written in (csrc/air_jit.cpp, JIT_PRELUDE: lz_mul_asm -- the 13-instruction asm product, the
// default for base AND extension-field gates since round 6 --, lz_mul_c, gl_mul and the extension-field forms over them) ON THE GPU against
// 128-bit host arithmetic: every pair of an edge set (0, 1, eps, 2^32, p - 1, p, p + 1, 2^64 - 1, ...), uniform 64-bit operands, operands
// with structure (32-bit values shifted, small values, non-canonical representatives in [p, 2^64)) and operands searched on the host
// for the product's RARE paths -- the borrow `bw` of r = t - hi.hi + c1 (2^32 - 1) and its correction `mk` -- which uniform operands
// reach about once in 2^32 products, i.e. a few times per 2^20-row proof of the core AIR and never in a small test.
//   tools/jit_mulcheck [n_random = 1 << 22]   -> "mismatches 0" and exit code 0, or the first mismatches per form and exit code 1
// tools/Makefile extracts the prelude text from air_jit.cpp into jit_prelude.inc (the same text hiprtc compiles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "jit_prelude.inc"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// four independent products per lane and form, interleaved by the scheduler the way a chunk interleaves its gates; the extension-field
// product over the same operands
__global__ void k_mul(const u64* a, const u64* b, u64* o3, u64* o1, u64* oc, u64* og, u64* oe, size_t n) {
  const size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4;
  if (i + 3 >= n) return;
  u64 x[4], y[4];
  for (int k = 0; k < 4; k++) { x[k] = a[i + k]; y[k] = b[i + k]; }
  u64 r3[4], r1[4], rc[4], rg[4];
  for (int k = 0; k < 4; k++) r3[k] = lz_mul(x[k], y[k]);   // what the generator emits for a base-field MUL gate
  for (int k = 0; k < 4; k++) r1[k] = lz_mul_asm(x[k], y[k]);
  for (int k = 0; k < 4; k++) rc[k] = lz_mul_c(x[k], y[k]);
  for (int k = 0; k < 4; k++) rg[k] = gl_mul(x[k], y[k]);
  for (int k = 0; k < 4; k++) { o3[i + k] = r3[k]; o1[i + k] = r1[k]; oc[i + k] = rc[k]; og[i + k] = rg[k]; }
  // (x0 + x1 X)(y0 + y1 X) and (x2 + x3 X)(y2 + y3 X), X^2 = 7, through lz_mul_ef = the default product
  const e2 p = lz_e2_mul(e2{x[0], x[1]}, e2{y[0], y[1]}), q = lz_e2_mul(e2{x[2], x[3]}, e2{y[2], y[3]});
  oe[i] = p.c0; oe[i + 1] = p.c1; oe[i + 2] = q.c0; oe[i + 3] = q.c1;
}

typedef unsigned __int128 u128h;
static const u64 P = 0xFFFFFFFF00000001ULL;
static u64 mulmod(u64 a, u64 b) { return (u64)(((u128h)(a % P) * (b % P)) % P); }
static u64 addmod(u64 a, u64 b) { return (u64)(((u128h)a + b) % P); }

// does a * b take the borrow path of the reduction?  (the prelude's own derivation: lo + hi.lo (2^32 - 1) = t (+ c1 2^64); r = t - hi.hi - c1
// + c1 2^32 borrows when it is negative)
static bool takes_borrow(u64 a, u64 b) {
  const u128h pr = (u128h)a * b;
  const u64 lo = (u64)pr, hi = (u64)(pr >> 64);
  const u64 hl = hi & 0xffffffffULL, hh = hi >> 32;
  const u128h t = (u128h)hl * 0xffffffffULL + lo;
  const u64 c1 = (u64)(t >> 64), t64 = (u64)t;
  // rl = t.lo - hh - c1, rh = t.hi + c1 - borrow: negative overall?
  const __int128 r = (__int128)t64 - (__int128)hh - (__int128)c1 + ((__int128)c1 << 32);
  return r < 0 || r >= ((__int128)1 << 64);
}

int main(int argc, char** argv) {
  const size_t n_random = argc > 1 ? strtoull(argv[1], nullptr, 0) : (size_t)1 << 22;
  std::mt19937_64 rng(0x6d68);
  std::vector<u64> A, B;
  const u64 eps = 0xffffffffULL;
  std::vector<u64> edge = {0, 1, 2, 7, eps - 1, eps, eps + 1, eps + 2, (u64)1 << 33, (u64)1 << 48, ((u64)1 << 63) - 1, (u64)1 << 63, ((u64)1 << 63) + 1,
                           P - 2, P - 1, P, P + 1, P + eps - 1, P + eps, ~(u64)0 - 1, ~(u64)0, 0xffffffff00000000ULL, 0xfffffffeffffffffULL,
                           0x00000000fffffffeULL, 0x0000000100000001ULL, 0x8000000080000000ULL, 0x7fffffff7fffffffULL, 0xfffffffefffffffeULL};
  for (u64 x : edge) for (u64 y : edge) { A.push_back(x); B.push_back(y); }
  for (size_t i = 0; i < n_random; i++) { A.push_back(rng()); B.push_back(rng()); }
  for (size_t i = 0; i < n_random / 8; i++) { A.push_back((rng() & eps) << 32); B.push_back(rng()); }
  for (size_t i = 0; i < n_random / 8; i++) { A.push_back(rng() & 0xffff); B.push_back(rng()); }
  for (size_t i = 0; i < n_random / 8; i++) { A.push_back(P + (rng() & eps) % eps); B.push_back(P + (rng() & eps) % eps); }
  for (size_t i = 0; i < n_random / 8; i++) { A.push_back((rng() & eps) << 32); B.push_back((rng() & eps) << 32 | (rng() & 0xff)); }
  // the rare paths: products whose high word has a zero low half and a large high half against a small low word
  size_t n_borrow = 0;
  for (size_t tries = 0; tries < ((size_t)1 << 24) && n_borrow < 20000; tries++) {
    const u64 u = (rng() & eps) | 0x80000000ULL, v = (rng() & eps) | 0x80000000ULL;
    const u64 a = u << 32 | (tries & 1 ? 0 : rng() & 0xff), b = v << 32 | (tries & 2 ? 0 : rng() & 0xff);
    if (takes_borrow(a, b)) { A.push_back(a); B.push_back(b); n_borrow++; }
  }
  // products of the form h 2^96 (+ nothing below): t = 0 and hi.hi = h > 0 -- the borrow for certain.  2^k b with b = h 2^(96 - k), and
  // (u 2^48)(v 2^48): what a constraint that scales a limb by a power of two multiplies
  for (int k = 33; k < 64; k++)
    for (int rep = 0; rep < 256; rep++) {
      const u64 h = (rng() % (((u64)1 << (k - 32)) - 1 + (k == 33))) + 1;   // 1 <= h < 2^(k - 32)
      A.push_back((u64)1 << k); B.push_back(h << (96 - k));
      A.push_back(h << (96 - k)); B.push_back((u64)1 << k);
    }
  for (size_t i = 0; i < 16384; i++) { A.push_back((rng() & 0xffff) << 48); B.push_back((rng() & 0xffff) << 48); }
  for (size_t i = 0; i < 16384; i++) { A.push_back((rng() & 0xffffff) << 40); B.push_back((rng() & 0xff) << 56); }
  for (size_t i = A.size() - 2 * 31 * 256 - 32768; i < A.size(); i++) n_borrow += takes_borrow(A[i], B[i]);
  for (u64 x : edge) for (size_t i = 0; i < 512; i++) { const u64 y = rng(); if (takes_borrow(x, y)) { A.push_back(x); B.push_back(y); n_borrow++; } }
  while (A.size() % 1024) { A.push_back(rng()); B.push_back(rng()); }
  const size_t n = A.size();
  u64 *da, *db, *d3, *d1, *dc, *dg, *de;
  for (u64** p : {&da, &db, &d3, &d1, &dc, &dg, &de}) CK(hipMalloc(p, n * 8));
  CK(hipMemcpy(da, A.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, B.data(), n * 8, hipMemcpyHostToDevice));
  k_mul<<<(unsigned)((n / 4 + 255) / 256), 256>>>(da, db, d3, d1, dc, dg, de, n);
  CK(hipDeviceSynchronize());
  std::vector<u64> o3(n), o1(n), oc(n), og(n), oe(n);
  CK(hipMemcpy(o3.data(), d3, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(o1.data(), d1, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(oc.data(), dc, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(og.data(), dg, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(oe.data(), de, n * 8, hipMemcpyDeviceToHost));
  size_t bad[5] = {0, 0, 0, 0, 0};
  const char* names[5] = {"lz_mul", "lz_mul_asm", "lz_mul_c", "gl_mul", "lz_e2_mul"};
  auto report = [&](int f, size_t i, u64 got, u64 want) {
    if (bad[f]++ < 5) fprintf(stderr, "%s: a %016llx b %016llx -> %016llx, want %016llx (mod p)\n", names[f], (unsigned long long)A[i], (unsigned long long)B[i], (unsigned long long)got, (unsigned long long)want);
  };
  for (size_t i = 0; i < n; i++) {
    const u64 want = mulmod(A[i], B[i]);
    if (o3[i] % P != want) report(0, i, o3[i], want);
    if (o1[i] % P != want) report(1, i, o1[i], want);
    if (oc[i] % P != want) report(2, i, oc[i], want);
    if (og[i] != want) report(3, i, og[i], want);   // gl_mul is canonical
  }
  for (size_t i = 0; i + 3 < n; i += 4)
    for (int h = 0; h < 2; h++) {
      const u64 x0 = A[i + 2 * h], x1 = A[i + 2 * h + 1], y0 = B[i + 2 * h], y1 = B[i + 2 * h + 1];
      const u64 c0 = addmod(mulmod(x0, y0), mulmod(7, mulmod(x1, y1))), c1 = addmod(mulmod(x0, y1), mulmod(x1, y0));
      if (oe[i + 2 * h] % P != c0) report(4, i + 2 * h, oe[i + 2 * h], c0);
      if (oe[i + 2 * h + 1] % P != c1) report(4, i + 2 * h + 1, oe[i + 2 * h + 1], c1);
    }
  const size_t total = bad[0] + bad[1] + bad[2] + bad[3] + bad[4];
  printf("products %zu (edge pairs %zu, borrow-path operands %zu) mismatches %zu [lz_mul %zu, asm %zu, c %zu, gl_mul %zu, e2_mul %zu]\n", n, edge.size() * edge.size(),
         n_borrow, total, bad[0], bad[1], bad[2], bad[3], bad[4]);
  return total ? 1 : 0;
}
