// ORACLE — TEST INFRASTRUCTURE ONLY (see gl.hpp header).
//
// DFT / coset-LDE over Goldilocks.  The reference calls the external crate p3-dft 0.6.2
// (`Radix2DitParallel::coset_lde_batch`, call site crates/lifted-stark/src/prover/commit.rs:173);
// its result is mathematically unique, so this restates the published definition:
//   coset_lde_batch(M, added_bits, shift): per column, interpolate the N values as evaluations
//   on H = <w_N> (natural order), and evaluate the degree-<N polynomial on shift*K,
//   K = <w_{N*2^added_bits}>; the returned matrix is a bit-reversed *view*, i.e. physical row r
//   holds the evaluation at shift*w_K^{bitrev(r)}   (commit.rs:83-106).
// `naive_dft` is the O(n^2) definition (the reference's own DFT oracle is p3 `NaiveDft`,
// e.g. crates/lifted-stark/src/prover/quotient.rs:236-272).
#pragma once
#include "gl.hpp"
#include <vector>

namespace oracle {

// out[i] = sum_k in[k] * w^(i*k), w = w_n (or w_n^-1 and /n when inverse)
static inline std::vector<uint64_t> naive_dft(const std::vector<uint64_t>& in, bool inverse) {
  size_t n = in.size();
  int lg = log2_strict(n);
  uint64_t w = two_adic_generator(lg);
  if (inverse) w = finv(w);
  std::vector<uint64_t> out(n);
  for (size_t i = 0; i < n; i++) {
    uint64_t wi = fpow(w, i), acc = 0, x = 1;
    for (size_t k = 0; k < n; k++) {
      acc = fadd(acc, fmul(in[k], x));
      x = fmul(x, wi);
    }
    out[i] = inverse ? fmul(acc, finv(n % P)) : acc;
  }
  return out;
}

// In-place iterative radix-2 (bit-reverse then DIT butterflies); natural in, natural out.
static inline void dft_inplace(uint64_t* a, size_t n, bool inverse) {
  int lg = log2_strict(n);
  for (size_t i = 0; i < n; i++) {
    size_t j = bitrev((uint32_t)i, lg);
    if (i < j) std::swap(a[i], a[j]);
  }
  for (int s = 1; s <= lg; s++) {
    size_t m = (size_t)1 << s, h = m >> 1;
    uint64_t wm = two_adic_generator(s);
    if (inverse) wm = finv(wm);
    std::vector<uint64_t> tw(h);
    tw[0] = 1;
    for (size_t k = 1; k < h; k++) tw[k] = fmul(tw[k - 1], wm);
    for (size_t b = 0; b < n; b += m)
      for (size_t k = 0; k < h; k++) {
        uint64_t u = a[b + k], v = fmul(a[b + k + h], tw[k]);
        a[b + k] = fadd(u, v);
        a[b + k + h] = fsub(u, v);
      }
  }
  if (inverse) {
    uint64_t ninv = finv(n % P);
    for (size_t i = 0; i < n; i++) a[i] = fmul(a[i], ninv);
  }
}

// Evaluate the interpolant of `col` (evals on H, natural order) on shift*K, natural order.
static inline std::vector<uint64_t> coset_lde_col(const std::vector<uint64_t>& col, int added_bits, uint64_t shift) {
  size_t n = col.size();
  std::vector<uint64_t> c(col);
  dft_inplace(c.data(), n, true);  // coefficients
  std::vector<uint64_t> e(n << added_bits, 0);
  uint64_t s = 1;
  for (size_t k = 0; k < n; k++) {
    e[k] = fmul(c[k], s);
    s = fmul(s, shift);
  }
  dft_inplace(e.data(), e.size(), false);
  return e;
}

// Row-major matrix (height n, width w) -> row-major LDE (height n<<added_bits), rows stored
// bit-reversed like the reference's committed matrices.
static inline std::vector<uint64_t> coset_lde_matrix_bitrev(const uint64_t* m, size_t n, size_t w, int added_bits, uint64_t shift) {
  size_t big = n << added_bits;
  int lg = log2_strict(big);
  std::vector<uint64_t> out(big * w);
#pragma omp parallel for schedule(dynamic)
  for (long c = 0; c < (long)w; c++) {
    std::vector<uint64_t> col(n);
    for (size_t r = 0; r < n; r++) col[r] = m[r * w + c];
    std::vector<uint64_t> e = coset_lde_col(col, added_bits, shift);
    for (size_t r = 0; r < big; r++) out[(size_t)r * w + c] = e[bitrev((uint32_t)r, lg)];
  }
  return out;
}

}  // namespace oracle
