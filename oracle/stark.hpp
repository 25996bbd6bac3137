// ORACLE — TEST INFRASTRUCTURE ONLY (see gl.hpp header).
//
// CPU restatement of the lifted-STARK prover and verifier, in the reference's own storage
// conventions (row-major matrices, bit-reversed physical rows, natural-order quotient vector):
//   prove ................ crates/lifted-stark/src/prover/mod.rs:230-578
//   commit_traces ........ crates/lifted-stark/src/prover/commit.rs:142-180
//   constraint eval ...... crates/lifted-stark/src/prover/constraints/mod.rs:83-278
//   selectors / 1/Z_H .... crates/lifted-stark/src/domain.rs:698-750
//   periodic LDE ......... crates/lifted-stark/src/prover/periodic.rs:49-77
//   accumulate / commit .. crates/lifted-stark/src/prover/quotient.rs:83-217
//   OOD sampling ......... crates/lifted-stark/src/domain.rs:539-553
//   PCS open ............. crates/lifted-stark/src/pcs/prover.rs:34-101
//   DEEP ................. crates/lifted-stark/src/pcs/deep/prover.rs:54-315, interpolate.rs:87-204
//   FRI .................. crates/lifted-stark/src/pcs/fri/prover.rs:93-269, fold/arity4.rs:46-121,
//                          fold/arity2.rs, fold/arity8.rs:35-138, fri/mod.rs:80-115
//   verifier ............. crates/lifted-stark/src/verifier/mod.rs, pcs/verifier.rs,
//                          pcs/deep/verifier.rs, pcs/fri/verifier.rs, lmcs/config.rs:172-211
// Every field operation is exact, so any evaluation order yields the reference's values; what
// matters for byte parity is the ORDER in which values enter the transcript.
#pragma once
#include "air.hpp"
#include "challenger.hpp"
#include "lmcs.hpp"
#include "ntt.hpp"
#include <algorithm>
#include <map>
#include <memory>
#include <numeric>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace oracle {

struct PcsParams {
  int log_blowup, log_folding_arity, log_final_degree, folding_pow_bits, deep_pow_bits, num_queries, query_pow_bits;
};
// PcsParams::new (crates/lifted-stark/src/pcs/params.rs:45-84): InvalidFoldingArity, ZeroBlowup, ZeroQueries, FinalDegreeUnreachable
// (log_final_degree + log_blowup >= log_folding_arity - 1).  nullptr = valid.
static inline const char* pcs_params_error(const PcsParams& p) {
  if (p.log_folding_arity < 1 || p.log_folding_arity > 3) return "invalid folding arity (log_arity must be 1, 2, or 3)";
  if (p.log_blowup <= 0) return "log_blowup must be > 0";
  if (p.num_queries <= 0) return "num_queries must be > 0";
  if (p.log_final_degree < 0 || p.log_final_degree + p.log_blowup < p.log_folding_arity - 1) return "final degree unreachable by fixed-arity folding";
  return nullptr;
}

// build_aux_trace callback (crates/lifted-air/src/air.rs LiftedAir::build_aux_trace): fills the
// flattened EF aux trace (n x 2*aux_width, row-major) and 2*num_aux_values felts. Non-zero = abort.
typedef int (*AuxBuilder)(void* user, int instance_idx, const uint64_t* randomness, uint64_t* aux_out, uint64_t* aux_values_out);
// Statement::eval_external as a callback: writes (c0, c1) per assertion, returns their number (<= cap) or < 0 on error.
typedef int (*ExternalAssertions)(void* user, const uint64_t* randomness, size_t n_randomness, const uint64_t* const* aux_values,
                                  const size_t* n_aux_values, const uint8_t* log_trace_heights, int n_airs, uint64_t* assertions_out,
                                  size_t cap);

struct Proof {
  std::vector<uint8_t> log_trace_heights;  // instance order
  std::vector<uint64_t> fields;
  std::vector<Digest> commitments;
  Digest digest;
};

// aligned_len(w, lmcs.alignment()) (util/align.rs:7-13); the name is from the sponge configuration, where the alignment is 8
static inline size_t align8(size_t w) {
  const size_t a = lmcs_alignment();
  return (w + a - 1) / a * a;
}

static inline void batch_inverse(std::vector<uint64_t>& v) {
  size_t n = v.size();
  if (!n) return;
  std::vector<uint64_t> pre(n);
  uint64_t acc = 1;
  for (size_t i = 0; i < n; i++) {
    pre[i] = acc;
    acc = fmul(acc, v[i]);
  }
  uint64_t inv = finv(acc);
  for (size_t i = n; i-- > 0;) {
    uint64_t x = v[i];
    v[i] = fmul(inv, pre[i]);
    inv = fmul(inv, x);
  }
}

static inline E2 horner_ef(const std::vector<uint64_t>& coeffs_ascending, E2 x) {
  E2 acc = e2(0);
  for (size_t k = coeffs_ascending.size(); k-- > 0;) acc = eadd(emul(acc, x), e2(coeffs_ascending[k]));
  return acc;
}

// proof order = stable sort by (log_height, instance index)   (order.rs)
static inline std::vector<int> proof_order(const std::vector<int>& log_heights) {
  std::vector<int> idx(log_heights.size());
  std::iota(idx.begin(), idx.end(), 0);
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return log_heights[a] < log_heights[b]; });
  return idx;
}

struct Selectors {
  E2 is_first, is_last, is_transition;
};
// domain.rs:518-531 selectors_at on the instance domain with trace height 2^log_n at point y.
static inline Selectors selectors_at(E2 y, int log_n) {
  E2 van = esub(eexp_pow2(y, log_n), e2(1));
  uint64_t wh_inv = finv(two_adic_generator(log_n));
  return {emul(van, einv(esub(y, e2(1)))), emul(van, einv(esub(y, e2(wh_inv)))), esub(y, e2(wh_inv))};
}

// Value at y of the degree < p polynomial interpolating `col` over the subgroup of order p.
static inline E2 periodic_eval(const std::vector<uint64_t>& col, E2 y) {
  std::vector<uint64_t> c(col);
  dft_inplace(c.data(), c.size(), true);
  return horner_ef(c, y);
}

static inline int fri_num_rounds(const PcsParams& p, int log_lde) {
  int log_max_final = p.log_final_degree + p.log_blowup;
  int steps = log_lde > log_max_final ? log_lde - log_max_final : 0;
  return (steps + p.log_folding_arity - 1) / p.log_folding_arity;
}
static inline int fri_log_final_poly_degree(const PcsParams& p, int log_lde) {
  int lf = log_lde - fri_num_rounds(p, log_lde) * p.log_folding_arity;
  if (lf < 0) lf = 0;
  int d = lf - p.log_blowup;
  return d < 0 ? 0 : d;
}

static inline E2 fri_fold_row(const E2* y, int log_arity, uint64_t s_inv, E2 beta) {
  E2 x = emulf(beta, s_inv);
  if (log_arity == 1) {
    E2 sum = eadd(y[0], y[1]), diff = esub(y[0], y[1]);
    E2 r = eadd(sum, emul(diff, x));
    return emulf(r, finv(2));
  }
  if (log_arity == 3) {
    // fold/arity8.rs:35-138: size-8 inverse FFT (DIT, bit-reversed input, unscaled) -> 8 * coefficients of f(sX),
    // then sum c_i x^i at x = beta / s, divided by 8.
    const uint64_t w8 = two_adic_generator(3), w4 = two_adic_generator(2);
    const uint64_t w8_3 = fmul(w4, w8), w8_5 = fmul(w8_3, w4), w8_6 = fmul(w8_3, w8_3), w8_7 = fmul(w8_6, w8);
    const uint64_t w4_inv = w8_6, i1 = w8_7, i2 = w8_6, i3 = w8_5;
    // row = [y0, y4, y2, y6, y1, y5, y3, y7]
    const E2 y0 = y[0], y4 = y[1], y2 = y[2], y6 = y[3], y1 = y[4], y5 = y[5], y3 = y[6], y7 = y[7];
    auto tf = [](E2 a, E2 b, E2& s, E2& d) { s = eadd(a, b); d = esub(a, b); };
    auto dit = [](E2 a, E2 b, uint64_t tw, E2& s, E2& d) { const E2 t = emulf(b, tw); s = eadd(a, t); d = esub(a, t); };
    E2 a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7, c[8];
    tf(y0, y4, a0, a1); tf(y2, y6, a2, a3); tf(y1, y5, a4, a5); tf(y3, y7, a6, a7);
    tf(a0, a2, b0, b2); dit(a1, a3, w4_inv, b1, b3); tf(a4, a6, b4, b6); dit(a5, a7, w4_inv, b5, b7);
    tf(b0, b4, c[0], c[4]); dit(b1, b5, i1, c[1], c[5]); dit(b2, b6, i2, c[2], c[6]); dit(b3, b7, i3, c[3], c[7]);
    E2 acc = c[7];
    for (int i = 6; i >= 0; i--) acc = eadd(emul(acc, x), c[i]);
    return emulf(acc, finv(8));
  }
  if (log_arity != 2) throw std::runtime_error("oracle: FRI arity must be 2, 4 or 8");
  // row = [y0, y2, y1, y3] (bit-reversed)
  E2 y0 = y[0], y2 = y[1], y1 = y[2], y3 = y[3];
  uint64_t w = two_adic_generator(2);
  E2 s02 = eadd(y0, y2), d02 = esub(y0, y2), s13 = eadd(y1, y3), d31w = emulf(esub(y3, y1), w);
  E2 c0 = eadd(s02, s13), c1 = eadd(d02, d31w), c2 = esub(s02, s13), c3 = esub(d02, d31w);
  E2 x2 = emul(x, x), x3 = emul(x2, x);
  E2 r = eadd(eadd(c0, emul(c1, x)), eadd(emul(c2, x2), emul(c3, x3)));
  return emulf(r, finv(4));
}

struct ProverInput {
  PcsParams params;
  std::vector<Air> airs;                 // instance order
  std::vector<const uint64_t*> traces;   // row-major natural order
  // per instance: the AIR's preprocessed matrix (same height as its trace, row-major) or nullptr.  The tree over
  // their LDEs is the setup-time commitment (preprocessed.rs:74-135); its root must already be in `challenger`
  // (observed after the protocol parameters, before the statement: prover/mod.rs:282-286).
  std::vector<const uint64_t*> preprocessed;
  std::vector<int> log_heights;
  std::vector<uint64_t> publics;
  Challenger challenger;                 // already bound to protocol params + statement
  AuxBuilder aux_builder = nullptr;
  void* aux_user = nullptr;
};

struct StageTimer {  // ORACLE_TIMING=1 prints per-stage wall time (used to keep the CPU baseline honest)
  bool on = getenv("ORACLE_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char* name) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[oracle] %-22s %8.3f s\n", name, std::chrono::duration<double>(n - t).count());
    t = n;
  }
};

static inline Proof prove(ProverInput& in) {
  StageTimer tm;
  const PcsParams& pp = in.params;
  const int lb = pp.log_blowup;
  const size_t B = (size_t)1 << lb;
  const size_t n_airs = in.airs.size();
  if (const char* e = pcs_params_error(pp)) throw std::runtime_error(e);
  // ProverStatement::new (crates/lifted-air: InstanceError::TraceHeightTooSmall): a trace needs a transition
  for (int lh : in.log_heights)
    if (lh < 1) throw std::runtime_error("trace height too small (at least 2 rows)");
  std::vector<int> order = proof_order(in.log_heights);
  const int log_n_max = in.log_heights[order.back()];
  const int L = log_n_max + lb;
  const size_t NB = (size_t)1 << L;
  if (L > TWO_ADICITY) throw std::runtime_error("LDE order exceeds two-adicity");
  const uint64_t g = canonical_lde_shift(L);

  ProverTranscript ch;
  ch.ch = in.challenger;
  // order.rs:154-163 observe_shape
  ch.ch.observe((uint64_t)n_airs);
  for (int lh : in.log_heights) ch.ch.observe((uint64_t)lh);

  int logD = 0;
  for (auto& a : in.airs) logD = std::max(logD, a.log_quotient_degree);
  if (logD > lb) throw std::runtime_error("constraint degree too high for blowup");
  const size_t D = (size_t)1 << logD;

  // ---- 1. main commitment ------------------------------------------------------------------
  auto commit = [&](const std::vector<const uint64_t*>& mats, const std::vector<size_t>& widths,
                    std::vector<std::vector<uint64_t>>& ldes) {
    std::vector<Mat> ms;
    ldes.resize(n_airs);
    for (size_t j = 0; j < n_airs; j++) {
      int lh = in.log_heights[order[j]];
      size_t n = (size_t)1 << lh;
      ldes[j] = coset_lde_matrix_bitrev(mats[j], n, widths[j], lb, canonical_lde_shift(lh + lb));
      ms.push_back(Mat{ldes[j].data(), n << lb, widths[j]});
    }
    return lmcs_build(ms);
  };
  // preprocessed tree: the AIRs that declare preprocessed columns, in proof order (preprocessed.rs:100-133)
  std::vector<int> prep_of(n_airs, -1);  // proof position j -> index in the preprocessed tree
  std::vector<std::vector<uint64_t>> prep_lde;
  std::vector<size_t> prep_w;
  std::unique_ptr<LmcsTree> prep_tree;
  int prep_depth = 0;
  {
    std::vector<Mat> ms;
    for (size_t j = 0; j < n_airs; j++) {
      const Air& a = in.airs[order[j]];
      if (!a.preprocessed_width) continue;
      if (in.preprocessed.size() != n_airs || !in.preprocessed[order[j]]) throw std::runtime_error("AIR declares preprocessed columns but none were supplied");
      int lh = in.log_heights[order[j]];
      size_t n = (size_t)1 << lh;
      prep_of[j] = (int)prep_lde.size();
      prep_lde.push_back(coset_lde_matrix_bitrev(in.preprocessed[order[j]], n, a.preprocessed_width, lb, canonical_lde_shift(lh + lb)));
      prep_w.push_back(a.preprocessed_width);
      prep_depth = lh + lb;
    }
    for (size_t k = 0; k < prep_lde.size(); k++) ms.push_back(Mat{prep_lde[k].data(), prep_lde[k].size() / prep_w[k], prep_w[k]});
    if (!ms.empty()) prep_tree.reset(new LmcsTree(lmcs_build(ms)));
  }
  std::vector<const uint64_t*> main_mats;
  std::vector<size_t> main_w;
  for (size_t j = 0; j < n_airs; j++) {
    main_mats.push_back(in.traces[order[j]]);
    main_w.push_back(in.airs[order[j]].main_width);
  }
  std::vector<std::vector<uint64_t>> main_lde, aux_lde;
  LmcsTree main_tree = commit(main_mats, main_w, main_lde);
  ch.send_commitment(main_tree.root());

  tm.lap("commit main");
  // ---- 2. randomness, aux traces -----------------------------------------------------------
  size_t max_rand = 0;
  for (auto& a : in.airs) max_rand = std::max(max_rand, a.num_randomness);
  std::vector<E2> randomness;
  for (size_t i = 0; i < max_rand; i++) randomness.push_back(ch.ch.sample_ef());
  std::vector<uint64_t> rand_flat;
  for (E2 r : randomness) { rand_flat.push_back(r.c0); rand_flat.push_back(r.c1); }
  std::vector<std::vector<uint64_t>> aux_tr(n_airs), aux_vals(n_airs);  // instance order
  for (size_t i = 0; i < n_airs; i++) {
    const Air& a = in.airs[i];
    size_t n = (size_t)1 << in.log_heights[i];
    aux_tr[i].assign(n * 2 * a.aux_width, 0);
    aux_vals[i].assign(2 * a.num_aux_values, 0);
    if (in.aux_builder) {
      int rc = in.aux_builder(in.aux_user, (int)i, rand_flat.data(), aux_tr[i].data(), aux_vals[i].data());
      if (rc) throw std::runtime_error("aux trace builder / external assertion failed");
    }
  }
  std::vector<const uint64_t*> aux_mats;
  std::vector<size_t> aux_w;
  for (size_t j = 0; j < n_airs; j++) {
    aux_mats.push_back(aux_tr[order[j]].data());
    aux_w.push_back(2 * in.airs[order[j]].aux_width);
  }
  LmcsTree aux_tree = commit(aux_mats, aux_w, aux_lde);
  ch.send_commitment(aux_tree.root());
  for (size_t j = 0; j < n_airs; j++)
    for (uint64_t v : aux_vals[order[j]]) ch.send_field(v);

  tm.lap("commit aux");
  // ---- 3. alpha, beta ----------------------------------------------------------------------
  E2 alpha = ch.ch.sample_ef();
  E2 beta = ch.ch.sample_ef();

  // ---- 4. constraint evaluation + accumulation ---------------------------------------------
  std::vector<E2> acc;
  for (size_t j = 0; j < n_airs; j++) {
    const Air& air = in.airs[order[j]];
    const int lh = in.log_heights[order[j]];
    const size_t n = (size_t)1 << lh;
    const int logDj = air.log_quotient_degree;
    const size_t Dj = (size_t)1 << logDj;
    const int Lj = lh + lb;
    const size_t nD = n * Dj, nBj = n << lb;
    const size_t step = B / Dj;
    const uint64_t gj = canonical_lde_shift(Lj);
    const uint64_t wJ = two_adic_generator(lh + logDj);
    const uint64_t wh_inv = finv(two_adic_generator(lh));
    // coset points, selectors (domain.rs:698-735)
    std::vector<uint64_t> xs(nD), d_first(nD), d_last(nD);
    uint64_t x = gj;
    for (size_t i = 0; i < nD; i++) {
      xs[i] = x;
      d_first[i] = fsub(x, 1);
      d_last[i] = fsub(x, wh_inv);
      x = fmul(x, wJ);
    }
    batch_inverse(d_first);
    batch_inverse(d_last);
    std::vector<uint64_t> zh(Dj), inv_zh(Dj);
    uint64_t s_pow_n = fexp_pow2(gj, lh), wd = two_adic_generator(logDj), t = 1;
    for (size_t k = 0; k < Dj; k++) {
      zh[k] = fsub(fmul(s_pow_n, t), 1);
      inv_zh[k] = zh[k];
      t = fmul(t, wd);
    }
    batch_inverse(inv_zh);
    // periodic LDE table (periodic.rs:49-77)
    size_t Pm = air.max_period();
    std::vector<std::vector<uint64_t>> ptab(air.periodic.size());
    if (Pm) {
      int logP = log2_strict(Pm);
      if (logP > lh) throw std::runtime_error("periodic column longer than trace");
      uint64_t pshift = fexp_pow2(gj, lh - logP);
      for (size_t c = 0; c < air.periodic.size(); c++) {
        std::vector<uint64_t> rep(Pm);
        for (size_t r = 0; r < Pm; r++) rep[r] = air.periodic[c][r % air.periodic[c].size()];
        ptab[c] = coset_lde_col(rep, logDj, pshift);  // natural order, Pm*Dj values
      }
    }
    const uint64_t* M = main_lde[j].data();
    const uint64_t* A = aux_lde[j].data();
    const uint64_t* PP = prep_of[j] >= 0 ? prep_lde[prep_of[j]].data() : nullptr;
    const size_t pw = air.preprocessed_width;
    const size_t mw = air.main_width, aw2 = 2 * air.aux_width;
    std::vector<E2> aux_values(air.num_aux_values);
    for (size_t k = 0; k < air.num_aux_values; k++) aux_values[k] = E2{aux_vals[order[j]][2 * k], aux_vals[order[j]][2 * k + 1]};
    std::vector<E2> q(nD);
#pragma omp parallel
    {
      std::vector<E2> scratch, ac(air.aux_width), an(air.aux_width), per(air.periodic.size());
#pragma omp for schedule(static)
      for (long ii = 0; ii < (long)nD; ii++) {
        size_t i = (size_t)ii;
        size_t r_cur = bitrev((uint32_t)((i * step) % nBj), Lj);
        size_t r_nxt = bitrev((uint32_t)(((i + Dj) * step) % nBj), Lj);
        for (size_t c = 0; c < air.aux_width; c++) {
          ac[c] = E2{A[r_cur * aw2 + 2 * c], A[r_cur * aw2 + 2 * c + 1]};
          an[c] = E2{A[r_nxt * aw2 + 2 * c], A[r_nxt * aw2 + 2 * c + 1]};
        }
        for (size_t c = 0; c < per.size(); c++) per[c] = e2(ptab[c][i % (Pm * Dj)]);
        EvalEnv e;
        e.main_cur = M + r_cur * mw; e.main_next = M + r_nxt * mw;
        if (PP) { e.prep_cur = PP + r_cur * pw; e.prep_next = PP + r_nxt * pw; }
        e.aux_cur = ac.data(); e.aux_next = an.data();
        e.publics = in.publics.data(); e.periodic = per.data();
        uint64_t z_h = zh[i % Dj];
        e.is_first = e2(fmul(z_h, d_first[i]));
        e.is_last = e2(fmul(z_h, d_last[i]));
        e.is_transition = e2(fsub(xs[i], wh_inv));
        e.randomness = randomness.data(); e.aux_values = aux_values.data();
        E2 folded = dag_fold(air, e, nullptr, alpha, scratch);
        q[i] = emulf(folded, inv_zh[i % Dj]);
      }
    }
    // upsample D_j -> D (quotient.rs:45-58): LDE of the EF vector, same shift
    if (logDj < logD) {
      int ab = logD - logDj;
      std::vector<uint64_t> c0(nD), c1(nD);
      for (size_t i = 0; i < nD; i++) { c0[i] = q[i].c0; c1[i] = q[i].c1; }
      std::vector<uint64_t> e0 = coset_lde_col(c0, ab, 1), e1 = coset_lde_col(c1, ab, 1);
      q.resize(nD << ab);
      for (size_t i = 0; i < q.size(); i++) q[i] = E2{e0[i], e1[i]};
    }
    // cyclic_extend_and_accumulate (quotient.rs:83-111)
    if (acc.empty()) acc = q;
    else {
      size_t old = acc.size();
      std::vector<E2> nacc(q.size());
      for (size_t i = 0; i < q.size(); i++) nacc[i] = eadd(emul(acc[i % old], beta), q[i]);
      acc.swap(nacc);
    }
  }
  const size_t N = (size_t)1 << log_n_max;
  if (acc.size() != N * D) throw std::runtime_error("internal: accumulator size");

  tm.lap("constraints");
  // ---- 5. commit quotient (quotient.rs:143-217) ---------------------------------------------
  std::vector<uint64_t> quot_lde(NB * 2 * D);
  {
    const uint64_t wJ_inv = finv(two_adic_generator(log_n_max + logD));
#pragma omp parallel for schedule(dynamic)
    for (long col = 0; col < (long)(2 * D); col++) {
      size_t tt = (size_t)col / 2, e = (size_t)col % 2;
      std::vector<uint64_t> v(N);
      for (size_t r = 0; r < N; r++) v[r] = e ? acc[r * D + tt].c1 : acc[r * D + tt].c0;
      dft_inplace(v.data(), N, true);
      uint64_t base = fpow(wJ_inv, tt), s = 1;
      for (size_t k = 0; k < N; k++) {
        v[k] = fmul(v[k], s);
        s = fmul(s, base);
      }
      v.resize(NB, 0);
      dft_inplace(v.data(), NB, false);
      for (size_t i = 0; i < NB; i++) quot_lde[(size_t)bitrev((uint32_t)i, L) * 2 * D + col] = v[i];
    }
  }
  LmcsTree quot_tree = lmcs_build({Mat{quot_lde.data(), NB, 2 * D}});
  ch.send_commitment(quot_tree.root());

  tm.lap("commit quotient");
  // ---- 6. OOD point (domain.rs:539-553) -----------------------------------------------------
  E2 z;
  const uint64_t g_inv = finv(g);
  for (;;) {
    z = ch.ch.sample_ef();
    if (eiszero(z)) continue;
    if (eeq(eexp_pow2(z, log_n_max), e2(1))) continue;
    if (eeq(eexp_pow2(emulf(z, g_inv), L), e2(1))) continue;
    break;
  }
  const uint64_t wH = two_adic_generator(log_n_max);
  E2 zs[2] = {z, emulf(z, wH)};

  // ---- 7. PCS open: DEEP --------------------------------------------------------------------
  struct OpenMat {
    const uint64_t* v;
    size_t h, w;
  };
  std::vector<OpenMat> mats;
  for (size_t k = 0; k < prep_lde.size(); k++) mats.push_back({prep_lde[k].data(), prep_lde[k].size() / prep_w[k], prep_w[k]});  // group order: [preprocessed?, main, aux, quotient]
  for (size_t j = 0; j < n_airs; j++) mats.push_back({main_lde[j].data(), main_lde[j].size() / main_w[j], main_w[j]});
  for (size_t j = 0; j < n_airs; j++)  // an AIR without aux columns keeps a width-0 slot of the height of its main LDE
    mats.push_back({aux_lde[j].data(), aux_w[j] ? aux_lde[j].size() / aux_w[j] : main_lde[j].size() / main_w[j], aux_w[j]});
  mats.push_back({quot_lde.data(), NB, 2 * D});
  size_t W = 0;
  for (auto& m : mats) W += align8(m.w);
  // OOD evaluations f(z_k^{r_m})  (interpolate.rs:127-204; value is unique)
  std::vector<E2> evals[2];
  evals[0].assign(W, e2(0));
  evals[1].assign(W, e2(0));
  {
    size_t off = 0;
    for (auto& m : mats) {
      int lhm = log2_strict(m.h) - lb;
      size_t nm = (size_t)1 << lhm;
      int lift = L - log2_strict(m.h);
      uint64_t gm_inv = finv(canonical_lde_shift(lhm + lb));
      E2 u[2] = {emulf(eexp_pow2(zs[0], lift), gm_inv), emulf(eexp_pow2(zs[1], lift), gm_inv)};
#pragma omp parallel for schedule(dynamic)
      for (long c = 0; c < (long)m.w; c++) {
        std::vector<uint64_t> col(nm);
        for (size_t r = 0; r < nm; r++) col[r] = m.v[(size_t)bitrev((uint32_t)r, lhm) * m.w + c];
        dft_inplace(col.data(), nm, true);
        evals[0][off + c] = horner_ef(col, u[0]);
        evals[1][off + c] = horner_ef(col, u[1]);
      }
      off += align8(m.w);
    }
  }
  tm.lap("ood evals");
  for (int k = 0; k < 2; k++)
    for (E2 v : evals[k]) ch.send_ef(v);
  ch.grind(pp.deep_pow_bits);
  E2 alpha_d = ch.ch.sample_ef();
  E2 beta_d = ch.ch.sample_ef();
  E2 fred[2];
  for (int k = 0; k < 2; k++) {
    E2 a = e2(0);
    for (size_t i = 0; i < W; i++) a = eadd(emul(a, alpha_d), evals[k][i]);
    fred[k] = a;
  }
  // negated Horner coefficients: column i gets -alpha^(W-1-i)
  std::vector<E2> negc(W);
  {
    E2 pw = e2(P - 1);
    for (size_t i = W; i-- > 0;) {
      negc[i] = pw;
      pw = emul(pw, alpha_d);
    }
  }
  std::vector<E2> ev(NB);
  {
    const uint64_t wK = two_adic_generator(L);
    std::vector<uint64_t> xs(NB);
    uint64_t x = g;
    for (size_t i = 0; i < NB; i++) {
      xs[bitrev((uint32_t)i, L)] = x;
      x = fmul(x, wK);
    }
#pragma omp parallel for schedule(static)
    for (long pp_ = 0; pp_ < (long)NB; pp_++) {
      size_t p = (size_t)pp_;
      E2 neg = e2(0);
      size_t off = 0;
      for (auto& m : mats) {
        int sh = L - log2_strict(m.h);
        const uint64_t* row = m.v + (p >> sh) * m.w;
        for (size_t c = 0; c < m.w; c++) neg = eadd(neg, emulf(negc[off + c], row[c]));
        off += align8(m.w);
      }
      E2 q0 = einv(esub(zs[0], e2(xs[p]))), q1 = einv(esub(zs[1], e2(xs[p])));
      E2 r = emul(q0, eadd(fred[0], neg));
      r = eadd(r, emul(emul(beta_d, q1), eadd(fred[1], neg)));
      ev[p] = r;
    }
  }

  tm.lap("deep");
  // ---- 8. FRI commit phase ------------------------------------------------------------------
  const int la = pp.log_folding_arity;
  const size_t arity = (size_t)1 << la;
  const int log_fpd = fri_log_final_poly_degree(pp, L);
  const size_t final_domain = ((size_t)1 << log_fpd) << lb;
  std::vector<LmcsTree> fri_trees;
  std::vector<std::vector<uint64_t>> fri_mats;
  int logn = L;
  while (((size_t)1 << logn) > final_domain) {
    size_t n = (size_t)1 << logn, rows = n >> la;
    fri_mats.emplace_back(rows * arity * 2);
    std::vector<uint64_t>& fm = fri_mats.back();
    for (size_t i = 0; i < n; i++) { fm[2 * i] = ev[i].c0; fm[2 * i + 1] = ev[i].c1; }
    fri_trees.push_back(lmcs_build({Mat{fm.data(), rows, arity * 2}}));
    ch.send_commitment(fri_trees.back().root());
    ch.grind(pp.folding_pow_bits);
    E2 b = ch.ch.sample_ef();
    uint64_t w_inv = finv(two_adic_generator(logn));
    std::vector<E2> next(rows);
#pragma omp parallel for schedule(static)
    for (long k = 0; k < (long)rows; k++) {
      uint64_t s_inv = fpow(w_inv, bitrev((uint32_t)k, logn - la));
      next[k] = fri_fold_row(&ev[(size_t)k * arity], la, s_inv, b);
    }
    ev.swap(next);
    logn -= la;
  }
  {
    size_t fpd = (size_t)1 << log_fpd;
    std::vector<uint64_t> c0(fpd), c1(fpd);
    for (size_t i = 0; i < fpd; i++) {
      c0[bitrev((uint32_t)i, log_fpd)] = ev[i].c0;
      c1[bitrev((uint32_t)i, log_fpd)] = ev[i].c1;
    }
    dft_inplace(c0.data(), fpd, true);
    dft_inplace(c1.data(), fpd, true);
    for (size_t i = fpd; i-- > 0;) ch.send_ef(E2{c0[i], c1[i]});
  }

  tm.lap("fri");
  // ---- 9. queries ---------------------------------------------------------------------------
  ch.grind(pp.query_pow_bits);
  std::vector<size_t> idx;
  for (int i = 0; i < pp.num_queries; i++) idx.push_back(ch.ch.sample_bits(L));
  std::sort(idx.begin(), idx.end());
  idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
  if (prep_tree) {  // a shorter tree is virtually lifted: indices fold to its depth by their low bits (tree_indices.rs:72-84)
    std::vector<size_t> pidx(idx);
    const size_t mask = ((size_t)1 << prep_depth) - 1;
    for (auto& i : pidx) i &= mask;
    std::sort(pidx.begin(), pidx.end());
    pidx.erase(std::unique(pidx.begin(), pidx.end()), pidx.end());
    std::vector<uint64_t> f;
    std::vector<Digest> c;
    lmcs_prove_batch(*prep_tree, pidx, lmcs_alignment(), f, c);
    ch.hint_fields(f);
    ch.hint_commitments(c);
  }
  for (const LmcsTree* t : {&main_tree, &aux_tree, &quot_tree}) {
    std::vector<uint64_t> f;
    std::vector<Digest> c;
    lmcs_prove_batch(*t, idx, lmcs_alignment(), f, c);
    ch.hint_fields(f);
    ch.hint_commitments(c);
  }
  int depth = L;
  for (auto& t : fri_trees) {
    depth -= la;
    size_t mask = ((size_t)1 << depth) - 1;
    for (auto& i : idx) i &= mask;
    std::sort(idx.begin(), idx.end());
    idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
    std::vector<uint64_t> f;
    std::vector<Digest> c;
    lmcs_prove_batch(t, idx, 1, f, c);
    ch.hint_fields(f);
    ch.hint_commitments(c);
  }
  tm.lap("queries");
  Proof pr;
  for (int lh : in.log_heights) pr.log_trace_heights.push_back((uint8_t)lh);
  pr.digest = ch.ch.finalize();
  pr.fields = std::move(ch.fields);
  pr.commitments = std::move(ch.commitments);
  return pr;
}

// ===============================================================================================
// Verifier
// ===============================================================================================
struct VerifyError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// lmcs/config.rs:172-211 open_batch: read opened (aligned) rows + missing siblings, rebuild root.
// Returns rows per index (concatenated aligned rows of every matrix).
static inline std::map<size_t, std::vector<uint64_t>> lmcs_verify_batch(VerifierTranscript& ch, const Digest& root,
                                                                       const std::vector<size_t>& aligned_widths,
                                                                       const std::vector<size_t>& sorted_idx, int depth) {
  std::map<size_t, std::vector<uint64_t>> rows;
  std::map<size_t, Digest> level;
  size_t tot = 0;
  for (size_t w : aligned_widths) tot += w;
  for (size_t i : sorted_idx) {
    std::vector<uint64_t> r(tot);
    for (auto& x : r) x = ch.hint_field();
    size_t off = 0;
    if (g_lmcs == LMCS_BLAKE3) {
      Digest st{0, 0, 0, 0};
      for (size_t w : aligned_widths) {
        st = b3_absorb(st, r.data() + off, w);
        off += w;
      }
      level[i] = st;
    } else if (g_lmcs == LMCS_KECCAK) {
      uint64_t st[25] = {0};
      for (size_t w : aligned_widths) {
        keccak_absorb(st, r.data() + off, w);
        off += w;
      }
      level[i] = Digest{st[0], st[1], st[2], st[3]};
    } else {
      uint64_t st[12] = {0};
      for (size_t w : aligned_widths) {
        sponge_absorb(st, r.data() + off, w);
        off += w;
      }
      level[i] = Digest{st[0], st[1], st[2], st[3]};
    }
    rows[i] = std::move(r);
  }
  for (int d = depth; d > 0; d--) {
    std::map<size_t, Digest> next;
    for (auto it = level.begin(); it != level.end();) {
      size_t node = it->first, sib = node ^ 1;
      Digest me = it->second, other;
      auto nx = std::next(it);
      if (nx != level.end() && nx->first == sib) {
        other = nx->second;
        it = std::next(nx);
      } else {
        other = ch.hint_commitment();
        it = nx;
      }
      Digest parent;
      if (g_lmcs == LMCS_BLAKE3) parent = (node & 1) ? b3_compress(other, me) : b3_compress(me, other);
      else if (g_lmcs == LMCS_KECCAK) parent = (node & 1) ? keccak_compress(other, me) : keccak_compress(me, other);
      else if (node & 1) compress(other.data(), me.data(), parent.data());
      else compress(me.data(), other.data(), parent.data());
      next[node >> 1] = parent;
    }
    level.swap(next);
  }
  if (level.size() != 1 || level.begin()->second != root) throw VerifyError("LMCS: root mismatch");
  return rows;
}

struct VerifierInput {
  PcsParams params;
  std::vector<Air> airs;
  std::vector<uint64_t> publics;
  Challenger challenger;  // preprocessed commitment (if any) already observed, like the statement
  bool has_preprocessed = false;
  Digest preprocessed_root{};
  // Statement::eval_external (crates/lifted-air/src/statement.rs:94-108): cross-AIR assertions over the challenges,
  // the aux values and the log heights, all in INSTANCE order; every returned value must be zero
  // (verifier/mod.rs:488-501).  Null = the default MultiAir (no assertions).
  ExternalAssertions external = nullptr;
  void* external_user = nullptr;
};

static inline Digest verify(const VerifierInput& in, const Proof& proof) {
  const PcsParams& pp = in.params;
  const int lb = pp.log_blowup;
  const size_t n_airs = in.airs.size();
  if (const char* e = pcs_params_error(pp)) throw VerifyError(e);
  if (proof.log_trace_heights.size() != n_airs) throw VerifyError("trace count mismatch");
  std::vector<int> lhs(proof.log_trace_heights.begin(), proof.log_trace_heights.end());
  for (size_t i = 0; i < n_airs; i++) {
    if (lhs[i] == 0) throw VerifyError("trace too small");
    if (((size_t)1 << lhs[i]) < in.airs[i].max_period()) throw VerifyError("trace shorter than periodic column");
  }
  std::vector<int> order = proof_order(lhs);
  const int log_n_max = lhs[order.back()];
  const int L = log_n_max + lb;
  if (L > TWO_ADICITY) throw VerifyError("LDE order too large");
  const uint64_t g = canonical_lde_shift(L);
  VerifierTranscript ch;
  ch.ch = in.challenger;
  ch.ch.observe((uint64_t)n_airs);
  for (int lh : lhs) ch.ch.observe((uint64_t)lh);
  ch.f = proof.fields.data(); ch.nf = proof.fields.size();
  ch.c = proof.commitments.data(); ch.nc = proof.commitments.size();

  int logD = 0;
  for (auto& a : in.airs) logD = std::max(logD, a.log_quotient_degree);
  if (logD > lb) throw VerifyError("constraint degree too high");
  const size_t D = (size_t)1 << logD;

  Digest main_root = ch.receive_commitment();
  size_t max_rand = 0;
  for (auto& a : in.airs) max_rand = std::max(max_rand, a.num_randomness);
  std::vector<E2> randomness;
  for (size_t i = 0; i < max_rand; i++) randomness.push_back(ch.ch.sample_ef());
  Digest aux_root = ch.receive_commitment();
  std::vector<std::vector<E2>> aux_values(n_airs);  // proof order
  for (size_t j = 0; j < n_airs; j++)
    for (size_t k = 0; k < in.airs[order[j]].num_aux_values; k++) aux_values[j].push_back(ch.receive_ef());
  E2 alpha = ch.ch.sample_ef();
  E2 beta = ch.ch.sample_ef();
  Digest quot_root = ch.receive_commitment();
  E2 z;
  const uint64_t g_inv = finv(g);
  for (;;) {
    z = ch.ch.sample_ef();
    if (eiszero(z)) continue;
    if (eeq(eexp_pow2(z, log_n_max), e2(1))) continue;
    if (eeq(eexp_pow2(emulf(z, g_inv), L), e2(1))) continue;
    break;
  }
  E2 zs[2] = {z, emulf(z, two_adic_generator(log_n_max))};

  // commitment groups: aligned widths (pcs/verifier.rs verify_aligned); [preprocessed?, main, aux, quotient] (proof.rs:326-375)
  std::vector<std::vector<size_t>> groups;
  std::vector<Digest> roots;
  std::vector<int> depths;
  bool any_prep = false;
  for (auto& a : in.airs) any_prep |= a.preprocessed_width > 0;
  if (any_prep != in.has_preprocessed) throw VerifyError("preprocessed commitment presence mismatch");
  if (any_prep) {
    groups.emplace_back();
    int dp = 0;
    for (size_t j = 0; j < n_airs; j++)
      if (in.airs[order[j]].preprocessed_width) {
        groups.back().push_back(align8(in.airs[order[j]].preprocessed_width));
        dp = std::max(dp, lhs[order[j]] + lb);
      }
    roots.push_back(in.preprocessed_root);
    depths.push_back(dp);
  }
  const size_t g_main = groups.size();
  groups.emplace_back(); groups.emplace_back(); groups.emplace_back();
  for (size_t j = 0; j < n_airs; j++) groups[g_main].push_back(align8(in.airs[order[j]].main_width));
  for (size_t j = 0; j < n_airs; j++) groups[g_main + 1].push_back(align8(2 * in.airs[order[j]].aux_width));
  groups[g_main + 2].push_back(align8(2 * D));
  roots.push_back(main_root); roots.push_back(aux_root); roots.push_back(quot_root);
  depths.push_back(L); depths.push_back(L); depths.push_back(L);
  size_t W = 0;
  for (auto& gset : groups)
    for (size_t w : gset) W += w;
  // DEEP oracle (deep/verifier.rs)
  std::vector<E2> evals[2];
  for (int k = 0; k < 2; k++)
    for (size_t i = 0; i < W; i++) evals[k].push_back(ch.receive_ef());
  ch.grind(pp.deep_pow_bits);
  E2 alpha_d = ch.ch.sample_ef();
  E2 beta_d = ch.ch.sample_ef();
  E2 fred[2];
  for (int k = 0; k < 2; k++) {
    E2 a = e2(0);
    for (size_t i = 0; i < W; i++) a = eadd(emul(a, alpha_d), evals[k][i]);
    fred[k] = a;
  }
  // FRI oracle (fri/verifier.rs)
  const int la = pp.log_folding_arity;
  const size_t arity = (size_t)1 << la;
  const int rounds = fri_num_rounds(pp, L);
  std::vector<Digest> fri_roots;
  std::vector<E2> fri_betas;
  for (int r = 0; r < rounds; r++) {
    fri_roots.push_back(ch.receive_commitment());
    ch.grind(pp.folding_pow_bits);
    fri_betas.push_back(ch.ch.sample_ef());
  }
  const size_t fpd = (size_t)1 << fri_log_final_poly_degree(pp, L);
  std::vector<E2> final_poly;  // descending degree
  for (size_t i = 0; i < fpd; i++) final_poly.push_back(ch.receive_ef());
  ch.grind(pp.query_pow_bits);
  std::vector<size_t> idx;
  for (int i = 0; i < pp.num_queries; i++) idx.push_back(ch.ch.sample_bits(L));
  std::sort(idx.begin(), idx.end());
  idx.erase(std::unique(idx.begin(), idx.end()), idx.end());

  // DEEP open_batch
  std::map<size_t, E2> reduced;
  for (size_t i : idx) reduced[i] = e2(0);
  for (size_t gi = 0; gi < groups.size(); gi++) {
    std::vector<size_t> gidx(idx);
    const size_t mask = ((size_t)1 << depths[gi]) - 1;
    for (auto& i : gidx) i &= mask;
    std::sort(gidx.begin(), gidx.end());
    gidx.erase(std::unique(gidx.begin(), gidx.end()), gidx.end());
    auto rows = lmcs_verify_batch(ch, roots[gi], groups[gi], gidx, depths[gi]);
    for (auto& kv : reduced)
      for (uint64_t v : rows[kv.first & mask]) kv.second = eadd(emul(kv.second, alpha_d), e2(v));
  }
  std::map<size_t, E2> fevals;
  const uint64_t wK = two_adic_generator(L);
  for (auto& kv : reduced) {
    E2 x = e2(fmul(g, fpow(wK, kv.first)));
    E2 acc = e2(0), bp = e2(1);
    for (int k = 0; k < 2; k++) {
      E2 den = esub(zs[k], x);
      if (eiszero(den)) throw VerifyError("eval point on domain");
      acc = eadd(acc, emul(emul(bp, esub(fred[k], kv.second)), einv(den)));
      bp = emul(bp, beta_d);
    }
    fevals[kv.first] = acc;
  }
  // FRI test_low_degree
  int logn = L;
  uint64_t gen_inv = finv(two_adic_generator(L));
  for (int r = 0; r < rounds; r++) {
    int logf = logn - la;
    size_t fsize = (size_t)1 << logf;
    for (auto& i : idx) i &= fsize - 1;
    std::sort(idx.begin(), idx.end());
    idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
    auto rows = lmcs_verify_batch(ch, fri_roots[r], {arity * 2}, idx, logf);
    std::map<size_t, E2> next;
    for (auto& kv : fevals) {
      size_t row_idx = kv.first & (fsize - 1);
      size_t position = bitrev((uint32_t)(kv.first >> logf), la);
      auto it = rows.find(row_idx);
      if (it == rows.end()) throw VerifyError("FRI: invalid opening");
      std::vector<E2> row(arity);
      for (size_t k = 0; k < arity; k++) row[k] = E2{it->second[2 * k], it->second[2 * k + 1]};
      if (!eeq(row[position], kv.second)) throw VerifyError("FRI: evaluation mismatch");
      uint64_t s_inv = fpow(gen_inv, row_idx);
      next[row_idx] = fri_fold_row(row.data(), la, s_inv, fri_betas[r]);
    }
    fevals.swap(next);
    logn = logf;
    gen_inv = fexp_pow2(gen_inv, la);
  }
  {
    uint64_t gen = two_adic_generator(logn);
    for (auto& kv : fevals) {
      uint64_t x = fpow(gen, kv.first);
      E2 acc = e2(0);
      for (E2 c : final_poly) acc = eadd(emulf(acc, x), c);
      if (!eeq(acc, kv.second)) throw VerifyError("FRI: final polynomial mismatch");
    }
  }

  // constraint identity (verifier/mod.rs step 9-12)
  E2 accumulated = e2(0);
  size_t off_prep = 0, off_main = 0, off_aux = 0;
  for (size_t j = 0; j < n_airs; j++) off_main += in.airs[order[j]].preprocessed_width ? align8(in.airs[order[j]].preprocessed_width) : 0;
  off_aux = off_main;
  for (size_t j = 0; j < n_airs; j++) off_aux += align8(in.airs[order[j]].main_width);
  size_t off_quot = off_aux;
  for (size_t j = 0; j < n_airs; j++) off_quot += align8(2 * in.airs[order[j]].aux_width);
  std::vector<E2> scratch;
  for (size_t j = 0; j < n_airs; j++) {
    const Air& air = in.airs[order[j]];
    int lh = lhs[order[j]];
    std::vector<E2> mc(air.main_width), mn(air.main_width), ac(air.aux_width), an(air.aux_width), per;
    std::vector<E2> pc(air.preprocessed_width), pn(air.preprocessed_width);
    for (size_t c = 0; c < air.preprocessed_width; c++) { pc[c] = evals[0][off_prep + c]; pn[c] = evals[1][off_prep + c]; }
    if (air.preprocessed_width) off_prep += align8(air.preprocessed_width);
    for (size_t c = 0; c < air.main_width; c++) { mc[c] = evals[0][off_main + c]; mn[c] = evals[1][off_main + c]; }
    for (size_t c = 0; c < air.aux_width; c++) {
      // EF value of an EF column from its two base-column openings: v = f0(z) + x*f1(z)
      auto rec = [&](int k) {
        E2 a = evals[k][off_aux + 2 * c], b = evals[k][off_aux + 2 * c + 1];
        return eadd(a, emul(b, E2{0, 1}));
      };
      ac[c] = rec(0); an[c] = rec(1);
    }
    off_main += align8(air.main_width);
    off_aux += align8(2 * air.aux_width);
    E2 y = eexp_pow2(z, log_n_max - lh);
    Selectors s = selectors_at(y, lh);
    for (auto& col : air.periodic) {
      int logp = log2_strict(col.size());
      per.push_back(periodic_eval(col, eexp_pow2(z, log_n_max - logp)));
    }
    EvalEnv e;
    e.main_cur = nullptr; e.main_next = nullptr;
    e.aux_cur = ac.data(); e.aux_next = an.data();
    e.publics = in.publics.data(); e.periodic = per.data();
    e.is_first = s.is_first; e.is_last = s.is_last; e.is_transition = s.is_transition;
    e.randomness = randomness.data(); e.aux_values = aux_values[j].data();
    EvalEnvExt em{mc.data(), mn.data(), pc.data(), pn.data()};
    E2 folded = dag_fold(air, e, &em, alpha, scratch);
    accumulated = eadd(emul(accumulated, beta), folded);
  }
  // external assertions (verifier/mod.rs:488-501): aux values back in instance order
  if (in.external) {
    std::vector<std::vector<uint64_t>> flat(n_airs);
    for (size_t j = 0; j < n_airs; j++)
      for (E2 v : aux_values[j]) { flat[order[j]].push_back(v.c0); flat[order[j]].push_back(v.c1); }
    std::vector<const uint64_t*> ptrs(n_airs);
    std::vector<size_t> cnt(n_airs);
    std::vector<uint8_t> lh8(n_airs);
    for (size_t i = 0; i < n_airs; i++) { ptrs[i] = flat[i].data(); cnt[i] = flat[i].size() / 2; lh8[i] = (uint8_t)lhs[i]; }
    std::vector<uint64_t> rflat;
    for (E2 r : randomness) { rflat.push_back(r.c0); rflat.push_back(r.c1); }
    std::vector<uint64_t> out(2 * 64, 0);
    const int k = in.external(in.external_user, rflat.data(), randomness.size(), ptrs.data(), cnt.data(), lh8.data(), (int)n_airs,
                              out.data(), 64);
    if (k < 0 || k > 64) throw VerifyError("external assertions could not be evaluated");
    for (int a = 0; a < k; a++)
      if (out[2 * a] % P || out[2 * a + 1] % P) throw VerifyError("external assertion " + std::to_string(a) + " failed");
  }
  // reconstruct_quotient (domain.rs:773-794)
  {
    std::vector<E2> chunks(D);
    for (size_t t = 0; t < D; t++) {
      E2 a = evals[0][off_quot + 2 * t], b = evals[0][off_quot + 2 * t + 1];
      chunks[t] = eadd(a, emul(b, E2{0, 1}));
    }
    uint64_t omega_s = two_adic_generator(logD);
    E2 u = eexp_pow2(emulf(z, g_inv), log_n_max);
    E2 num = e2(0), den = e2(0);
    uint64_t wt = 1;
    for (size_t t = 0; t < D; t++) {
      E2 a_t = esub(u, e2(wt));
      E2 w_t = emulf(einv(a_t), wt);
      num = eadd(num, emul(w_t, chunks[t]));
      den = eadd(den, w_t);
      wt = fmul(wt, omega_s);
    }
    E2 qz = emul(num, einv(den));
    E2 van = esub(eexp_pow2(z, log_n_max), e2(1));
    if (!eeq(accumulated, emul(qz, van))) throw VerifyError("constraint mismatch");
  }
  if (!ch.exhausted()) throw VerifyError("transcript has trailing data");
  return ch.ch.finalize();
}

}  // namespace oracle
