// K4 / K5 / K12: constraint evaluation on the quotient coset, alpha-fold, 1/Z_H, and the
// cross-AIR beta accumulation, gfx950.
//
// Replaces crates/lifted-stark/src/prover/constraints/mod.rs:83-278 (evaluate_constraints_into),
// constraints/folder.rs:88-105 (finalize_constraints), domain.rs:698-750 (selectors,
// inv_vanishing_evals), prover/quotient.rs:83-111 (cyclic_extend_and_accumulate).
//
// Layout: the quotient coset gJ (size n*D) is the union of the D committed cosets
// jc = t*B/D of the coset-major LDE, so point (t, r) reads row r of coset jc of every column
// (unit stride across lanes) and its "next row" is r+1 of the same coset -- no bit-reversed
// gather (the reference's packed_row_bitrev.rs) exists here.  The accumulator is the N x 2D
// column-major base matrix the quotient commit consumes directly: acc[(2t+e)*n + r].
// The AIR is an interpreter program (air.hpp); its slot file lives in LDS, SoA across lanes.
// Roofline: HBM (reads the touched main/aux columns once per AIR, writes 16 B per point); for real
// AIRs with thousands of gates the interpreter is VALU bound.
#include "air.hpp"
#include "air_jit.hpp"
#include "gl.cuh"
#include "kernels.hpp"

struct QuotArgs {
  const AirIns* code;
  u32 n_ins, n_slots;
  const u64* main_lde;
  const u64* aux_lde;
  const u64* prep_lde;  // preprocessed columns of this AIR (or null)
  int log_n, log_cosets, log_d, log_dl, jc_shift;  // log_d: quotient degree; log_dl: quotient cosets on this rank
  u32 t0;                // global index of this rank's first quotient coset
  const u64* tw;         // w_n^k (k < n/2)
  const u64* coset_tab;  // [3][Dl]: x-coordinate of local coset t at r = 0, Z_H on it, 1/Z_H
  u64 wh_inv;
  const u64* inv_first;  // [D*n] 1/(x-1)      (null when the AIR never asks)
  const u64* inv_last;   // [D*n] 1/(x-w_H^-1)
  const u64* periodic;   // [n_periodic][periodic_rows], natural gJ index mod periodic_rows
  u32 periodic_rows;
  const u64* publics;
  const u64* randomness;  // EF pairs
  const u64* aux_values;  // EF pairs
  const u64* alpha_pows;  // EF pairs: alpha^(K-1-k) at index k
  const u64* acc_in;      // [2D][n_prev] or null
  int log_n_prev;
  e2 beta;
  u64* acc_out;           // [2D][n]
};

__device__ __forceinline__ u64 coset_point(const u64* tw, int log_n, u64 cx, size_t r) {
  const size_t half = ((size_t)1 << log_n) >> 1;
  u64 w = (r < half || half == 0) ? tw[half ? r : 0] : gl_neg(tw[r - half]);
  return gl_mul(cx, w);
}

__global__ __launch_bounds__(256) void k_eval_quotient(QuotArgs a) {
  extern __shared__ u64 slots[];
  const u32 T = blockDim.x, tid = threadIdx.x;
  const size_t n = (size_t)1 << a.log_n;
  const size_t D = (size_t)1 << a.log_d, Dl = (size_t)1 << a.log_dl;
  const size_t q = blockIdx.x * (size_t)T + tid;
  if (q >= n * Dl) return;
  const size_t t = q >> a.log_n, r = q & (n - 1);  // t: local quotient coset
  const size_t r_next = (r + 1) & (n - 1);
  const size_t jc = t << a.jc_shift;               // local index of the LDE coset it lives on
  const size_t B = (size_t)1 << a.log_cosets;
  const u64 x = coset_point(a.tw, a.log_n, a.coset_tab[t], r);
  e2 acc = e2_make(0);
  // selectors (domain.rs:698-735): Z_H(x)/(x-1), Z_H(x)/(x-w_H^-1), x - w_H^-1
  u64 sel_first = 0, sel_last = 0;
  if (a.inv_first) {
    sel_first = gl_mul(a.coset_tab[Dl + t], a.inv_first[q]);
    sel_last = gl_mul(a.coset_tab[Dl + t], a.inv_last[q]);
  }
  const u64 sel_trans = gl_sub(x, a.wh_inv);
#define SLOT0(s) slots[(size_t)(2 * (s)) * T + tid]
#define SLOT1(s) slots[(size_t)(2 * (s) + 1) * T + tid]
  auto fetch = [&](uint8_t kind, u32 idx, bool ext, u64 imm) -> e2 {
    switch (kind) {
      case OPK_SLOT: return e2{SLOT0(idx), ext ? SLOT1(idx) : 0};
      case DOP_CONST: return e2_make(imm);
      case DOP_MAIN: return e2_make(a.main_lde[(((size_t)(idx & 0x7FFFFFFFu) * B + jc) << a.log_n) + ((idx >> 31) ? r_next : r)]);
      case DOP_AUX: {
        const size_t rr = (idx >> 31) ? r_next : r, cc = idx & 0x7FFFFFFFu;
        return e2{a.aux_lde[(((size_t)(2 * cc) * B + jc) << a.log_n) + rr], a.aux_lde[(((size_t)(2 * cc + 1) * B + jc) << a.log_n) + rr]};
      }
      case DOP_PREP: return e2_make(a.prep_lde[(((size_t)(idx & 0x7FFFFFFFu) * B + jc) << a.log_n) + ((idx >> 31) ? r_next : r)]);
      case DOP_PUBLIC: return e2_make(a.publics[idx]);
      case DOP_PERIODIC: return e2_make(a.periodic[(size_t)idx * a.periodic_rows + ((r * D + a.t0 + t) % a.periodic_rows)]);
      case DOP_IS_FIRST: return e2_make(sel_first);
      case DOP_IS_LAST: return e2_make(sel_last);
      case DOP_IS_TRANSITION: return e2_make(sel_trans);
      case DOP_RANDOMNESS: return e2{a.randomness[2 * idx], a.randomness[2 * idx + 1]};
      default: return e2{a.aux_values[2 * idx], a.aux_values[2 * idx + 1]};  // DOP_AUX_VALUE
    }
  };
#pragma unroll 1
  for (u32 pc = 0; pc < a.n_ins; pc++) {
    const AirIns ins = a.code[pc];
    const bool a_ext = ins.ext & 1, b_ext = ins.ext & 2;
    const e2 va = fetch(ins.a_kind, ins.a, a_ext, ins.imm);
    if (ins.op == DOP_FOLD) {
      const e2 pw = e2{a.alpha_pows[2 * ins.b], a.alpha_pows[2 * ins.b + 1]};
      acc = e2_add(acc, a_ext ? e2_mul(pw, va) : e2_mulf(pw, va.c0));
      continue;
    }
    e2 v;
    if (ins.op == DOP_NEG) {
      v = a_ext ? e2_neg(va) : e2_make(gl_neg(va.c0));
    } else {
      const e2 vb = fetch(ins.b_kind, ins.b, b_ext, ins.imm);
      const bool ext = a_ext || b_ext;
      if (ins.op == DOP_ADD) v = ext ? e2_add(va, vb) : e2_make(gl_add(va.c0, vb.c0));
      else if (ins.op == DOP_SUB) v = ext ? e2_sub(va, vb) : e2_make(gl_sub(va.c0, vb.c0));
      else {  // MUL
        if (a_ext && b_ext) v = e2_mul(va, vb);
        else if (a_ext) v = e2_mulf(va, vb.c0);
        else if (b_ext) v = e2_mulf(vb, va.c0);
        else v = e2_make(gl_mul(va.c0, vb.c0));
      }
    }
    SLOT0(ins.dst) = v.c0;
    if (a_ext || b_ext) SLOT1(ins.dst) = v.c1;
  }
#undef SLOT0
#undef SLOT1
  e2 qv = e2_mulf(acc, a.coset_tab[2 * Dl + t]);  // * 1/Z_H
  if (a.acc_in) {
    const size_t n_prev = (size_t)1 << a.log_n_prev;
    const size_t rp = r & (n_prev - 1);
    e2 old = e2{a.acc_in[((2 * t) << a.log_n_prev) + rp], a.acc_in[((2 * t + 1) << a.log_n_prev) + rp]};
    qv = e2_add(e2_mul(old, a.beta), qv);
  }
  a.acc_out[((2 * t) << a.log_n) + r] = qv.c0;
  a.acc_out[((2 * t + 1) << a.log_n) + r] = qv.c1;
}

// After the compiled chunks (air_jit.cpp) left sum_k alpha^(K-1-k) C_k in `acc`: * 1/Z_H, + beta * previous AIRs.
__global__ void k_quot_finish(u64* __restrict__ acc, const u64* __restrict__ coset_tab, int log_n, int log_dl, const u64* __restrict__ acc_in,
                              int log_n_prev, e2 beta) {
  const size_t n = (size_t)1 << log_n, Dl = (size_t)1 << log_dl;
  const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (q >= n * Dl) return;
  const size_t t = q >> log_n, r = q & (n - 1);
  u64* p0 = acc + ((2 * t) << log_n) + r;
  u64* p1 = acc + ((2 * t + 1) << log_n) + r;
  e2 qv = e2_mulf(e2{*p0, *p1}, coset_tab[2 * Dl + t]);
  if (acc_in) {
    const size_t n_prev = (size_t)1 << log_n_prev, rp = r & (n_prev - 1);
    e2 old = e2{acc_in[((2 * t) << log_n_prev) + rp], acc_in[((2 * t + 1) << log_n_prev) + rp]};
    qv = e2_add(e2_mul(old, beta), qv);
  }
  *p0 = qv.c0;
  *p1 = qv.c1;
}

// 1/(x - 1) and 1/(x - w_H^-1) for every point of the quotient coset (batch inversion: 4 points per
// lane, one Fermat inversion per 8 denominators).
__global__ __launch_bounds__(256) void k_selector_inverses(const u64* tw, const u64* coset_tab, int log_n, int log_d, u64 wh_inv,
                                                           u64* inv_first, u64* inv_last) {
  const size_t n = (size_t)1 << log_n, total = n << log_d;
  const size_t base = (size_t)blockIdx.x * (4 * 256) + threadIdx.x;
  u64 d[8], pre[8];
  u64 run = 1;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    size_t q = base + (size_t)k * 256;
    u64 x = 2;  // harmless filler for out-of-range lanes
    if (q < total) x = coset_point(tw, log_n, coset_tab[q >> log_n], q & (n - 1));
    d[2 * k] = gl_sub(x, 1);
    d[2 * k + 1] = gl_sub(x, wh_inv);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    pre[k] = run;
    run = gl_mul(run, d[k]);
  }
  u64 inv = gl_inv(run);
#pragma unroll
  for (int k = 7; k >= 0; k--) {
    u64 v = gl_mul(inv, pre[k]);
    inv = gl_mul(inv, d[k]);
    size_t q = base + (size_t)(k >> 1) * 256;
    if (q < total) {
      if (k & 1) inv_last[q] = v;
      else inv_first[q] = v;
    }
  }
}

// PeriodicLde::build (prover/periodic.rs:49-77): every periodic column on the n * 2^log_d-point quotient coset, as `Pm * 2^log_d` values per column
// (Pm = the longest period; a column of period P repeats).  O(Pm^2 2^log_d) field operations per column on the host, and a function of the AIR
// and the domain only -- never of a challenge -- so it is computed once per (log_n, log_blowup, log_d) and kept in the mh_air (bounded).
static std::vector<u64> periodic_lde_table(const mh_air* air, int log_n, int log_blowup, int log_d) {
  const size_t Pm = air->max_period();
  const size_t Dg = (size_t)1 << log_d, prow = Pm ? Pm * Dg : 1;
  const std::array<int, 3> key{log_n, log_blowup, log_d};
  {
    std::lock_guard<std::mutex> lk(air->ptab_mu);
    auto it = air->ptab_cache.find(key);
    if (it != air->ptab_cache.end()) return it->second;
  }
  std::vector<u64> ptab(std::max<size_t>(1, air->periodic.size() * prow), 0);
  if (Pm) {
    int logP = 0;
    while (((size_t)1 << logP) < Pm) logP++;
    MH_REQUIRE(logP <= log_n, "periodic column longer than the trace");
    const u64 g = gl_lde_shift(log_n + log_blowup);
    const u64 pshift = gl_exp_pow2(g, log_n - logP);
    const u64 wP = gl_two_adic_generator(logP), wPD = gl_two_adic_generator(logP + log_d);
    const u64 pinv = gl_inv((u64)Pm);
    for (size_t col = 0; col < air->periodic.size(); col++) {
      const auto& pc = air->periodic[col];
      // coefficients of the interpolant over the order-Pm subgroup (column repeated to Pm)
      std::vector<u64> coef(Pm);
      for (size_t k = 0; k < Pm; k++) {
        u64 s = 0, wk = gl_inv(gl_pow(wP, k)), x = 1;
        for (size_t r = 0; r < Pm; r++) {
          s = gl_add(s, gl_mul(pc[r % pc.size()] % GL_P, x));
          x = gl_mul(x, wk);
        }
        coef[k] = gl_mul(s, pinv);
      }
      u64 y = pshift;
      for (size_t m = 0; m < prow; m++) {
        u64 v = 0;
        for (size_t k = Pm; k-- > 0;) v = gl_add(gl_mul(v, y), coef[k]);
        ptab[col * prow + m] = v;
        y = gl_mul(y, wPD);
      }
    }
  }
  std::lock_guard<std::mutex> lk(air->ptab_mu);
  if (air->ptab_cache.size() >= 16) air->ptab_cache.clear();  // a service proving many heights with one AIR
  air->ptab_cache[key] = ptab;
  return ptab;
}

// Evaluate AIR `air` (trace height 2^log_n, LDE matrices main/aux) on its quotient coset and fold the
// result into the accumulator.  Requires the AIR's quotient degree to equal the batch degree
// (`log_d`); see prover.cpp for the upsample path.
void quotient_eval_accumulate(mh_ctx* c, const mh_air* air, const LdeMatrix& main, const LdeMatrix& aux, const LdeMatrix* prep,
                              int log_blowup, int log_d,
                              const std::vector<u64>& publics, const std::vector<e2>& randomness, const std::vector<e2>& aux_values,
                              e2 alpha, const u64* acc_in, int log_n_prev, e2 beta, u64* acc_out) {
  const int log_n = main.log_n;
  const size_t n = (size_t)1 << log_n, Dg = (size_t)1 << log_d, B = (size_t)1 << log_blowup;
  MH_REQUIRE(log_d <= log_blowup, "quotient degree exceeds blowup");
  MH_REQUIRE((air->preprocessed_width == 0) == (prep == nullptr), "internal: preprocessed matrix presence");
  if (prep)
    MH_REQUIRE(prep->log_n == main.log_n && prep->width == air->preprocessed_width && prep->log_cosets == main.log_cosets &&
                   prep->coset0 == main.coset0,
               "preprocessed matrix does not match the AIR / trace shape");
  // this rank stores 2^log_cosets of the 2^log_blowup cosets: it evaluates the quotient cosets among them
  const int G = log_blowup - main.log_cosets;
  MH_REQUIRE(G >= 0 && aux.log_cosets == main.log_cosets && aux.coset0 == main.coset0, "internal: main / aux shard shapes differ");
  // more ranks than quotient cosets (log_d < G): coset t lives on the rank whose first LDE coset is t * B / D -- that rank evaluates
  // it (one local quotient coset = its local LDE coset 0), the others hold none and must not get here (quotient_rank_owns)
  MH_REQUIRE(log_d >= G || (main.coset0 & ((B >> log_d) - 1)) == 0, "internal: this rank holds no quotient coset of this AIR");
  const int log_dl = std::max(0, log_d - G);
  const size_t D = (size_t)1 << log_dl;  // local quotient cosets
  const size_t t0 = main.coset0 >> (log_blowup - log_d);
  const int L = log_n + log_blowup;
  const u64 g = gl_lde_shift(L), wK = gl_two_adic_generator(L);
  // per-coset tables
  std::vector<u64> tab(3 * D);
  const u64 g_pow_n = gl_exp_pow2(g, log_n);
  const u64 wd = gl_two_adic_generator(log_d);
  u64 wt = gl_pow(wd, t0);
  for (size_t t = 0; t < D; t++) {
    tab[t] = gl_mul(g, gl_pow(wK, (t0 + t) * (B / Dg)));
    u64 zh = gl_sub(gl_mul(g_pow_n, wt), 1);
    tab[D + t] = zh;
    tab[2 * D + t] = gl_inv(zh);
    wt = gl_mul(wt, wd);
  }
  // alpha powers: constraint k gets alpha^(K-1-k)
  const size_t K = air->n_constraints;
  std::vector<u64> apow(2 * std::max<size_t>(K, 1));
  {
    e2 p = e2_make(1);
    for (size_t k = K; k-- > 0;) {
      apow[2 * k] = p.c0;
      apow[2 * k + 1] = p.c1;
      p = e2_mul(p, alpha);
    }
  }
  // periodic table on the quotient coset (prover/periodic.rs:49-77): once per (AIR, domain), cached in the AIR
  const size_t Pm = air->max_period();
  const size_t prow = Pm ? Pm * Dg : 1;
  const std::vector<u64> ptab = periodic_lde_table(air, log_n, log_blowup, log_d);
  std::vector<u64> pub(std::max<size_t>(1, publics.size())), rnd(2 * std::max<size_t>(1, randomness.size())),
      av(2 * std::max<size_t>(1, aux_values.size()));
  for (size_t i = 0; i < publics.size(); i++) pub[i] = gl_canon(publics[i]);
  for (size_t i = 0; i < randomness.size(); i++) { rnd[2 * i] = randomness[i].c0; rnd[2 * i + 1] = randomness[i].c1; }
  for (size_t i = 0; i < aux_values.size(); i++) { av[2 * i] = aux_values[i].c0; av[2 * i + 1] = aux_values[i].c1; }
  // one upload for all the small tables
  std::vector<u64> blob;
  auto put = [&](const std::vector<u64>& v) {
    size_t off = blob.size();
    blob.insert(blob.end(), v.begin(), v.end());
    return off;
  };
  const size_t o_tab = put(tab), o_ap = put(apow), o_pt = put(ptab), o_pub = put(pub), o_rnd = put(rnd), o_av = put(av);
  DevBuf dblob(blob.size() * 8);
  c->h2d(dblob.p, blob.data(), blob.size() * 8);
  const u64* tw = log_n ? c->twiddles(log_n, false) : nullptr;
  DevBuf one;
  if (!tw) {  // n = 1: a one-entry table holding w^0
    one.alloc(8);
    u64 v = 1;
    c->h2d(one.p, &v, 8);
    tw = one.u();
  }
  const u64 wh_inv = gl_inv(gl_two_adic_generator(log_n));
  DevBuf inv_first, inv_last;
  if (air->uses_first_last) {
    inv_first.alloc(n * D * 8);
    inv_last.alloc(n * D * 8);
    ProfScope ps(c, "quotient_selectors", 16.0 * n * D);
    MH_LAUNCH(k_selector_inverses, dim3((unsigned)((n * D + 1023) / 1024)), dim3(256), 0, c->stream, tw, dblob.u() + o_tab,
                       log_n, log_dl, wh_inv, inv_first.u(), inv_last.u());
  }
  // compiled kernels: large constraint systems (a fused program stages 256-row tiles: traces below 2^8 rows take the interpreter)
  if (air->jit && !(jit_program_fused(air->jit) && log_n < 8)) {
    JitArgs j{};
    j.main_lde = main.lde.u(); j.aux_lde = aux.lde.u();
    j.prep_lde = prep ? prep->lde.u() : nullptr;
    j.acc = acc_out;
    j.tw = tw; j.coset_tab = dblob.u() + o_tab;
    j.inv_first = inv_first.u(); j.inv_last = inv_last.u();
    j.periodic = dblob.u() + o_pt; j.periodic_rows = (u32)prow;
    j.publics = dblob.u() + o_pub; j.randomness = dblob.u() + o_rnd; j.aux_values = dblob.u() + o_av;
    j.alpha_pows = dblob.u() + o_ap;
    j.wh_inv = wh_inv;
    j.log_n = log_n; j.log_cosets = main.log_cosets; j.log_d = log_d; j.log_dl = log_dl;
    j.jc_shift = log_blowup - log_d;
    j.t0 = (u32)t0;
    ProfScope ps(c, "quotient_eval", (double)n * D * (8.0 * air->touched_base_columns + 16.0));
    j.acc_in = acc_in; j.log_n_prev = log_n_prev; j.beta0 = beta.c0; j.beta1 = beta.c1;
    jit_quotient_run(c, air->jit, j, n * D);
    if (!jit_program_fused(air->jit))  // a fused program has applied 1/Z_H and the beta accumulation itself
      MH_LAUNCH(k_quot_finish, dim3((unsigned)((n * D + 255) / 256)), dim3(256), 0, c->stream, acc_out, dblob.u() + o_tab, log_n,
                         log_dl, acc_in, log_n_prev, beta);
    return;  // no host wait: the tables are pool buffers (stream-ordered reuse), the parameter block went through the h2d ring
  }
  QuotArgs a{};
  a.code = (const AirIns*)air->d_code.p;
  a.n_ins = (u32)air->code.size();
  a.n_slots = air->n_slots;
  a.main_lde = main.lde.u();
  a.aux_lde = aux.lde.u();
  a.prep_lde = prep ? prep->lde.u() : nullptr;
  a.log_n = log_n; a.log_cosets = main.log_cosets; a.log_d = log_d; a.log_dl = log_dl;
  a.jc_shift = log_blowup - log_d;
  a.t0 = (u32)t0;
  a.tw = tw;
  a.coset_tab = dblob.u() + o_tab;
  a.wh_inv = wh_inv;
  a.inv_first = inv_first.u(); a.inv_last = inv_last.u();
  a.periodic = dblob.u() + o_pt;
  a.periodic_rows = (u32)prow;
  a.publics = dblob.u() + o_pub; a.randomness = dblob.u() + o_rnd; a.aux_values = dblob.u() + o_av;
  a.alpha_pows = dblob.u() + o_ap;
  a.acc_in = acc_in; a.log_n_prev = log_n_prev; a.beta = beta; a.acc_out = acc_out;
  // slot file: 16 B per slot per lane in LDS
  unsigned T = 256;
  while (T > 64 && (size_t)air->n_slots * 16 * T > 48 * 1024) T >>= 1;
  const size_t lds = (size_t)air->n_slots * 16 * T;
  MH_REQUIRE(lds <= 160 * 1024, "constraint DAG needs more live values than fit in LDS (160 KiB per workgroup)");
  if (lds > 64 * 1024)
    HIP_CHECK(hipFuncSetAttribute((const void*)k_eval_quotient, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  {
    // algorithmic bytes: every main/aux/preprocessed column the live DAG touches is read once, 16 B written
    ProfScope ps(c, "quotient_eval", (double)n * D * (8.0 * air->touched_base_columns + 16.0));
    MH_LAUNCH(k_eval_quotient, dim3((unsigned)((n * D + T - 1) / T)), dim3(T), lds, c->stream, a);
  }
  // no host wait here either: the next AIR's tables are prepared while this one's kernels run (the tables of this scope are pool
  // buffers: their reuse is ordered on the stream; copy-stream users fence on it, prover.hip)
}

// ---- per-AIR quotient degree below the batch degree (prover/mod.rs:520-528, quotient.rs:45-58) ------
// Q_j is known on the n*Dj-point coset g_j*J_j (cosets t' = 0..Dj-1 in `q_small`, planes [2Dj][n]);
// the batch needs it on the n*D-point coset (same shift, denser subgroup).  Same polynomial, so:
// natural-order column -> LDE by (log D - log Dj) bits -> regroup into the D cosets, fused with the
// beta accumulation.
__global__ void k_quot_to_natural(const u64* __restrict__ q_small, u64* __restrict__ nat, int log_n, int log_dj) {
  const size_t n = (size_t)1 << log_n, Dj = (size_t)1 << log_dj;
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n * Dj) return;
  const size_t r = i >> log_dj, tp = i & (Dj - 1);
  nat[i] = q_small[((2 * tp) << log_n) + r];
  nat[n * Dj + i] = q_small[((2 * tp + 1) << log_n) + r];
}
__global__ void k_quot_regroup_accumulate(const u64* __restrict__ lde, int log_n, int log_dj, int log_d, const u64* __restrict__ acc_in,
                                          int log_n_prev, e2 beta, u64* __restrict__ acc_out, size_t t_first, size_t n_local) {
  const size_t n = (size_t)1 << log_n;
  const int ab = log_d - log_dj;
  const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (q >= (n_local << log_n)) return;
  const size_t tl = q >> log_n, r = q & (n - 1);  // tl: chunk index on this rank (acc_in / acc_out planes), t: in the batch
  const size_t t = t_first + tl;
  const size_t tp = t >> ab, u = t & (((size_t)1 << ab) - 1);
  const size_t nd = n << log_dj;                       // column length of the small coset
  const size_t src = (u * nd) + (r << log_dj) + tp;    // column e at offset e * 2^ab * nd
  e2 v = e2{lde[src], lde[((size_t)1 << ab) * nd + src]};
  if (acc_in) {
    const size_t n_prev = (size_t)1 << log_n_prev, rp = r & (n_prev - 1);
    e2 old = e2{acc_in[((2 * tl) << log_n_prev) + rp], acc_in[((2 * tl + 1) << log_n_prev) + rp]};
    v = e2_add(e2_mul(old, beta), v);
  }
  acc_out[((2 * tl) << log_n) + r] = v.c0;
  acc_out[((2 * tl + 1) << log_n) + r] = v.c1;
}

// Output: the `n_local` batch chunks t_first .. t_first + n_local - 1 (a sharded proof: this rank's chunks; else all 2^log_d).
void quotient_upsample_accumulate(mh_ctx* c, const u64* q_small, int log_n, int log_blowup, int log_dj, int log_d, const u64* acc_in,
                                  int log_n_prev, e2 beta, u64* acc_out, size_t t_first, size_t n_local) {
  const size_t n = (size_t)1 << log_n, nd = n << log_dj;
  const int ab = log_d - log_dj;
  MH_REQUIRE(ab > 0, "internal: nothing to upsample");
  DevBuf nat(2 * nd * 8), scratch(2 * nd * 8), lde((2 * nd << ab) * 8);
  MH_LAUNCH(k_quot_to_natural, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, c->stream, q_small, nat.u(), log_n, log_dj);
  // evaluations on g_j * <w_{n Dj}>  ->  on g_j * w_{n D}^u * <w_{n Dj}>, u < 2^ab
  const u64 gj = gl_lde_shift(log_n + log_blowup), w = gl_two_adic_generator(log_n + log_d);
  std::vector<u64> outs((size_t)1 << ab);
  u64 x = gj;
  for (auto& v : outs) {
    v = x;
    x = gl_mul(x, w);
  }
  lde_columns(c, nat.u(), 2, log_n + log_dj, gj, outs, lde.u(), scratch.u());
  MH_REQUIRE(t_first + n_local <= ((size_t)1 << log_d), "internal: chunk range");
  const size_t total = n * n_local;
  if (total)
    MH_LAUNCH(k_quot_regroup_accumulate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, lde.u(), log_n, log_dj,
                       log_d, acc_in, log_n_prev, beta, acc_out, t_first, n_local);
}
