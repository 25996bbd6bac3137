// LogUp auxiliary trace on the device, gfx950.
//
// Replaces air/src/lookup/aux_builder.rs `build_logup_aux_trace` (:49-96) for AIRs whose bus messages are
// exported as a lookup program (air.hpp mh_lookup):
//   collection  (lookup/prover.rs build_lookup_fractions)   -> the compiled lookup program writes every
//               fraction's multiplicity m and denominator d as planes [2 * K][n]  (air_jit.cpp, output mode);
//   accumulate  (aux_builder.rs:202-258 accumulate_slow is the semantics; :290-330 the fused form)
//               f_c(r) = sum_j m_j(r) / d_j(r);  aux[r][c >= 1] = f_c(r);
//               aux[r][0] = sum_{r' < r} sum_c f_c(r')  (row 0 = 0);  acc_final = the sum over all rows;
//   registers   (optional, behind the LogUp columns) linear recurrences over the rows, by a scan over affine maps (below).
// A fraction whose multiplicity is zero contributes zero, so the reference's conditional pushes and the
// program's always-evaluated fractions give the same sums.
// The aux trace is produced column-major in HBM (an mh_trace): it goes straight into the aux commitment, no
// host round trip.  HBM-bound: 24 B read per fraction and 16 B written per cell; the per-fraction EF inversion
// (~130 multiplications) is hidden behind that at this arithmetic intensity.
#include "air.hpp"
#include "air_jit.hpp"
#include "gl.cuh"
#include "kernels.hpp"

struct LogupArgs {
  const u64* planes;     // [2K][n]: m_0, d_0, m_1, d_1, ... (c0 plane, c1 plane each)
  const u32* col_count;  // [num_cols]
  const unsigned char* out_ext;  // [2K]
  u32 num_cols;
  int log_n;
  u64* aux;      // [2 * num_cols][n]; columns >= 1 written here, column 0 by the scan
  u64* totals;   // [2][n] row totals t(r)
  u32* err;      // set when a denominator is zero
};

__global__ __launch_bounds__(256) void k_logup_rows(LogupArgs a) {
  const size_t n = (size_t)1 << a.log_n;
  const size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  e2 total = e2_make(0);
  u32 k = 0;
  for (u32 c = 0; c < a.num_cols; c++) {
    e2 sum = e2_make(0);
    const u32 cnt = a.col_count[c];
    for (u32 j = 0; j < cnt; j++, k++) {
      const u32 om = 2 * k, od = 2 * k + 1;  // output indices of m and d
      const e2 m = e2{a.planes[((size_t)(2 * om) << a.log_n) + r], a.out_ext[om] ? a.planes[((size_t)(2 * om + 1) << a.log_n) + r] : 0};
      const e2 d = e2{a.planes[((size_t)(2 * od) << a.log_n) + r], a.out_ext[od] ? a.planes[((size_t)(2 * od + 1) << a.log_n) + r] : 0};
      if (e2_is_zero(m)) continue;  // contributes nothing (and the reference never pushed it)
      if (e2_is_zero(d)) {
        atomicOr(a.err, 1u);
        continue;
      }
      sum = e2_add(sum, e2_mul(e2_inv(d), m));
    }
    if (c > 0) {
      a.aux[((size_t)(2 * c) << a.log_n) + r] = sum.c0;
      a.aux[((size_t)(2 * c + 1) << a.log_n) + r] = sum.c1;
    }
    total = e2_add(total, sum);
  }
  a.totals[r] = total.c0;
  a.totals[n + r] = total.c1;
}

// ---- exclusive prefix sums over the rows (field addition; the two EF coordinates are independent) ----
static constexpr int SCAN_T = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_T * SCAN_ITEMS;

// phase 1: per tile of 2048 rows: exclusive scan into `out`, tile sum into `tile_sums`
__global__ __launch_bounds__(SCAN_T) void k_scan_tiles(const u64* __restrict__ in, u64* __restrict__ out, u64* __restrict__ tile_sums,
                                                       size_t n, size_t plane_stride_in, size_t plane_stride_out, size_t tiles) {
  __shared__ u64 part[SCAN_T];
  const size_t plane = blockIdx.y;
  const u64* src = in + plane * plane_stride_in;
  u64* dst = out + plane * plane_stride_out;
  const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  u64 v[SCAN_ITEMS];
  u64 s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    v[i] = base + i < n ? src[base + i] : 0;
    s = gl_add(s, v[i]);
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < SCAN_T; off <<= 1) {  // Hillis-Steele over the 256 thread sums
    u64 x = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] = gl_add(part[threadIdx.x], x);
    __syncthreads();
  }
  u64 run = threadIdx.x ? part[threadIdx.x - 1] : 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    if (base + i < n) dst[base + i] = run;
    run = gl_add(run, v[i]);
  }
  if (threadIdx.x == SCAN_T - 1) tile_sums[plane * tiles + blockIdx.x] = part[SCAN_T - 1];
}
// phase 2: one workgroup per plane turns the tile sums into exclusive tile offsets; the grand total goes to `totals_out`
__global__ __launch_bounds__(SCAN_T) void k_scan_tile_sums(u64* __restrict__ tile_sums, size_t tiles, u64* __restrict__ totals_out) {
  __shared__ u64 part[SCAN_T];
  __shared__ u64 carry;
  u64* s = tile_sums + (size_t)blockIdx.x * tiles;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (size_t base = 0; base < tiles; base += SCAN_T) {
    const size_t i = base + threadIdx.x;
    const u64 v = i < tiles ? s[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_T; off <<= 1) {
      u64 x = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] = gl_add(part[threadIdx.x], x);
      __syncthreads();
    }
    const u64 incl = gl_add(carry, part[threadIdx.x]);
    if (i < tiles) s[i] = gl_sub(incl, v);  // exclusive
    __syncthreads();
    if (threadIdx.x == SCAN_T - 1) carry = incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals_out[blockIdx.x] = carry;
}
// phase 3: add the tile offsets
__global__ __launch_bounds__(SCAN_T) void k_scan_add_offsets(u64* __restrict__ out, const u64* __restrict__ tile_sums, size_t n,
                                                             size_t plane_stride_out, size_t tiles) {
  const size_t plane = blockIdx.y;
  u64* dst = out + plane * plane_stride_out;
  const u64 off = tile_sums[plane * tiles + blockIdx.x];
  const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++)
    if (base + i < n) dst[base + i] = gl_add(dst[base + i], off);
}

// ---- register columns: r[0] = 0, r[i + 1] = keep(i) r[i] + build(i) -- an exclusive scan over the affine maps x -> keep(i) x + build(i) ----
// (precompiles-prover/src/uint/store_mul/trace.rs:74-218 computes its three registers row by row on the CPU; tests/aux_register.rs is the
// smallest case.)  The composition (k2, b2) o (k1, b1) = (k2 k1, k2 b1 + b2) is associative, so the same three phases as the sums above
// apply: tile aggregates, a scan over the tiles' aggregates, the tiles again from their start values.  `build` already holds the contributions of
// the earlier registers (k_reg_build).  Planes may be absent: keep0 == nullptr is keep = 1, a missing c1 plane is a base-field value.
struct AffArgs {
  const u64 *k0, *k1, *b0, *b1;
  u64 *out0, *out1;
  u64* tile_k;  // [2][tiles] aggregate keep of a tile
  u64* tile_b;  // [2][tiles] aggregate build of a tile (phase 1), then the register's value at the tile's first row (phase 2)
  size_t n, tiles;
};
struct Aff {
  e2 k, b;
};
__device__ __forceinline__ Aff aff_then(Aff first, Aff then) { return {e2_mul(then.k, first.k), e2_add(e2_mul(then.k, first.b), then.b)}; }
__device__ __forceinline__ Aff aff_load(const AffArgs& a, size_t i) {
  if (i >= a.n) return {e2_make(1), e2_make(0)};
  return {a.k0 ? e2{a.k0[i], a.k1 ? a.k1[i] : 0} : e2_make(1), e2{a.b0[i], a.b1 ? a.b1[i] : 0}};
}
template <bool APPLY>
__global__ __launch_bounds__(SCAN_T) void k_affine_tiles(AffArgs a) {
  __shared__ u64 part[4][SCAN_T];
  const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  Aff v[SCAN_ITEMS];
  Aff acc = {e2_make(1), e2_make(0)};
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    v[i] = aff_load(a, base + i);
    acc = aff_then(acc, v[i]);
  }
  auto put = [&](Aff x) {
    part[0][threadIdx.x] = x.k.c0; part[1][threadIdx.x] = x.k.c1; part[2][threadIdx.x] = x.b.c0; part[3][threadIdx.x] = x.b.c1;
  };
  auto get = [&](unsigned t) { return Aff{e2{part[0][t], part[1][t]}, e2{part[2][t], part[3][t]}}; };
  put(acc);
  __syncthreads();
  for (int off = 1; off < SCAN_T; off <<= 1) {  // Hillis-Steele over the 256 thread aggregates (earlier rows first)
    const bool has = threadIdx.x >= (unsigned)off;
    Aff before = has ? get(threadIdx.x - off) : Aff{e2_make(1), e2_make(0)};
    __syncthreads();
    if (has) put(aff_then(before, get(threadIdx.x)));
    __syncthreads();
  }
  if (!APPLY) {
    if (threadIdx.x == SCAN_T - 1) {
      const Aff t = get(SCAN_T - 1);
      a.tile_k[blockIdx.x] = t.k.c0; a.tile_k[a.tiles + blockIdx.x] = t.k.c1;
      a.tile_b[blockIdx.x] = t.b.c0; a.tile_b[a.tiles + blockIdx.x] = t.b.c1;
    }
    return;
  }
  const e2 start = e2{a.tile_b[blockIdx.x], a.tile_b[a.tiles + blockIdx.x]};
  const Aff pre = threadIdx.x ? get(threadIdx.x - 1) : Aff{e2_make(1), e2_make(0)};
  e2 run = e2_add(e2_mul(pre.k, start), pre.b);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    if (base + i < a.n) {
      a.out0[base + i] = run.c0;
      a.out1[base + i] = run.c1;
    }
    run = e2_add(e2_mul(v[i].k, run), v[i].b);
  }
}
// phase 2: the register's value at the first row of every tile: one workgroup scans the tile aggregates, 256 tiles per round
__global__ __launch_bounds__(SCAN_T) void k_affine_tile_starts(AffArgs a) {
  __shared__ u64 part[4][SCAN_T];
  __shared__ u64 carry[2];
  auto put = [&](Aff x) {
    part[0][threadIdx.x] = x.k.c0; part[1][threadIdx.x] = x.k.c1; part[2][threadIdx.x] = x.b.c0; part[3][threadIdx.x] = x.b.c1;
  };
  auto get = [&](unsigned t) { return Aff{e2{part[0][t], part[1][t]}, e2{part[2][t], part[3][t]}}; };
  if (threadIdx.x == 0) carry[0] = carry[1] = 0;
  __syncthreads();
  for (size_t base = 0; base < a.tiles; base += SCAN_T) {
    const size_t t = base + threadIdx.x;
    const Aff own = t < a.tiles ? Aff{e2{a.tile_k[t], a.tile_k[a.tiles + t]}, e2{a.tile_b[t], a.tile_b[a.tiles + t]}} : Aff{e2_make(1), e2_make(0)};
    put(own);
    __syncthreads();
    for (int off = 1; off < SCAN_T; off <<= 1) {
      const bool has = threadIdx.x >= (unsigned)off;
      Aff before = has ? get(threadIdx.x - off) : Aff{e2_make(1), e2_make(0)};
      __syncthreads();
      if (has) put(aff_then(before, get(threadIdx.x)));
      __syncthreads();
    }
    const e2 start = e2{carry[0], carry[1]};
    const Aff pre = threadIdx.x ? get(threadIdx.x - 1) : Aff{e2_make(1), e2_make(0)};
    const Aff all = get(SCAN_T - 1);
    __syncthreads();
    if (t < a.tiles) {
      const e2 v = e2_add(e2_mul(pre.k, start), pre.b);
      a.tile_b[t] = v.c0;
      a.tile_b[a.tiles + t] = v.c1;
    }
    if (threadIdx.x == 0) {
      const e2 nx = e2_add(e2_mul(all.k, start), all.b);
      carry[0] = nx.c0;
      carry[1] = nx.c1;
    }
    __syncthreads();
  }
}
// build(i) += sum_j coeff_j(i) r_j[i] over the earlier registers a register reads
struct RegBuildArgs {
  const u64 *v0, *v1;
  const u64 *u0[8], *u1[8], *r0[8], *r1[8];
  int n_terms;
  u64 *b0, *b1;
  size_t n;
};
__global__ __launch_bounds__(256) void k_reg_build(RegBuildArgs a) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  e2 acc = e2{a.v0[i], a.v1 ? a.v1[i] : 0};
  for (int t = 0; t < a.n_terms; t++)
    acc = e2_add(acc, e2_mul(e2{a.u0[t][i], a.u1[t] ? a.u1[t][i] : 0}, e2{a.r0[t][i], a.r1[t][i]}));
  a.b0[i] = acc.c0;
  a.b1[i] = acc.c1;
}

// Build the aux trace of `lk` over `main` with the lookup challenges `randomness` (EF pairs).
mh_trace* lookup_build_aux(mh_ctx* c, const mh_lookup* lk, const mh_trace* main, const mh_trace* prep, const std::vector<e2>& randomness,
                           e2* acc_final) {
  MH_REQUIRE(main->width == lk->main_width, "lookup program was exported for a different trace width");
  MH_REQUIRE(!lk->preprocessed_width || (prep && prep->width == lk->preprocessed_width && prep->log_n == main->log_n),
             "lookup program reads preprocessed columns: the preprocessed matrix (same height as the trace) is required");
  MH_REQUIRE(randomness.size() >= lk->num_randomness, "not enough lookup challenges");
  const int log_n = main->log_n;
  const size_t n = (size_t)1 << log_n;
  const size_t K = lk->n_fractions();
  size_t pm = 0;
  for (auto& col : lk->periodic) pm = std::max(pm, col.size());
  MH_REQUIRE(pm <= n, "trace shorter than a periodic column");
  // small tables: periodic columns tiled to the longest period, challenges, per-column counts, output kinds
  std::vector<u64> blob;
  const size_t prow = pm ? pm : 1;
  for (auto& col : lk->periodic)
    for (size_t i = 0; i < prow; i++) blob.push_back(col[i % col.size()] % GL_P);
  const size_t o_rnd = blob.size();
  for (size_t i = 0; i < std::max<size_t>(1, lk->num_randomness); i++) {
    blob.push_back(i < randomness.size() ? randomness[i].c0 : 0);
    blob.push_back(i < randomness.size() ? randomness[i].c1 : 0);
  }
  if (blob.empty()) blob.push_back(0);
  DevBuf dblob(blob.size() * 8), dcount(lk->col_count.size() * 4), dext(lk->out_ext.size()), derr(4);
  c->h2d(dblob.p, blob.data(), blob.size() * 8);
  c->h2d(dcount.p, lk->col_count.data(), lk->col_count.size() * 4);
  c->h2d(dext.p, lk->out_ext.data(), lk->out_ext.size());
  HIP_CHECK(hipMemsetAsync(derr.p, 0, 4, c->stream));

  trace_wait_ready(c, main);
  trace_wait_ready(c, prep);
  const size_t n_out = lk->out_ext.size();
  DevBuf planes(2 * n_out * n * 8);  // outputs m_j, d_j, then the registers' keep / build / coefficients: two planes each
  JitArgs j{};
  j.main_lde = main->cols.u();  // the trace itself: one "coset", B = 1
  j.aux_lde = main->cols.u();   // never read (a lookup program has no aux inputs)
  j.prep_lde = lk->preprocessed_width ? prep->cols.u() : nullptr;  // the preprocessed matrix itself, same layout
  j.acc = planes.u();
  j.periodic = dblob.u();
  j.periodic_rows = (u32)prow;
  j.publics = dblob.u(); j.aux_values = dblob.u(); j.alpha_pows = dblob.u(); j.tw = dblob.u(); j.coset_tab = dblob.u();
  j.inv_first = dblob.u(); j.inv_last = dblob.u();
  j.randomness = dblob.u() + o_rnd;
  j.log_n = log_n;  // log_cosets = log_d = log_dl = jc_shift = t0 = 0: point q IS row r
  std::unique_ptr<mh_trace> aux(new mh_trace());
  aux->ctx = c; aux->log_n = log_n; aux->width = 2 * lk->num_aux_cols();
  aux->cols.alloc(aux->width * n * 8);
  DevBuf totals(2 * n * 8);
  const size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  DevBuf tile_sums(2 * tiles * 8), grand(2 * 8);
  {
    ProfScope ps(c, "logup_aux", (double)n * (24.0 * K + 16.0 * lk->num_cols + 8.0 * lk->main_width));
    jit_quotient_run(c, lk->jit, j, n);
    LogupArgs a{};
    a.planes = planes.u(); a.col_count = (const u32*)dcount.p; a.out_ext = (const unsigned char*)dext.p;
    a.num_cols = (u32)lk->num_cols; a.log_n = log_n; a.aux = aux->cols.u(); a.totals = totals.u(); a.err = (u32*)derr.p;
    MH_LAUNCH(k_logup_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, a);
    // column 0 (planes 0 and 1 of the aux trace) = exclusive prefix sums of the row totals
    MH_LAUNCH(k_scan_tiles, dim3((unsigned)tiles, 2), dim3(SCAN_T), 0, c->stream, totals.u(), aux->cols.u(), tile_sums.u(), n, n, n,
                       tiles);
    MH_LAUNCH(k_scan_tile_sums, dim3(2), dim3(SCAN_T), 0, c->stream, tile_sums.u(), tiles, grand.u());
    MH_LAUNCH(k_scan_add_offsets, dim3((unsigned)tiles, 2), dim3(SCAN_T), 0, c->stream, aux->cols.u(), tile_sums.u(), n, n, tiles);
    if (!lk->regs.empty()) {
      DevBuf tile_k(2 * tiles * 8), tile_b(2 * tiles * 8), tmp(2 * n * 8);
      auto plane = [&](int out, int coord) -> const u64* {  // coordinate plane of a program output (nullptr: a base-field value's c1)
        return coord && !lk->out_ext[out] ? nullptr : planes.u() + ((size_t)(2 * out + coord) << log_n);
      };
      for (const size_t k : lk->reg_order) {
        const mh_lookup::Reg& g = lk->regs[k];
        AffArgs a{};
        a.n = n; a.tiles = tiles; a.tile_k = tile_k.u(); a.tile_b = tile_b.u();
        a.out0 = aux->cols.u() + ((size_t)(2 * (lk->num_cols + k)) << log_n);
        a.out1 = a.out0 + n;
        if (g.keep_out >= 0) { a.k0 = plane(g.keep_out, 0); a.k1 = plane(g.keep_out, 1); }
        a.b0 = plane(g.build_out, 0); a.b1 = plane(g.build_out, 1);
        if (!g.terms.empty()) {
          RegBuildArgs rb{};
          rb.v0 = a.b0; rb.v1 = a.b1; rb.n_terms = (int)g.terms.size(); rb.n = n;
          rb.b0 = tmp.u(); rb.b1 = tmp.u() + n;
          for (size_t t = 0; t < g.terms.size(); t++) {
            rb.u0[t] = plane(g.terms[t].second, 0); rb.u1[t] = plane(g.terms[t].second, 1);
            rb.r0[t] = aux->cols.u() + ((size_t)(2 * (lk->num_cols + g.terms[t].first)) << log_n);
            rb.r1[t] = rb.r0[t] + n;
          }
          MH_LAUNCH(k_reg_build, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, rb);
          a.b0 = rb.b0; a.b1 = rb.b1;
        }
        MH_LAUNCH(k_affine_tiles<false>, dim3((unsigned)tiles), dim3(SCAN_T), 0, c->stream, a);
        MH_LAUNCH(k_affine_tile_starts, dim3(1), dim3(SCAN_T), 0, c->stream, a);
        MH_LAUNCH(k_affine_tiles<true>, dim3((unsigned)tiles), dim3(SCAN_T), 0, c->stream, a);
      }
    }
  }
  u64 fin[2];
  u32 err = 0;
  HIP_CHECK(hipMemcpyAsync(fin, grand.p, 16, hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipMemcpyAsync(&err, derr.p, 4, hipMemcpyDeviceToHost, c->stream));
  c->sync();
  MH_REQUIRE(err == 0, "LogUp denominator is zero (aux_builder.rs:226-228: bus_prefix is never zero)");
  *acc_final = e2{fin[0], fin[1]};
  return aux.release();
}
