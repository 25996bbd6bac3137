// LogUp auxiliary trace on the device, gfx950.
//
// Replaces air/src/lookup/aux_builder.rs `build_logup_aux_trace` (:49-96) for AIRs whose bus messages are
// exported as a lookup program (air.hpp mh_lookup):
//   collection  (lookup/prover.rs build_lookup_fractions)   -> the compiled lookup program writes every
//               fraction's multiplicity m and denominator d as planes [2 * K][n]  (air_jit.cpp, output mode);
//   accumulate  (aux_builder.rs:202-258 accumulate_slow is the semantics; :290-330 the fused form)
//               f_c(r) = sum_j m_j(r) / d_j(r);  aux[r][c >= 1] = f_c(r);
//               aux[r][0] = sum_{r' < r} sum_c f_c(r')  (row 0 = 0);  acc_final = the sum over all rows.
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
  DevBuf planes(4 * K * n * 8);  // outputs m_j, d_j: two planes each
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
  aux->ctx = c; aux->log_n = log_n; aux->width = 2 * lk->num_cols;
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
