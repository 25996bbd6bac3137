// K7 / K8: out-of-domain evaluation and the DEEP quotient, gfx950.
//
// Replaces crates/lifted-stark/src/pcs/deep/interpolate.rs:87-204 (PointQuotients::new,
// batch_eval_lifted) and crates/lifted-stark/src/pcs/deep/prover.rs:115-315 (reduce + assemble).
//
// OOD: f(z^r) for every committed column by the barycentric formula over the FIRST coset of the
// matrix (coset 0 of the coset-major LDE = evaluations on g_m*H_m, contiguous, natural order):
//   f(y) = ((y/g)^n - 1)/n * sum_r  x_r/(y - x_r) * f(x_r),   x_r = g*w^r.
// The value is unique, so folding weights (the reference) or evaluating per height (here) agree.
// DEEP: one pass over every LDE column:  neg(x) = sum_i -alpha^(W-1-i) f_i(x)  over the ALIGNED
// column order, then Q(x) = sum_k beta^k (f_red(z_k) + neg(x)) / (z_k - x); the 1/(z_k - x) are
// produced in-register (4 points x 2 OOD points per lane, one Fermat inversion per 8 norms) instead
// of the reference's 268 MB point-quotient table.
// Roofline: HBM-bound stream (8 B per LDE felt read once, 16 B per point written).
#include "gl.cuh"
#include "kernels.hpp"

// ---- barycentric weights ------------------------------------------------------------------------
// w[k][r] = x_r / (y_k - x_r), r < n.  y_k in EF.
__global__ __launch_bounds__(256) void k_bary_weights(const u64* tw, int log_n, u64 g, e2 y0, e2 y1, u64* w0, u64* w1) {
  const size_t n = (size_t)1 << log_n;
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  const size_t half = n >> 1;
  u64 xs[4], nrm[8], pre[8];
  e2 den[8];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    size_t r = base + (size_t)k * 256;
    u64 x = 0;
    if (r < n) x = gl_mul(g, half ? (r < half ? tw[r] : gl_neg(tw[r - half])) : 1);
    xs[k] = x;
    den[2 * k] = e2_sub(y0, e2_make(x));
    den[2 * k + 1] = e2_sub(y1, e2_make(x));
  }
  u64 run = 1;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    nrm[k] = gl_sub(gl_sqr(den[k].c0), gl_mul7(gl_sqr(den[k].c1)));
    pre[k] = run;
    run = gl_mul(run, nrm[k]);
  }
  u64 inv = gl_inv(run);
#pragma unroll
  for (int k = 7; k >= 0; k--) {
    u64 ni = gl_mul(inv, pre[k]);
    inv = gl_mul(inv, nrm[k]);
    size_t r = base + (size_t)(k >> 1) * 256;
    if (r < n) {
      u64 s = gl_mul(ni, xs[k >> 1]);
      u64* w = (k & 1) ? w1 : w0;
      w[2 * r] = gl_mul(den[k].c0, s);
      w[2 * r + 1] = gl_mul(gl_neg(den[k].c1), s);
    }
  }
}

// ---- column dot products against the weights -----------------------------------------------------
// grid = (row chunk, column); partial[(col*chunks + chunk)*4 .. +4] = sum over the chunk of
// f_col(x_r) * w_k[r] for k = 0, 1 (EF).  The host adds the chunks.
static constexpr int OOD_ROWS_PER_BLOCK = 8192;
__global__ __launch_bounds__(256) void k_ood_partial(const u64* lde, int log_n, int log_blowup, const u64* w0, const u64* w1,
                                                     u64* partial, unsigned chunks) {
  __shared__ u64 red[4][256];
  const size_t n = (size_t)1 << log_n;
  const size_t col = blockIdx.y, chunk = blockIdx.x;
  const u64* f = lde + ((col << log_blowup) << log_n);  // coset 0 of this column
  const size_t r0 = chunk * OOD_ROWS_PER_BLOCK;
  const size_t r1 = r0 + OOD_ROWS_PER_BLOCK < n ? r0 + OOD_ROWS_PER_BLOCK : n;
  e2 a0 = e2_make(0), a1 = e2_make(0);
  for (size_t r = r0 + threadIdx.x; r < r1; r += 256) {
    u64 v = f[r];
    a0 = e2_add(a0, e2_mulf(e2{w0[2 * r], w0[2 * r + 1]}, v));
    a1 = e2_add(a1, e2_mulf(e2{w1[2 * r], w1[2 * r + 1]}, v));
  }
  red[0][threadIdx.x] = a0.c0; red[1][threadIdx.x] = a0.c1; red[2][threadIdx.x] = a1.c0; red[3][threadIdx.x] = a1.c1;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int k = 0; k < 4; k++) red[k][threadIdx.x] = gl_add(red[k][threadIdx.x], red[k][threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) partial[(col * chunks + chunk) * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// evals_out[k][col] (EF) for one matrix; y_k = z_k^(lift) supplied by the caller.
// Only columns [col_begin, col_end) are evaluated (the others come back as zero): a rank of a sharded proof evaluates its
// share of the columns -- on its own first coset -- and the ranks add their vectors up.
// All matrices of a statement in ONE pass over the host: every job's kernels are queued, the partial sums of all of them come back
// in one blocking copy (per matrix that was a round trip of ~45 us), and matrices that share (height, coset, points) share one
// set of barycentric weights (at 2^20 rows: three matrices, one k_bary_weights instead of three).
void deep_ood_eval_batch(mh_ctx* c, std::vector<OodJob>& jobs, int log_blowup) {
  struct Weights {
    int log_n;
    size_t coset0;
    e2 y0, y1;
    DevBuf w0, w1;
  };
  std::vector<std::unique_ptr<Weights>> weights;
  struct Slot {
    size_t off = 0, ncol = 0;
    unsigned chunks = 0;
  };
  std::vector<Slot> slots(jobs.size());
  size_t total = 0;
  for (size_t k = 0; k < jobs.size(); k++) {
    OodJob& j = jobs[k];
    const LdeMatrix& m = *j.m;
    if (j.col_end > m.width) j.col_end = m.width;
    j.out0.assign(m.width, e2_make(0));
    j.out1.assign(m.width, e2_make(0));
    if (j.col_begin >= j.col_end) continue;
    const size_t n = (size_t)1 << m.log_n;
    slots[k].ncol = j.col_end - j.col_begin;
    slots[k].chunks = (unsigned)((n + OOD_ROWS_PER_BLOCK - 1) / OOD_ROWS_PER_BLOCK);
    slots[k].off = total;
    total += slots[k].ncol * slots[k].chunks * 4;
  }
  if (!total) return;
  DevBuf partial(total * 8), one;
  for (size_t k = 0; k < jobs.size(); k++) {
    if (!slots[k].ncol) continue;
    const OodJob& j = jobs[k];
    const LdeMatrix& m = *j.m;
    const int log_n = m.log_n;
    const size_t n = (size_t)1 << log_n;
    // any coset of H determines the polynomial: use the first one this rank stores
    // (shift g_m * w_{K_m}^coset0; coset 0 on a single GPU, as the reference does)
    const u64 g = gl_mul(gl_lde_shift(log_n + log_blowup), gl_pow(gl_two_adic_generator(log_n + log_blowup), m.coset0));
    const u64* tw = log_n ? c->twiddles(log_n, false) : nullptr;
    if (!tw) {
      if (!one.p) {
        one.alloc(8);
        u64 v = 1;
        c->h2d(one.p, &v, 8);
      }
      tw = one.u();
    }
    ProfScope ps(c, "deep_ood_eval", (double)n * 8.0 * slots[k].ncol + 64.0 * n);
    Weights* w = nullptr;
    for (auto& cand : weights)
      if (cand->log_n == log_n && cand->coset0 == m.coset0 && e2_eq(cand->y0, j.y0) && e2_eq(cand->y1, j.y1)) w = cand.get();
    if (!w) {
      weights.emplace_back(new Weights{log_n, m.coset0, j.y0, j.y1, DevBuf(n * 16), DevBuf(n * 16)});
      w = weights.back().get();
      MH_LAUNCH(k_bary_weights, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, c->stream, tw, log_n, g, j.y0, j.y1, w->w0.u(), w->w1.u());
    }
    MH_LAUNCH(k_ood_partial, dim3(slots[k].chunks, (unsigned)slots[k].ncol), dim3(256), 0, c->stream,
              m.lde.u() + ((j.col_begin << m.log_cosets) << log_n), log_n, m.log_cosets, w->w0.u(), w->w1.u(), partial.u() + slots[k].off,
              slots[k].chunks);
  }
  std::vector<u64> host(total);
  c->d2h(host.data(), partial.p, total * 8);
  for (size_t k = 0; k < jobs.size(); k++) {
    if (!slots[k].ncol) continue;
    OodJob& j = jobs[k];
    const LdeMatrix& m = *j.m;
    const int log_n = m.log_n;
    const size_t n = (size_t)1 << log_n;
    const u64 g = gl_mul(gl_lde_shift(log_n + log_blowup), gl_pow(gl_two_adic_generator(log_n + log_blowup), m.coset0));
    // scaling s(y) = ((y/g)^n - 1)/n
    const u64 g_inv = gl_inv(g), n_inv = gl_inv((u64)n % GL_P);
    const e2 s0 = e2_mulf(e2_sub(e2_exp_pow2(e2_mulf(j.y0, g_inv), log_n), e2_make(1)), n_inv);
    const e2 s1 = e2_mulf(e2_sub(e2_exp_pow2(e2_mulf(j.y1, g_inv), log_n), e2_make(1)), n_inv);
    for (size_t col = 0; col < slots[k].ncol; col++) {
      e2 a0 = e2_make(0), a1 = e2_make(0);
      for (unsigned ch = 0; ch < slots[k].chunks; ch++) {
        const u64* p = host.data() + slots[k].off + (col * slots[k].chunks + ch) * 4;
        a0 = e2_add(a0, e2{p[0], p[1]});
        a1 = e2_add(a1, e2{p[2], p[3]});
      }
      j.out0[j.col_begin + col] = e2_mul(a0, s0);
      j.out1[j.col_begin + col] = e2_mul(a1, s1);
    }
  }
}
void deep_ood_eval_matrix(mh_ctx* c, const LdeMatrix& m, int log_blowup, e2 y0, e2 y1, std::vector<e2>& out0, std::vector<e2>& out1,
                          size_t col_begin, size_t col_end) {
  std::vector<OodJob> jobs(1);
  jobs[0].m = &m; jobs[0].y0 = y0; jobs[0].y1 = y1; jobs[0].col_begin = col_begin; jobs[0].col_end = col_end;
  deep_ood_eval_batch(c, jobs, log_blowup);
  out0.swap(jobs[0].out0);
  out1.swap(jobs[0].out1);
}

// ---- DEEP reduce + assemble ----------------------------------------------------------------------
// one descriptor per committed matrix travels in the kernel arguments (4 KB): a chiplet-stack statement of twelve AIRs
// (precompiles-prover/src/session/prove.rs) commits 12 x {main, aux, quotient} + its setup matrices
static constexpr int DEEP_MAX_MATS = 128;
static constexpr unsigned DEEP_FLUSH = 128;
#ifndef DEEP_UNROLL
#define DEEP_UNROLL 2  // columns per trip of the dot-product loop (loads in flight per lane = DEEP_UNROLL x DEEP_PTS)
#endif
#ifndef DEEP_PTS_N
#define DEEP_PTS_N 1  // measured: 1 point per lane 1.69 ms, 2 points 1.96 ms, 4 points 3.12 ms (occupancy beats the shared inversion)
#endif
static constexpr int DEEP_PTS = DEEP_PTS_N;  // points per lane (they share one Fermat inversion); more costs occupancy
struct DeepMat {
  const u64* lde;
  u32 width, coef_off;  // coef_off: index of this matrix's first column in the aligned coefficient list
  int log_n;
};
struct DeepArgs {
  DeepMat m[DEEP_MAX_MATS];
  int n_mats, log_n, log_blowup;  // log_n = max trace height; log_blowup = coset bits stored on this rank
  const u64* negc;                // EF pairs per aligned column
  const u64* tw;                  // w_N^k
  const u64* coset_x;             // [B] g*w_K^j
  e2 z0, z1, fred0, fred1, beta;
  u64* out;                       // EF pairs, coset-major [B][N]
};

__global__ __launch_bounds__(256) void k_deep_assemble(DeepArgs a) {
  const size_t N = (size_t)1 << a.log_n;
  const size_t j = blockIdx.y;
  const size_t base = (size_t)blockIdx.x * (256 * DEEP_PTS) + threadIdx.x;
  size_t r[DEEP_PTS];
#pragma unroll
  for (int k = 0; k < DEEP_PTS; k++) r[k] = base + (size_t)k * 256;
  // neg[k] = sum over columns of cf_col * v_col(r[k]) with the modular reduction DELAYED: cf (uniform, EF) is cut
  // into THREE 22-bit limbs, v into 32-bit halves, and the 54-bit partial products are summed by weight
  // 2^0, 2^22, 2^44 (x v.lo) and 2^32, 2^54, 2^76 (x v.hi) in plain 64-bit accumulators (one v_mad_u64_u32 each, no
  // carries); one reduction per <= DEEP_FLUSH columns.  12 multiply-adds per (column, point) instead of two modular
  // multiplications (16 with the four 16-bit limbs of rounds 2-3: the kernel is bound by these mads).
  e2 neg[DEEP_PTS];
  u64 w[DEEP_PTS][2][6];
#pragma unroll
  for (int k = 0; k < DEEP_PTS; k++) {
    neg[k] = e2_make(0);
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
      for (int i = 0; i < 6; i++) w[k][e][i] = 0;
  }
  auto flush = [&]() {
    // weights of the six accumulators mod p: 2^0, 2^22, 2^44, 2^32, 2^54, 2^76 = 2^12 (2^32 - 1)
    const u64 C[6] = {1ULL, 1ULL << 22, 1ULL << 44, 1ULL << 32, 1ULL << 54, GL_EPS << 12};
#pragma unroll
    for (int k = 0; k < DEEP_PTS; k++) {
      u64 s0 = w[k][0][0], s1 = w[k][1][0];  // < 2^61: canonical
#pragma unroll
      for (int i = 1; i < 6; i++) {
        s0 = gl_add(s0, gl_mul(w[k][0][i], C[i]));
        s1 = gl_add(s1, gl_mul(w[k][1][i], C[i]));
      }
      neg[k] = e2_add(neg[k], e2{s0, s1});
#pragma unroll
      for (int e = 0; e < 2; e++)
#pragma unroll
        for (int i = 0; i < 6; i++) w[k][e][i] = 0;
    }
  };
  u32 pending = 0;
#pragma unroll 1
  for (int mi = 0; mi < a.n_mats; mi++) {
    const DeepMat m = a.m[mi];
    const size_t nm_mask = ((size_t)1 << m.log_n) - 1;
    const u64* colp = m.lde + (j << m.log_n);
    const size_t cstride = (size_t)1 << (m.log_n + a.log_blowup);
#pragma unroll DEEP_UNROLL
    for (u32 cidx = 0; cidx < m.width; cidx++) {
      const u64 cf0 = a.negc[2 * (m.coef_off + cidx)], cf1 = a.negc[2 * (m.coef_off + cidx) + 1];
      u32 al[2][3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        al[0][i] = (u32)(cf0 >> (22 * i)) & 0x3FFFFFu;
        al[1][i] = (u32)(cf1 >> (22 * i)) & 0x3FFFFFu;
      }
#pragma unroll
      for (int k = 0; k < DEEP_PTS; k++) {
        if (r[k] < N) {
          const u64 v = colp[r[k] & nm_mask];
          const u32 v0 = (u32)v, v1 = (u32)(v >> 32);
#pragma unroll
          for (int e = 0; e < 2; e++) {
            w[k][e][0] += (u64)al[e][0] * v0;
            w[k][e][1] += (u64)al[e][1] * v0;
            w[k][e][2] += (u64)al[e][2] * v0;
            w[k][e][3] += (u64)al[e][0] * v1;
            w[k][e][4] += (u64)al[e][1] * v1;
            w[k][e][5] += (u64)al[e][2] * v1;
          }
        }
      }
      colp += cstride;
      if (++pending == DEEP_FLUSH) {  // DEEP_FLUSH products of < 2^54 per accumulator stay below 2^61
        flush();
        pending = 0;
      }
    }
  }
  flush();
  const size_t half = N >> 1;
  const u64 cx = a.coset_x[j];
  e2 den[2 * DEEP_PTS];
  u64 nrm[2 * DEEP_PTS], pre[2 * DEEP_PTS];
#pragma unroll
  for (int k = 0; k < DEEP_PTS; k++) {
    u64 x = 0;
    if (r[k] < N) x = gl_mul(cx, half ? (r[k] < half ? a.tw[r[k]] : gl_neg(a.tw[r[k] - half])) : 1);
    den[2 * k] = e2_sub(a.z0, e2_make(x));
    den[2 * k + 1] = e2_sub(a.z1, e2_make(x));
  }
  u64 run = 1;
#pragma unroll
  for (int k = 0; k < 2 * DEEP_PTS; k++) {
    nrm[k] = gl_sub(gl_sqr(den[k].c0), gl_mul7(gl_sqr(den[k].c1)));
    pre[k] = run;
    run = gl_mul(run, nrm[k]);
  }
  u64 inv = gl_inv(run);
  e2 qinv[2 * DEEP_PTS];
#pragma unroll
  for (int k = 2 * DEEP_PTS - 1; k >= 0; k--) {
    u64 ni = gl_mul(inv, pre[k]);
    inv = gl_mul(inv, nrm[k]);
    qinv[k] = e2{gl_mul(den[k].c0, ni), gl_mul(gl_neg(den[k].c1), ni)};
  }
#pragma unroll
  for (int k = 0; k < DEEP_PTS; k++) {
    if (r[k] < N) {
      e2 v = e2_mul(qinv[2 * k], e2_add(a.fred0, neg[k]));
      v = e2_add(v, e2_mul(e2_mul(a.beta, qinv[2 * k + 1]), e2_add(a.fred1, neg[k])));
      ulonglong2* o = reinterpret_cast<ulonglong2*>(a.out + 2 * ((j << a.log_n) + r[k]));
      *o = make_ulonglong2(v.c0, v.c1);
    }
  }
}

// mats: every committed matrix in transcript order (main.., aux.., quotient); negc: -alpha^(W-1-i)
// per ALIGNED column index.  out: EF pairs coset-major [B][N].
void deep_assemble(mh_ctx* c, const std::vector<const LdeMatrix*>& mats, const std::vector<u32>& coef_off, int log_n, int log_blowup,
                   const std::vector<e2>& negc, e2 z0, e2 z1, e2 fred0, e2 fred1, e2 beta, u64* out) {
  MH_REQUIRE(mats.size() <= (size_t)DEEP_MAX_MATS, "too many committed matrices for one DEEP pass");
  const int lbl = mats[0]->log_cosets;  // cosets stored on this rank (all matrices alike)
  const size_t coset0 = mats[0]->coset0;
  for (auto* m : mats) MH_REQUIRE(m->log_cosets == lbl && m->coset0 == coset0, "internal: matrices cover different cosets");
  const size_t N = (size_t)1 << log_n, B = (size_t)1 << lbl;
  const int L = log_n + log_blowup;
  std::vector<u64> blob;
  for (e2 v : negc) { blob.push_back(v.c0); blob.push_back(v.c1); }
  const size_t o_cx = blob.size();
  const u64 g = gl_lde_shift(L), wK = gl_two_adic_generator(L);
  u64 x = gl_mul(g, gl_pow(wK, coset0));
  for (size_t j = 0; j < B; j++) {
    blob.push_back(x);
    x = gl_mul(x, wK);
  }
  DevBuf dblob(blob.size() * 8), one;
  c->h2d(dblob.p, blob.data(), blob.size() * 8);
  const u64* tw = log_n ? c->twiddles(log_n, false) : nullptr;
  if (!tw) {
    one.alloc(8);
    u64 v = 1;
    c->h2d(one.p, &v, 8);
    tw = one.u();
  }
  DeepArgs a{};
  double bytes = 16.0 * N * B;
  for (size_t i = 0; i < mats.size(); i++) {
    a.m[i] = DeepMat{mats[i]->lde.u(), (u32)mats[i]->width, coef_off[i], mats[i]->log_n};
    bytes += 8.0 * (double)mats[i]->width * (double)(((size_t)1 << mats[i]->log_n) << lbl);
  }
  a.n_mats = (int)mats.size();
  a.log_n = log_n; a.log_blowup = lbl;
  a.negc = dblob.u(); a.tw = tw; a.coset_x = dblob.u() + o_cx;
  a.z0 = z0; a.z1 = z1; a.fred0 = fred0; a.fred1 = fred1; a.beta = beta; a.out = out;
  {
    ProfScope ps(c, "deep_assemble", bytes);
    MH_LAUNCH(k_deep_assemble, dim3((unsigned)((N + 256 * DEEP_PTS - 1) / (256 * DEEP_PTS)), (unsigned)B), dim3(256), 0, c->stream, a);
  }
  HIP_CHECK(hipStreamSynchronize(c->stream));
}
