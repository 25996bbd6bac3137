// The lifted-STARK prover on the MI355X: host orchestration of the device kernels.
//
// Replaces `prove` of crates/lifted-stark/src/prover/mod.rs:230-578 as reached from
// miden_prover::prove_stark (prover/src/lib.rs:317-355), including commit_traces
// (prover/commit.rs:142-180), commit_quotient (prover/quotient.rs:143-217), sample_ood_point
// (domain.rs:539-553) and pcs::open_with_channel (pcs/prover.rs:34-101).
// Every pass over LDE-sized data is a device kernel on resident buffers; the host keeps the
// Fiat-Shamir transcript (challenger.hpp) and only ever receives the values the transcript observes:
// 3 + #FRI-rounds roots, the OOD evaluations, PoW witnesses, the final polynomial, query openings.
// Transcript order: SURVEY.md Appendix A.
#include "../../include/midenhip.h"
#include "air.hpp"
#include "air_jit.hpp"
#include "challenger.hpp"
#include "ctx.hpp"
#include "gl.cuh"
#include "kernels.hpp"
#include "poseidon2.cuh"
#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>
#include <memory>
#include <numeric>

// -------------------------------------------------------------------------------------------------
std::vector<u64> coset_shifts(int log_n, int lb) {
  // shift * w_K^j for j < B, K of order 2^(log_n+lb), canonical shift of that order (domain.rs:358-361)
  u64 g = gl_lde_shift(log_n + lb);
  u64 wk = gl_two_adic_generator(log_n + lb);
  std::vector<u64> s((size_t)1 << lb);
  u64 x = g;
  for (auto& v : s) {
    v = x;
    x = gl_mul(x, wk);
  }
  return s;
}

mh_trace* trace_upload(mh_ctx* c, const u64* rowmajor, int log_n, size_t width) {
  size_t n = (size_t)1 << log_n;
  std::unique_ptr<mh_trace> t(new mh_trace());
  t->ctx = c; t->log_n = log_n; t->width = width;
  if (width == 0) return t.release();
  DevBuf staging(n * width * 8);
  t->cols.alloc(n * width * 8);
  HIP_CHECK(hipMemcpyAsync(staging.p, rowmajor, n * width * 8, hipMemcpyHostToDevice, c->stream));
  {
    ProfScope ps(c, "transpose_in", 16.0 * n * width);
    launch_transpose_rm_to_cm(c, staging.u(), t->cols.u(), n, width);
  }
  c->sync();
  return t.release();
}

// The same without blocking the caller and without occupying the compute stream: the DMA copy and the transpose go to the
// context's copy stream, `ready` is recorded behind them, and every consumer orders itself after it on the GPU
// (trace_wait_ready).  With several traces in flight -- the three matrices of a Miden statement -- the LDE and leaf sponges of
// matrix k run while matrices k+1.. are still on the PCIe link; the first matrix of the proof order is the only exposed copy.
// (A single row-major matrix cannot be split further: column windows of a row-major host buffer move at 16-36 GB/s, the PCIe
// read-tag limit for 64-216 B segments, tools/h2dbench, and every column NTT needs all rows.)
mh_trace* trace_upload_async(mh_ctx* c, const u64* rowmajor, int log_n, size_t width) {
  size_t n = (size_t)1 << log_n;
  std::unique_ptr<mh_trace> t(new mh_trace());
  t->ctx = c; t->log_n = log_n; t->width = width;
  if (width == 0) return t.release();
  if (!c->copy_stream) HIP_CHECK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  t->staging.alloc(n * width * 8);
  t->cols.alloc(n * width * 8);
  // pooled buffers: whatever the compute stream still does with their previous contents comes first
  hipEvent_t fence = c->get_event();
  HIP_CHECK(hipEventRecord(fence, c->stream));
  HIP_CHECK(hipStreamWaitEvent(c->copy_stream, fence, 0));
  c->event_pool.push_back(fence);
  HIP_CHECK(hipMemcpyAsync(t->staging.p, rowmajor, n * width * 8, hipMemcpyHostToDevice, c->copy_stream));
  launch_transpose_rm_to_cm(c, t->staging.u(), t->cols.u(), n, width, c->copy_stream);
  HIP_CHECK(hipEventCreateWithFlags(&t->ready, hipEventDisableTiming));
  HIP_CHECK(hipEventRecord(t->ready, c->copy_stream));
  return t.release();
}
// A COLUMN-major host matrix ([width][2^log_n], e.g. a trace builder that writes columns: SURVEY 8(f) #4): every column is one
// contiguous 8 * N-byte DMA, there is no transpose, and the matrix CAN be pipelined -- groups of eight columns, one event each; the
// LDE of a group (lde_trace_cosets) waits for that group only.  Values are canonicalised in place behind the copy.
__global__ void k_canon_inplace(u64* p, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = gl_canon(p[i]);
}
mh_trace* trace_upload_cols_async(mh_ctx* c, const u64* colmajor, int log_n, size_t width) {
  const size_t n = (size_t)1 << log_n;
  std::unique_ptr<mh_trace> t(new mh_trace());
  t->ctx = c; t->log_n = log_n; t->width = width;
  if (width == 0) return t.release();
  if (!c->copy_stream) HIP_CHECK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  t->cols.alloc(n * width * 8);
  hipEvent_t fence = c->get_event();  // pooled buffer: whatever the compute stream still does with its previous contents comes first
  HIP_CHECK(hipEventRecord(fence, c->stream));
  HIP_CHECK(hipStreamWaitEvent(c->copy_stream, fence, 0));
  c->event_pool.push_back(fence);
  t->col_group = 8;
  for (size_t g0 = 0; g0 < width; g0 += t->col_group) {
    const size_t gc = std::min(t->col_group, width - g0), words = gc * n;
    HIP_CHECK(hipMemcpyAsync(t->cols.u() + g0 * n, colmajor + g0 * n, words * 8, hipMemcpyHostToDevice, c->copy_stream));
    MH_LAUNCH(k_canon_inplace, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, c->copy_stream, t->cols.u() + g0 * n, words);
    hipEvent_t e = nullptr;
    HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    t->col_ready.push_back(e);
    HIP_CHECK(hipEventRecord(e, c->copy_stream));
    t->ready = e;  // always the last RECORDED event: a failure further down leaves a trace whose destructor still waits for the copies
  }
  return t.release();
}
void trace_wait_ready(mh_ctx* c, const mh_trace* t) {
  if (!t || !t->ready) return;
  HIP_CHECK(hipStreamWaitEvent(c->stream, t->ready, 0));
  // the transpose that read the landing buffer lies before `ready`; the row-major copy does not have to live as long as the trace
  // (2^24 x 51: 6.8 GB).  POOL INVARIANT: a buffer returned to the pool may be taken by any later allocation of this context, whose
  // first write is ordered only against the PRIMARY stream's position at that moment -- so the buffer goes back only when the
  // primary stream itself has passed `ready`.  Inside commit_traces_pipelined `c->stream` is the side stream: the primary stream
  // is fenced on the event first (a wait on an already-signalled event costs nothing).
  if (t->staging.p && c == t->ctx) {
    if (c->stream != c->primary_stream && c->primary_stream) HIP_CHECK(hipStreamWaitEvent(c->primary_stream, t->ready, 0));
    t->staging.release();
  }
}

mh_trace* trace_zeros(mh_ctx* c, int log_n, size_t width) {
  size_t n = (size_t)1 << log_n;
  std::unique_ptr<mh_trace> t(new mh_trace());
  t->ctx = c; t->log_n = log_n; t->width = width;
  t->cols.alloc(n * width * 8);
  if (width) HIP_CHECK(hipMemsetAsync(t->cols.p, 0, n * width * 8, c->stream));
  return t.release();
}

// ---- collectives of a sharded proof ------------------------------------------------------------------
// Host-synchronous communicators (callbacks of a host layer) get a drained stream and must finish before returning;
// the in-library RCCL communicator (comm_rccl.cpp, stream_ordered) enqueues on c->stream and needs neither.
// A collective whose peer never arrives (a rank that died, a fabric link that does not come up) would leave this rank in
// hipStreamSynchronize for ever: the first multi-GPU run of a deployment must end in an error code, not in a hang.  After a
// stream-ordered collective has been enqueued the host waits for it with a bound ($MH_COMM_TIMEOUT_S seconds, default 120, 0 = no
// watchdog: fully asynchronous as before); the transcript needs the roots a few microseconds later anyway, so the wait costs nothing
// that was not already paid.  Host-synchronous communicators (callbacks) return when done and bound themselves.
static void comm_bounded_wait(mh_ctx* c, const char* what) {
  static const double limit_s = [] {
    const char* v = getenv("MH_COMM_TIMEOUT_S");
    return v && *v ? atof(v) : 120.0;
  }();
  if (limit_s <= 0) return;
  hipEvent_t ev;
  HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, c->stream);
  const auto t0 = std::chrono::steady_clock::now();
  long spins = 0;
  while (e == hipSuccess) {
    e = hipEventQuery(ev);
    if (e != hipErrorNotReady) break;
    e = hipSuccess;
    if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) {
      (void)hipEventDestroy(ev);
      char msg[160];
      snprintf(msg, sizeof msg, "collective %s did not complete within %.0f s (MH_COMM_TIMEOUT_S): a peer rank is missing or the fabric is down", what, limit_s);
      throw MhError(MH_ERR_COMM, msg);
    }
  }
  (void)hipEventDestroy(ev);
  HIP_CHECK(e);
}
void Dist::all_to_all(mh_ctx* c, const void* send, void* recv, size_t bytes_per_peer) const {
  if (!on()) {
    HIP_CHECK(hipMemcpyAsync(recv, send, bytes_per_peer, hipMemcpyDeviceToDevice, c->stream));
    return;
  }
  if (!comm->stream_ordered) c->sync();
  ProfScope ps(c, "comm_all_to_all", (double)bytes_per_peer * world);
  MH_REQUIRE(comm->all_to_all(comm->user, send, recv, bytes_per_peer) == 0, "all_to_all failed: " + c->err);
  if (comm->stream_ordered) comm_bounded_wait(c, "all_to_all");
}
void Dist::all_gather(mh_ctx* c, const void* send, void* recv, size_t bytes_per_rank) const {
  if (!on()) {
    HIP_CHECK(hipMemcpyAsync(recv, send, bytes_per_rank, hipMemcpyDeviceToDevice, c->stream));
    return;
  }
  if (!comm->stream_ordered) c->sync();
  ProfScope ps(c, "comm_all_gather", (double)bytes_per_rank * world);
  MH_REQUIRE(comm->all_gather(comm->user, send, recv, bytes_per_rank) == 0, "all_gather failed: " + c->err);
  if (comm->stream_ordered) comm_bounded_wait(c, "all_gather");
}
void Dist::all_reduce_sum(mh_ctx* c, u64* buf, size_t n) const {
  if (!on()) return;
  if (!comm->stream_ordered) c->sync();
  ProfScope ps(c, "comm_all_reduce", (double)n * 8);
  MH_REQUIRE(comm->all_reduce_sum_u64(comm->user, buf, n) == 0, "all_reduce failed: " + c->err);
  if (comm->stream_ordered) comm_bounded_wait(c, "all_reduce");
}

// LDE of one uploaded trace into coset-major layout on the canonical shift of its own LDE order;
// only cosets [first, first + count) are produced (all of them for a single-GPU commitment).
LdeMatrix lde_trace_cosets(mh_ctx* c, const mh_trace* tr, int lb, size_t first, size_t count) {
  LdeMatrix m;
  m.log_n = tr->log_n; m.width = tr->width;
  size_t N = (size_t)1 << tr->log_n;
  MH_REQUIRE(tr->log_n + lb <= 32, "LDE order exceeds the field's two-adicity");
  MH_REQUIRE(count > 0 && (count & (count - 1)) == 0 && first + count <= ((size_t)1 << lb), "coset range out of bounds");
  m.coset0 = first;
  while (((size_t)1 << m.log_cosets) < count) m.log_cosets++;
  if (tr->width == 0) return m;  // an AIR without aux columns still owns a (width-0) slot of the aux tree
  m.lde.alloc(N * count * tr->width * 8);
  DevBuf scratch(N * tr->width * 8);
  std::vector<u64> all = coset_shifts(tr->log_n, lb);
  std::vector<u64> mine(all.begin() + first, all.begin() + first + count);
  ProfScope ps(c, "lde", (double)(1 + count) * N * tr->width * 8.0);
  if (tr->col_group && !tr->col_ready.empty()) {
    // a column-major upload still in flight: extend each group of columns as soon as it has landed
    for (size_t g = 0, g0 = 0; g0 < tr->width; g++, g0 += tr->col_group) {
      const size_t gc = std::min(tr->col_group, tr->width - g0);
      HIP_CHECK(hipStreamWaitEvent(c->stream, tr->col_ready[g], 0));
      lde_columns(c, tr->cols.u() + g0 * N, gc, tr->log_n, 1, mine, m.lde.u() + g0 * count * N, scratch.u() + g0 * N);
    }
    return m;
  }
  trace_wait_ready(c, tr);
  lde_columns(c, tr->cols.u(), tr->width, tr->log_n, 1, mine, m.lde.u(), scratch.u());
  return m;
}
static LdeMatrix lde_trace(mh_ctx* c, const mh_trace* tr, int lb) { return lde_trace_cosets(c, tr, lb, 0, (size_t)1 << lb); }

// ---- commit_traces with the forward transforms pipelined under the leaf hashing, by groups of cosets -------------------------
// A leaf is one row of ONE coset, so the leaves of cosets [0, k) can be hashed as soon as those cosets exist: the forward NTTs of the
// next group run on a second stream meanwhile.  The strided NTT pass is short of HBM bandwidth and issues only 60-75 % of its wall
// time; the Poseidon2 sponge is pure VALU issue (and the Blake3 one pure HBM streaming next to a VALU-bound NTT): their waves share
// the SIMDs (4 sponge waves + 1 NTT wave fit a SIMD's registers) and fill each other's stalls.  Only when every matrix of the tree
// has one height (no sponge state carried between launches).  An option (MH_PIPELINE, below), not the default.
struct StreamSwap {
  mh_ctx* c;
  hipStream_t prev;
  StreamSwap(mh_ctx* ctx, hipStream_t s) : c(ctx), prev(ctx->stream) { c->stream = s; }
  ~StreamSwap() { c->stream = prev; }
};
// MH_PIPELINE: unset / 0 = off (the default); 1 = geometric groups of cosets 1, 1, 2, 4, ... ; n >= 2 = n equal groups.
// Measured at 2^20 rows (same box, ms per proof, off / 4 groups): Poseidon2 48.31 / 47.80, 49.25 / 48.33 on another box (2 groups 48.70,
// 8 groups 49.07, geometric 48.01 vs 48.27); three proofs in flight 44.3 / 42.3; Blake3 15.64 / 15.79; three matrices uploaded inside
// the proof 80.3 / 85.3 (the coefficient pass waits for every upload).  A gain of 1-2 % for one proof of one matrix, at the price of
// per-kernel times that no longer add up (both kernels are stretched while they share the SIMDs): off unless asked for.
static int pipeline_mode() {
  static const int v = [] {
    const char* e = getenv("MH_PIPELINE");
    return e ? atoi(e) : 0;
  }();
  return v;
}
static std::vector<std::pair<size_t, size_t>> pipeline_groups(int lb) {  // (first coset, count), counts powers of two
  std::vector<std::pair<size_t, size_t>> g;
  const size_t B = (size_t)1 << lb;
  const int mode = pipeline_mode();
  if (mode <= 0 || B < 2) return g;
  if (mode == 1) {
    g.emplace_back(0, 1);
    for (size_t z = 1; z < B; z *= 2) g.emplace_back(z, z);
    return g;
  }
  size_t n = (size_t)mode;
  while (n > B) n >>= 1;
  if (n < 2 || (n & (n - 1))) return {};
  for (size_t i = 0; i < n; i++) g.emplace_back(i * (B / n), B / n);
  return g;
}
static bool commit_traces_pipelined(mh_ctx* c, mh_tree* t, const std::vector<const mh_trace*>& traces, int lb) {
  if (traces.empty()) return false;
  const std::vector<std::pair<size_t, size_t>> groups = pipeline_groups(lb);
  if (groups.empty()) return false;
  const int log_n = traces[0]->log_n;
  if (log_n < 17) return false;  // small proofs live on launch latencies: nothing to hide, more launches to pay
  for (const mh_trace* tr : traces)
    if (tr->log_n != log_n || tr->width == 0) return false;
  if (c->lmcs != MH_LMCS_POSEIDON2 && c->lmcs != MH_LMCS_BLAKE3) return false;
  if (traces.size() > 8) return false;
  MH_REQUIRE(log_n + lb <= 32, "LDE order exceeds the field's two-adicity");
  const size_t N = (size_t)1 << log_n, B = (size_t)1 << lb;
  for (const mh_trace* tr : traces) {
    LdeMatrix m;
    m.log_n = log_n; m.width = tr->width; m.log_cosets = lb; m.coset0 = 0;
    m.lde.alloc(N * B * tr->width * 8);
    t->mats.push_back(std::move(m));
  }
  if (!lmcs_leaves_rangeable(c, t->mats)) {
    t->mats.clear();
    return false;
  }
  if (!c->side_stream) HIP_CHECK(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
  const std::vector<u64> shifts = coset_shifts(log_n, lb);
  std::vector<DevBuf> coef(traces.size());
  std::vector<hipEvent_t> done;
  hipEvent_t fence = c->get_event();
  HIP_CHECK(hipEventRecord(fence, c->stream));  // the side stream starts after everything already queued (and so after the pool's last users)
  try {
    {
      StreamSwap sw(c, c->side_stream);
      HIP_CHECK(hipStreamWaitEvent(c->stream, fence, 0));
      for (size_t i = 0; i < traces.size(); i++) {
        trace_wait_ready(c, traces[i]);
        coef[i].alloc(N * traces[i]->width * 8);
        ProfScope ps(c, "lde", (double)N * traces[i]->width * 8.0);
        lde_coefficients(c, traces[i]->cols.u(), traces[i]->width, log_n, coef[i].u());
      }
      for (const auto& g : groups) {
        const std::vector<u64> mine(shifts.begin() + g.first, shifts.begin() + g.first + g.second);
        for (size_t i = 0; i < traces.size(); i++) {
          ProfScope ps(c, "lde", (double)g.second * N * traces[i]->width * 8.0);
          lde_forward_group(c, coef[i].u(), traces[i]->width, log_n, 1, mine, t->mats[i].lde.u() + g.first * N, B * N);
        }
        done.push_back(c->get_event());
        HIP_CHECK(hipEventRecord(done.back(), c->stream));
      }
    }
    lmcs_alloc_layers(t, log_n + lb);
    for (size_t k = 0; k < groups.size(); k++) {
      HIP_CHECK(hipStreamWaitEvent(c->stream, done[k], 0));
      lmcs_hash_leaves_range(c, t->mats, lb, lmcs_leaf_layer(t), groups[k].first * N, groups[k].second * N);
    }
    lmcs_compress_layers(c, t);  // ends with a blocking copy of the root: both streams are idle when the coefficient buffers go back to the pool
  } catch (...) {
    (void)hipStreamSynchronize(c->side_stream);  // the side stream may still read the coefficient buffers this frame is about to free
    c->event_pool.push_back(fence);
    for (hipEvent_t e : done) c->event_pool.push_back(e);
    throw;
  }
  c->event_pool.push_back(fence);
  for (hipEvent_t e : done) c->event_pool.push_back(e);
  return true;
}

mh_tree* commit_traces(mh_ctx* c, const std::vector<const mh_trace*>& traces, int log_blowup) {
  std::unique_ptr<mh_tree> t(new mh_tree());
  t->ctx = c; t->log_blowup = log_blowup;
  if (commit_traces_pipelined(c, t.get(), traces, log_blowup)) return t.release();
  for (const mh_trace* tr : traces) t->mats.push_back(lde_trace(c, tr, log_blowup));
  lmcs_build_tree(c, t.get());
  return t.release();
}

// commit_traces of a sharded proof: this rank's cosets only, then the digest exchange.
static mh_tree* commit_traces_dist(mh_ctx* c, const std::vector<const mh_trace*>& traces, int lb, const Dist& dist) {
  if (!dist.on()) return commit_traces(c, traces, lb);
  std::unique_ptr<mh_tree> t(new mh_tree());
  t->ctx = c; t->log_blowup = lb;
  const size_t per = (size_t)1 << (lb - dist.logG);
  for (const mh_trace* tr : traces) t->mats.push_back(lde_trace_cosets(c, tr, lb, (size_t)dist.rank * per, per));
  const int log_n = t->mats.back().log_n;
  DevBuf dig((per << log_n) * 32);
  lmcs_hash_leaves(c, t->mats, lb - dist.logG, dig.u());
  lmcs_build_sharded(c, t.get(), dist, dig.u(), log_n);
  return t.release();
}

// -------------------------------------------------------------------------------------------------
struct mh_proof {
  std::vector<uint8_t> log_trace_heights;  // instance order
  std::vector<u64> fields;
  std::vector<u64> commitments;  // 4 felts each
  u64 digest[4];
};

static size_t align8(size_t w) { return (w + 7) / 8 * 8; }

static int fri_num_rounds(const mh_pcs_params& p, int log_lde) {
  int log_max_final = p.log_final_degree + p.log_blowup;
  int steps = log_lde > log_max_final ? log_lde - log_max_final : 0;
  return (steps + p.log_folding_arity - 1) / p.log_folding_arity;
}

// One proof in flight: every device stage of `prove` (prover/mod.rs:230-578) as a method that takes the
// challenges the transcript produced and returns the values the transcript must observe next.  mh_prove
// drives it with the built-in challenger; the mh_session_* entry points expose the same methods so a
// host that owns the Fiat-Shamir state (the Rust shim with p3's DuplexChallenger) can drive it itself.
struct mh_session {
  mh_ctx* c;
  mh_pcs_params pp;
  Dist dist;
  int n_airs = 0;
  std::vector<mh_air*> airs;        // instance order
  std::vector<mh_trace*> traces;
  std::vector<u64> publics;
  std::vector<int> lhs, order;      // log heights (instance order), proof order -> instance index
  int lb = 0, log_N = 0, L = 0, logD = 0, lbl = 0;
  size_t N = 0, D = 0, B_loc = 0, coset0 = 0, D_loc = 0, D_slot = 1;
  // chunks of a 2^log_d-coset quotient domain that live on this rank, and the first of them
  size_t chunks_on_rank(int log_d) const {
    if (log_d >= dist.logG) return (size_t)1 << (log_d - dist.logG);
    return (dist.rank & ((1 << (dist.logG - log_d)) - 1)) == 0 ? 1 : 0;
  }
  size_t first_chunk_on_rank(int log_d) const {
    return log_d >= dist.logG ? ((size_t)dist.rank << (log_d - dist.logG)) : ((size_t)dist.rank >> (dist.logG - log_d));
  }
  // where chunk t of a 2^log_d-chunk set sits in a buffer all-gathered with `slot` chunks per rank
  size_t gathered_chunk(int log_d, size_t slot, size_t t) const {
    if (log_d >= dist.logG) return t;  // rank r contributed chunks r * slot ..: the natural order
    return (t << (dist.logG - log_d)) * slot;
  }
  size_t max_rand = 0;
  int stage = 0;                    // protocol position, enforced on every call
  std::unique_ptr<mh_tree> main_tree, aux_tree, quot_tree;
  const mh_tree* prep_tree = nullptr;   // setup-time tree of the preprocessed LDEs (borrowed), or null
  std::vector<int> prep_of;             // proof position j -> matrix index in prep_tree, -1 if the AIR has none
  std::vector<e2> randomness;
  std::vector<std::vector<e2>> aux_vals;  // instance order
  // DEEP
  std::vector<const LdeMatrix*> mats;
  std::vector<u32> coef_off;
  size_t W = 0;
  e2 z, z_next;
  std::vector<e2> ev0, ev1;
  // FRI
  DevBuf layer;
  std::vector<std::unique_ptr<mh_tree>> fri_trees;
  int rounds = 0, log_rows = 0, cbits = 0, cb_loc = 0;
  size_t fri_c0 = 0;
  bool sharded = false, round_committed = false;

  void begin(mh_ctx* ctx, const mh_pcs_params& params, int n, mh_air* const* airs_in, mh_trace* const* traces_in,
             const u64* publics_in, size_t n_publics, const Dist& d) {
    c = ctx; pp = params; dist = d; n_airs = n;
    lmcs0 = c->lmcs;
    MH_REQUIRE(n_airs > 0 && n_airs <= 256, "need between 1 and 256 AIR instances");
    lb = pp.log_blowup;
    MH_REQUIRE(lb > 0 && lb <= 8, "log_blowup must be in 1..8");
    MH_REQUIRE(pp.log_folding_arity >= 1 && pp.log_folding_arity <= 3, "FRI folding arity must be 2, 4 or 8");
    MH_REQUIRE(pp.num_queries > 0, "num_queries must be > 0");
    for (int b : {pp.deep_pow_bits, pp.folding_pow_bits, pp.query_pow_bits})
      MH_REQUIRE(b >= 0 && b <= 32, "proof-of-work bits must be in 0..32 (sample_bits reads the low 32 bits of a sample)");
    MH_REQUIRE(pp.log_final_degree >= 0 && pp.log_final_degree <= 32, "log_final_degree must be in 0..32");
    MH_REQUIRE(pp.log_final_degree + lb >= pp.log_folding_arity - 1, "final degree unreachable by fixed-arity folding");
    // ---- trust boundary (prover/mod.rs:199-214) ----
    lhs.resize(n_airs);
    for (int i = 0; i < n_airs; i++) {
      MH_REQUIRE(airs_in[i] && traces_in[i], "null AIR or trace");
      MH_REQUIRE(traces_in[i]->width == airs_in[i]->main_width, "trace width does not match the AIR");
      MH_REQUIRE(airs_in[i]->num_public == n_publics, "AIR expects a different number of public values");
      MH_REQUIRE(traces_in[i]->log_n >= 1, "trace needs at least 2 rows");
      MH_REQUIRE(((size_t)1 << traces_in[i]->log_n) >= airs_in[i]->max_period(), "trace shorter than a periodic column");
      lhs[i] = traces_in[i]->log_n;
      airs.push_back(airs_in[i]);
      traces.push_back(traces_in[i]);
      max_rand = std::max(max_rand, airs_in[i]->num_randomness);
    }
    order.resize(n_airs);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lhs[a] < lhs[b]; });
    log_N = lhs[order.back()];
    L = log_N + lb;
    MH_REQUIRE(L <= 32, "LDE order exceeds the field's two-adicity");
    N = (size_t)1 << log_N;
    for (int i = 0; i < n_airs; i++) logD = std::max(logD, airs[i]->log_quotient_degree);
    MH_REQUIRE(logD <= lb, "constraint degree too high for the blowup");
    D = (size_t)1 << logD;
    publics.assign(publics_in, publics_in + n_publics);
    if (dist.on()) {
      MH_REQUIRE(dist.logG <= lb, "more ranks than cosets");
      for (int i = 0; i < n_airs; i++) MH_REQUIRE(lhs[i] >= dist.logG, "trace shorter than the number of ranks");
    }
    lbl = lb - dist.logG;  // coset bits stored on this rank
    B_loc = (size_t)1 << lbl;
    coset0 = (size_t)dist.rank * B_loc;
    // quotient chunks owned by this rank: chunk t IS the LDE coset t * B / D.  G <= D: D / G chunks per rank; G > D: the
    // chunk lives on rank t * G / D, the ranks in between own none (they idle through constraint evaluation, contribute an empty slot to
    // the chunk gather and transform all D chunks onto their cosets like everybody else)
    D_loc = chunks_on_rank(logD);
    D_slot = std::max<size_t>(1, D >> dist.logG);
    // ---- preprocessed columns (crates/lifted-stark/src/preprocessed.rs validate_preprocessed): every AIR that
    // declares some must point at ONE setup tree whose matrices are those AIRs' LDEs in proof order ----
    prep_of.assign(n_airs, -1);
    int n_prep = 0;
    for (int j = 0; j < n_airs; j++) {
      const mh_air* a = airs[order[j]];
      if (!a->preprocessed_width) continue;
      MH_REQUIRE(a->prep_tree, "AIR declares preprocessed columns but no preprocessed tree is attached");
      MH_REQUIRE(!prep_tree || prep_tree == a->prep_tree, "all AIRs must share one preprocessed tree");
      prep_tree = a->prep_tree;
      MH_REQUIRE(a->prep_index == n_prep, "preprocessed matrices must be committed in proof order (ascending height, ties by instance)");
      MH_REQUIRE((size_t)a->prep_index < prep_tree->mats.size(), "preprocessed matrix index out of range");
      const LdeMatrix& m = prep_tree->mats[a->prep_index];
      MH_REQUIRE(m.width == a->preprocessed_width, "preprocessed matrix width differs from the AIR's declaration");
      MH_REQUIRE(m.log_n == lhs[order[j]], "preprocessed matrix height differs from the main trace height");
      prep_of[j] = n_prep++;
    }
    if (prep_tree) {
      MH_REQUIRE(prep_tree->log_blowup == lb, "preprocessed tree was committed under a different blowup");
      MH_REQUIRE(prep_tree->shard_logG == dist.logG && (!dist.on() || prep_tree->shard_rank == dist.rank),
                 "preprocessed tree was committed for a different sharding (use mh_commit_traces_sharded with the same communicator)");
      MH_REQUIRE((int)prep_tree->mats.size() == n_prep, "preprocessed tree holds matrices no AIR declares");
      MH_REQUIRE(prep_tree->lmcs == c->lmcs, "the preprocessed (setup) tree was committed with another LMCS hasher than this context's");
    }
    rounds = fri_num_rounds(pp, L);
    stage = 1;
  }
  // aligned_len(w, lmcs.alignment()) (util/align.rs:7-13): the sponge's rate -- 8 (Poseidon2), 17 (Keccak) -- or 1 for the chaining
  // hasher of the Blake3 LMCS
  size_t alignment() const { return c->lmcs == MH_LMCS_BLAKE3 ? 1 : (c->lmcs == MH_LMCS_KECCAK ? 17 : 8); }
  size_t al(size_t w) const {
    const size_t a = alignment();
    return (w + a - 1) / a * a;
  }
  size_t ood_width() const {
    size_t w = 0;
    for (int i = 0; i < n_airs; i++)
      w += al(airs[i]->main_width) + al(2 * airs[i]->aux_width) + (airs[i]->preprocessed_width ? al(airs[i]->preprocessed_width) : 0);
    return w + al(2 * D);
  }
  size_t num_aux_values() const {
    size_t k = 0;
    for (int i = 0; i < n_airs; i++) k += airs[i]->num_aux_values;
    return k;
  }
  size_t final_poly_len() const { return (size_t)1 << std::max(0, L - rounds * pp.log_folding_arity - lb); }
  int lmcs0 = 0;  // the context's LMCS hasher when the session began: one configuration per proof
  void expect(int s, const char* what) {
    MH_REQUIRE(stage == s, std::string("session call out of protocol order: ") + what);
    MH_REQUIRE(c->lmcs == lmcs0, "the context's LMCS hasher changed during the session");
  }

  // ---- 1. main commitment ----
  void commit_main(u64 root[4]) {
    expect(1, "commit_main");
    ProfScope span(c, "span:commit to main traces");  // prover/mod.rs:339
    std::vector<const mh_trace*> po;
    for (int j = 0; j < n_airs; j++) po.push_back(traces[order[j]]);
    main_tree.reset(commit_traces_dist(c, po, lb, dist));
    memcpy(root, main_tree->root, 32);
    stage = 2;
  }
  // ---- 2. aux traces (built by the caller from the sampled randomness, instance order) + commitment ----
  // aux_values_out: the aux values in PROOF order, flattened EF, exactly as they enter the transcript.
  void commit_aux(const std::vector<e2>& rnd, mh_aux_builder cb, void* user, u64 root[4], std::vector<e2>& aux_values_out) {
    expect(2, "commit_aux");
    ProfScope span(c, "span:build aux traces + commit to aux traces");  // prover/mod.rs:355, :412
    MH_REQUIRE(rnd.size() == max_rand, "wrong number of randomness elements");
    randomness = rnd;
    std::vector<u64> rand_flat;
    for (e2 r : randomness) { rand_flat.push_back(r.c0); rand_flat.push_back(r.c1); }
    if (rand_flat.empty()) rand_flat.push_back(0);
    std::vector<std::unique_ptr<mh_trace>> aux_tr(n_airs);
    // host-built aux traces land in page-locked scratch and upload on the copy stream: the callback of instance i + 1 runs (on the
    // CPU) while instance i's matrix is on the PCIe link, and the LDE of instance i does not wait for the later uploads
    struct HostScratch {
      mh_ctx* c;
      std::vector<std::pair<void*, size_t>> bufs;
      ~HostScratch() {
        if (!bufs.empty() && c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);  // error paths: the DMA may still be reading
        for (auto& b : bufs) c->host_give(b.first, b.second);
      }
    } scratch{c, {}};
    aux_vals.assign(n_airs, {});
    for (int i = 0; i < n_airs; i++) {
      const mh_air* a = airs[i];
      const size_t n = (size_t)1 << lhs[i], w = 2 * a->aux_width;
      aux_vals[i].assign(a->num_aux_values, e2_make(0));
      if (a->lookup) {  // LogUp aux trace built on the device from the AIR's lookup program; value = acc_final
        e2 fin;
        MH_REQUIRE(!a->lookup->preprocessed_width || a->prep_raw, "the lookup program reads preprocessed columns: attach the preprocessed matrix too");
        aux_tr[i].reset(lookup_build_aux(c, a->lookup, traces[i], a->prep_raw, randomness, &fin));
        aux_vals[i][0] = fin;
      } else if (cb) {
        std::vector<u64> vals(2 * std::max<size_t>(1, a->num_aux_values), 0);
        const size_t bytes = std::max<size_t>(8, n * w * 8);
        u64* host = static_cast<u64*>(c->host_take(bytes));
        scratch.bufs.emplace_back(host, bytes);
        memset(host, 0, n * w * 8);
        int rc = cb(user, i, rand_flat.data(), host, vals.data());
        MH_REQUIRE(rc == 0, "aux trace builder / external assertion failed");
        for (size_t k = 0; k < a->num_aux_values; k++) aux_vals[i][k] = e2{gl_canon(vals[2 * k]), gl_canon(vals[2 * k + 1])};
        aux_tr[i].reset(w ? trace_upload_async(c, host, lhs[i], w) : trace_upload(c, host, lhs[i], w));
      } else {
        aux_tr[i].reset(trace_zeros(c, lhs[i], w));  // DummyMidenAir::build_aux_trace (testing/airs/miden.rs:79-89)
      }
    }
    std::vector<const mh_trace*> po;
    for (int j = 0; j < n_airs; j++) po.push_back(aux_tr[order[j]].get());
    aux_tree.reset(commit_traces_dist(c, po, lb, dist));
    memcpy(root, aux_tree->root, 32);
    aux_values_out.clear();
    for (int j = 0; j < n_airs; j++)
      for (e2 v : aux_vals[order[j]]) aux_values_out.push_back(v);
    stage = 3;
  }
  // ---- 4. + 5. constraint evaluation, accumulation, quotient commitment ----
  void commit_quotient(e2 alpha, e2 beta, u64 root[4]) {
    expect(3, "commit_quotient");
    ProfScope span(c, "span:evaluate constraints + commit to quotient poly chunks");  // prover/mod.rs:445, :542
    DevBuf acc;
    int log_n_prev = 0;
    for (int j = 0; j < n_airs; j++) {
      const mh_air* a = airs[order[j]];
      const int ln = lhs[order[j]];
      DevBuf out(std::max<size_t>(8, ((size_t)2 * D_loc << ln) * 8));
      std::vector<e2> rnd(randomness.begin(), randomness.begin() + a->num_randomness);
      const int logDj = a->log_quotient_degree;
      const LdeMatrix* prep = prep_of[j] >= 0 ? &prep_tree->mats[prep_of[j]] : nullptr;
      if (logDj == logD) {
        if (D_loc)
          quotient_eval_accumulate(c, a, main_tree->mats[j], aux_tree->mats[j], prep, lb, logD, publics, rnd, aux_vals[order[j]], alpha,
                                   j ? acc.u() : nullptr, log_n_prev, beta, out.u());
      } else {
        // native coset of n * Dj points, then upsample to n * D (prover/mod.rs:520-528).  Sharded: the Dj native chunks are the LDE
        // cosets t' * B / Dj, each evaluated by the rank that stores it; every rank gets all of them (16 B * n per chunk), runs the
        // small LDE of the upsample itself (two columns of n * Dj) and keeps the batch chunks it owns
        const size_t dj_loc = chunks_on_rank(logDj), dj_slot = std::max<size_t>(1, ((size_t)1 << logDj) >> dist.logG);
        const size_t chunk_words = (size_t)2 << ln;
        DevBuf mine(dj_slot * chunk_words * 8);
        if (dj_loc)
          quotient_eval_accumulate(c, a, main_tree->mats[j], aux_tree->mats[j], prep, lb, logDj, publics, rnd, aux_vals[order[j]], alpha,
                                   nullptr, 0, beta, mine.u());
        DevBuf small;
        if (dist.on()) {
          small.alloc(((size_t)2 << (logDj + ln)) * 8);
          DevBuf all((size_t)dist.world * dj_slot * chunk_words * 8);
          dist.all_gather(c, mine.u(), all.p, dj_slot * chunk_words * 8);
          for (size_t t = 0; t < ((size_t)1 << logDj); t++)
            HIP_CHECK(hipMemcpyAsync(small.u() + t * chunk_words, all.u() + gathered_chunk(logDj, dj_slot, t) * chunk_words, chunk_words * 8,
                                     hipMemcpyDeviceToDevice, c->stream));
          c->sync();
        } else {
          small = std::move(mine);
        }
        quotient_upsample_accumulate(c, small.u(), ln, lb, logDj, logD, j ? acc.u() : nullptr, log_n_prev, beta, out.u(),
                                     first_chunk_on_rank(logD), D_loc);
      }
      acc = std::move(out);
      log_n_prev = ln;
    }
    // quotient commitment (quotient.rs:143-217): chunk t = columns 2t, 2t+1.  acc holds this rank's
    // chunks as evaluations on the cosets g*w_J^t*H: inverse NTT in place, gather every chunk's
    // coefficients (sharded proofs: 16 B * N per chunk), forward NTT onto the local cosets.
    quot_tree.reset(new mh_tree());
    quot_tree->ctx = c; quot_tree->log_blowup = lb;
    LdeMatrix qm;
    qm.log_n = log_N; qm.width = 2 * D;
    qm.log_cosets = lbl; qm.coset0 = coset0;
    qm.lde.alloc(N * B_loc * 2 * D * 8);
    const u64 g = gl_lde_shift(L);
    const u64 wJ = gl_two_adic_generator(log_N + logD);
    const std::vector<u64> all_outs = coset_shifts(log_N, lb);
    DevBuf gathered, slot_buf;
    const u64* coef = acc.u();
    if (D_loc) {
      ProfScope ps(c, "lde", (double)N * 2 * D_loc * 8.0);
      ntt_inverse_dif_inplace(c, acc.u(), 2 * D_loc, log_N);
    }
    if (dist.on()) {
      const u64* mine = acc.u();
      if (!D_loc) {  // a rank without a chunk still fills its slot of the gather
        slot_buf.alloc(2 * D_slot * N * 8);
        HIP_CHECK(hipMemsetAsync(slot_buf.p, 0, 2 * D_slot * N * 8, c->stream));
        mine = slot_buf.u();
      }
      gathered.alloc((size_t)dist.world * 2 * D_slot * N * 8);
      dist.all_gather(c, mine, gathered.p, 2 * D_slot * N * 8);
      coef = gathered.u();
    }
    {
      ProfScope ps(c, "lde", (double)B_loc * N * 2 * D * 8.0);
      std::vector<u64> bases(D * B_loc);
      bool contiguous = true;  // chunk t's coefficients lie at 2 t N (always, unless a gather put them in rank order)
      for (size_t t = 0; t < D; t++) {
        const u64 in_inv = gl_inv(gl_mul(g, gl_pow(wJ, t)));
        for (size_t zc = 0; zc < B_loc; zc++) bases[t * B_loc + zc] = gl_mul(all_outs[coset0 + zc], in_inv);
        contiguous = contiguous && (dist.on() ? gathered_chunk(logD, D_slot, t) : t) == t;
      }
      static const int grouped = [] { const char* e = getenv("MH_QUOTIENT_LDE_GROUPED"); return e ? atoi(e) : 1; }();
      if (grouped && contiguous && D > 1) {  // all chunks in one pair of launches, every chunk with its own coset shifts
        ntt_forward_cosets(c, coef, 2 * D, log_N, bases, qm.lde.u(), 0, 2);
      } else {
        for (size_t t = 0; t < D; t++) {
          const size_t src_chunk = dist.on() ? gathered_chunk(logD, D_slot, t) : t;
          const std::vector<u64> bt(bases.begin() + t * B_loc, bases.begin() + (t + 1) * B_loc);
          ntt_forward_cosets(c, coef + 2 * src_chunk * N, 2, log_N, bt, qm.lde.u() + 2 * t * B_loc * N);
        }
      }
    }
    quot_tree->mats.push_back(std::move(qm));
    if (dist.on()) {
      DevBuf dig((B_loc << log_N) * 32);
      lmcs_hash_leaves(c, quot_tree->mats, lbl, dig.u());
      lmcs_build_sharded(c, quot_tree.get(), dist, dig.u(), log_N);
    } else {
      lmcs_build_tree(c, quot_tree.get());
    }
    memcpy(root, quot_tree->root, 32);
    stage = 4;
  }
  // ---- 6. is this OOD candidate acceptable? (domain.rs:539-553: nonzero, outside H and gK) ----
  bool ood_point_ok(e2 cand) const {
    if (e2_is_zero(cand)) return false;
    if (e2_eq(e2_exp_pow2(cand, log_N), e2_make(1))) return false;
    const u64 g_inv = gl_inv(gl_lde_shift(L));
    if (e2_eq(e2_exp_pow2(e2_mulf(cand, g_inv), L), e2_make(1))) return false;
    return true;
  }
  // ---- 7a. OOD evaluations at z and z*w_H: all trees, all matrices, aligned to 8 columns ----
  void ood(e2 zp) {
    expect(4, "ood");
    ProfScope span(c, "span:evaluate at OOD points");  // pcs/deep/prover.rs:88
    MH_REQUIRE(ood_point_ok(zp), "OOD point lies on the trace domain or the LDE coset");
    z = zp;
    z_next = e2_mulf(z, gl_two_adic_generator(log_N));
    mats.clear(); coef_off.clear();
    if (prep_tree)  // group order [preprocessed?, main, aux, quotient] (prover/mod.rs:552-560)
      for (auto& m : prep_tree->mats) mats.push_back(&m);
    for (auto& m : main_tree->mats) mats.push_back(&m);
    for (auto& m : aux_tree->mats) mats.push_back(&m);
    mats.push_back(&quot_tree->mats[0]);
    W = 0;
    for (auto* m : mats) {
      coef_off.push_back((u32)W);
      W += al(m->width);
    }
    ev0.assign(W, e2_make(0));
    ev1.assign(W, e2_make(0));
    std::vector<OodJob> jobs(mats.size());
    for (size_t i = 0; i < mats.size(); i++) {
      const int lift = log_N - mats[i]->log_n;
      // sharded proof: every rank holds every column (on its own cosets), so the columns of a matrix are split between the ranks
      // and the vectors added up below -- the barycentric sums are not repeated G times
      size_t cb = 0, ce = mats[i]->width;
      if (dist.on()) {
        const size_t per = (mats[i]->width + dist.world - 1) / dist.world;
        cb = std::min(mats[i]->width, per * dist.rank);
        ce = std::min(mats[i]->width, cb + per);
      }
      jobs[i].m = mats[i];
      jobs[i].y0 = e2_exp_pow2(z, lift);
      jobs[i].y1 = e2_exp_pow2(z_next, lift);
      jobs[i].col_begin = cb;
      jobs[i].col_end = ce;
    }
    deep_ood_eval_batch(c, jobs, lb);  // one read-back for all matrices
    for (size_t i = 0; i < mats.size(); i++)
      for (size_t k = 0; k < jobs[i].out0.size(); k++) {
        ev0[coef_off[i] + k] = jobs[i].out0[k];
        ev1[coef_off[i] + k] = jobs[i].out1[k];
      }
    if (dist.on() && W) {  // every slot was computed by exactly one rank: a plain sum puts the full vectors on every rank
      std::vector<u64> flat(4 * W);
      for (size_t i = 0; i < W; i++) {
        flat[4 * i] = ev0[i].c0; flat[4 * i + 1] = ev0[i].c1; flat[4 * i + 2] = ev1[i].c0; flat[4 * i + 3] = ev1[i].c1;
      }
      DevBuf d(flat.size() * 8);
      c->h2d(d.p, flat.data(), flat.size() * 8);
      dist.all_reduce_sum(c, d.u(), flat.size());
      c->d2h(flat.data(), d.p, flat.size() * 8);
      for (size_t i = 0; i < W; i++) {
        ev0[i] = e2{flat[4 * i], flat[4 * i + 1]};
        ev1[i] = e2{flat[4 * i + 2], flat[4 * i + 3]};
      }
    }
    stage = 5;
  }
  // ---- 7b. DEEP quotient ----
  void deep(e2 alpha_d, e2 beta_d) {
    expect(5, "deep");
    ProfScope span(c, "span:DEEP quotient");  // pcs/prover.rs:57
    e2 fred0 = e2_make(0), fred1 = e2_make(0);
    for (size_t i = 0; i < W; i++) {
      fred0 = e2_add(e2_mul(fred0, alpha_d), ev0[i]);
      fred1 = e2_add(e2_mul(fred1, alpha_d), ev1[i]);
    }
    std::vector<e2> negc(W);
    e2 pw = e2_make(GL_P - 1);
    for (size_t i = W; i-- > 0;) {
      negc[i] = pw;
      pw = e2_mul(pw, alpha_d);
    }
    layer.alloc((N << lbl) * 16);
    deep_assemble(c, mats, coef_off, log_N, lb, negc, z, z_next, fred0, fred1, beta_d, layer.u());
    // cbits = coset bits of the whole layer, cb_loc = those stored on this rank (cosets fri_c0 ..)
    log_rows = log_N; cbits = lb; cb_loc = lbl; fri_c0 = coset0;
    sharded = dist.on();
    stage = 6;
  }
  // ---- 8. FRI: commit the current layer, then fold it ----
  void fri_commit(u64 root[4]) {
    expect(6, "fri_commit");
    ProfScope span(c, "span:FRI round commit");  // pcs/fri/prover.rs:164
    MH_REQUIRE((int)fri_trees.size() < rounds && !round_committed, "no FRI round left to commit");
    const int la = pp.log_folding_arity;
    if (sharded && log_rows - la < dist.logG) {
      // fewer leaf rows per coset than ranks: the row-range split of the tree is over; every rank takes
      // the whole (small) layer and continues redundantly
      DevBuf full(((size_t)1 << (log_rows + cbits)) * 16);
      dist.all_gather(c, layer.p, full.p, ((size_t)1 << (log_rows + cb_loc)) * 16);
      layer = std::move(full);
      cb_loc = cbits;
      fri_c0 = 0;
      sharded = false;
    }
    if (log_rows < la) {  // tiny layer: fewer than `arity` rows per coset -> single-coset (natural) layout
      MH_REQUIRE(!sharded, "internal: sharded FRI layer shorter than the arity");
      DevBuf nat(((size_t)1 << (log_rows + cbits)) * 16);
      fri_to_natural(c, layer.u(), log_rows, cbits, nat.u());
      c->sync();
      layer = std::move(nat);
      log_rows += cbits;
      cbits = 0;
      cb_loc = 0;
    }
    std::unique_ptr<mh_tree> t(new mh_tree());
    t->ctx = c; t->log_blowup = cbits;
    t->fri_log_rows = log_rows; t->fri_log_arity = la;
    t->fri_log_cosets = cb_loc; t->fri_coset0 = fri_c0;
    if (sharded) {
      DevBuf dig(((size_t)1 << (log_rows - la + cb_loc)) * 32);
      fri_leaf_hash(c, layer.u(), log_rows, cb_loc, la, dig.u());
      lmcs_build_sharded(c, t.get(), dist, dig.u(), log_rows - la);
    } else {
      lmcs_alloc_layers(t.get(), log_rows + cbits - la);
      fri_leaf_hash(c, layer.u(), log_rows, cbits, la, lmcs_leaf_layer(t.get()));
      lmcs_compress_layers(c, t.get());
    }
    memcpy(root, t->root, 32);
    fri_trees.push_back(std::move(t));
    round_committed = true;
  }
  void fri_fold_round(e2 fb) {
    expect(6, "fri_fold");
    ProfScope span(c, "span:FRI fold");  // pcs/fri/prover.rs:183
    MH_REQUIRE(round_committed, "fold before the round's commitment");
    const int la = pp.log_folding_arity;
    DevBuf next(((size_t)1 << (log_rows + cb_loc - la)) * 16);
    fri_fold(c, layer.u(), log_rows, cb_loc, cbits, fri_c0, la, fb, next.u());
    fri_trees.back()->fri_layer = std::move(layer);
    layer = std::move(next);
    log_rows -= la;
    round_committed = false;
  }
  // final polynomial (fri/prover.rs:212-239): it has degree < fpd = n_f / B, so the fpd evaluations on
  // ONE coset s*<w_fpd> of the final layer determine it (s = w_{n_f}^(first local coset); s = 1 on a
  // single GPU = the reference's first fpd bit-reversed entries).  Interpolated on the host, shift undone,
  // returned in descending degree order.
  void fri_final(std::vector<e2>& desc) {
    expect(6, "fri_final");
    ProfScope span(c, "span:idft final poly");  // pcs/fri/prover.rs:231
    MH_REQUIRE((int)fri_trees.size() == rounds && !round_committed, "FRI rounds not finished");
    const int logn_f = log_rows + cbits;
    const int log_fpd = std::max(0, logn_f - lb);
    const size_t fpd = (size_t)1 << log_fpd;
    const size_t n_loc = (size_t)1 << (log_rows + cb_loc);
    std::vector<u64> host(2 * n_loc);
    c->d2h(host.data(), layer.p, n_loc * 16);
    std::vector<e2> vals(fpd);
    u64 s_shift = 1;
    if (sharded) {
      MH_REQUIRE(cbits == lb && fpd == ((size_t)1 << log_rows), "internal: final layer shape");
      for (size_t r = 0; r < fpd; r++) vals[r] = e2{host[2 * r], host[2 * r + 1]};  // first local coset
      s_shift = gl_pow(gl_two_adic_generator(logn_f), fri_c0);
    } else {
      for (size_t r = 0; r < fpd; r++) {
        size_t i = r << (logn_f - log_fpd);
        size_t slot = ((i & (((size_t)1 << cbits) - 1)) << log_rows) + (i >> cbits);
        vals[r] = e2{host[2 * slot], host[2 * slot + 1]};
      }
    }
    const u64 w_inv = gl_inv(gl_two_adic_generator(log_fpd)), n_inv = gl_inv((u64)fpd);
    const u64 s_inv = gl_inv(s_shift);
    std::vector<e2> coef(fpd);
    u64 sk = 1;
    for (size_t k = 0; k < fpd; k++) {
      e2 s = e2_make(0);
      u64 wk = gl_pow(w_inv, k), x = 1;
      for (size_t r = 0; r < fpd; r++) {
        s = e2_add(s, e2_mulf(vals[r], x));
        x = gl_mul(x, wk);
      }
      coef[k] = e2_mulf(s, gl_mul(n_inv, sk));
      sk = gl_mul(sk, s_inv);
    }
    desc.assign(coef.rbegin(), coef.rend());
    layer.release();
    stage = 7;
  }
  // ---- 9. openings of every tree at the sampled domain indices, in transcript (hint) order ----
  void open(std::vector<size_t> idx, std::vector<u64>& fields, std::vector<u64>& commitments) {
    expect(7, "open");
    ProfScope span(c, "span:query phase");  // pcs/prover.rs:89
    std::sort(idx.begin(), idx.end());
    idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
    for (size_t i : idx) MH_REQUIRE(i < ((size_t)1 << L), "query index out of range");
    // every tree's gather list first, then ONE gather / read-back (/ all-reduce) for all of them
    std::vector<const u64*> ptrs;
    std::vector<OpenPlan> plans;
    if (prep_tree) {  // a tree shorter than the max domain is virtually lifted: indices fold by their low bits
      std::vector<size_t> pidx(idx);
      const size_t mask = ((size_t)1 << (prep_tree->log_height + prep_tree->shard_logG)) - 1;  // full depth (a rank stores a subtree)
      for (auto& i : pidx) i &= mask;
      std::sort(pidx.begin(), pidx.end());
      pidx.erase(std::unique(pidx.begin(), pidx.end()), pidx.end());
      plans.push_back(lmcs_open_plan(prep_tree, pidx, alignment(), &dist, ptrs));
    }
    for (const mh_tree* t : {main_tree.get(), aux_tree.get(), quot_tree.get()}) plans.push_back(lmcs_open_plan(t, idx, alignment(), &dist, ptrs));
    int depth = L;
    for (auto& t : fri_trees) {
      depth -= pp.log_folding_arity;
      const size_t mask = ((size_t)1 << depth) - 1;
      for (auto& i : idx) i &= mask;
      std::sort(idx.begin(), idx.end());
      idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
      plans.push_back(lmcs_open_plan(t.get(), idx, 1, &dist, ptrs));
    }
    std::vector<u64> host;
    lmcs_open_run(c, ptrs, &dist, host);
    for (const OpenPlan& plan : plans) {
      std::vector<u64> f, cm;
      lmcs_open_take(plan, host, f, cm);
      fields.insert(fields.end(), f.begin(), f.end());
      commitments.insert(commitments.end(), cm.begin(), cm.end());
    }
    stage = 8;
  }
};

static void do_grind(mh_ctx* c, HostTranscript& tr, int bits) {
  if (bits == 0) {
    tr.fields.push_back(0);
    return;
  }
  if (bits <= 5 && tr.ch.hash == MH_LMCS_POSEIDON2 && p2_host_simd_available()) {
    // A trial is ONE permutation whose input differs from the next trial's in the witness word only (observe(w), then the duplexing
    // that sample_bits or the full rate triggers; the sampled word is st[7]): eight trials per AVX-512 permutation, smallest hit first
    // -- the same witness the scalar loop finds.  ~16 trials per FRI round, seven rounds per proof.
    HostChallenger& ch = tr.ch;
    const size_t k = ch.in.size() + 1;  // rate words after observing the witness: 1..8
    u64 base[12];
    for (int i = 0; i < 12; i++) base[i] = ch.st[i];
    for (size_t i = 0; i + 1 < k; i++) base[i] = ch.in[i];
    for (size_t i = k; i < 8; i++) base[i] = 0;
    base[8] = gl_add(base[8], (u64)k);
    const u64 mask = ((u64)1 << bits) - 1;
    for (u64 w0 = 0;; w0 += 8) {
      u64 states[8][12];
      for (int j = 0; j < 8; j++) {
        memcpy(states[j], base, 96);
        states[j][k - 1] = w0 + (u64)j;
      }
      p2_host_permute8(&states[0][0]);
      for (int j = 0; j < 8; j++)
        if (((states[j][7] & 0xFFFFFFFFULL) & mask) == 0) {
          MH_REQUIRE(ch.check_witness(bits, w0 + (u64)j), "internal: vectorised PoW witness rejected by the challenger");
          tr.fields.push_back(w0 + (u64)j);
          return;
        }
    }
  }
  if (bits <= 5) {  // ~2^bits trials: cheaper on the host than one kernel launch + round trip
    for (u64 w = 0;; w++) {
      HostChallenger trial = tr.ch;
      if (trial.check_witness(bits, w)) {
        tr.ch = trial;
        tr.fields.push_back(w);
        return;
      }
    }
  }
  u64 w = tr.ch.bytes() ? fri_grind_bytes(c, tr.ch.hash, tr.ch.bin, bits)
                        : fri_grind(c, tr.ch.st, tr.ch.in.data(), (int)tr.ch.in.size(), bits);
  MH_REQUIRE(tr.ch.check_witness(bits, w), "internal: device PoW witness rejected by the host challenger");
  tr.fields.push_back(w);
}

// `prove` with the built-in transcript: the exact sequence of SURVEY.md Appendix A.
static void prove_impl(mh_ctx* c, const mh_pcs_params& pp, int n_airs, mh_air* const* airs_in, mh_trace* const* traces_in,
                       const u64* publics_in, size_t n_publics, const u64 init_state[12], const u64* pre_observe, size_t n_pre,
                       mh_aux_builder cb, void* user, mh_proof& proof, const Dist& dist) {
  // the one-shot prover owns the transcript, and the library's challenger is the duplex sponge of the Poseidon2 (algebraic)
  // configuration; the Blake3 configuration goes through the staged session, where the host owns the challenger
  // every configuration of air/src/config.rs: the duplex sponge over the context's permutation, or -- Blake3, Keccak -- the
  // library's restatement of p3's serializing hash challenger (challenger.hpp; unpinned: a shim that wants p3's own keeps the
  // transcript on the host and drives the staged session instead)
  mh_session s;
  s.begin(c, pp, n_airs, airs_in, traces_in, publics_in, n_publics, dist);
  HostTranscript tr;
  tr.ch.hash = c->lmcs;
  tr.ch.init_from_state(init_state);
  for (size_t i = 0; i < n_pre; i++) tr.ch.observe_framing(pre_observe[i]);
  tr.ch.observe((u64)n_airs);  // order.rs:154-163
  for (int i = 0; i < n_airs; i++) tr.ch.observe((u64)s.lhs[i]);

  u64 root[4];
  s.commit_main(root);
  tr.send_commitment(root);
  std::vector<e2> rnd;
  for (size_t i = 0; i < s.max_rand; i++) rnd.push_back(tr.ch.sample_ef());
  std::vector<e2> aux_values;
  s.commit_aux(rnd, cb, user, root, aux_values);
  tr.send_commitment(root);
  for (e2 v : aux_values) tr.send_ef(v);
  const e2 alpha = tr.ch.sample_ef();
  const e2 beta = tr.ch.sample_ef();
  s.commit_quotient(alpha, beta, root);
  tr.send_commitment(root);
  e2 z;
  do {
    z = tr.ch.sample_ef();
  } while (!s.ood_point_ok(z));
  s.ood(z);
  for (e2 v : s.ev0) tr.send_ef(v);
  for (e2 v : s.ev1) tr.send_ef(v);
  do_grind(c, tr, pp.deep_pow_bits);
  const e2 alpha_d = tr.ch.sample_ef();
  const e2 beta_d = tr.ch.sample_ef();
  s.deep(alpha_d, beta_d);
  for (int r = 0; r < s.rounds; r++) {
    s.fri_commit(root);
    tr.send_commitment(root);
    do_grind(c, tr, pp.folding_pow_bits);
    s.fri_fold_round(tr.ch.sample_ef());
  }
  std::vector<e2> final_poly;
  s.fri_final(final_poly);
  for (e2 v : final_poly) tr.send_ef(v);
  do_grind(c, tr, pp.query_pow_bits);
  std::vector<size_t> idx;
  for (int i = 0; i < pp.num_queries; i++) idx.push_back(tr.ch.sample_bits(s.L));
  std::vector<u64> f, cm;
  s.open(idx, f, cm);
  tr.hint_fields(f);
  tr.hint_commitments(cm);
  // finalize (CanFinalizeDigest, external; "unconditionally applies a final state transition before extracting the
  // digest", crates/stark-transcript/src/prover.rs:31-35): one more duplexing whatever is buffered, then 4 felts
  tr.ch.finalize(proof.digest);
  for (int i = 0; i < n_airs; i++) proof.log_trace_heights.push_back((uint8_t)s.lhs[i]);
  proof.fields = std::move(tr.fields);
  for (auto& d : tr.commitments) proof.commitments.insert(proof.commitments.end(), d.begin(), d.end());
}

// -------------------------------------------------------------------------------------------------
#define MH_TRY(ctx_expr) mh_ctx* _c = (ctx_expr); PoolScope _ps(_c); try {
#define MH_CATCH                                                   \
  }                                                                \
  catch (const MhError& e) {                                       \
    if (_c) _c->err = e.what();                                    \
    return e.code;                                                 \
  }                                                                \
  catch (const std::exception& e) {                                \
    if (_c) _c->err = e.what();                                    \
    return MH_ERR_INTERNAL;                                        \
  }                                                                \
  return MH_OK;

extern "C" {

int mh_air_load(mh_ctx* c, const uint64_t* blob, size_t n_words, mh_air** out) {
  MH_TRY(c)
  MH_REQUIRE(c && blob && out, "null argument");
  HIP_CHECK(hipSetDevice(c->device));
  *out = mh_air::load(c, blob, n_words);
  MH_CATCH
}
void mh_air_free(mh_air* a) {
  if (!a) return;
  (void)hipSetDevice(a->ctx->device);
  PoolScope ps(a->ctx);
  delete a;
}
int mh_air_log_quotient_degree(const mh_air* a) { return a ? a->log_quotient_degree : -1; }
int mh_jit_precompile(const uint64_t* blob, size_t n_words, int* n_chunks) {
  try {
    if (!blob) return MH_ERR_INVALID;
    const int k = jit_precompile_blob(blob, n_words);
    if (n_chunks) *n_chunks = k;
    return MH_OK;
  } catch (const MhError& e) {
    return e.code;
  } catch (const std::exception&) {
    return MH_ERR_INTERNAL;
  }
}
// ---- LogUp lookup programs -----------------------------------------------------------------------------
int mh_lookup_load(mh_ctx* c, const uint64_t* blob, size_t n_words, mh_lookup** out) {
  MH_TRY(c)
  MH_REQUIRE(c && blob && out, "null argument");
  HIP_CHECK(hipSetDevice(c->device));
  *out = mh_lookup::load(c, blob, n_words);
  MH_CATCH
}
void mh_lookup_free(mh_lookup* l) {
  if (!l) return;
  (void)hipSetDevice(l->ctx->device);
  PoolScope ps(l->ctx);
  delete l;
}
int mh_air_attach_lookup(mh_air* a, const mh_lookup* l) {
  MH_TRY(a ? a->ctx : nullptr)
  MH_REQUIRE(a, "null argument");
  if (l) {
    MH_REQUIRE(l->main_width == a->main_width, "lookup program and AIR disagree on the trace width");
    MH_REQUIRE(l->num_aux_cols() == a->aux_width, "lookup program and AIR disagree on the number of aux columns");
    MH_REQUIRE(a->num_aux_values == 1, "a LogUp AIR commits exactly one aux value (the accumulator's final)");
    MH_REQUIRE(l->num_randomness <= a->num_randomness, "lookup program needs more challenges than the AIR samples");
    MH_REQUIRE(l->preprocessed_width == 0 || l->preprocessed_width == a->preprocessed_width,
               "lookup program and AIR disagree on the preprocessed width");
  }
  a->lookup = l;
  MH_CATCH
}
int mh_lookup_build_aux(mh_ctx* c, const mh_lookup* l, const mh_trace* main_trace, const mh_trace* preprocessed, const uint64_t* randomness,
                        size_t n_randomness,
                        mh_trace** aux_out, uint64_t acc_final[2]) {
  MH_TRY(c)
  MH_REQUIRE(c && l && main_trace && aux_out && acc_final && (randomness || !n_randomness), "null argument");
  HIP_CHECK(hipSetDevice(c->device));
  std::vector<e2> rnd;
  for (size_t i = 0; i < n_randomness; i++) rnd.push_back(e2{gl_canon(randomness[2 * i]), gl_canon(randomness[2 * i + 1])});
  e2 fin;
  *aux_out = lookup_build_aux(c, l, main_trace, preprocessed, rnd, &fin);
  acc_final[0] = fin.c0;
  acc_final[1] = fin.c1;
  MH_CATCH
}
int mh_trace_download(mh_ctx* c, const mh_trace* t, uint64_t* rowmajor_out) {
  MH_TRY(c)
  MH_REQUIRE(c && t && rowmajor_out, "null argument");
  HIP_CHECK(hipSetDevice(c->device));
  const size_t n = (size_t)1 << t->log_n;
  std::vector<u64> cm(n * t->width);
  trace_wait_ready(c, t);
  HIP_CHECK(hipMemcpyAsync(cm.data(), t->cols.p, cm.size() * 8, hipMemcpyDeviceToHost, c->stream));
  c->sync();
  for (size_t col = 0; col < t->width; col++)
    for (size_t r = 0; r < n; r++) rowmajor_out[r * t->width + col] = cm[col * n + r];
  MH_CATCH
}

int mh_air_attach_preprocessed(mh_air* a, const mh_tree* tree, int matrix_index, const mh_trace* raw) {
  MH_TRY(a ? a->ctx : nullptr)
  MH_REQUIRE(a, "null argument");
  if (tree) {
    MH_REQUIRE(a->preprocessed_width > 0, "this AIR declares no preprocessed columns");
    MH_REQUIRE(matrix_index >= 0 && (size_t)matrix_index < tree->mats.size(), "preprocessed matrix index out of range");
    MH_REQUIRE(tree->mats[matrix_index].width == a->preprocessed_width, "preprocessed matrix width differs from the AIR's declaration");
  }
  if (tree && raw) MH_REQUIRE(raw->width == a->preprocessed_width, "preprocessed matrix width differs from the AIR's declaration");
  a->prep_tree = tree;
  a->prep_index = tree ? matrix_index : -1;
  a->prep_raw = tree ? raw : nullptr;
  MH_CATCH
}
int mh_air_compiled_chunks(const mh_air* a) { return a ? (int)jit_program_chunks(a->jit) : -1; }
int mh_air_compiled_max_vgprs(const mh_air* a) { return a ? jit_program_max_vgprs(a->jit) : -1; }

// ---- coset-sharded commitment (one process per GPU) -------------------------
struct mh_shard {
  mh_ctx* ctx;
  int log_blowup, log_n, rank, world, log_world;
  std::vector<LdeMatrix> mats;  // this rank's 2^(lb - log_world) cosets of every matrix
  DevBuf leaf_digests;          // [cosets_local][N] digests (4 felts each)
  std::unique_ptr<mh_tree> subtree;
};

int mh_shard_commit_leaves(mh_ctx* c, int n_traces, mh_trace* const* traces, int log_blowup, int rank, int world, mh_shard** out) {
  MH_TRY(c)
  MH_REQUIRE(c && traces && out && n_traces > 0, "null/empty argument");
  int lw = 0;
  while ((1 << lw) < world) lw++;
  MH_REQUIRE(world >= 1 && (1 << lw) == world && lw <= log_blowup, "world must be a power of two not larger than the blowup");
  MH_REQUIRE(rank >= 0 && rank < world, "rank out of range");
  HIP_CHECK(hipSetDevice(c->device));
  std::unique_ptr<mh_shard> s(new mh_shard());
  s->ctx = c; s->log_blowup = log_blowup; s->rank = rank; s->world = world; s->log_world = lw;
  const size_t per = (size_t)1 << (log_blowup - lw);
  for (int i = 0; i < n_traces; i++) {
    MH_REQUIRE(traces[i], "null trace");
    MH_REQUIRE(traces[i]->log_n >= lw, "trace too short to shard by rows");
    s->mats.push_back(lde_trace_cosets(c, traces[i], log_blowup, (size_t)rank * per, per));
  }
  s->log_n = s->mats.back().log_n;
  s->leaf_digests.alloc((per << s->log_n) * 32);
  lmcs_hash_leaves(c, s->mats, log_blowup - lw, s->leaf_digests.u());
  c->sync();
  *out = s.release();
  MH_CATCH
}
void mh_shard_free(mh_shard* s) {
  if (!s) return;
  (void)hipSetDevice(s->ctx->device);
  PoolScope ps(s->ctx);
  delete s;
}
uint64_t* mh_shard_leaf_digests(mh_shard* s, size_t* n_digests) {
  if (!s) return nullptr;
  if (n_digests) *n_digests = s->leaf_digests.bytes / 32;
  return s->leaf_digests.u();
}
int mh_shard_build_subtree(mh_ctx* c, mh_shard* s, const uint64_t* digests_device, uint64_t subroot[4]) {
  MH_TRY(c)
  MH_REQUIRE(c && s && digests_device && subroot, "null argument");
  HIP_CHECK(hipSetDevice(c->device));
  std::unique_ptr<mh_tree> t(new mh_tree());
  t->ctx = c; t->log_blowup = s->log_blowup;
  lmcs_alloc_layers(t.get(), s->log_n - s->log_world + s->log_blowup);
  const size_t leaves = (size_t)1 << t->log_height;
  HIP_CHECK(hipMemcpyAsync(lmcs_leaf_layer(t.get()), digests_device, leaves * 32, hipMemcpyDeviceToDevice, c->stream));
  lmcs_compress_layers(c, t.get());
  memcpy(subroot, t->root, 32);
  s->subtree = std::move(t);
  MH_CATCH
}
// Root of the cap: `world` subtree roots (rank order = domain order) -> the commitment (host only), under the hasher the
// subtrees were built with (MH_LMCS_*; mh_shard_commit_leaves / mh_shard_build_subtree follow the context's).
int mh_merkle_cap_root_lmcs(int lmcs, const uint64_t* subroots, int world, uint64_t root[4]) {
  if (!subroots || !root || world < 1 || (world & (world - 1)) || lmcs < 0 || lmcs > MH_LMCS_RPX) return MH_ERR_INVALID;
  std::vector<u64> cur(subroots, subroots + 4 * (size_t)world);
  const bool felts = lmcs != MH_LMCS_BLAKE3 && lmcs != MH_LMCS_KECCAK;
  if (felts)
    for (auto& x : cur) x = gl_canon(x);
  for (int n = world; n > 1; n >>= 1) {
    std::vector<u64> next(4 * (size_t)(n / 2));
    lmcs_host_compress_level(lmcs, cur.data(), (size_t)(n / 2), next.data());
    cur.swap(next);
  }
  memcpy(root, cur.data(), 32);
  return MH_OK;
}
int mh_merkle_cap_root(const uint64_t* subroots, int world, uint64_t root[4]) {
  return mh_merkle_cap_root_lmcs(MH_LMCS_POSEIDON2, subroots, world, root);
}

// commit_traces for one rank of a sharded prover: the setup-time commitment of preprocessed matrices when proofs are
// sharded (same root as mh_commit_traces; this rank keeps its cosets and its slice of the tree).
// The trace of a sharded proof without G full uploads: every rank moves only its 1/G of the ROWS over its own PCIe link (a
// contiguous slice of the row-major host matrix: full DMA rate), the slices are all-gathered over xGMI, and each
// rank transposes the assembled matrix.  At 2^24 x 51 on 8 GPUs: 0.85 GB of PCIe + 6 GB of xGMI receives per rank instead of 6.8 GB
// of PCIe each (all eight links of the host at once).  `rowmajor` must be readable by every rank (threads of one process, or a
// shared mapping); a rank only reads its own slice.
int mh_trace_upload_sharded(mh_ctx* c, const mh_comm* comm, const uint64_t* rowmajor, int log_n, size_t width, mh_trace** out) {
  MH_TRY(c)
  MH_REQUIRE(c && comm && rowmajor && out, "null argument");
  MH_REQUIRE(log_n >= 0 && log_n <= 29 && width > 0, "bad trace shape");
  MH_REQUIRE(comm->world >= 1 && (comm->world & (comm->world - 1)) == 0 && comm->rank >= 0 && comm->rank < comm->world,
             "world must be a power of two and 0 <= rank < world");
  HIP_CHECK(hipSetDevice(c->device));
  const size_t n = (size_t)1 << log_n;
  if (comm->world == 1 || n < (size_t)comm->world) {
    *out = trace_upload(c, rowmajor, log_n, width);
  } else {
    MH_REQUIRE(comm->all_gather, "missing all_gather callback");
    Dist d;
    d.comm = comm; d.rank = comm->rank; d.world = comm->world;
    while ((1 << d.logG) < d.world) d.logG++;
    const size_t rows = n / comm->world, slice = rows * width * 8;
    std::unique_ptr<mh_trace> t(new mh_trace());
    t->ctx = c; t->log_n = log_n; t->width = width;
    DevBuf mine(slice), full(n * width * 8);
    t->cols.alloc(n * width * 8);
    HIP_CHECK(hipMemcpyAsync(mine.p, rowmajor + (size_t)comm->rank * rows * width, slice, hipMemcpyHostToDevice, c->stream));
    d.all_gather(c, mine.p, full.p, slice);  // rank order = row order
    {
      ProfScope ps(c, "transpose_in", 16.0 * n * width);
      launch_transpose_rm_to_cm(c, full.u(), t->cols.u(), n, width);
    }
    c->sync();
    *out = t.release();
  }
  MH_CATCH
}

int mh_commit_traces_sharded(mh_ctx* c, const mh_comm* comm, int n_traces, mh_trace* const* traces, int log_blowup, mh_tree** out,
                             uint64_t root[4]) {
  MH_TRY(c)
  MH_REQUIRE(c && comm && traces && out && n_traces > 0, "null/empty argument");
  MH_REQUIRE(log_blowup >= 0 && log_blowup <= 8, "bad log_blowup");
  MH_REQUIRE(comm->world >= 1 && (comm->world & (comm->world - 1)) == 0 && comm->rank >= 0 && comm->rank < comm->world,
             "world must be a power of two and 0 <= rank < world");
  MH_REQUIRE(comm->world == 1 || (comm->all_to_all && comm->all_gather && comm->all_reduce_sum_u64), "missing collective callbacks");
  HIP_CHECK(hipSetDevice(c->device));
  Dist d;
  if (comm->world > 1) {
    d.comm = comm; d.rank = comm->rank; d.world = comm->world;
    while ((1 << d.logG) < d.world) d.logG++;
    MH_REQUIRE(d.logG <= log_blowup, "more ranks than cosets");
  }
  std::vector<const mh_trace*> v;
  for (int i = 0; i < n_traces; i++) {
    MH_REQUIRE(traces[i] && traces[i]->log_n >= d.logG, "null trace / trace shorter than the number of ranks");
    v.push_back(traces[i]);
  }
  std::unique_ptr<mh_tree> t(commit_traces_dist(c, v, log_blowup, d));
  if (root) memcpy(root, t->root, 32);
  *out = t.release();
  MH_CATCH
}

int mh_prove(mh_ctx* c, const mh_pcs_params* params, int n_airs, mh_air* const* airs, mh_trace* const* traces,
             const uint64_t* public_values, size_t n_public_values, const uint64_t challenger_state[12],
             const uint64_t* pre_observe, size_t n_pre_observe, mh_aux_builder aux_builder, void* user, mh_proof** out) {
  MH_TRY(c)
  MH_REQUIRE(c && params && airs && traces && challenger_state && out, "null argument");
  MH_REQUIRE(public_values || !n_public_values, "null public values");
  MH_REQUIRE(pre_observe || !n_pre_observe, "null pre_observe");
  HIP_CHECK(hipSetDevice(c->device));
  std::unique_ptr<mh_proof> p(new mh_proof());
  prove_impl(c, *params, n_airs, airs, traces, public_values, n_public_values, challenger_state, pre_observe, n_pre_observe,
             aux_builder, user, *p, Dist{});
  *out = p.release();
  MH_CATCH
}

// prove_stark's shape exactly (prover/src/lib.rs:317-355): HOST RowMajorMatrix values in, proof out.  The uploads are started in
// proof order (ascending height, ties by instance index: the order the matrices are extended and absorbed in), so matrix k + 1
// is on the PCIe link while matrix k is extended and hashed; the traces are freed before returning.
int mh_prove_host(mh_ctx* c, const mh_pcs_params* params, int n_airs, mh_air* const* airs, const uint64_t* const* traces_rowmajor,
                  const int* log_heights, const uint64_t* public_values, size_t n_public_values, const uint64_t challenger_state[12],
                  const uint64_t* pre_observe, size_t n_pre_observe, mh_aux_builder aux_builder, void* user, mh_proof** out) {
  MH_TRY(c)
  MH_REQUIRE(c && params && airs && traces_rowmajor && log_heights && challenger_state && out, "null argument");
  MH_REQUIRE(n_airs > 0 && n_airs <= 256, "need between 1 and 256 AIR instances");
  MH_REQUIRE(public_values || !n_public_values, "null public values");
  MH_REQUIRE(pre_observe || !n_pre_observe, "null pre_observe");
  HIP_CHECK(hipSetDevice(c->device));
  std::vector<int> order(n_airs);
  std::iota(order.begin(), order.end(), 0);
  for (int i = 0; i < n_airs; i++) MH_REQUIRE(airs[i] && traces_rowmajor[i] && log_heights[i] >= 0 && log_heights[i] <= 29, "bad trace argument");
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return log_heights[a] < log_heights[b]; });
  std::vector<std::unique_ptr<mh_trace>> owned(n_airs);
  for (int i : order) owned[i].reset(trace_upload_async(c, traces_rowmajor[i], log_heights[i], airs[i]->main_width));
  std::vector<mh_trace*> tr(n_airs);
  for (int i = 0; i < n_airs; i++) tr[i] = owned[i].get();
  std::unique_ptr<mh_proof> p(new mh_proof());
  prove_impl(c, *params, n_airs, airs, tr.data(), public_values, n_public_values, challenger_state, pre_observe, n_pre_observe,
             aux_builder, user, *p, Dist{});
  *out = p.release();
  MH_CATCH
}

int mh_prove_sharded(mh_ctx* c, const mh_comm* comm, const mh_pcs_params* params, int n_airs, mh_air* const* airs,
                     mh_trace* const* traces, const uint64_t* public_values, size_t n_public_values,
                     const uint64_t challenger_state[12], const uint64_t* pre_observe, size_t n_pre_observe,
                     mh_aux_builder aux_builder, void* user, mh_proof** out) {
  MH_TRY(c)
  MH_REQUIRE(c && comm && params && airs && traces && challenger_state && out, "null argument");
  MH_REQUIRE(public_values || !n_public_values, "null public values");
  MH_REQUIRE(pre_observe || !n_pre_observe, "null pre_observe");
  MH_REQUIRE(comm->world >= 1 && (comm->world & (comm->world - 1)) == 0 && comm->rank >= 0 && comm->rank < comm->world,
             "world must be a power of two and 0 <= rank < world");
  MH_REQUIRE(comm->world == 1 || (comm->all_to_all && comm->all_gather && comm->all_reduce_sum_u64), "missing collective callbacks");
  HIP_CHECK(hipSetDevice(c->device));
  Dist d;
  d.comm = comm; d.rank = comm->rank; d.world = comm->world;
  while ((1 << d.logG) < d.world) d.logG++;
  std::unique_ptr<mh_proof> p(new mh_proof());
  prove_impl(c, *params, n_airs, airs, traces, public_values, n_public_values, challenger_state, pre_observe, n_pre_observe,
             aux_builder, user, *p, d);
  *out = p.release();
  MH_CATCH
}

// ---- staged session: the host owns the transcript --------------------------
#define MH_STRY MH_TRY(s ? s->c : nullptr) MH_REQUIRE(s, "null session"); HIP_CHECK(hipSetDevice(s->c->device));
static e2 e2_in(const uint64_t v[2]) { return e2{gl_canon(v[0]), gl_canon(v[1])}; }
static void e2_out(const std::vector<e2>& v, uint64_t* out) {
  for (size_t i = 0; i < v.size(); i++) { out[2 * i] = v[i].c0; out[2 * i + 1] = v[i].c1; }
}

int mh_session_begin(mh_ctx* c, const mh_comm* comm, const mh_pcs_params* params, int n_airs, mh_air* const* airs,
                     mh_trace* const* traces, const uint64_t* public_values, size_t n_public_values, mh_session** out) {
  MH_TRY(c)
  MH_REQUIRE(c && params && airs && traces && out, "null argument");
  MH_REQUIRE(public_values || !n_public_values, "null public values");
  HIP_CHECK(hipSetDevice(c->device));
  Dist d;
  if (comm && comm->world > 1) {
    MH_REQUIRE((comm->world & (comm->world - 1)) == 0 && comm->rank >= 0 && comm->rank < comm->world,
               "world must be a power of two and 0 <= rank < world");
    MH_REQUIRE(comm->all_to_all && comm->all_gather && comm->all_reduce_sum_u64, "missing collective callbacks");
    d.comm = comm; d.rank = comm->rank; d.world = comm->world;
    while ((1 << d.logG) < d.world) d.logG++;
  }
  std::unique_ptr<mh_session> s(new mh_session());
  s->begin(c, *params, n_airs, airs, traces, public_values, n_public_values, d);
  *out = s.release();
  MH_CATCH
}
void mh_session_free(mh_session* s) {
  if (!s) return;
  (void)hipSetDevice(s->c->device);
  PoolScope ps(s->c);
  delete s;
}
int mh_session_shape(const mh_session* s, mh_session_shape_t* out) {
  if (!s || !out) return MH_ERR_INVALID;
  out->log_lde_height = s->L;
  out->num_randomness = s->max_rand;
  out->num_aux_values = s->num_aux_values();
  out->ood_width = s->ood_width();
  out->num_fri_rounds = s->rounds;
  out->final_poly_len = s->final_poly_len();
  return MH_OK;
}
int mh_session_commit_main(mh_session* s, uint64_t root[4]) {
  MH_STRY
  MH_REQUIRE(root, "null argument");
  s->commit_main(root);
  MH_CATCH
}
int mh_session_commit_aux(mh_session* s, const uint64_t* randomness, mh_aux_builder aux_builder, void* user, uint64_t root[4],
                          uint64_t* aux_values_out) {
  MH_STRY
  MH_REQUIRE(root && (randomness || !s->max_rand) && (aux_values_out || !s->num_aux_values()), "null argument");
  std::vector<e2> rnd, vals;
  for (size_t i = 0; i < s->max_rand; i++) rnd.push_back(e2_in(randomness + 2 * i));
  s->commit_aux(rnd, aux_builder, user, root, vals);
  e2_out(vals, aux_values_out);
  MH_CATCH
}
int mh_session_commit_quotient(mh_session* s, const uint64_t alpha[2], const uint64_t beta[2], uint64_t root[4]) {
  MH_STRY
  MH_REQUIRE(alpha && beta && root, "null argument");
  s->commit_quotient(e2_in(alpha), e2_in(beta), root);
  MH_CATCH
}
int mh_session_ood_point_ok(const mh_session* s, const uint64_t z[2]) { return s && z && s->ood_point_ok(e2_in(z)) ? 1 : 0; }
int mh_session_ood(mh_session* s, const uint64_t z[2], uint64_t* evals_out) {
  MH_STRY
  MH_REQUIRE(z && evals_out, "null argument");
  s->ood(e2_in(z));
  e2_out(s->ev0, evals_out);
  e2_out(s->ev1, evals_out + 2 * s->W);
  MH_CATCH
}
int mh_session_deep(mh_session* s, const uint64_t alpha[2], const uint64_t beta[2]) {
  MH_STRY
  MH_REQUIRE(alpha && beta, "null argument");
  s->deep(e2_in(alpha), e2_in(beta));
  MH_CATCH
}
int mh_session_fri_commit(mh_session* s, uint64_t root[4]) {
  MH_STRY
  MH_REQUIRE(root, "null argument");
  s->fri_commit(root);
  MH_CATCH
}
int mh_session_fri_fold(mh_session* s, const uint64_t beta[2]) {
  MH_STRY
  MH_REQUIRE(beta, "null argument");
  s->fri_fold_round(e2_in(beta));
  MH_CATCH
}
int mh_session_fri_final(mh_session* s, uint64_t* coeffs_out) {
  MH_STRY
  MH_REQUIRE(coeffs_out, "null argument");
  std::vector<e2> desc;
  s->fri_final(desc);
  e2_out(desc, coeffs_out);
  MH_CATCH
}
// The hints of all openings, in transcript order, as an mh_proof holding only hinted fields/commitments.
int mh_session_open(mh_session* s, const uint64_t* indices, size_t n_indices, mh_proof** out) {
  MH_STRY
  MH_REQUIRE(indices && n_indices && out, "null/empty argument");
  std::unique_ptr<mh_proof> p(new mh_proof());
  memset(p->digest, 0, sizeof p->digest);
  std::vector<size_t> idx(indices, indices + n_indices);
  s->open(idx, p->fields, p->commitments);
  for (int i = 0; i < s->n_airs; i++) p->log_trace_heights.push_back((uint8_t)s->lhs[i]);
  *out = p.release();
  MH_CATCH
}
// Proof-of-work: the smallest witness w such that observing w (after the `n_pending` absorbed-but-not-yet-
// permuted felts) and sampling `bits` bits yields zero  (p3 GrindingChallenger::grind, any valid witness verifies).
int mh_grind(mh_ctx* c, const uint64_t state[12], const uint64_t* pending, size_t n_pending, int bits, uint64_t* witness) {
  MH_TRY(c)
  MH_REQUIRE(c && state && witness && (pending || !n_pending), "null argument");
  MH_REQUIRE(n_pending < 8 && bits >= 0 && bits <= 32, "pending input must be shorter than the rate; bits in 0..32");
  HIP_CHECK(hipSetDevice(c->device));
  HostTranscript tr;
  tr.ch.hash = c->lmcs;
  MH_REQUIRE(c->lmcs == MH_LMCS_POSEIDON2 || c->lmcs == MH_LMCS_RPO || c->lmcs == MH_LMCS_RPX, "mh_grind searches the duplex sponge's witness: algebraic configurations only");
  for (int i = 0; i < 12; i++) tr.ch.st[i] = gl_canon(state[i]);
  for (size_t i = 0; i < n_pending; i++) tr.ch.in.push_back(gl_canon(pending[i]));
  do_grind(c, tr, bits);
  *witness = tr.fields.back();
  MH_CATCH
}

int mh_grind_bytes(mh_ctx* c, const uint8_t* input, size_t n_input, int bits, uint64_t* witness) {
  MH_TRY(c)
  MH_REQUIRE(c && witness && (input || !n_input), "null argument");
  MH_REQUIRE(bits >= 0 && bits <= 32, "bits in 0..32");
  MH_REQUIRE(c->lmcs == MH_LMCS_BLAKE3 || c->lmcs == MH_LMCS_KECCAK, "mh_grind_bytes searches a hash challenger's witness: Blake3 / Keccak contexts");
  HIP_CHECK(hipSetDevice(c->device));
  if (bits == 0) {
    *witness = 0;
  } else {
    std::vector<uint8_t> prefix(input, input + n_input);
    *witness = fri_grind_bytes(c, c->lmcs, prefix, bits);
  }
  MH_CATCH
}

void mh_proof_free(mh_proof* p) { delete p; }
size_t mh_proof_num_fields(const mh_proof* p) { return p ? p->fields.size() : 0; }
size_t mh_proof_num_commitments(const mh_proof* p) { return p ? p->commitments.size() / 4 : 0; }
const uint64_t* mh_proof_fields(const mh_proof* p) { return p ? p->fields.data() : nullptr; }
const uint64_t* mh_proof_commitments(const mh_proof* p) { return p ? p->commitments.data() : nullptr; }
const uint64_t* mh_proof_digest(const mh_proof* p) { return p ? p->digest : nullptr; }
size_t mh_proof_num_traces(const mh_proof* p) { return p ? p->log_trace_heights.size() : 0; }
const uint8_t* mh_proof_log_trace_heights(const mh_proof* p) { return p ? p->log_trace_heights.data() : nullptr; }

// StarkProofData { log_trace_heights: Vec<u8>, transcript: { fields: Vec<Felt>, commitments: Vec<[Felt;4]> } }
// (crates/lifted-stark/src/proof.rs:58-63; TranscriptData: crates/stark-transcript/src/data.rs:8-14) as the reference's
// wincode::config::Configuration::default() + serde-wincode SerdeCompat writes it (prover/src/lib.rs:347-353): wincode 0.5.5
// (external, Cargo.lock) is the bincode-compatible fixed-width little-endian encoding -- a Vec is a u64 LE length followed by
// its elements, a u8 is one byte, a Felt (serde newtype over p3 Goldilocks) its canonical u64 LE, a commitment [Felt; 4] four
// of those without a length.  PARITY UNPINNED against reference bytes until tools/ref_fixtures has been run.
size_t mh_proof_serialize(const mh_proof* p, uint8_t* out, size_t cap) {
  if (!p) return 0;
  const size_t need = 8 + p->log_trace_heights.size() + 8 + 8 * p->fields.size() + 8 + 8 * p->commitments.size();
  if (!out || cap < need) return need;
  uint8_t* o = out;
  auto put64 = [&](u64 v) {
    memcpy(o, &v, 8);
    o += 8;
  };
  put64(p->log_trace_heights.size());
  memcpy(o, p->log_trace_heights.data(), p->log_trace_heights.size());
  o += p->log_trace_heights.size();
  put64(p->fields.size());
  memcpy(o, p->fields.data(), 8 * p->fields.size());
  o += 8 * p->fields.size();
  put64(p->commitments.size() / 4);
  memcpy(o, p->commitments.data(), 8 * p->commitments.size());
  return need;
}

// The inverse of mh_proof_serialize (what the reference's verifier entry point does with the bytes before anything else:
// verifier/src/lib.rs:320-330, crates/test-utils/src/recursive_verifier.rs:84-96, both under a 64 MiB limit).  Rejects
// truncated input, trailing bytes, length prefixes that do not fit the input and non-canonical field elements (serde's
// Goldilocks deserialiser refuses values >= p).  The digest is not part of StarkProofData: it comes back zeroed.
int mh_proof_deserialize(const uint8_t* bytes, size_t len, mh_proof** out) {
  if (!bytes || !out || len > ((size_t)64 << 20)) return MH_ERR_INVALID;
  size_t pos = 0;
  auto get64 = [&](u64& v) {
    if (len - pos < 8) return false;
    memcpy(&v, bytes + pos, 8);
    pos += 8;
    return true;
  };
  std::unique_ptr<mh_proof> p(new mh_proof());
  memset(p->digest, 0, sizeof p->digest);
  u64 n = 0;
  if (!get64(n) || n > len - pos || n == 0 || n > 256) return MH_ERR_INVALID;
  p->log_trace_heights.assign(bytes + pos, bytes + pos + n);
  pos += n;
  if (!get64(n) || n > (len - pos) / 8) return MH_ERR_INVALID;
  p->fields.resize(n);
  memcpy(p->fields.data(), bytes + pos, 8 * n);
  pos += 8 * n;
  if (!get64(n) || n > (len - pos) / 32) return MH_ERR_INVALID;
  p->commitments.resize(4 * n);
  memcpy(p->commitments.data(), bytes + pos, 32 * n);
  pos += 32 * n;
  if (pos != len) return MH_ERR_INVALID;
  for (u64 v : p->fields)
    if (v >= GL_P) return MH_ERR_INVALID;
  // commitments are 32 opaque bytes here: [Felt; 4] in the algebraic configurations (the verifier rejects a non-canonical word
  // there), [u8; 32] / [u64; 4] under Blake3 / Keccak -- the same framing
  *out = p.release();
  return MH_OK;
}

}  // extern "C"
