// Internal interfaces between the translation units of libmidenhip (not part of the C ABI).
#pragma once
#include "ctx.hpp"
#include <vector>

// ---- ntt.hip ---------------------------------------------------------------------------------
void launch_transpose_rm_to_cm(mh_ctx* c, const u64* in_rowmajor, u64* out_colmajor, size_t n, size_t w, hipStream_t stream = nullptr);  // null: the compute stream
void ntt_inverse_dif_inplace(mh_ctx* c, u64* cols, size_t n_cols, int log_n);
void ntt_inverse_dif(mh_ctx* c, const u64* src, u64* dst, size_t n_cols, int log_n);  // src == dst: in place
void ntt_forward_cosets(mh_ctx* c, const u64* coef_br, size_t n_cols, int log_n, const std::vector<u64>& bases, u64* out, size_t out_col_stride = 0,
                        size_t group_cols = 0);
void lde_columns(mh_ctx* c, const u64* cols_in, size_t n_cols, int log_n, u64 in_shift, const std::vector<u64>& out_shifts,
                 u64* out, u64* scratch);
void lde_coefficients(mh_ctx* c, const u64* cols_in, size_t n_cols, int log_n, u64* coef_br);
void lde_forward_group(mh_ctx* c, const u64* coef_br, size_t n_cols, int log_n, u64 in_shift, const std::vector<u64>& group_shifts,
                       u64* out, size_t out_col_stride);

// ---- device-resident objects -------------------------------------------------------------------
// A trace matrix as uploaded: column-major, natural row order, canonical felts.
struct mh_trace {
  mh_ctx* ctx;
  int log_n;
  size_t width;
  DevBuf cols;  // [width][N]
  // mh_trace_upload_async: the H2D copy + transpose run on the context's copy stream; `ready` is recorded behind them.  Every
  // consumer on the compute stream orders itself after it with trace_wait_ready() (a stream-side wait, no host block).
  hipEvent_t ready = nullptr;
  mutable DevBuf staging;  // the row-major landing buffer of the DMA: back to the pool when the first consumer has ordered itself after `ready`
  // mh_trace_upload_cols_async (a COLUMN-major host matrix): the columns arrive in groups of `col_group`; group g is complete behind
  // col_ready[g] (`ready` = the last of them), so the LDE of group g runs while group g + 1 is still on the PCIe link.
  size_t col_group = 0;
  std::vector<hipEvent_t> col_ready;
  ~mh_trace() {
    // the copy stream may still be writing cols / reading staging -- also when an upload failed half-way (`ready` not recorded yet)
    if (ready) (void)hipEventSynchronize(ready);
    else if ((staging.p || !col_ready.empty()) && ctx && ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    for (hipEvent_t e : col_ready)
      if (e != ready) (void)hipEventDestroy(e);
    if (ready) (void)hipEventDestroy(ready);
  }
};
// Order everything enqueued on c->stream from here on after the trace's upload (no-op for a synchronously uploaded trace).
void trace_wait_ready(mh_ctx* c, const mh_trace* t);
mh_trace* trace_upload_cols_async(mh_ctx* c, const u64* colmajor, int log_n, size_t width);

// A committed (LDE'd) matrix: coset-major column-major: lde[(c*B + j)*N + r] = f_c(shift*w_K^j*w_H^r)
// = evaluation at natural index i = r*B + j of the max-domain-lifted polynomial.
// A rank of a sharded proof stores only cosets [coset0, coset0 + 2^log_cosets) of the 2^log_blowup.
struct LdeMatrix {
  int log_n;  // trace height
  size_t width;
  DevBuf lde;
  int log_cosets = 0;  // coset bits stored here (= log_blowup on a single GPU)
  size_t coset0 = 0;   // global index of the first stored coset
};

// Collectives of one sharded proof (include/midenhip.h mh_comm); world == 1: every call is a no-op.
struct mh_comm;
struct Dist {
  const mh_comm* comm = nullptr;
  int rank = 0, world = 1, logG = 0;
  bool on() const { return world > 1; }
  void all_to_all(mh_ctx* c, const void* send, void* recv, size_t bytes_per_peer) const;
  void all_gather(mh_ctx* c, const void* send, void* recv, size_t bytes_per_rank) const;
  void all_reduce_sum(mh_ctx* c, u64* buf, size_t n) const;
};

// LMCS tree over a group of LDE matrices (ascending heights).  Node (depth d, natural position p)
// lives at layer_ptr(d) + 4*node_slot(d,p): leaf-side layers are coset-major (see lmcs.hip).
struct mh_tree {
  mh_ctx* ctx;
  int log_blowup;  // number of coset bits in the leaf layer layout (may be 0)
  int log_height;  // tree depth L (leaves = 2^L)
  int lmcs = 0;    // MH_LMCS_* hasher the layers were built with (the context's, at build time)
  std::vector<LdeMatrix> mats;
  // FRI round trees (fri.hip) commit one EF layer instead of LDE matrices: rows are rebuilt from it
  DevBuf fri_layer;      // EF pairs, coset-major [2^fri_log_cosets][2^fri_log_rows] (this rank's cosets)
  int fri_log_rows = -1; // rows per coset of the layer (before grouping by arity); -1 = not a FRI tree
  int fri_log_arity = 0;
  int fri_log_cosets = 0;
  size_t fri_coset0 = 0;
  // Sharded proofs: `nodes` is the subtree over this rank's ROW range (all cosets), of height
  // log_height = full height - shard_logG; `cap` holds the top shard_logG + 1 levels on the host
  // (node (d, p) at cap[4 * ((1 << d) - 1 + p)]).  shard_logG == 0: an ordinary full tree.
  int shard_logG = 0, shard_rank = 0;
  std::vector<u64> cap;
  DevBuf nodes;                    // all layers, leaf layer first
  std::vector<size_t> layer_off;   // layer_off[d] = element offset (in digests) of depth-d layer
  u64 root[4];
  size_t node_slot(int d, size_t p) const {
    int cbits = d - (log_height - log_blowup);
    if (cbits <= 0) return p;
    size_t j = p & (((size_t)1 << cbits) - 1), r = p >> cbits;
    return (j << (log_height - log_blowup)) + r;
  }
};

// ---- lmcs.hip --------------------------------------------------------------------------------
void poseidon2_permute_device(mh_ctx* c, u64* states_soa, size_t n);  // [12][n]
double poseidon2_register_rate(mh_ctx* c);  // permutations/s with the state held in registers (VALU ceiling)
// Build leaf digests + all layers for `t->mats` (already filled); sets t->root.
void lmcs_build_tree(mh_ctx* c, mh_tree* t);
// Pieces of the above for trees whose leaf digests come from another kernel (FRI rounds):
void lmcs_hash_leaves(mh_ctx* c, const std::vector<LdeMatrix>& mats, int lb, u64* digests);
// the leaves [q_begin, q_begin + q_count) only (a group of cosets of a tree whose matrices all have one height)
void lmcs_hash_leaves_range(mh_ctx* c, const std::vector<LdeMatrix>& mats, int lb, u64* digests, size_t q_begin, size_t q_count);
bool lmcs_leaves_rangeable(mh_ctx* c, const std::vector<LdeMatrix>& mats);
void lmcs_alloc_layers(mh_tree* t, int log_height);  // sets log_height, layer_off, nodes
u64* lmcs_leaf_layer(mh_tree* t);                     // device pointer of the leaf digest layer
void lmcs_compress_layers(mh_ctx* c, mh_tree* t);     // leaf layer -> root (copies root to host)
// Gather opened rows (aligned, per sorted unique index) and missing siblings.
void lmcs_open(mh_ctx* c, const mh_tree* t, const std::vector<size_t>& sorted_unique_idx, size_t alignment,
               std::vector<u64>& fields, std::vector<u64>& commitments, const Dist* dist = nullptr);
// the same in three steps, so that the openings of several trees share one gather, one read-back and (sharded) one all-reduce
struct OpenPlan {
  size_t first = 0, n_fields = 0, n = 0;                     // this tree's slice of the gather list: rows, then sibling digests
  std::vector<std::pair<size_t, const u64*>> cap_fill;       // (offset in the slice, host digest of the sharded tree's cap)
};
OpenPlan lmcs_open_plan(const mh_tree* t, const std::vector<size_t>& idx, size_t alignment, const Dist* dist, std::vector<const u64*>& ptrs);
void lmcs_open_run(mh_ctx* c, const std::vector<const u64*>& ptrs, const Dist* dist, std::vector<u64>& host);
void lmcs_open_take(const OpenPlan& plan, std::vector<u64>& host, std::vector<u64>& fields, std::vector<u64>& commitments);
// Sharded tree build: local leaf digests [2^lbl][2^log_rows] -> all-to-all -> subtree over this rank's
// row range -> all-gather of subroots -> cap on the host.  t->log_blowup = global coset bits.
void lmcs_build_sharded(mh_ctx* c, mh_tree* t, const Dist& dist, const u64* local_digests, int log_rows);
void lmcs_host_compress(int lmcs, const u64* pair, u64* out);
// one tree level on the host: out[q] = compress(children[2q], children[2q + 1]), q < n_out (natural order); eight nodes per call of
// the AVX-512 Poseidon2 (p2_host_simd.cpp) where the CPU has it, the scalar code otherwise
void lmcs_host_compress_level(int lmcs, const u64* children, size_t n_out, u64* out);
bool p2_host_simd_available();
void p2_host_compress8(const u64* pairs, int n, u64* out);
void p2_host_permute8(u64* states /* [8][12] */);
std::vector<std::pair<int, size_t>> lmcs_missing_siblings(const std::vector<size_t>& sorted_unique_idx, int depth);

// ---- prover.hip (commit helpers shared with the C ABI) -------------------------------------------
std::vector<u64> coset_shifts(int log_n, int lb);  // g*w_K^j, j < 2^lb, for the canonical shift of order log_n+lb
mh_trace* trace_upload(mh_ctx* c, const u64* rowmajor, int log_n, size_t width);
mh_trace* trace_upload_async(mh_ctx* c, const u64* rowmajor, int log_n, size_t width);
mh_trace* trace_zeros(mh_ctx* c, int log_n, size_t width);
mh_tree* commit_traces(mh_ctx* c, const std::vector<const mh_trace*>& traces, int log_blowup);
// LDE of `tr` onto the cosets [first, first + count) only of the 2^lb cosets (count a power of two)
LdeMatrix lde_trace_cosets(mh_ctx* c, const mh_trace* tr, int lb, size_t first, size_t count);
// ---- quotient.hip ------------------------------------------------------------------------------
struct mh_air;
#include "gl.cuh"
// log_d = quotient degree of the evaluation (global); the rank evaluates the 2^(log_d - logG) cosets it
// stores (all of them on one GPU).  acc layouts are [2 * D_local][n].
void quotient_eval_accumulate(mh_ctx* c, const mh_air* air, const LdeMatrix& main, const LdeMatrix& aux, const LdeMatrix* prep,
                              int log_blowup, int log_d,
                              const std::vector<u64>& publics, const std::vector<e2>& randomness, const std::vector<e2>& aux_values,
                              e2 alpha, const u64* acc_in, int log_n_prev, e2 beta, u64* acc_out);
void quotient_upsample_accumulate(mh_ctx* c, const u64* q_small, int log_n, int log_blowup, int log_dj, int log_d, const u64* acc_in,
                                  int log_n_prev, e2 beta, u64* acc_out, size_t t_first, size_t n_local);
// ---- deep.hip ----------------------------------------------------------------------------------
struct OodJob {
  const LdeMatrix* m = nullptr;
  e2 y0, y1;                            // the two evaluation points, already lifted to this matrix's height
  size_t col_begin = 0, col_end = ~(size_t)0;
  std::vector<e2> out0, out1;           // per column (zero outside [col_begin, col_end))
};
void deep_ood_eval_batch(mh_ctx* c, std::vector<OodJob>& jobs, int log_blowup);
void deep_ood_eval_matrix(mh_ctx* c, const LdeMatrix& m, int log_blowup, e2 y0, e2 y1, std::vector<e2>& out0, std::vector<e2>& out1,
                          size_t col_begin = 0, size_t col_end = (size_t)-1);
void deep_assemble(mh_ctx* c, const std::vector<const LdeMatrix*>& mats, const std::vector<uint32_t>& coef_off, int log_n, int log_blowup,
                   const std::vector<e2>& negc, e2 z0, e2 z1, e2 fred0, e2 fred1, e2 beta, u64* out);
// ---- fri.hip -----------------------------------------------------------------------------------
void fri_leaf_hash(mh_ctx* c, const u64* ev, int log_rows, int cbits, int log_arity, u64* digests);
// cbits = coset bits stored locally, cbits_global / coset0 locate them in the whole layer
void fri_fold(mh_ctx* c, const u64* ev, int log_rows, int cbits, int cbits_global, size_t coset0, int log_arity, e2 beta, u64* out);
void fri_to_natural(mh_ctx* c, const u64* ev, int log_rows, int cbits, u64* out);
u64 fri_grind(mh_ctx* c, const u64 st[12], const u64* in, int n_in, int bits);
// byte challengers (Blake3 / Keccak): smallest w with sample_bits(bits) == 0 after observing it; prefix = the input buffer
u64 fri_grind_bytes(mh_ctx* c, int lmcs, const std::vector<uint8_t>& prefix, int bits);

// logup.hip
struct mh_lookup;
mh_trace* lookup_build_aux(mh_ctx* c, const mh_lookup* lk, const mh_trace* main, const mh_trace* prep, const std::vector<e2>& randomness,
                           e2* acc_final);
