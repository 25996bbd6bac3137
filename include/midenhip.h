/* libmidenhip — C ABI of the MI355X-native STARK proving backend for Miden VM.
 *
 * Drop-in boundary: these entry points are what a Rust FFI shim under
 * `miden_prover::prove_stark()` (reference prover/src/lib.rs:317-355) binds; see INTEGRATION.md
 * for the `extern "C"` block and the cfg-gated `prove_stark_hip`.  Conventions, following the
 * reference's own C-ABI precedent (crates/crypto/src/hash/algebraic_sponge/rescue/arch/mod.rs:12-21):
 *   - plain pointers and sizes only; all field elements are uint64_t Goldilocks values
 *     (p = 2^64 - 2^32 + 1); inputs may be non-canonical (< 2^64), outputs are canonical;
 *   - extension-field elements are 2 consecutive uint64_t [c0, c1] (x^2 = 7);
 *   - every function returns 0 on success or an MH_ERR_* code; mh_last_error(ctx) gives the
 *     message; no exception or panic crosses the boundary;
 *   - the caller owns host buffers; the library owns device buffers until the matching *_free;
 *   - one ctx per proving thread, calls on one ctx are blocking and not re-entrant.
 */
#ifndef MIDENHIP_H
#define MIDENHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_OK 0
#define MH_ERR_INVALID 1
#define MH_ERR_HIP 2
#define MH_ERR_OOM 3
#define MH_ERR_INTERNAL 4
#define MH_ERR_COMM 5      /* a collective of a sharded proof failed or did not complete within $MH_COMM_TIMEOUT_S (default 120 s) */

typedef struct mh_ctx mh_ctx;
typedef struct mh_trace mh_trace; /* device-resident trace matrix (column-major) */
typedef struct mh_tree mh_tree;   /* device-resident LMCS tree + its LDE matrices */
typedef struct mh_air mh_air;     /* an AIR: constraint DAG compiled for the device interpreter */
typedef struct mh_proof mh_proof; /* host-resident proof: transcript fields + commitments */

/* ---- context ------------------------------------------------------------------------------ */
int mh_ctx_create(int device_id, mh_ctx** out);
void mh_ctx_destroy(mh_ctx* ctx);
/* Release the device buffers the context keeps pooled between proofs (freed automatically on OOM and at destroy). */
int mh_ctx_trim(mh_ctx* ctx);
/* Device memory as a long-lived service sees it: out[0] = bytes cached in the context's buffer pool (returned by mh_ctx_trim),
 * out[1] = bytes of twiddle / coset tables the context keeps, out[2] = free and out[3] = total bytes of the device (hipMemGetInfo). */
int mh_ctx_mem_stats(mh_ctx* ctx, uint64_t out[4]);
const char* mh_last_error(const mh_ctx* ctx);
int mh_device_count(void);

/* Kernel profiler: HIP events recorded on the ctx's private stream around each kernel class.
 * mh_prof_get returns accumulated milliseconds, attributed algorithmic bytes and launch count. */
int mh_prof_enable(mh_ctx* ctx, int on);
/* Restrict the profiler to ONE kernel class (e.g. "lmcs_leaf_absorb"); NULL or "" = every class and span again.  An event
 * record is a barrier packet on the stream: recording every class costs a 2^20-row proof ~0.5 ms (a proof has ~130 scopes),
 * one class ~10 us. */
int mh_prof_filter(mh_ctx* ctx, const char* name);
int mh_prof_reset(mh_ctx* ctx);
int mh_prof_get(mh_ctx* ctx, const char* name, double* ms, double* bytes, long* count);
/* writes up to cap bytes of a '\n'-separated "name ms bytes count" listing */
int mh_prof_dump(mh_ctx* ctx, char* buf, size_t cap);

/* ---- unit-parity entry points --------------------------------------------------------------- */
/* Poseidon2 permutation (replaces Poseidon2Permutation256::permute_mut,
 * crates/crypto/src/hash/algebraic_sponge/poseidon2/mod.rs:340,380-386) on n states of 12 felts,
 * host array-of-states layout [n][12], in place. */
int mh_poseidon2_permute(mh_ctx* ctx, uint64_t* states, size_t n);
/* Measurement aid: permutations per second of the same device permutation with the state held in
 * registers (no memory traffic) = the VALU ceiling bench.py reports hash kernels against. */
int mh_poseidon2_register_rate(mh_ctx* ctx, double* perms_per_second);

/* Coset LDE of a host row-major matrix (replaces Radix2DitParallel::coset_lde_batch as called at
 * crates/lifted-stark/src/prover/commit.rs:173): `out` receives the (n<<added_bits) x width
 * row-major matrix in the REFERENCE's storage order (physical row r = evaluation at
 * shift * w^bitrev(r)), for parity checks. */
int mh_coset_lde_batch(mh_ctx* ctx, const uint64_t* rowmajor, int log_n, size_t width, int added_bits,
                       uint64_t shift, uint64_t* out);

/* ---- device-resident traces ----------------------------------------------------------------- */
/* Upload a host row-major RowMajorMatrix<Felt> (values, width) of height 2^log_n: one H2D copy +
 * an on-device transpose to column-major; canonicalises felts. */
int mh_trace_upload(mh_ctx* ctx, const uint64_t* rowmajor, int log_n, size_t width, mh_trace** out);
/* The same without blocking: returns as soon as the copy is enqueued.  The DMA and the transpose run on the context's COPY
 * stream, under whatever the compute stream is doing; every consumer of the trace (mh_prove, mh_session_*, mh_commit_traces,
 * mh_lookup_build_aux, mh_trace_download) orders itself after the upload on the GPU, no host wait.  Start the uploads of all
 * the matrices of a statement in proof order (ascending height, ties by instance index), then call mh_prove: the LDE and the leaf
 * sponges of matrix k run while matrices k+1.. are still on the PCIe link; only the first matrix's copy is exposed.  `rowmajor`
 * should be page-locked (mh_host_alloc) -- pageable memory makes the copy synchronous -- and must stay valid and unmodified
 * until the proof has been made (or mh_trace_wait has returned).  Reference: prover/src/lib.rs:317-355 hands over host
 * RowMajorMatrix values. */
int mh_trace_upload_async(mh_ctx* ctx, const uint64_t* rowmajor, int log_n, size_t width, mh_trace** out);
/* The same for a COLUMN-major host matrix, colmajor[c * 2^log_n + r] (a trace builder that writes columns: SURVEY 8(f) #4).
 * Every column is one contiguous copy and there is no transpose, so ONE matrix pipelines too: the columns go up in groups of
 * eight, and the LDE of a group starts when that group has landed -- the exposed part of the upload is the first group.  (A
 * row-major matrix cannot be split like this: its column windows move at 16-36 GB/s.)  Same lifetime rules as above. */
int mh_trace_upload_cols_async(mh_ctx* ctx, const uint64_t* colmajor, int log_n, size_t width, mh_trace** out);
/* Blocks until the upload of `t` has landed (the host buffer may be reused) and releases its landing buffer. */
int mh_trace_wait(mh_ctx* ctx, mh_trace* t);
/* The same for a row-major matrix that is already in device memory (a GPU trace generator): no PCIe traffic. */
int mh_trace_from_device(mh_ctx* ctx, const uint64_t* device_rowmajor, int log_n, size_t width, mh_trace** out);
void mh_trace_free(mh_trace* t);
/* Page-locked host memory (hipHostMalloc): a trace built in it uploads by direct DMA at PCIe line rate; any
 * other host pointer works too, staged by the runtime (several times slower).  NULL on failure. */
void* mh_host_alloc(size_t bytes);
void mh_host_free(void* p);

/* ---- commitments (K1-K3) -------------------------------------------------------------------- */
/* The commitment scheme's hasher (the reference's StarkConfig::Lmcs, air/src/config.rs:212-305).  MH_LMCS_POSEIDON2 (default):
 * StatefulSponge leaves + TruncatedPermutation nodes.  MH_LMCS_BLAKE3: the Blake3_256 configuration's LMCS (config.rs:275-289,
 * ProvingOptions::default()): leaf = chain over the matrices of blake3(state || row felts as 8 LE bytes each) from a zero
 * state (crates/stateful-hasher/src/chaining.rs:32-50, alignment 1), node = blake3(left || right); a digest travels as
 * four uint64_t = its 32 bytes little-endian.  MH_LMCS_KECCAK: the Keccak configuration's LMCS (config.rs:307-353): the
 * overwrite-mode sponge over 64-bit lanes with Keccak-f[1600] (25 lanes, 17 felts as canonical u64 per permutation, digest =
 * lanes 0..3; alignment 17), node = one permutation over left || right in a zero state.  MH_LMCS_RPO / MH_LMCS_RPX: the
 * other two ALGEBRAIC configurations (config.rs:224-245): the Poseidon2 LMCS and duplex challenger with the Rescue Prime
 * permutations (crates/crypto/src/hash/algebraic_sponge/rescue/).
 * The setting is the context's StarkConfig: it applies to everything the context commits, opens and proves -- mh_commit_traces,
 * mh_commit_traces_sharded, mh_tree_open, mh_prove / mh_prove_sharded and the staged session.  The row alignment follows the
 * hasher (lmcs.alignment(): the sponge's rate, 8 or 17; 1 for the chaining hasher: OOD blocks and opened rows are then
 * unpadded), the FRI leaves use it, and so does the challenger of the one-shot prover: the duplex sponge over the configuration's
 * permutation (Poseidon2, RPO, RPX) or, for Blake3 and Keccak, the library's restatement of p3's SerializingChallenger64 over a
 * HashChallenger (external code with no in-tree mirror: a shim that wants p3's own keeps the transcript on the host and drives
 * mh_session_*, which never sees a challenger).  mh_verify / mh_verify_ex are the Poseidon2 configuration; mh_verify_lmcs takes
 * the id.  mh_grind is the duplex sponge's PoW search (algebraic configurations). */
#define MH_LMCS_POSEIDON2 0
#define MH_LMCS_BLAKE3 1
#define MH_LMCS_KECCAK 2
#define MH_LMCS_RPO 3
#define MH_LMCS_RPX 4
int mh_ctx_set_lmcs(mh_ctx* ctx, int lmcs);
int mh_ctx_get_lmcs(const mh_ctx* ctx);
/* blake3(data) (host only; unit-parity entry point, like mh_poseidon2_permute) */
void mh_blake3(const uint8_t* data, size_t n, uint8_t out32[32]);

/* commit_traces (crates/lifted-stark/src/prover/commit.rs:142-180): per trace (proof order =
 * ascending height) coset-LDE by 2^log_blowup on the canonical shift of ITS lde order, then the
 * aligned LMCS tree (Lmcs::build_aligned_tree, lmcs/config.rs:125-137).  root = 4 felts. */
int mh_commit_traces(mh_ctx* ctx, int n_traces, mh_trace* const* traces, int log_blowup, mh_tree** out,
                     uint64_t root[4]);
void mh_tree_free(mh_tree* t);
int mh_tree_root(const mh_tree* t, uint64_t root[4]);
int mh_tree_log_height(const mh_tree* t);

/* LmcsTree::prove_batch (lmcs/lifted_tree.rs:155-180): for sorted, de-duplicated domain indices
 * (this call sorts/dedups), the opened rows of every matrix padded to `alignment`, then the
 * missing sibling digests bottom-up, left-to-right.  fields/commits must hold
 * n_idx * sum(aligned widths) and n_idx * depth * 4 felts at most. */
int mh_tree_open(mh_ctx* ctx, const mh_tree* t, const uint64_t* indices, size_t n_idx, size_t alignment,
                 uint64_t* fields, size_t* n_fields, uint64_t* commits, size_t* n_commit_felts);

/* Parity/debug: download matrix `mat` of the tree in the reference's storage order
 * (row-major, bit-reversed rows). */
int mh_tree_download_lde(mh_ctx* ctx, const mh_tree* t, int mat, uint64_t* out_rowmajor_bitrev);
/* Parity/debug: download all digest layers, leaf layer (domain order) first: (2H-1)*4 felts. */
int mh_tree_download_layers(mh_ctx* ctx, const mh_tree* t, uint64_t* out);

/* ---- coset-sharded commitment across the GPUs of a node (one process + one ctx per GPU) -------- */
/* The LDE is stored coset-major, so rank k of `world` (a power of two <= 2^log_blowup) owns cosets
 * [k*B/world, (k+1)*B/world) of every column: it runs the (replicated) inverse NTT, the forward NTT
 * of ITS cosets only and the leaf sponges of its leaves.  The Merkle tree is indexed in domain order
 * (crates/lifted-stark/src/lmcs/lifted_tree.rs:247-258), i.e. by rows first, so one all-to-all of
 * leaf digests follows (done by the host layer with RCCL / torch.distributed, see
 * miden-vm_amd/sharding.py): rank d receives rows [d*N/world, (d+1)*N/world) of every coset, laid
 * out [B][N/world] digests, builds that subtree, and the `world` subroots (all-gathered, 32 B each)
 * are combined on the host.  The result equals mh_commit_traces' root on one GPU. */
typedef struct mh_shard mh_shard;
int mh_shard_commit_leaves(mh_ctx* ctx, int n_traces, mh_trace* const* traces, int log_blowup, int rank, int world,
                           mh_shard** out);
void mh_shard_free(mh_shard* s);
/* DEVICE pointer to this rank's leaf digests, [cosets_local][N] x 4 felts (coset-major). */
uint64_t* mh_shard_leaf_digests(mh_shard* s, size_t* n_digests);
/* digests_device: DEVICE pointer to the exchanged digests [B][N/world] x 4 felts. */
int mh_shard_build_subtree(mh_ctx* ctx, mh_shard* s, const uint64_t* digests_device, uint64_t subroot[4]);
/* Host only (no GPU needed): Merkle root over `world` subroots given in rank order, Poseidon2 nodes ... */
int mh_merkle_cap_root(const uint64_t* subroots, int world, uint64_t root[4]);
/* ... and under any LMCS hasher (MH_LMCS_*): the one the context of mh_shard_commit_leaves / mh_shard_build_subtree was set to. */
int mh_merkle_cap_root_lmcs(int lmcs, const uint64_t* subroots, int world, uint64_t root[4]);

/* ---- AIRs as data: the constraint DAG blob "MHDAG001" ------------------------------------------ */
/* The Rust side captures `air.eval` once on a symbolic builder (the route of
 * crates/ace-codegen/src/pipeline.rs:71-123) and ships a flat u64 blob:
 *   [0] magic 0x4d48444147303031  [1] main_width  [2] aux_width (EF columns)  [3] num_randomness
 *   [4] num_aux_values  [5] num_public_values  [6] n_periodic  [7] log_quotient_degree
 *   [8] n_nodes  [9] n_constraints  [10] preprocessed_width  [11] reserved
 *   n_periodic x { len, values[len] }            periodic columns (power-of-two lengths)
 *   n_nodes x { op | a << 8 | b << 36, const }   ops: 0 CONST(const) 1 MAIN(a=col,b=row offset)
 *        2 AUX(a=EF col,b=row) 3 PUBLIC(a) 4 PERIODIC(a) 5 IS_FIRST 6 IS_LAST 7 IS_TRANSITION
 *        8 RANDOMNESS(a) 9 AUX_VALUE(a) 10 ADD(a,b) 11 SUB(a,b) 12 MUL(a,b) 13 NEG(a) 14 PREPROCESSED(a=col,b=row);
 *        a, b < node id for the four gates
 *   n_constraints x node id                      in emission order (constraint k folds with alpha^(K-1-k))
 * miden-vm_amd/dag.py is the reference exporter used by tests and bench. */
int mh_air_load(mh_ctx* ctx, const uint64_t* blob, size_t n_words, mh_air** out);
void mh_air_free(mh_air* air);
int mh_air_log_quotient_degree(const mh_air* air);
/* Number of specialised kernels the DAG was compiled into (hiprtc at load time, code objects cached on disk
 * under $MH_JIT_CACHE_DIR or ~/.cache/midenhip); 0 = small DAG, evaluated by the generic interpreter kernel.
 * MH_JIT=0 / MH_JIT=1 in the environment force either path (both are bit-identical). */
int mh_air_compiled_chunks(const mh_air* air);
/* Offline precompilation, host only (no GPU, no context): compiles the chunk kernels of a constraint-DAG blob or of a lookup program
 * ("MHLKP001") for gfx950 into the cache directory ($MH_JIT_CACHE_DIR, else ~/.cache/midenhip), where a later mh_air_load /
 * mh_lookup_load finds them -- a prover service ships the directory and never runs hiprtc on the request path (measured for the
 * chiplets AIR: 5.7 s cold, 3 ms from the cache).  *n_chunks = kernels of the program (0: small DAG, interpreted). */
int mh_jit_precompile(const uint64_t* blob, size_t n_words, int* n_chunks);
/* Largest VGPR count over those kernels (their occupancy is 512 / VGPRs waves per SIMD); 0 when nothing was compiled. */
int mh_air_compiled_max_vgprs(const mh_air* air);
/* Preprocessed columns (fixed circuit data committed once at setup: crates/lifted-stark/src/preprocessed.rs; blob word
 * [10] = preprocessed width, DAG op 14 PREPROCESSED(a = col, b = row offset)).  Setup = mh_commit_traces of the
 * preprocessed matrices of the AIRs that declare some, in PROOF order (ascending height, ties by instance index), with
 * the proving blowup; its root is observed by the caller right after the protocol parameters and before the statement
 * (prover/mod.rs:282-286), i.e. it belongs in `pre_observe`.  Each such AIR is then pointed at its matrix of that tree;
 * proofs open the tree first (`[preprocessed?, main, aux, quotient]`).  The tree must outlive the proofs. */
int mh_air_attach_preprocessed(mh_air* air, const mh_tree* tree, int matrix_index, const mh_trace* raw);
/* tree = NULL detaches.  `raw` = the uploaded preprocessed matrix itself (trace domain); only needed when a lookup
 * program attached to the same AIR reads preprocessed columns (table lookups), NULL otherwise. */

/* ---- LogUp aux trace on the device -------------------------------------------------------------------------
 * Replaces `build_logup_aux_trace` (air/src/lookup/aux_builder.rs:49-96) for an AIR whose bus messages are
 * exported as a lookup program: the "MHLKP001" blob = the constraint-DAG blob's header (w[2] = number of aux EF
 * columns, w[4] = w[5] = w[7] = 0, w[9] ignored, w[10] = preprocessed width), periodic tables and node list (ops CONST,
 * MAIN, PREPROCESSED, PERIODIC, RANDOMNESS, ADD, SUB, MUL, NEG), followed per aux column by: count, then `count` pairs (multiplicity node id,
 * denominator node id).  Semantics (aux_builder.rs:202-258): f_c(r) = sum_j m_j(r) / d_j(r) over the fractions
 * of column c with m_j(r) != 0;  aux[r][c >= 1] = f_c(r);  aux[r][0] = sum_{r' < r} sum_c f_c(r');  the one
 * aux value = the sum over all rows (`committed_finals`).  A zero denominator is MH_ERR_INVALID.
 * Optional tail, REGISTER columns behind the w[2] LogUp columns (precompiles-prover/src/tests/aux_register.rs; the extension-field
 * accumulators of precompiles-prover/src/uint/store_mul/mod.rs:118-121, which that AIR's `build_aux_trace` computes on the CPU):
 * count, then per register: keep node id (0xFFFFFFFF = the constant 1), build node id, n_terms <= 8, n_terms pairs (another register,
 * coefficient node id; no cycles).  Semantics: r_k[0] = 0,  r_k[i + 1] = keep(i) r_k[i] + sum_j coeff_j(i) r_j[i] + build(i);  aux
 * column w[2] + k = r_k.  Registers stay out of the running sum and of the aux value; the AIR's own constraints tie them down.
 * `mh_air_attach_lookup` makes mh_prove / mh_session_commit_aux build that instance's aux trace on the device
 * (its `mh_aux_builder` callback is not called); the lookup must outlive the AIR's proofs. */
typedef struct mh_lookup mh_lookup;
int mh_lookup_load(mh_ctx* ctx, const uint64_t* blob, size_t n_words, mh_lookup** out);
void mh_lookup_free(mh_lookup* l);
int mh_air_attach_lookup(mh_air* air, const mh_lookup* l); /* l = NULL detaches */
/* Stand-alone: aux trace (device resident, 2 * (LogUp columns + registers) base columns) + accumulator final of `main_trace`. */
int mh_lookup_build_aux(mh_ctx* ctx, const mh_lookup* l, const mh_trace* main_trace, const mh_trace* preprocessed /* or NULL */,
                        const uint64_t* randomness, size_t n_randomness, mh_trace** aux_out, uint64_t acc_final[2]);
/* Copy a device trace back as a row-major [2^log_n][width] matrix (tests). */
int mh_trace_download(mh_ctx* ctx, const mh_trace* t, uint64_t* rowmajor_out);

/* ---- the whole proof: miden_prover::prove_stark (prover/src/lib.rs:317-355) -> ------------------ */
/* ProverInstance::prove (crates/lifted-stark/src/prover/mod.rs:230-578).                           */
typedef struct mh_pcs_params { /* PcsParams::new argument order regrouped (pcs/params.rs:52-96) */
  int log_blowup, log_folding_arity, log_final_degree, folding_pow_bits, deep_pow_bits, num_queries, query_pow_bits;
} mh_pcs_params;

/* LiftedAir::build_aux_trace (crates/lifted-air/src/air.rs): called once per AIR instance, in
 * instance order, after the main commitment.  randomness = 2*max_num_randomness felts; the callee
 * fills the flattened EF aux trace (rows x 2*aux_width, row-major) and 2*num_aux_values felts.
 * A non-zero return aborts the proof (this is also where Statement::eval_external failures go).
 * NULL = every aux trace and aux value is zero (DummyMidenAir), generated on the device. */
typedef int (*mh_aux_builder)(void* user, int instance_idx, const uint64_t* randomness, uint64_t* aux_out,
                              uint64_t* aux_values_out);

/* airs / traces in INSTANCE order.  challenger_state = the 12-felt sponge state of the prototype
 * challenger (air/src/config.rs:255-273: RELATION_DIGEST in state[8..12]); pre_observe = every felt
 * observed before observe_shape (protocol parameters, air/src/config.rs:188-198, then
 * Statement::observe).  The transcript, PoW witnesses (smallest valid witness) and query openings
 * are produced inside; the result is the reference's StarkProofData. */
int mh_prove(mh_ctx* ctx, const mh_pcs_params* params, int n_airs, mh_air* const* airs, mh_trace* const* traces,
             const uint64_t* public_values, size_t n_public_values, const uint64_t challenger_state[12],
             const uint64_t* pre_observe, size_t n_pre_observe, mh_aux_builder aux_builder, void* user, mh_proof** out);
/* The shape of prove_stark itself (prover/src/lib.rs:317-355): HOST row-major matrices in (instance order; page-locked memory from
 * mh_host_alloc for full overlap), proof out.  The library starts every upload at once in proof order and proves: matrix k + 1 is on
 * the PCIe link while matrix k is extended and hashed.  Equivalent to mh_trace_upload_async x n, mh_prove, mh_trace_free x n. */
int mh_prove_host(mh_ctx* ctx, const mh_pcs_params* params, int n_airs, mh_air* const* airs, const uint64_t* const* traces_rowmajor,
                  const int* log_heights, const uint64_t* public_values, size_t n_public_values, const uint64_t challenger_state[12],
                  const uint64_t* pre_observe, size_t n_pre_observe, mh_aux_builder aux_builder, void* user, mh_proof** out);
/* One proof sharded over `world` GPUs (one process + one ctx per GPU, all ranks call this with the same
 * arguments and traces; every rank returns the same proof).  The coset-major layout makes every pass over
 * LDE-sized data local to a rank's cosets; what crosses ranks goes through these three collectives on
 * DEVICE buffers (implemented by the host layer over RCCL, see miden-vm_amd/sharding.py):
 *   all_to_all  (leaf digests of every commitment, 32 B per leaf),
 *   all_gather  (subtree roots; quotient-chunk coefficients; a FRI layer once it has fewer rows per
 *                coset than ranks),
 *   all_reduce_sum_u64 (query openings: every value is contributed by exactly one rank).
 * Each callback returns 0 on success; unless `stream_ordered` is set the library synchronises its stream before calling
 * and expects the collective to be complete on return.  world must be a power of two <= 2^log_blowup and <= the shortest trace.
 * Any mix of per-AIR quotient degrees is accepted (an AIR's native quotient chunks are gathered with all_gather and upsampled on
 * every rank), and world may exceed the number of quotient chunks D: chunk t then lives on rank t * world / D, the other ranks
 * idle through constraint evaluation (if D < B only world * D / B ranks hold quotient-coset rows). */
typedef struct mh_comm {
  int rank, world;
  void* user;
  int (*all_to_all)(void* user, const void* send_dev, void* recv_dev, size_t bytes_per_peer);
  int (*all_gather)(void* user, const void* send_dev, void* recv_dev, size_t bytes_per_rank);
  int (*all_reduce_sum_u64)(void* user, uint64_t* buf_dev, size_t n);
  /* 0: host-synchronous callbacks (the contract above: stream synchronised before the call, complete on return);
   * 1: the callbacks ENQUEUE on the ctx's stream and return at once (mh_comm_create_rccl): no host synchronisation. */
  int stream_ordered;
} mh_comm;
/* The trace of a sharded proof: every rank uploads only its 1/world of the ROWS (a contiguous slice of the row-major host matrix)
 * over its own PCIe link, the slices are all-gathered over the communicator and each rank transposes the whole.  Collective: every
 * rank calls it with the same shape; `rowmajor` must be readable by every rank (threads of one process, or a shared mapping), a
 * rank reads only its slice.  The result is an ordinary mh_trace (the full matrix on every rank: the inverse NTT needs all rows). */
int mh_trace_upload_sharded(mh_ctx* ctx, const mh_comm* comm, const uint64_t* rowmajor, int log_n, size_t width, mh_trace** out);
/* ---- the communicator inside the library: RCCL over xGMI ------------------------------------------------------
 * One process + one ctx per GPU.  Rank 0 calls mh_rccl_unique_id and hands the 128 bytes to every rank (by whatever
 * started the ranks); every rank calls mh_comm_create_rccl with its ctx (ncclCommInitRank: collective, blocks until all
 * `world` ranks arrive).  The returned mh_comm runs the three collectives as RCCL calls on the ctx's own stream and on the
 * library's own device buffers (leaf-digest all-to-all = grouped ncclSend/ncclRecv, ncclAllGather, ncclAllReduce(sum, u64)).
 * RCCL is opened lazily from $MH_RCCL_LIB, $ROCM_PATH/lib/librccl.so.1 or /opt/rocm/lib/librccl.so.1.
 * mh_comm_selftest moves known patterns through all three collectives of ANY mh_comm and checks them (every rank calls it). */
#define MH_RCCL_ID_BYTES 128
int mh_rccl_unique_id(uint8_t id[MH_RCCL_ID_BYTES]);
int mh_comm_create_rccl(mh_ctx* ctx, const uint8_t id[MH_RCCL_ID_BYTES], int rank, int world, mh_comm** out);
void mh_comm_destroy(mh_comm* comm); /* communicators made by mh_comm_create_rccl or mh_comm_create_local */
/* The same collectives between the contexts of ONE process (one thread + one ctx per rank; ranks on different GPUs use peer
 * copies over xGMI, ranks sharing a GPU plain device copies): create one fabric, then every rank's thread calls
 * mh_comm_create_local (collective: returns when all `world` ranks have joined).  Stream ordered like the RCCL communicator.
 * No RCCL, no launcher: the shape for a host that drives all GPUs of a node from one process. */
typedef struct mh_local_fabric mh_local_fabric;
mh_local_fabric* mh_local_fabric_create(int world);
void mh_local_fabric_destroy(mh_local_fabric* f); /* after every rank's communicator has been destroyed */
int mh_comm_create_local(mh_ctx* ctx, mh_local_fabric* f, int rank, mh_comm** out);
/* Marks the fabric dead and wakes every rank waiting in a collective (they return an error).  Call it from a rank's error path
 * -- a failed session step, a rank that will never reach the next collective -- so that its peers do not block for ever.  A failed
 * mh_comm_create_local and a failed collective do this themselves.  Sticky: create a new fabric to continue. */
void mh_local_fabric_abort(mh_local_fabric* f);
int mh_comm_selftest(mh_ctx* ctx, const mh_comm* comm);
/* mh_commit_traces for one rank of a sharded prover (every rank calls it with the same traces): the setup commitment
 * of preprocessed matrices for mh_prove_sharded / sharded sessions.  Same root as mh_commit_traces. */
int mh_commit_traces_sharded(mh_ctx* ctx, const mh_comm* comm, int n_traces, mh_trace* const* traces, int log_blowup,
                             mh_tree** out, uint64_t root[4]);
int mh_prove_sharded(mh_ctx* ctx, const mh_comm* comm, const mh_pcs_params* params, int n_airs, mh_air* const* airs,
                     mh_trace* const* traces, const uint64_t* public_values, size_t n_public_values,
                     const uint64_t challenger_state[12], const uint64_t* pre_observe, size_t n_pre_observe,
                     mh_aux_builder aux_builder, void* user, mh_proof** out);
/* ---- staged proof session: the caller owns the Fiat-Shamir transcript ------------------------------
 * The same device stages `mh_prove` runs, one entry point per step of ProverInstance::prove
 * (crates/lifted-stark/src/prover/mod.rs:230-578), for a host that keeps p3's DuplexChallenger /
 * ProverTranscript itself: every call takes the challenges the transcript sampled and returns what the
 * transcript must observe next.  Calls out of protocol order return MH_ERR_INVALID.  EF values are
 * (c0, c1) pairs of canonical felts.  `comm` NULL (or world 1) = single GPU; otherwise every rank makes the
 * same calls with the same challenges (as in mh_prove_sharded).  The session BORROWS the airs and traces:
 * they must outlive it.
 *
 *   begin                    channel.observe(n_airs, log heights...)               order.rs:154-163
 *   commit_main   -> root    channel.send_commitment(root)                         mod.rs:300-318
 *   commit_aux(randomness[num_randomness]) -> root, aux values (proof order)       mod.rs:330-412
 *   commit_quotient(alpha, beta) -> root                                           mod.rs:420-560
 *   ood_point_ok(z) / ood(z) -> evals[2][ood_width] (row z, then row z*w_H)        pcs/prover.rs:70-136
 *   [grind deep_pow_bits]  deep(alpha_deep, beta_deep)                             deep/prover.rs:147-193
 *   num_fri_rounds x { fri_commit -> root, [grind folding_pow_bits], fri_fold(beta) }   fri/prover.rs:113-205
 *   fri_final -> final_poly_len coefficients, descending degree                    fri/prover.rs:212-239
 *   [grind query_pow_bits]  open(indices) -> hinted felts + digests, transcript order   pcs/prover.rs:138-195 */
typedef struct mh_session mh_session;
typedef struct mh_session_shape_t {
  int log_lde_height;    /* query indices are sampled with this many bits */
  size_t num_randomness; /* EF challenges to sample after the main commitment (max over the AIRs) */
  size_t num_aux_values; /* EF aux values sent after the aux commitment (all instances, proof order) */
  size_t ood_width;      /* EF evaluations per OOD point: every committed matrix, each padded to 8 columns */
  int num_fri_rounds;
  size_t final_poly_len; /* EF coefficients of the final polynomial */
} mh_session_shape_t;
int mh_session_begin(mh_ctx* ctx, const mh_comm* comm, const mh_pcs_params* params, int n_airs, mh_air* const* airs,
                     mh_trace* const* traces, const uint64_t* public_values, size_t n_public_values, mh_session** out);
void mh_session_free(mh_session* s);
int mh_session_shape(const mh_session* s, mh_session_shape_t* out);
int mh_session_commit_main(mh_session* s, uint64_t root[4]);
int mh_session_commit_aux(mh_session* s, const uint64_t* randomness, mh_aux_builder aux_builder, void* user,
                          uint64_t root[4], uint64_t* aux_values_out);
int mh_session_commit_quotient(mh_session* s, const uint64_t alpha[2], const uint64_t beta[2], uint64_t root[4]);
int mh_session_ood_point_ok(const mh_session* s, const uint64_t z[2]); /* 1 = acceptable, 0 = resample */
int mh_session_ood(mh_session* s, const uint64_t z[2], uint64_t* evals_out);
int mh_session_deep(mh_session* s, const uint64_t alpha[2], const uint64_t beta[2]);
int mh_session_fri_commit(mh_session* s, uint64_t root[4]);
int mh_session_fri_fold(mh_session* s, const uint64_t beta[2]);
int mh_session_fri_final(mh_session* s, uint64_t* coeffs_out);
/* The result holds only the hints: mh_proof_fields / mh_proof_commitments = what ProverTranscript::hint_* receives. */
int mh_session_open(mh_session* s, const uint64_t* indices, size_t n_indices, mh_proof** out);
/* GrindingChallenger::grind on the device: `state` = the sponge state, `pending` = felts observed since the
 * last permutation (< 8).  Returns the smallest witness; the caller replays check_witness on its challenger. */
int mh_grind(mh_ctx* ctx, const uint64_t state[12], const uint64_t* pending, size_t n_pending, int bits, uint64_t* witness);
/* The same search for the byte challengers of the Blake3 / Keccak configurations (context set accordingly): `input` = the hash
 * challenger's input buffer as it stands before the witness is observed (the chaining value of the last flush followed by every
 * byte observed since); returns the smallest felt w such that, after observing w as 8 little-endian bytes, the u64 built from the
 * LAST eight digest bytes (last byte lowest: HashChallenger pops from the end) has `bits` low zero bits.  The caller replays
 * check_witness on its own challenger. */
int mh_grind_bytes(mh_ctx* ctx, const uint8_t* input, size_t n_input, int bits, uint64_t* witness);
/* ---- verifier (host only: no GPU, no ctx) ---------------------------------------------------------------------
 * Replays a proof against the AIRs' constraint-DAG blobs: crates/lifted-stark/src/verifier/mod.rs (flow, constraint
 * identity), pcs/verifier.rs + lmcs/config.rs:172-211 (openings), pcs/deep/verifier.rs, pcs/fri/verifier.rs.
 * Inputs mirror mh_prove: AIR blobs and log heights in INSTANCE order, the same challenger state / pre-observed
 * felts, and the proof's two streams (mh_proof_fields / mh_proof_commitments).  MH_OK + the transcript digest, or
 * MH_ERR_INVALID with the reason in `err`. */
int mh_verify(const mh_pcs_params* params, int n_airs, const uint64_t* const* air_blobs, const size_t* air_blob_words,
              const uint8_t* log_trace_heights, const uint64_t* public_values, size_t n_public_values,
              const uint64_t challenger_state[12], const uint64_t* pre_observe, size_t n_pre_observe,
              const uint64_t* fields, size_t n_fields, const uint64_t* commitments, size_t n_commitments,
              const uint64_t* preprocessed_root /* [4]: the setup commitment, NULL if no AIR has preprocessed columns */,
              uint64_t digest[4], char* err, size_t err_cap);
/* Statement::eval_external (crates/lifted-air/src/statement.rs:94-108; MultiAir::eval_external, crates/lifted-air/src/air.rs:247-287):
 * the statement's cross-AIR assertions, checked by the reference verifier as its step 11 (crates/lifted-stark/src/verifier/mod.rs:
 * 488-501, ExternalAssertionFailed) and by its prover before the aux commitment is used (prover/mod.rs:383-399).  The AIRs of this
 * library are data (constraint DAGs); a statement-level hook is code, so it is a callback: it receives the shared challenges
 * (n_randomness EF values), every instance's committed aux values and the log heights, all in INSTANCE order, writes one EF value
 * (c0, c1) per assertion into assertions_out (room for `cap`) and returns their number, or a negative value for the reference's
 * ReductionError.  mh_verify_ex rejects the proof unless every value is zero.  mh_verify == mh_verify_ex(external = NULL) verifies a
 * statement WITHOUT cross-AIR assertions (the default MultiAir): a caller whose statement has some -- every LogUp / bus statement,
 * e.g. Miden's "sum of the committed finals plus boundary corrections = 0" (air/src/lib.rs:854-1000) -- MUST use mh_verify_ex;
 * with mh_verify the bus balance is unchecked.  On the proving side the same check belongs in the `mh_aux_builder` callback /
 * after mh_session_commit_aux, which hand the aux values to the caller for exactly this purpose. */
typedef int (*mh_external_assertions)(void* user, const uint64_t* randomness, size_t n_randomness, const uint64_t* const* aux_values,
                                      const size_t* n_aux_values, const uint8_t* log_trace_heights, int n_airs, uint64_t* assertions_out,
                                      size_t cap);
int mh_verify_ex(const mh_pcs_params* params, int n_airs, const uint64_t* const* air_blobs, const size_t* air_blob_words,
                 const uint8_t* log_trace_heights, const uint64_t* public_values, size_t n_public_values,
                 const uint64_t challenger_state[12], const uint64_t* pre_observe, size_t n_pre_observe, const uint64_t* fields,
                 size_t n_fields, const uint64_t* commitments, size_t n_commitments, const uint64_t* preprocessed_root,
                 mh_external_assertions external, void* external_user, uint64_t digest[4], char* err, size_t err_cap);
/* mh_verify_ex under any of the five configurations: lmcs = MH_LMCS_* (prover/src/lib.rs:246-300 dispatches the same five). */
int mh_verify_lmcs(int lmcs, const mh_pcs_params* params, int n_airs, const uint64_t* const* air_blobs, const size_t* air_blob_words,
                   const uint8_t* log_trace_heights, const uint64_t* public_values, size_t n_public_values,
                   const uint64_t challenger_state[12], const uint64_t* pre_observe, size_t n_pre_observe, const uint64_t* fields,
                   size_t n_fields, const uint64_t* commitments, size_t n_commitments, const uint64_t* preprocessed_root,
                   mh_external_assertions external, void* external_user, uint64_t digest[4], char* err, size_t err_cap);
/* A ready-made mh_external_assertions: one assertion, the sum over the instances of their aux value 0 (the LogUp accumulator
 * final, `committed_finals` of air/src/lookup/aux_builder.rs) -- the balance of a bus statement with no boundary corrections. */
int mh_external_logup_balance(void* user, const uint64_t* randomness, size_t n_randomness, const uint64_t* const* aux_values,
                              const size_t* n_aux_values, const uint8_t* log_trace_heights, int n_airs, uint64_t* assertions_out,
                              size_t cap);
/* A ready-made mh_external_assertions for the second client: `ChipletMultiAir::eval_external` of the precompile prover's session
 * (precompiles-prover/src/session/prove.rs:243-256) = the sum of the committed sigmas + `fixed_boundary_correction` (:205-216), the
 * verifier's consumes of the session's fixed environment (session/fixed.rs: the VM-owned curve group's `EcGroup` tuple, the five fixed
 * uints' `UintVal` tuples).  Every AIR must expose exactly one aux value (its sigma): any other shape returns -1.  `user` is unused.
 * `mh_external_precompile_session_ec_only` is the reduced form for statements that leave the uint store out (the EcGroup part of the
 * correction only) -- a callback of its own name so that the real statement cannot be weakened by a flag passed by mistake. */
int mh_external_precompile_session(void* user, const uint64_t* randomness, size_t n_randomness, const uint64_t* const* aux_values,
                                   const size_t* n_aux_values, const uint8_t* log_trace_heights, int n_airs, uint64_t* assertions_out,
                                   size_t cap);
int mh_external_precompile_session_ec_only(void* user, const uint64_t* randomness, size_t n_randomness, const uint64_t* const* aux_values,
                                           const size_t* n_aux_values, const uint8_t* log_trace_heights, int n_airs,
                                           uint64_t* assertions_out, size_t cap);

/* ---- the precompile prover's session: SessionTraces::prove_stark's own shape (precompiles-prover/src/session/prove.rs:295-330, 385-416)
 * Twelve main traces + the transcript root in, proof out -- the statement layer `ChipletMultiAir` adds to the proof system, in the
 * library (csrc/precompile.cpp) so that a C caller restates nothing:
 *   - the twelve AIRs of `ChipletAir::all()` (session/prove.rs:111-126: ChunkNode, Poseidon2, KeccakRound, BytePairLut, KeccakSponge,
 *     TranscriptEval, UintStoreMul, UintAdd, EcGroups, EcPointStore, EcGroupAdd, EcMsm) with their lookup programs, embedded as the
 *     blobs of miden-vm_amd/blobs/precompile; every aux column (LogUp sums and UintStoreMul's three registers) is built on the device;
 *   - BytePairLutAir's preprocessed 2^16 x 4 table (primitives/byte_pair_lut.rs:262-277), generated by mh_precompile_load and
 *     committed once per hash function, on first use (the reference's session/preprocessed_cache.rs);
 *   - `precompile_pcs_params()` (stark_config.rs:60-71), the placeholder relation digest 0^4 (session/prove.rs:40), the transcript
 *     prefix `observe_protocol_params` | preprocessed commitment | len(air_inputs) = 4, the root, 0, 0 (the default `MultiAir::observe`,
 *     crates/lifted-air/src/air.rs:307-324);
 *   - `ChipletMultiAir::eval_external` = mh_external_precompile_session.
 * mains_rowmajor[i]: AIR i's main trace, row-major, height 2^log_heights[i], width as the AIR declares (42, 32, 68, 3, 67, 39, 44,
 * 30, 6, 14, 21, 38), canonical felts; index 3 (BytePairLut) has 2^16 rows.  public_root: the transcript root (`air_inputs`).
 * hash_fn = MH_LMCS_*. */
#define MH_PRECOMPILE_NUM_AIRS 12
#define MH_PRECOMPILE_PRE_OBSERVE_FELTS 19 /* 8 protocol parameters + 4 (preprocessed commitment) + 1 + 4 + 2 (statement framing) */
typedef struct mh_precompile mh_precompile; /* the twelve AIRs loaded on a context, lookups attached, the byte-pair table uploaded */
void mh_precompile_pcs_params(mh_pcs_params* out);
int mh_precompile_load(mh_ctx* ctx, mh_precompile** out);
void mh_precompile_free(mh_precompile* s);
/* the embedded blobs, for a caller that drives the generic entry points itself: lookup = 0 the constraint DAG, 1 the lookup program */
int mh_precompile_air_blob(int which, int lookup, const uint64_t** words_out, size_t* n_words);
/* the setup commitment of the byte-pair table under hash_fn (made on first use, then cached in `s`): what a verifier must be given */
int mh_precompile_preprocessed_root(mh_precompile* s, int hash_fn, uint64_t root[4]);
int mh_precompile_pre_observe(const mh_pcs_params* p, const uint64_t preprocessed_root[4], const uint64_t public_root[4],
                              uint64_t out[MH_PRECOMPILE_PRE_OBSERVE_FELTS]);
int mh_prove_precompile(mh_ctx* ctx, mh_precompile* s, int hash_fn, const uint64_t* const mains_rowmajor[MH_PRECOMPILE_NUM_AIRS],
                        const int log_heights[MH_PRECOMPILE_NUM_AIRS], const uint64_t public_root[4], mh_proof** out);
/* the same over device-resident traces, in `ChipletAir::all()` order */
int mh_prove_precompile_traces(mh_ctx* ctx, mh_precompile* s, int hash_fn, mh_trace* const traces[MH_PRECOMPILE_NUM_AIRS],
                               const uint64_t public_root[4], mh_proof** out);
/* `verify_stark` (session/prove.rs:365-383, 386-425): StarkProofData bytes -> MH_OK + the transcript digest, or MH_ERR_INVALID + reason.
 * Host only (no context, no GPU); preprocessed_root = mh_precompile_preprocessed_root of a prover-side context (the reference's
 * verifier recomputes it from the table; this library has no CPU path for an LDE + Merkle commitment, by design). */
int mh_verify_precompile(int hash_fn, const uint64_t preprocessed_root[4], const uint64_t public_root[4], const uint8_t* proof_bytes,
                         size_t n_bytes, uint64_t digest[4], char* err, size_t err_cap);

/* ---- the Miden VM statement: prove_stark's own shape (prover/src/lib.rs:317-355) --------------------------------------------------
 * Three matrices + 32 public values + aux inputs in, proof out -- the statement layer the reference's `MidenMultiAir` adds to the
 * proof system, in the library (csrc/miden.cpp) so that a C caller restates nothing:
 *   - the three AIRs [CoreAir, ChipletsAir, Poseidon2PermutationAir] (air/src/lib.rs:560-650) with their LogUp lookup programs, embedded
 *     as the blobs of miden-vm_amd/blobs (all eight aux columns are built on the device);
 *   - `MidenMultiAir::observe` (air/src/lib.rs:805-849): [kernel_H | program_hash] [deferred_root | 0^4] [stack inputs 16] [stack
 *     outputs 16], kernel_H = hash_kernel_digests (:946-961) = Poseidon2 hash_elements of the kernel-digest felts;
 *   - `MidenMultiAir::eval_external` (:854-933): the three committed LogUp finals + the boundary corrections (block-hash seed,
 *     deferred-root log, one KernelRomInit per kernel digest) must vanish -- mh_verify_miden checks it, no proof of this statement
 *     verifies without it;
 *   - pcs_params() and RELATION_DIGEST of air/src/config.rs:54-67, 93-98; hash_fn = MH_LMCS_* as `prove_miden_vm_execution_trace`
 *     dispatches on ProvingOptions::hash_fn (prover/src/lib.rs:246-300).
 * public_values: stack inputs (16) ++ stack outputs (16).  aux_inputs: program hash (4) | deferred root (4) | kernel procedure
 * digests (4 each, at most 255).  Matrices row-major, heights 2^log_*, widths 51 / 22 / 16, canonical felts. */
#define MH_MIDEN_NUM_PUBLIC_VALUES 32
#define MH_MIDEN_PRE_OBSERVE_FELTS 56 /* 8 protocol parameters + the 48-felt statement schedule */
typedef struct mh_miden mh_miden; /* the three AIRs loaded on a context, lookups attached */
int mh_miden_load(mh_ctx* ctx, mh_miden** out);
void mh_miden_free(mh_miden* m);
int mh_prove_miden(mh_ctx* ctx, const mh_miden* m, int hash_fn, const uint64_t* core_rowmajor, int log_core,
                   const uint64_t* chiplets_rowmajor, int log_chiplets, const uint64_t* poseidon2_rowmajor, int log_poseidon2,
                   const uint64_t* public_values /* [32] */, const uint64_t* aux_inputs, size_t n_aux_inputs, mh_proof** out);
/* the same over device-resident traces (mh_trace_upload* / mh_trace_from_device), instance order core, chiplets, poseidon2 */
int mh_prove_miden_traces(mh_ctx* ctx, const mh_miden* m, int hash_fn, mh_trace* const traces[3], const uint64_t* public_values,
                          const uint64_t* aux_inputs, size_t n_aux_inputs, mh_proof** out);
/* verifier/src/lib.rs:320-330 for this statement: StarkProofData bytes -> MH_OK + the transcript digest, or MH_ERR_INVALID + reason.
 * Host only (no context, no GPU). */
int mh_verify_miden(int hash_fn, const uint64_t* public_values /* [32] */, const uint64_t* aux_inputs, size_t n_aux_inputs,
                    const uint8_t* proof_bytes, size_t n_bytes, uint64_t digest[4], char* err, size_t err_cap);
/* the pieces, for a caller that drives mh_prove / mh_session_* / mh_verify_ex itself */
void mh_miden_pcs_params(mh_pcs_params* out);
void mh_miden_challenger_state(uint64_t state[12]);
int mh_miden_hash_kernel_digests(const uint64_t* kernel_felts, size_t n_felts, uint64_t out[4]);
int mh_miden_pre_observe(const mh_pcs_params* params, const uint64_t* public_values, const uint64_t* aux_inputs, size_t n_aux_inputs,
                         uint64_t out[MH_MIDEN_PRE_OBSERVE_FELTS]);
/* randomness = (alpha, beta) as 4 felts; aux_values[i] = the i-th AIR's committed values (2 felts each); out = the one assertion */
int mh_miden_eval_external(const uint64_t randomness[4], const uint64_t* aux_inputs, size_t n_aux_inputs,
                           const uint64_t* const* aux_values, const size_t* n_aux_values, int n_airs, uint64_t out[2]);
int mh_miden_air_blob(int which /* 0 core, 1 chiplets, 2 poseidon2 */, const uint64_t** words_out, size_t* n_words);

void mh_proof_free(mh_proof* p);
size_t mh_proof_num_fields(const mh_proof* p);
size_t mh_proof_num_commitments(const mh_proof* p);
const uint64_t* mh_proof_fields(const mh_proof* p);       /* TranscriptData::fields */
const uint64_t* mh_proof_commitments(const mh_proof* p);  /* TranscriptData::commitments, 4 felts each.  Either pointer may be NULL when its
                                                             count is 0 (mh_session_open of openings that need no sibling) */
const uint64_t* mh_proof_digest(const mh_proof* p);       /* StarkOutput::digest, 4 felts */
size_t mh_proof_num_traces(const mh_proof* p);
const uint8_t* mh_proof_log_trace_heights(const mh_proof* p);
/* Serialise StarkProofData; returns the byte length needed (writes only if cap is large enough). */
size_t mh_proof_serialize(const mh_proof* p, uint8_t* out, size_t cap);
/* The inverse: StarkProofData bytes -> proof (what verifier/src/lib.rs:320-330 does first; 64 MiB limit).  MH_ERR_INVALID on
 * truncated input, trailing bytes, oversized length prefixes or non-canonical field elements.  The digest is not part of
 * StarkProofData and comes back zeroed (mh_verify recomputes it). */
int mh_proof_deserialize(const uint8_t* bytes, size_t len, mh_proof** out);

#ifdef __cplusplus
}
#endif
#endif /* MIDENHIP_H */
