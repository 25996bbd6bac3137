//! Raw FFI declarations for libmidenhip (include/midenhip.h) — the `-sys` layer of the shim described in
//! INTEGRATION.md.  One declaration per exported symbol; no logic.  NOT COMPILED IN THIS REPOSITORY: the build image
//! has no Rust toolchain, so this file is written against the header by hand (tests/test_abi.py checks that the symbol
//! list here, the header and the shared library agree).  Link with `-lmidenhip`.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_double, c_int, c_long, c_void};

macro_rules! opaque { ($($n:ident),*) => { $( #[repr(C)] pub struct $n { _p: [u8; 0] } )* } }
opaque!(mh_ctx, mh_trace, mh_tree, mh_air, mh_proof, mh_shard, mh_lookup, mh_session, mh_miden, mh_precompile);

pub const MH_OK: c_int = 0;
pub const MH_ERR_INVALID: c_int = 1;
pub const MH_ERR_HIP: c_int = 2;
pub const MH_ERR_OOM: c_int = 3;
pub const MH_ERR_INTERNAL: c_int = 4;
pub const MH_ERR_COMM: c_int = 5;

/// PcsParams (crates/lifted-stark/src/pcs/params.rs:52-96).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct mh_pcs_params {
    pub log_blowup: c_int,
    pub log_folding_arity: c_int,
    pub log_final_degree: c_int,
    pub folding_pow_bits: c_int,
    pub deep_pow_bits: c_int,
    pub num_queries: c_int,
    pub query_pow_bits: c_int,
}

/// LiftedAir::build_aux_trace as a callback (prover/mod.rs:355-381); non-zero aborts the proof.
#[repr(C)]
pub struct mh_local_fabric {
    _private: [u8; 0],
}
pub type mh_aux_builder = Option<
    unsafe extern "C" fn(user: *mut c_void, instance_idx: c_int, randomness: *const u64, aux_out: *mut u64, aux_values_out: *mut u64) -> c_int,
>;
/// Statement::eval_external as a callback (see midenhip.h): writes one EF value per assertion, returns their number or < 0.
pub type mh_external_assertions = Option<
    unsafe extern "C" fn(user: *mut c_void, randomness: *const u64, n_randomness: usize, aux_values: *const *const u64, n_aux_values: *const usize, log_trace_heights: *const u8, n_airs: c_int, assertions_out: *mut u64, cap: usize) -> c_int,
>;

/// Collectives of a sharded proof, on DEVICE buffers (RCCL): 0 = success.
#[repr(C)]
pub struct mh_comm {
    pub rank: c_int,
    pub world: c_int,
    pub user: *mut c_void,
    pub all_to_all: Option<unsafe extern "C" fn(user: *mut c_void, send_dev: *const c_void, recv_dev: *mut c_void, bytes_per_peer: usize) -> c_int>,
    pub all_gather: Option<unsafe extern "C" fn(user: *mut c_void, send_dev: *const c_void, recv_dev: *mut c_void, bytes_per_rank: usize) -> c_int>,
    pub all_reduce_sum_u64: Option<unsafe extern "C" fn(user: *mut c_void, buf_dev: *mut u64, n: usize) -> c_int>,
    /// 0: host-synchronous callbacks; 1: enqueue on the ctx's stream (the library's own RCCL communicator)
    pub stream_ordered: c_int,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct mh_session_shape_t {
    pub log_lde_height: c_int,
    pub num_randomness: usize,
    pub num_aux_values: usize,
    pub ood_width: usize,
    pub num_fri_rounds: c_int,
    pub final_poly_len: usize,
}

/// mh_ctx_set_lmcs / mh_verify_lmcs: the StarkConfig (air/src/config.rs:212-353) by its hash function
pub const MH_LMCS_POSEIDON2: c_int = 0;
pub const MH_LMCS_BLAKE3: c_int = 1;
pub const MH_LMCS_KECCAK: c_int = 2;
pub const MH_LMCS_RPO: c_int = 3;
pub const MH_LMCS_RPX: c_int = 4;

#[link(name = "midenhip")]
unsafe extern "C" {
    // ---- context ----
    pub fn mh_ctx_create(device_id: c_int, out: *mut *mut mh_ctx) -> c_int;
    pub fn mh_ctx_destroy(ctx: *mut mh_ctx);
    pub fn mh_ctx_trim(ctx: *mut mh_ctx) -> c_int;
    /// out = [pool bytes, table bytes, device free, device total]
    pub fn mh_ctx_mem_stats(ctx: *mut mh_ctx, out: *mut u64) -> c_int;
    pub fn mh_last_error(ctx: *const mh_ctx) -> *const c_char;
    pub fn mh_device_count() -> c_int;
    /// MH_LMCS_POSEIDON2 = 0, MH_LMCS_BLAKE3 = 1 (the hasher of mh_commit_traces / mh_tree_open on this context)
    pub fn mh_ctx_set_lmcs(ctx: *mut mh_ctx, lmcs: c_int) -> c_int;
    pub fn mh_ctx_get_lmcs(ctx: *const mh_ctx) -> c_int;
    pub fn mh_blake3(data: *const u8, n: usize, out32: *mut u8);
    pub fn mh_prof_enable(ctx: *mut mh_ctx, on: c_int) -> c_int;
    pub fn mh_prof_filter(ctx: *mut mh_ctx, name: *const c_char) -> c_int;
    pub fn mh_prof_reset(ctx: *mut mh_ctx) -> c_int;
    pub fn mh_prof_get(ctx: *mut mh_ctx, name: *const c_char, ms: *mut c_double, bytes: *mut c_double, count: *mut c_long) -> c_int;
    pub fn mh_prof_dump(ctx: *mut mh_ctx, buf: *mut c_char, cap: usize) -> c_int;
    // ---- unit-parity entry points ----
    pub fn mh_poseidon2_permute(ctx: *mut mh_ctx, states: *mut u64, n: usize) -> c_int;
    pub fn mh_poseidon2_register_rate(ctx: *mut mh_ctx, perms_per_second: *mut c_double) -> c_int;
    pub fn mh_coset_lde_batch(ctx: *mut mh_ctx, rowmajor: *const u64, log_n: c_int, width: usize, added_bits: c_int, shift: u64, out: *mut u64) -> c_int;
    // ---- traces ----
    pub fn mh_trace_upload(ctx: *mut mh_ctx, rowmajor: *const u64, log_n: c_int, width: usize, out: *mut *mut mh_trace) -> c_int;
    pub fn mh_trace_upload_async(ctx: *mut mh_ctx, rowmajor: *const u64, log_n: c_int, width: usize, out: *mut *mut mh_trace) -> c_int;
    pub fn mh_trace_upload_cols_async(ctx: *mut mh_ctx, colmajor: *const u64, log_n: c_int, width: usize, out: *mut *mut mh_trace) -> c_int;
    pub fn mh_trace_wait(ctx: *mut mh_ctx, t: *mut mh_trace) -> c_int;
    pub fn mh_trace_from_device(ctx: *mut mh_ctx, device_rowmajor: *const u64, log_n: c_int, width: usize, out: *mut *mut mh_trace) -> c_int;
    pub fn mh_trace_download(ctx: *mut mh_ctx, t: *const mh_trace, rowmajor_out: *mut u64) -> c_int;
    pub fn mh_trace_free(t: *mut mh_trace);
    pub fn mh_host_alloc(bytes: usize) -> *mut c_void;
    pub fn mh_host_free(p: *mut c_void);
    // ---- commitments ----
    pub fn mh_commit_traces(ctx: *mut mh_ctx, n_traces: c_int, traces: *const *mut mh_trace, log_blowup: c_int, out: *mut *mut mh_tree, root: *mut u64) -> c_int;
    pub fn mh_tree_free(t: *mut mh_tree);
    pub fn mh_tree_root(t: *const mh_tree, root: *mut u64) -> c_int;
    pub fn mh_tree_log_height(t: *const mh_tree) -> c_int;
    pub fn mh_tree_open(ctx: *mut mh_ctx, t: *const mh_tree, indices: *const u64, n_idx: usize, alignment: usize, fields: *mut u64, n_fields: *mut usize, commits: *mut u64, n_commit_felts: *mut usize) -> c_int;
    pub fn mh_tree_download_lde(ctx: *mut mh_ctx, t: *const mh_tree, mat: c_int, out_rowmajor_bitrev: *mut u64) -> c_int;
    pub fn mh_tree_download_layers(ctx: *mut mh_ctx, t: *const mh_tree, out: *mut u64) -> c_int;
    pub fn mh_shard_commit_leaves(ctx: *mut mh_ctx, n_traces: c_int, traces: *const *mut mh_trace, log_blowup: c_int, rank: c_int, world: c_int, out: *mut *mut mh_shard) -> c_int;
    pub fn mh_shard_free(s: *mut mh_shard);
    pub fn mh_shard_leaf_digests(s: *mut mh_shard, n_digests: *mut usize) -> *mut u64;
    pub fn mh_shard_build_subtree(ctx: *mut mh_ctx, s: *mut mh_shard, digests_device: *const u64, subroot: *mut u64) -> c_int;
    pub fn mh_merkle_cap_root(subroots: *const u64, world: c_int, root: *mut u64) -> c_int;
    pub fn mh_merkle_cap_root_lmcs(lmcs: c_int, subroots: *const u64, world: c_int, root: *mut u64) -> c_int;
    // ---- AIRs, lookups ----
    pub fn mh_air_load(ctx: *mut mh_ctx, blob: *const u64, n_words: usize, out: *mut *mut mh_air) -> c_int;
    pub fn mh_air_free(air: *mut mh_air);
    pub fn mh_air_log_quotient_degree(air: *const mh_air) -> c_int;
    pub fn mh_air_compiled_chunks(air: *const mh_air) -> c_int;
    pub fn mh_air_compiled_max_vgprs(air: *const mh_air) -> c_int;
    pub fn mh_jit_precompile(blob: *const u64, n_words: usize, n_chunks: *mut c_int) -> c_int;
    pub fn mh_air_attach_preprocessed(air: *mut mh_air, tree: *const mh_tree, matrix_index: c_int, raw: *const mh_trace) -> c_int;
    pub fn mh_air_attach_lookup(air: *mut mh_air, l: *const mh_lookup) -> c_int;
    pub fn mh_lookup_load(ctx: *mut mh_ctx, blob: *const u64, n_words: usize, out: *mut *mut mh_lookup) -> c_int;
    pub fn mh_lookup_free(l: *mut mh_lookup);
    pub fn mh_lookup_build_aux(ctx: *mut mh_ctx, l: *const mh_lookup, main_trace: *const mh_trace, preprocessed: *const mh_trace, randomness: *const u64, n_randomness: usize, aux_out: *mut *mut mh_trace, acc_final: *mut u64) -> c_int;
    // ---- proofs ----
    pub fn mh_prove(ctx: *mut mh_ctx, params: *const mh_pcs_params, n_airs: c_int, airs: *const *mut mh_air, traces: *const *mut mh_trace, public_values: *const u64, n_public_values: usize, challenger_state: *const u64, pre_observe: *const u64, n_pre_observe: usize, aux_builder: mh_aux_builder, user: *mut c_void, out: *mut *mut mh_proof) -> c_int;
    pub fn mh_commit_traces_sharded(ctx: *mut mh_ctx, comm: *const mh_comm, n_traces: c_int, traces: *const *mut mh_trace, log_blowup: c_int, out: *mut *mut mh_tree, root: *mut u64) -> c_int;
    pub fn mh_prove_host(ctx: *mut mh_ctx, params: *const mh_pcs_params, n_airs: c_int, airs: *const *mut mh_air, traces_rowmajor: *const *const u64,
                         log_heights: *const c_int, public_values: *const u64, n_public_values: usize, challenger_state: *const u64,
                         pre_observe: *const u64, n_pre_observe: usize, aux_builder: mh_aux_builder, user: *mut c_void, out: *mut *mut mh_proof) -> c_int;
    pub fn mh_prove_sharded(ctx: *mut mh_ctx, comm: *const mh_comm, params: *const mh_pcs_params, n_airs: c_int, airs: *const *mut mh_air, traces: *const *mut mh_trace, public_values: *const u64, n_public_values: usize, challenger_state: *const u64, pre_observe: *const u64, n_pre_observe: usize, aux_builder: mh_aux_builder, user: *mut c_void, out: *mut *mut mh_proof) -> c_int;
    pub fn mh_session_begin(ctx: *mut mh_ctx, comm: *const mh_comm, params: *const mh_pcs_params, n_airs: c_int, airs: *const *mut mh_air, traces: *const *mut mh_trace, public_values: *const u64, n_public_values: usize, out: *mut *mut mh_session) -> c_int;
    pub fn mh_session_free(s: *mut mh_session);
    pub fn mh_session_shape(s: *const mh_session, out: *mut mh_session_shape_t) -> c_int;
    pub fn mh_session_commit_main(s: *mut mh_session, root: *mut u64) -> c_int;
    pub fn mh_session_commit_aux(s: *mut mh_session, randomness: *const u64, aux_builder: mh_aux_builder, user: *mut c_void, root: *mut u64, aux_values_out: *mut u64) -> c_int;
    pub fn mh_session_commit_quotient(s: *mut mh_session, alpha: *const u64, beta: *const u64, root: *mut u64) -> c_int;
    pub fn mh_session_ood_point_ok(s: *const mh_session, z: *const u64) -> c_int;
    pub fn mh_session_ood(s: *mut mh_session, z: *const u64, evals_out: *mut u64) -> c_int;
    pub fn mh_session_deep(s: *mut mh_session, alpha: *const u64, beta: *const u64) -> c_int;
    pub fn mh_session_fri_commit(s: *mut mh_session, root: *mut u64) -> c_int;
    pub fn mh_session_fri_fold(s: *mut mh_session, beta: *const u64) -> c_int;
    pub fn mh_session_fri_final(s: *mut mh_session, coeffs_out: *mut u64) -> c_int;
    pub fn mh_session_open(s: *mut mh_session, indices: *const u64, n_indices: usize, out: *mut *mut mh_proof) -> c_int;
    pub fn mh_grind(ctx: *mut mh_ctx, state: *const u64, pending: *const u64, n_pending: usize, bits: c_int, witness: *mut u64) -> c_int;
    pub fn mh_grind_bytes(ctx: *mut mh_ctx, input: *const u8, n_input: usize, bits: c_int, witness: *mut u64) -> c_int;
    pub fn mh_verify(params: *const mh_pcs_params, n_airs: c_int, air_blobs: *const *const u64, air_blob_words: *const usize, log_trace_heights: *const u8, public_values: *const u64, n_public_values: usize, challenger_state: *const u64, pre_observe: *const u64, n_pre_observe: usize, fields: *const u64, n_fields: usize, commitments: *const u64, n_commitments: usize, preprocessed_root: *const u64, digest: *mut u64, err: *mut c_char, err_cap: usize) -> c_int;
    pub fn mh_verify_ex(params: *const mh_pcs_params, n_airs: c_int, air_blobs: *const *const u64, air_blob_words: *const usize, log_trace_heights: *const u8, public_values: *const u64, n_public_values: usize, challenger_state: *const u64, pre_observe: *const u64, n_pre_observe: usize, fields: *const u64, n_fields: usize, commitments: *const u64, n_commitments: usize, preprocessed_root: *const u64, external: mh_external_assertions, external_user: *mut c_void, digest: *mut u64, err: *mut c_char, err_cap: usize) -> c_int;
    /// mh_verify_ex for MH_LMCS_RPO (3) / MH_LMCS_RPX (4) / MH_LMCS_POSEIDON2 (0)
    pub fn mh_verify_lmcs(lmcs: c_int, params: *const mh_pcs_params, n_airs: c_int, air_blobs: *const *const u64, air_blob_words: *const usize, log_trace_heights: *const u8, public_values: *const u64, n_public_values: usize, challenger_state: *const u64, pre_observe: *const u64, n_pre_observe: usize, fields: *const u64, n_fields: usize, commitments: *const u64, n_commitments: usize, preprocessed_root: *const u64, external: mh_external_assertions, external_user: *mut c_void, digest: *mut u64, err: *mut c_char, err_cap: usize) -> c_int;
    pub fn mh_external_logup_balance(user: *mut c_void, randomness: *const u64, n_randomness: usize, aux_values: *const *const u64, n_aux_values: *const usize, log_trace_heights: *const u8, n_airs: c_int, assertions_out: *mut u64, cap: usize) -> c_int;
    pub fn mh_external_precompile_session(user: *mut c_void, randomness: *const u64, n_randomness: usize, aux_values: *const *const u64, n_aux_values: *const usize, log_trace_heights: *const u8, n_airs: c_int, assertions_out: *mut u64, cap: usize) -> c_int;
    pub fn mh_external_precompile_session_ec_only(user: *mut c_void, randomness: *const u64, n_randomness: usize, aux_values: *const *const u64, n_aux_values: *const usize, log_trace_heights: *const u8, n_airs: c_int, assertions_out: *mut u64, cap: usize) -> c_int;
    // the precompile prover's session: SessionTraces::prove_stark's own shape (csrc/precompile.cpp)
    pub fn mh_precompile_pcs_params(out: *mut mh_pcs_params);
    pub fn mh_precompile_load(ctx: *mut mh_ctx, out: *mut *mut mh_precompile) -> c_int;
    pub fn mh_precompile_free(s: *mut mh_precompile);
    pub fn mh_precompile_air_blob(which: c_int, lookup: c_int, words_out: *mut *const u64, n_words: *mut usize) -> c_int;
    pub fn mh_precompile_preprocessed_root(s: *mut mh_precompile, hash_fn: c_int, root: *mut u64) -> c_int;
    pub fn mh_precompile_pre_observe(p: *const mh_pcs_params, preprocessed_root: *const u64, public_root: *const u64, out: *mut u64) -> c_int;
    pub fn mh_prove_precompile(ctx: *mut mh_ctx, s: *mut mh_precompile, hash_fn: c_int, mains_rowmajor: *const *const u64, log_heights: *const c_int, public_root: *const u64, out: *mut *mut mh_proof) -> c_int;
    pub fn mh_prove_precompile_traces(ctx: *mut mh_ctx, s: *mut mh_precompile, hash_fn: c_int, traces: *const *mut mh_trace, public_root: *const u64, out: *mut *mut mh_proof) -> c_int;
    pub fn mh_verify_precompile(hash_fn: c_int, preprocessed_root: *const u64, public_root: *const u64, proof_bytes: *const u8, n_bytes: usize, digest: *mut u64, err: *mut c_char, err_cap: usize) -> c_int;
    pub fn mh_proof_deserialize(bytes: *const u8, len: usize, out: *mut *mut mh_proof) -> c_int;
    pub fn mh_trace_upload_sharded(ctx: *mut mh_ctx, comm: *const mh_comm, rowmajor: *const u64, log_n: c_int, width: usize, out: *mut *mut mh_trace) -> c_int;
    pub fn mh_rccl_unique_id(id: *mut u8) -> c_int;
    pub fn mh_comm_create_rccl(ctx: *mut mh_ctx, id: *const u8, rank: c_int, world: c_int, out: *mut *mut mh_comm) -> c_int;
    pub fn mh_comm_destroy(comm: *mut mh_comm);
    pub fn mh_local_fabric_create(world: c_int) -> *mut mh_local_fabric;
    pub fn mh_local_fabric_destroy(f: *mut mh_local_fabric);
    pub fn mh_local_fabric_abort(f: *mut mh_local_fabric);
    pub fn mh_comm_create_local(ctx: *mut mh_ctx, f: *mut mh_local_fabric, rank: c_int, out: *mut *mut mh_comm) -> c_int;
    pub fn mh_comm_selftest(ctx: *mut mh_ctx, comm: *const mh_comm) -> c_int;
    // ---- the Miden statement in the library (prove_stark's own shape, prover/src/lib.rs:317-355); a Rust shim that keeps
    // `MidenMultiAir` on its side uses mh_prove / mh_verify_ex instead
    pub fn mh_miden_load(ctx: *mut mh_ctx, out: *mut *mut mh_miden) -> c_int;
    pub fn mh_miden_free(m: *mut mh_miden);
    pub fn mh_prove_miden(ctx: *mut mh_ctx, m: *const mh_miden, hash_fn: c_int, core_rowmajor: *const u64, log_core: c_int, chiplets_rowmajor: *const u64, log_chiplets: c_int, poseidon2_rowmajor: *const u64, log_poseidon2: c_int, public_values: *const u64, aux_inputs: *const u64, n_aux_inputs: usize, out: *mut *mut mh_proof) -> c_int;
    pub fn mh_prove_miden_traces(ctx: *mut mh_ctx, m: *const mh_miden, hash_fn: c_int, traces: *const *mut mh_trace, public_values: *const u64, aux_inputs: *const u64, n_aux_inputs: usize, out: *mut *mut mh_proof) -> c_int;
    pub fn mh_verify_miden(hash_fn: c_int, public_values: *const u64, aux_inputs: *const u64, n_aux_inputs: usize, proof_bytes: *const u8, n_bytes: usize, digest: *mut u64, err: *mut c_char, err_cap: usize) -> c_int;
    pub fn mh_miden_pcs_params(out: *mut mh_pcs_params);
    pub fn mh_miden_challenger_state(state: *mut u64);
    pub fn mh_miden_hash_kernel_digests(kernel_felts: *const u64, n_felts: usize, out: *mut u64) -> c_int;
    pub fn mh_miden_pre_observe(params: *const mh_pcs_params, public_values: *const u64, aux_inputs: *const u64, n_aux_inputs: usize, out: *mut u64) -> c_int;
    pub fn mh_miden_eval_external(randomness: *const u64, aux_inputs: *const u64, n_aux_inputs: usize, aux_values: *const *const u64, n_aux_values: *const usize, n_airs: c_int, out: *mut u64) -> c_int;
    pub fn mh_miden_air_blob(which: c_int, words_out: *mut *const u64, n_words: *mut usize) -> c_int;
    pub fn mh_proof_free(p: *mut mh_proof);
    pub fn mh_proof_num_fields(p: *const mh_proof) -> usize;
    pub fn mh_proof_num_commitments(p: *const mh_proof) -> usize;
    pub fn mh_proof_fields(p: *const mh_proof) -> *const u64;
    pub fn mh_proof_commitments(p: *const mh_proof) -> *const u64;
    pub fn mh_proof_digest(p: *const mh_proof) -> *const u64;
    pub fn mh_proof_num_traces(p: *const mh_proof) -> usize;
    pub fn mh_proof_log_trace_heights(p: *const mh_proof) -> *const u8;
    pub fn mh_proof_serialize(p: *const mh_proof, out: *mut u8, cap: usize) -> usize;
}
