// The Miden VM statement above the proof system, in C++ behind the C ABI: what a caller of `miden_prover::prove_stark`
// (prover/src/lib.rs:317-355: three matrices + public values + aux inputs -> proof bytes) gets from `MidenMultiAir` in the reference,
// with no Python and no Rust in between.
//
//   * the three AIRs (CoreAir, ChipletsAir, Poseidon2PermutationAir: air/src/lib.rs:560-650) and their LogUp lookup programs, embedded
//     as the constraint-DAG / lookup blobs of miden-vm_amd/blobs (tests keep those equal to what core_air.py / chiplets_air.py /
//     miden_air.py generate);
//   * `MidenMultiAir::observe` (air/src/lib.rs:805-849): the 48-felt rate-aligned schedule, `hash_kernel_digests` (:946-961) =
//     Poseidon2 `hash_elements` of the kernel-digest felts (crates/crypto/src/hash/algebraic_sponge/mod.rs:215-265);
//   * `MidenMultiAir::eval_external` (air/src/lib.rs:854-933) with `MidenAir::boundary_correction` (:620-650) over
//     `emit_core_boundary` / `emit_chiplets_boundary` (air/src/constraints/lookup/miden_air.rs:32-66), challenges
//     `Challenges::new(alpha, beta, 16, 25)` (air/src/lookup/challenges.rs:14-36);
//   * the production configuration: `pcs_params()` and RELATION_DIGEST (air/src/config.rs:54-67, 93-98), the hash function chosen as
//     `prove_miden_vm_execution_trace` does (prover/src/lib.rs:246-300).
//
// Written against the public ABI only (include/midenhip.h): everything here could live in the caller; it is in the library so that a
// C caller has nothing to restate.  miden-vm_amd/miden_statement.py is the same layer in Python (tests compare the two).
#include "../../include/midenhip.h"
#include "ctx.hpp"
#include "gl.cuh"
#include "poseidon2.cuh"
#include <cstring>
#include <memory>
#include <string>
#include <vector>

// ---- the shipped blobs, linked into the library ------------------------------------------------------------------------------------
#define MH_EMBED(sym, path)                                                                                           \
  asm(".section .rodata\n.balign 8\n.global " #sym "\n" #sym ":\n.incbin \"" path "\"\n.global " #sym "_end\n" #sym "_end:\n.previous\n"); \
  extern "C" const unsigned char sym[], sym##_end[];
MH_EMBED(mh_blob_core_dag, "../blobs/core.dag")
MH_EMBED(mh_blob_core_lkp, "../blobs/core.lkp")
MH_EMBED(mh_blob_chiplets_dag, "../blobs/chiplets.dag")
MH_EMBED(mh_blob_chiplets_lkp, "../blobs/chiplets.lkp")
MH_EMBED(mh_blob_p2_dag, "../blobs/poseidon2_permutation.dag")
MH_EMBED(mh_blob_p2_lkp, "../blobs/poseidon2_permutation.lkp")

namespace {

constexpr size_t NUM_PUBLIC = MH_MIDEN_NUM_PUBLIC_VALUES;  // air/src/lib.rs:270
constexpr size_t AUX_PROGRAM_HASH = 0, AUX_DEFERRED_ROOT = 4, AUX_KERNEL_DIGESTS = 8;  // lib.rs:279-281
constexpr size_t MAX_KERNEL_PROCS = 255;                    // KernelDescriptor::MAX_NUM_PROCEDURES
constexpr int MAX_MESSAGE_WIDTH = 16, NUM_BUS_IDS = 25;     // MIDEN_MAX_MESSAGE_WIDTH, BusId::COUNT (air/src/constraints/lookup/messages.rs:55-107)
constexpr int BUS_KERNEL_ROM_INIT = 0, BUS_BLOCK_HASH_TABLE = 1, BUS_LOG_DEFERRED_ROOT = 2;
// RELATION_DIGEST (air/src/config.rs:93-98), held to the reference by tests/golden/kat.json
const u64 RELATION_DIGEST[4] = {837197885082815666ULL, 17812429367884914ULL, 12945170128166309606ULL, 6547471563106428306ULL};

std::vector<u64> words(const unsigned char* b, const unsigned char* e) {
  std::vector<u64> v((size_t)(e - b) / 8);
  memcpy(v.data(), b, v.size() * 8);
  return v;
}
struct Blobs {
  std::vector<u64> dag[3], lkp[3];
  Blobs() {
    dag[0] = words(mh_blob_core_dag, mh_blob_core_dag_end); lkp[0] = words(mh_blob_core_lkp, mh_blob_core_lkp_end);
    dag[1] = words(mh_blob_chiplets_dag, mh_blob_chiplets_dag_end); lkp[1] = words(mh_blob_chiplets_lkp, mh_blob_chiplets_lkp_end);
    dag[2] = words(mh_blob_p2_dag, mh_blob_p2_dag_end); lkp[2] = words(mh_blob_p2_lkp, mh_blob_p2_lkp_end);
    // inside the statement every MidenAir declares NUM_PUBLIC_VALUES (air/src/lib.rs:660-662); the stand-alone permutation AIR that
    // is shipped declares none and reads none: header word [5] is the only difference
    for (auto& d : dag) d[5] = NUM_PUBLIC;
  }
};
const Blobs& blobs() {
  static const Blobs b;
  return b;
}

void hash_elements(const u64* xs, size_t n, u64 out[4]) {
  u64 s[12] = {0};
  if (n) {
    s[8] = n % 8;
    for (size_t i = 0; i < n; i += 8) {
      for (size_t k = 0; k < 8; k++) s[k] = i + k < n ? gl_canon(xs[i + k]) : 0;
      p2_permute(s);
    }
  }
  for (int k = 0; k < 4; k++) out[k] = s[k];
}

struct Challenges {  // air/src/lookup/challenges.rs:14-36
  e2 beta_pow[MAX_MESSAGE_WIDTH], prefix[NUM_BUS_IDS];
  Challenges(e2 alpha, e2 beta) {
    beta_pow[0] = e2_make(1);
    for (int i = 1; i < MAX_MESSAGE_WIDTH; i++) beta_pow[i] = e2_mul(beta_pow[i - 1], beta);
    const e2 gamma = e2_mul(beta_pow[MAX_MESSAGE_WIDTH - 1], beta);
    for (int i = 0; i < NUM_BUS_IDS; i++) prefix[i] = e2_add(alpha, e2_mulf(gamma, (u64)(i + 1)));
  }
  e2 encode(int bus, const u64* f, int n) const {
    e2 acc = prefix[bus];
    for (int i = 0; i < n; i++) acc = e2_add(acc, e2_mulf(beta_pow[i], gl_canon(f[i])));
    return acc;
  }
};

// -> 0 ok, -1 a denominator is zero (the reference's ReductionError), -2 shape error
int eval_external(const u64* rnd, size_t n_rnd, const u64* aux_inputs, size_t n_aux, const u64* const* aux_values, const size_t* n_aux_values,
                  int n_airs, u64 out[2]) {
  if (n_airs != 3 || n_rnd != 2 || n_aux < AUX_KERNEL_DIGESTS || n_aux > AUX_KERNEL_DIGESTS + MAX_KERNEL_PROCS * 4 ||
      (n_aux - AUX_KERNEL_DIGESTS) % 4)
    return -2;
  const Challenges ch(e2{gl_canon(rnd[0]), gl_canon(rnd[1])}, e2{gl_canon(rnd[2]), gl_canon(rnd[3])});
  e2 total = e2_make(0);
  bool zero_den = false;
  auto add_inv = [&](e2 d, bool negate) {
    if (e2_is_zero(d)) { zero_den = true; return; }
    const e2 v = e2_inv(d);
    total = negate ? e2_sub(total, v) : e2_add(total, v);
  };
  // Core (emit_core_boundary): the block-hash seed Child{parent 0, program_hash, is_first_child 0, is_loop_body 0}; the deferred-root
  // log's initial entry added, its final entry removed
  u64 seed[7] = {aux_inputs[0], aux_inputs[1], aux_inputs[2], aux_inputs[3], 0, 0, 0};
  add_inv(ch.encode(BUS_BLOCK_HASH_TABLE, seed, 7), false);
  const u64 zero4[4] = {0, 0, 0, 0};
  add_inv(ch.encode(BUS_LOG_DEFERRED_ROOT, zero4, 4), false);
  add_inv(ch.encode(BUS_LOG_DEFERRED_ROOT, aux_inputs + AUX_DEFERRED_ROOT, 4), true);
  // Chiplets (emit_chiplets_boundary): one KernelRomInit per kernel digest; Poseidon2Permutation: nothing
  for (size_t i = AUX_KERNEL_DIGESTS; i < n_aux; i += 4) add_inv(ch.encode(BUS_KERNEL_ROM_INIT, aux_inputs + i, 4), false);
  if (zero_den) return -1;
  for (int a = 0; a < 3; a++) {
    if (n_aux_values[a] != 1) return -2;  // every Miden AIR commits exactly one LogUp final
    total = e2_add(total, e2{gl_canon(aux_values[a][0]), gl_canon(aux_values[a][1])});
  }
  out[0] = total.c0;
  out[1] = total.c1;
  return 0;
}

struct Statement {
  const u64* aux_inputs;
  size_t n_aux;
};
int external_cb(void* user, const uint64_t* randomness, size_t n_randomness, const uint64_t* const* aux_values, const size_t* n_aux_values,
                const uint8_t* log_trace_heights, int n_airs, uint64_t* assertions_out, size_t cap) {
  (void)log_trace_heights;
  const Statement* st = (const Statement*)user;
  if (cap < 1) return -1;
  u64 v[2];
  const int rc = eval_external(randomness, n_randomness, st->aux_inputs, st->n_aux, aux_values, n_aux_values, n_airs, v);
  if (rc) return -1;
  assertions_out[0] = v[0];
  assertions_out[1] = v[1];
  return 1;
}

bool statement_shape_ok(const u64* pv, const u64* aux, size_t n_aux) {
  return pv && aux && n_aux >= AUX_KERNEL_DIGESTS && n_aux <= AUX_KERNEL_DIGESTS + MAX_KERNEL_PROCS * 4 && (n_aux - AUX_KERNEL_DIGESTS) % 4 == 0;
}

}  // namespace

struct mh_miden {
  mh_ctx* ctx = nullptr;
  mh_air* airs[3] = {nullptr, nullptr, nullptr};
  mh_lookup* lookups[3] = {nullptr, nullptr, nullptr};
};

extern "C" {

void mh_miden_pcs_params(mh_pcs_params* out) {  // air/src/config.rs:54-67
  if (!out) return;
  out->log_blowup = 3; out->log_folding_arity = 2; out->log_final_degree = 7; out->folding_pow_bits = 4;
  out->deep_pow_bits = 12; out->num_queries = 27; out->query_pow_bits = 16;
}
void mh_miden_challenger_state(uint64_t state[12]) {  // air/src/config.rs:255-273: RELATION_DIGEST in the capacity
  if (!state) return;
  for (int i = 0; i < 8; i++) state[i] = 0;
  for (int i = 0; i < 4; i++) state[8 + i] = RELATION_DIGEST[i];
}
int mh_miden_hash_kernel_digests(const uint64_t* kernel_felts, size_t n_felts, uint64_t out[4]) {
  if (!out || (n_felts && !kernel_felts) || n_felts % 4 || n_felts > MAX_KERNEL_PROCS * 4) return MH_ERR_INVALID;
  hash_elements(kernel_felts, n_felts, out);
  return MH_OK;
}
int mh_miden_pre_observe(const mh_pcs_params* p, const uint64_t* public_values, const uint64_t* aux_inputs, size_t n_aux_inputs,
                         uint64_t out[MH_MIDEN_PRE_OBSERVE_FELTS]) {
  if (!p || !out || !statement_shape_ok(public_values, aux_inputs, n_aux_inputs)) return MH_ERR_INVALID;
  // observe_protocol_params (air/src/config.rs:188-198)
  const u64 head[8] = {(u64)p->num_queries, (u64)p->query_pow_bits, (u64)p->deep_pow_bits, (u64)p->folding_pow_bits, (u64)p->log_blowup,
                       (u64)p->log_final_degree, (u64)1 << p->log_folding_arity, 0};
  size_t k = 0;
  for (u64 v : head) out[k++] = v;
  u64 kh[4];
  hash_elements(aux_inputs + AUX_KERNEL_DIGESTS, n_aux_inputs - AUX_KERNEL_DIGESTS, kh);
  for (int i = 0; i < 4; i++) out[k++] = kh[i];
  for (int i = 0; i < 4; i++) out[k++] = gl_canon(aux_inputs[AUX_PROGRAM_HASH + i]);
  for (int i = 0; i < 4; i++) out[k++] = gl_canon(aux_inputs[AUX_DEFERRED_ROOT + i]);
  for (int i = 0; i < 4; i++) out[k++] = 0;
  for (size_t i = 0; i < NUM_PUBLIC; i++) out[k++] = gl_canon(public_values[i]);
  return MH_OK;
}
int mh_miden_eval_external(const uint64_t randomness[4], const uint64_t* aux_inputs, size_t n_aux_inputs, const uint64_t* const* aux_values,
                           const size_t* n_aux_values, int n_airs, uint64_t out[2]) {
  if (!randomness || !aux_inputs || !aux_values || !n_aux_values || !out) return MH_ERR_INVALID;
  return eval_external(randomness, 2, aux_inputs, n_aux_inputs, aux_values, n_aux_values, n_airs, out) ? MH_ERR_INVALID : MH_OK;
}

int mh_miden_load(mh_ctx* ctx, mh_miden** out) {
  if (!ctx || !out) return MH_ERR_INVALID;
  std::unique_ptr<mh_miden, void (*)(mh_miden*)> m(new mh_miden(), mh_miden_free);
  m->ctx = ctx;
  const Blobs& b = blobs();
  for (int i = 0; i < 3; i++) {
    int rc = mh_air_load(ctx, b.dag[i].data(), b.dag[i].size(), &m->airs[i]);
    if (rc == MH_OK) rc = mh_lookup_load(ctx, b.lkp[i].data(), b.lkp[i].size(), &m->lookups[i]);
    if (rc == MH_OK) rc = mh_air_attach_lookup(m->airs[i], m->lookups[i]);  // the LogUp columns are built on the device
    if (rc != MH_OK) return rc;
  }
  *out = m.release();
  return MH_OK;
}
void mh_miden_free(mh_miden* m) {
  if (!m) return;
  for (int i = 0; i < 3; i++) {
    if (m->airs[i]) mh_air_free(m->airs[i]);
    if (m->lookups[i]) mh_lookup_free(m->lookups[i]);
  }
  delete m;
}
int mh_miden_air_blob(int which, const uint64_t** words_out, size_t* n_words) {
  if (which < 0 || which > 2 || !words_out || !n_words) return MH_ERR_INVALID;
  *words_out = blobs().dag[which].data();
  *n_words = blobs().dag[which].size();
  return MH_OK;
}

static int prove_common(mh_ctx* ctx, const mh_miden* m, int hash_fn, const uint64_t* const* host_rm, const int* log_heights, mh_trace* const* traces,
                        const uint64_t* public_values, const uint64_t* aux_inputs, size_t n_aux_inputs, mh_proof** out) {
  if (!ctx || !m || m->ctx != ctx || !out) return MH_ERR_INVALID;
  mh_pcs_params prm;
  mh_miden_pcs_params(&prm);
  u64 pre[MH_MIDEN_PRE_OBSERVE_FELTS], state[12];
  if (mh_miden_pre_observe(&prm, public_values, aux_inputs, n_aux_inputs, pre) != MH_OK) {
    ctx->err = "mh_prove_miden: 32 public values and aux inputs = program hash (4) | deferred root (4) | kernel digests (4 each, <= 255) expected";
    return MH_ERR_INVALID;
  }
  mh_miden_challenger_state(state);
  const int old = mh_ctx_get_lmcs(ctx);
  int rc = mh_ctx_set_lmcs(ctx, hash_fn);
  if (rc != MH_OK) return rc;
  if (traces)
    rc = mh_prove(ctx, &prm, 3, m->airs, traces, public_values, NUM_PUBLIC, state, pre, MH_MIDEN_PRE_OBSERVE_FELTS, nullptr, nullptr, out);
  else
    rc = mh_prove_host(ctx, &prm, 3, m->airs, host_rm, log_heights, public_values, NUM_PUBLIC, state, pre, MH_MIDEN_PRE_OBSERVE_FELTS, nullptr,
                       nullptr, out);
  const std::string err = ctx->err;
  (void)mh_ctx_set_lmcs(ctx, old);
  if (rc != MH_OK) ctx->err = err;
  return rc;
}

int mh_prove_miden(mh_ctx* ctx, const mh_miden* m, int hash_fn, const uint64_t* core_rowmajor, int log_core, const uint64_t* chiplets_rowmajor,
                   int log_chiplets, const uint64_t* poseidon2_rowmajor, int log_poseidon2, const uint64_t* public_values,
                   const uint64_t* aux_inputs, size_t n_aux_inputs, mh_proof** out) {
  const uint64_t* rm[3] = {core_rowmajor, chiplets_rowmajor, poseidon2_rowmajor};
  const int lh[3] = {log_core, log_chiplets, log_poseidon2};
  return prove_common(ctx, m, hash_fn, rm, lh, nullptr, public_values, aux_inputs, n_aux_inputs, out);
}
int mh_prove_miden_traces(mh_ctx* ctx, const mh_miden* m, int hash_fn, mh_trace* const traces[3], const uint64_t* public_values,
                          const uint64_t* aux_inputs, size_t n_aux_inputs, mh_proof** out) {
  if (!traces) return MH_ERR_INVALID;
  return prove_common(ctx, m, hash_fn, nullptr, nullptr, traces, public_values, aux_inputs, n_aux_inputs, out);
}

int mh_verify_miden(int hash_fn, const uint64_t* public_values, const uint64_t* aux_inputs, size_t n_aux_inputs, const uint8_t* proof_bytes,
                    size_t n_bytes, uint64_t digest[4], char* err, size_t err_cap) {
  auto fail = [&](const char* msg) {
    if (err && err_cap) snprintf(err, err_cap, "%s", msg);
    return MH_ERR_INVALID;
  };
  if (!proof_bytes || !digest) return fail("null argument");
  mh_pcs_params prm;
  mh_miden_pcs_params(&prm);
  u64 pre[MH_MIDEN_PRE_OBSERVE_FELTS], state[12];
  if (mh_miden_pre_observe(&prm, public_values, aux_inputs, n_aux_inputs, pre) != MH_OK) return fail("malformed public values / aux inputs");
  mh_miden_challenger_state(state);
  mh_proof* p = nullptr;
  if (mh_proof_deserialize(proof_bytes, n_bytes, &p) != MH_OK) return fail("malformed proof bytes");
  std::unique_ptr<mh_proof, void (*)(mh_proof*)> hold(p, mh_proof_free);
  if (mh_proof_num_traces(p) != 3) return fail("a Miden proof has three traces");
  const Blobs& b = blobs();
  const uint64_t* blob_ptr[3] = {b.dag[0].data(), b.dag[1].data(), b.dag[2].data()};
  const size_t blob_len[3] = {b.dag[0].size(), b.dag[1].size(), b.dag[2].size()};
  Statement st{aux_inputs, n_aux_inputs};
  return mh_verify_lmcs(hash_fn, &prm, 3, blob_ptr, blob_len, mh_proof_log_trace_heights(p), public_values, NUM_PUBLIC, state, pre,
                        MH_MIDEN_PRE_OBSERVE_FELTS, mh_proof_fields(p), mh_proof_num_fields(p), mh_proof_commitments(p),
                        mh_proof_num_commitments(p), nullptr, external_cb, &st, digest, err, err_cap);
}

}  // extern "C"
