// ORACLE — TEST INFRASTRUCTURE ONLY (see gl.hpp header).  C entry points for ctypes (tests/,
// __graft_entry__.smoke(), bench.py cpu_baseline).  Parity pinned against the reference's in-tree
// known answers (tests/test_oracle_kat.py); the protocol-level pieces (transcript bytes, PoW
// witness, proof framing) are "parity unpinned" -- see DESIGN.md.
#include "gl.hpp"
#include "poseidon2.hpp"
#include "ntt.hpp"
#include "lmcs.hpp"
#include "stark.hpp"
#include "lookup.hpp"
#include "blake3.hpp"
#include "keccak.hpp"
#include <cstring>
#include <cstdio>

using namespace oracle;

extern "C" {

uint64_t orc_fmul(uint64_t a, uint64_t b) { return fmul(a, b); }
uint64_t orc_fadd(uint64_t a, uint64_t b) { return fadd(a, b); }
uint64_t orc_fsub(uint64_t a, uint64_t b) { return fsub(a, b); }
uint64_t orc_finv(uint64_t a) { return finv(a); }
uint64_t orc_fpow(uint64_t a, uint64_t e) { return fpow(a, e); }
uint64_t orc_two_adic_generator(int k) { return two_adic_generator(k); }
uint64_t orc_canonical_lde_shift(int log_lde) { return canonical_lde_shift(log_lde); }
void orc_emul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
  E2 r = emul(E2{a[0], a[1]}, E2{b[0], b[1]});
  out[0] = r.c0; out[1] = r.c1;
}
void orc_einv(const uint64_t a[2], uint64_t out[2]) {
  E2 r = einv(E2{a[0], a[1]});
  out[0] = r.c0; out[1] = r.c1;
}

// n states of 12 felts, in place
void orc_permute(uint64_t* states, size_t n) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)n; i++) p2_permute(states + 12 * i);
}
void orc_hash_elements(const uint64_t* in, size_t n, uint64_t out[4]) { hash_elements(in, n, out); }
void orc_compress(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) { compress(l, r, out); }
void orc_keccak_f1600(uint64_t st[25]) { kk::f1600(st); }
void orc_keccak256(const uint8_t* data, size_t n, int pad, uint8_t out[32]) { kk::hash256(data, n, (uint8_t)pad, out); }
void orc_blake3(const uint8_t* data, size_t n, uint8_t out[32]) { b3::hash(data, n, out); }
void orc_sponge_absorb(uint64_t state[12], const uint64_t* in, size_t n) { sponge_absorb(state, in, n); }

void orc_naive_dft(const uint64_t* in, size_t n, int inverse, uint64_t* out) {
  std::vector<uint64_t> v(in, in + n);
  auto r = naive_dft(v, inverse != 0);
  memcpy(out, r.data(), n * 8);
}
void orc_dft(uint64_t* a, size_t n, int inverse) { dft_inplace(a, n, inverse != 0); }

// row-major n x w -> row-major (n<<added_bits) x w, physical rows bit-reversed
void orc_coset_lde_bitrev(const uint64_t* m, size_t n, size_t w, int added_bits, uint64_t shift, uint64_t* out) {
  auto r = coset_lde_matrix_bitrev(m, n, w, added_bits, shift);
  memcpy(out, r.data(), r.size() * 8);
}

// LMCS over already bit-reversed row-major matrices (ascending heights).
// layers_out (optional): all digest layers bottom(leaf, domain order)-up concatenated: 2H-1 digests.
void orc_set_lmcs(int hash) {  // the configuration of every later call (tests are serial)
  g_lmcs = hash;
  g_alg_perm = (hash == LMCS_RPO || hash == LMCS_RPX) ? hash : 0;
}
// the 2-to-1 node function of the configuration set by orc_set_lmcs (lifted_tree.rs:472-511 compress_uniform's `compress`)
void orc_lmcs_compress(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
  Digest a{l[0], l[1], l[2], l[3]}, b{r[0], r[1], r[2], r[3]}, o;
  if (g_lmcs == LMCS_BLAKE3) o = b3_compress(a, b);
  else if (g_lmcs == LMCS_KECCAK) o = keccak_compress(a, b);
  else compress(l, r, o.data());
  memcpy(out, o.data(), 32);
}
void orc_rescue_permute(int which, uint64_t st[12]) {
  if (which == LMCS_RPX) rpx_permute(st);
  else rpo_permute(st);
}
void orc_lmcs_build(int n_mats, const uint64_t* const* ptrs, const size_t* heights, const size_t* widths,
                    uint64_t root_out[4], uint64_t* layers_out) {
  std::vector<Mat> mats;
  for (int i = 0; i < n_mats; i++) mats.push_back(Mat{ptrs[i], heights[i], widths[i]});
  LmcsTree t = lmcs_build(mats);
  Digest r = t.root();
  memcpy(root_out, r.data(), 32);
  if (layers_out) {
    size_t off = 0;
    for (int d = (int)t.layers.size() - 1; d >= 0; d--) {
      memcpy(layers_out + off, t.layers[d].data(), t.layers[d].size() * 32);
      off += t.layers[d].size() * 4;
    }
  }
}

// commit_traces (prover/commit.rs:142-180): natural-order row-major traces (ascending heights)
// -> per-trace coset LDE with the canonical shift for ITS OWN lde order -> LMCS.
// Optionally returns the LDE matrices (bit-reversed row-major) via lde_out[i] (may be null),
// opened aligned rows + siblings for `n_idx` domain indices via fields_out/commit_out.
void orc_commit_traces(int n_mats, const uint64_t* const* ptrs, const int* log_heights, const size_t* widths,
                       int log_blowup, uint64_t root_out[4], uint64_t* const* lde_out,
                       const size_t* indices, size_t n_idx, size_t alignment,
                       uint64_t* fields_out, size_t* n_fields, uint64_t* commit_out, size_t* n_commit) {
  std::vector<std::vector<uint64_t>> ldes(n_mats);
  std::vector<Mat> mats;
  for (int i = 0; i < n_mats; i++) {
    size_t n = (size_t)1 << log_heights[i];
    uint64_t shift = canonical_lde_shift(log_heights[i] + log_blowup);
    ldes[i] = coset_lde_matrix_bitrev(ptrs[i], n, widths[i], log_blowup, shift);
    mats.push_back(Mat{ldes[i].data(), n << log_blowup, widths[i]});
    if (lde_out && lde_out[i]) memcpy(lde_out[i], ldes[i].data(), ldes[i].size() * 8);
  }
  LmcsTree t = lmcs_build(mats);
  Digest r = t.root();
  memcpy(root_out, r.data(), 32);
  if (n_idx) {
    std::vector<uint64_t> f;
    std::vector<Digest> c;
    lmcs_prove_batch(t, std::vector<size_t>(indices, indices + n_idx), alignment, f, c);
    memcpy(fields_out, f.data(), f.size() * 8);
    *n_fields = f.size();
    memcpy(commit_out, c.data(), c.size() * 32);
    *n_commit = c.size();
  }
}


// ---- whole-protocol entry points (stark.hpp) ----------------------------------------------------
static Challenger make_challenger(const uint64_t init_state[12], const uint64_t* pre_observe, size_t n_pre) {
  Challenger c;
  c.init(init_state);
  for (size_t i = 0; i < n_pre; i++) c.observe(pre_observe[i]);
  return c;
}
static PcsParams make_params(const int p[7]) { return PcsParams{p[0], p[1], p[2], p[3], p[4], p[5], p[6]}; }
static void set_err(char* err, size_t cap, const char* msg) {
  if (err && cap) snprintf(err, cap, "%s", msg);
}

// params = {log_blowup, log_folding_arity, log_final_degree, folding_pow_bits, deep_pow_bits,
//           num_queries, query_pow_bits}
int orc_prove(const int params[7], int n_airs, const uint64_t* const* dags, const size_t* dag_lens,
              const uint64_t* const* traces, const int* log_heights, const uint64_t* publics, size_t n_publics,
              const uint64_t init_state[12], const uint64_t* pre_observe, size_t n_pre, AuxBuilder cb, void* user,
              uint64_t* fields_out, size_t fields_cap, size_t* n_fields, uint64_t* commits_out, size_t commits_cap,
              size_t* n_commits, uint64_t digest[4], char* err, size_t errcap,
              const uint64_t* const* preprocessed /* per instance (NULL entries allowed), or NULL */) {
  try {
    ProverInput in;
    in.params = make_params(params);
    for (int i = 0; i < n_airs; i++) {
      in.airs.push_back(Air::parse(dags[i], dag_lens[i]));
      in.traces.push_back(traces[i]);
      in.log_heights.push_back(log_heights[i]);
      in.preprocessed.push_back(preprocessed ? preprocessed[i] : nullptr);
    }
    in.publics.assign(publics, publics + n_publics);
    in.challenger = make_challenger(init_state, pre_observe, n_pre);
    in.aux_builder = cb;
    in.aux_user = user;
    Proof p = prove(in);
    if (p.fields.size() > fields_cap || p.commitments.size() > commits_cap) {
      set_err(err, errcap, "output buffers too small");
      return 2;
    }
    memcpy(fields_out, p.fields.data(), p.fields.size() * 8);
    memcpy(commits_out, p.commitments.data(), p.commitments.size() * 32);
    *n_fields = p.fields.size();
    *n_commits = p.commitments.size();
    memcpy(digest, p.digest.data(), 32);
    return 0;
  } catch (const std::exception& e) {
    set_err(err, errcap, e.what());
    return 1;
  }
}

int orc_verify(const int params[7], int n_airs, const uint64_t* const* dags, const size_t* dag_lens, const int* log_heights,
               const uint64_t* publics, size_t n_publics, const uint64_t init_state[12], const uint64_t* pre_observe,
               size_t n_pre, const uint64_t* fields, size_t n_fields, const uint64_t* commits, size_t n_commits,
               uint64_t digest[4], char* err, size_t errcap, const uint64_t* preprocessed_root /* [4] or NULL */,
               ExternalAssertions external /* or NULL */, void* external_user) {
  try {
    VerifierInput in;
    in.external = external;
    in.external_user = external_user;
    in.params = make_params(params);
    Proof p;
    for (int i = 0; i < n_airs; i++) {
      in.airs.push_back(Air::parse(dags[i], dag_lens[i]));
      p.log_trace_heights.push_back((uint8_t)log_heights[i]);
    }
    in.publics.assign(publics, publics + n_publics);
    in.challenger = make_challenger(init_state, pre_observe, n_pre);
    if (preprocessed_root) {
      in.has_preprocessed = true;
      memcpy(in.preprocessed_root.data(), preprocessed_root, 32);
    }
    p.fields.assign(fields, fields + n_fields);
    p.commitments.resize(n_commits);
    memcpy(p.commitments.data(), commits, n_commits * 32);
    Digest d = verify(in, p);
    memcpy(digest, d.data(), 32);
    return 0;
  } catch (const std::exception& e) {
    set_err(err, errcap, e.what());
    return 1;
  }
}

// fold_evals (crates/lifted-stark/src/pcs/fri/fold/mod.rs:70-84): one bit-reversed row of 2^log_arity EF values -> g(s^arity)
int orc_fri_fold_row(const uint64_t* y_flat, int log_arity, uint64_t s_inv, const uint64_t beta[2], uint64_t out[2]) {
  try {
    std::vector<E2> y((size_t)1 << log_arity);
    for (size_t i = 0; i < y.size(); i++) y[i] = E2{y_flat[2 * i] % P, y_flat[2 * i + 1] % P};
    const E2 r = fri_fold_row(y.data(), log_arity, s_inv % P, E2{beta[0] % P, beta[1] % P});
    out[0] = r.c0;
    out[1] = r.c1;
    return 0;
  } catch (const std::exception&) {
    return 1;
  }
}

// ---- the duplex challenger as an object (tests drive the staged device session with it) ----------
void* orc_ch_new(const uint64_t state[12]) {
  Challenger* c = new Challenger();
  uint64_t st[12];
  for (int i = 0; i < 12; i++) st[i] = state[i] % P;
  c->init(st);
  return c;
}
void orc_ch_free(void* h) { delete (Challenger*)h; }
void orc_ch_observe(void* h, const uint64_t* x, size_t n) {
  for (size_t i = 0; i < n; i++) ((Challenger*)h)->observe(x[i] % P);
}
uint64_t orc_ch_sample(void* h) { return ((Challenger*)h)->sample(); }
uint64_t orc_ch_sample_bits(void* h, int bits) { return ((Challenger*)h)->sample_bits(bits); }
uint64_t orc_ch_grind(void* h, int bits) { return ((Challenger*)h)->grind(bits); }
int orc_ch_check_witness(void* h, int bits, uint64_t w) { return ((Challenger*)h)->check_witness(bits, w) ? 1 : 0; }
size_t orc_ch_state(void* h, uint64_t state[12], uint64_t pending[8]) {
  Challenger* c = (Challenger*)h;
  for (int i = 0; i < 12; i++) state[i] = c->st[i];
  for (size_t i = 0; i < c->in.size(); i++) pending[i] = c->in[i];
  return c->in.size();
}
void orc_ch_finalize(void* h, uint64_t digest[4]) {
  Digest d = ((Challenger*)h)->finalize();
  memcpy(digest, d.data(), 32);
}

// ---- LogUp aux trace (oracle/lookup.hpp) ----
int orc_lookup_build_aux(const uint64_t* blob, size_t n_words, const uint64_t* main_rowmajor, int log_n, const uint64_t* randomness,
                         size_t n_rand, uint64_t* aux_rowmajor, uint64_t acc_final[2], char* err, size_t errcap,
                         const uint64_t* preprocessed_rowmajor /* or NULL */) {
  try {
    Lookup lk = Lookup::parse(blob, n_words);
    std::vector<E2> rnd;
    for (size_t i = 0; i < n_rand; i++) rnd.push_back(E2{randomness[2 * i] % P, randomness[2 * i + 1] % P});
    rnd.resize(std::max(rnd.size(), lk.dag.num_randomness), e2(0));
    E2 f = lookup_build_aux(lk, main_rowmajor, preprocessed_rowmajor, (size_t)1 << log_n, rnd.data(), aux_rowmajor);
    acc_final[0] = f.c0;
    acc_final[1] = f.c1;
    return 0;
  } catch (const std::exception& e) {
    set_err(err, errcap, e.what());
    return 1;
  }
}

// ---- row-by-row constraint check on concrete values (crates/lifted-stark/src/debug.rs:147-232 check_single_trace) ----
// Every constraint of the DAG is evaluated on every row window (next row wraps around) with is_first / is_last /
// is_transition as 0/1 values and periodic columns indexed by row % period.  Returns the number of (row, constraint) pairs
// that are non-zero; the first one is written to first_bad = {row, constraint index}.  -1 on malformed input.
long orc_check_constraints(const uint64_t* blob, size_t n_words, const uint64_t* main_rowmajor, int log_n,
                           const uint64_t* aux_rowmajor /* [n][2*aux_width] or NULL */, const uint64_t* aux_values,
                           const uint64_t* publics, const uint64_t* randomness, const uint64_t* preprocessed_rowmajor,
                           uint64_t first_bad[2], char* err, size_t errcap) {
  try {
    Air air = Air::parse(blob, n_words);
    size_t n = (size_t)1 << log_n, w = air.main_width, aw = air.aux_width, pw = air.preprocessed_width;
    if (aw && !aux_rowmajor) throw std::runtime_error("aux trace missing");
    if (pw && !preprocessed_rowmajor) throw std::runtime_error("preprocessed trace missing");
    std::vector<E2> rnd(air.num_randomness), av(air.num_aux_values);
    for (size_t i = 0; i < rnd.size(); i++) rnd[i] = E2{randomness[2 * i] % P, randomness[2 * i + 1] % P};
    for (size_t i = 0; i < av.size(); i++) av[i] = E2{aux_values[2 * i] % P, aux_values[2 * i + 1] % P};
    long bad = 0;
    std::vector<E2> scratch, ac(aw), an(aw), per(air.periodic.size());
    for (size_t r = 0; r < n; r++) {
      size_t rn = (r + 1) % n;
      for (size_t c = 0; c < aw; c++) {
        ac[c] = E2{aux_rowmajor[(r * aw + c) * 2], aux_rowmajor[(r * aw + c) * 2 + 1]};
        an[c] = E2{aux_rowmajor[(rn * aw + c) * 2], aux_rowmajor[(rn * aw + c) * 2 + 1]};
      }
      for (size_t c = 0; c < per.size(); c++) per[c] = e2(air.periodic[c][r % air.periodic[c].size()] % P);
      EvalEnv e;
      e.main_cur = main_rowmajor + r * w; e.main_next = main_rowmajor + rn * w;
      e.prep_cur = pw ? preprocessed_rowmajor + r * pw : nullptr; e.prep_next = pw ? preprocessed_rowmajor + rn * pw : nullptr;
      e.aux_cur = ac.data(); e.aux_next = an.data();
      e.publics = publics; e.periodic = per.data();
      e.is_first = e2(r == 0); e.is_last = e2(r == n - 1); e.is_transition = e2(r != n - 1);
      e.randomness = rnd.data(); e.aux_values = av.data();
      dag_fold(air, e, nullptr, e2(0), scratch);  // fills scratch with every node's value
      for (size_t k = 0; k < air.constraints.size(); k++) {
        E2 v = scratch[air.constraints[k]];
        if (v.c0 != 0 || v.c1 != 0) {
          if (bad == 0) { first_bad[0] = r; first_bad[1] = k; }
          bad++;
        }
      }
    }
    return bad;
  } catch (const std::exception& e) {
    set_err(err, errcap, e.what());
    return -1;
  }
}

}  // extern "C"
