// Host-only verifier of the lifted STARK: accepts or rejects the proofs mh_prove
// produces without a GPU, a ctx or a Rust toolchain.  Restated from the reference's verifier:
//   crates/lifted-stark/src/verifier/mod.rs      protocol flow, constraint identity, quotient reconstruction
//   crates/lifted-stark/src/pcs/verifier.rs      aligned openings of the three trace trees
//   crates/lifted-stark/src/pcs/deep/verifier.rs reduced openings -> DEEP quotient values at the query points
//   crates/lifted-stark/src/pcs/fri/verifier.rs  per-round openings, folding consistency, final polynomial
//   crates/lifted-stark/src/lmcs/config.rs:172-211 batch opening against a root
// Fiat-Shamir is challenger.hpp (the same HostChallenger mh_prove uses); AIRs are the constraint-DAG blobs of
// include/midenhip.h, evaluated at the out-of-domain point over the extension field.
#include "../../include/midenhip.h"
#include "air.hpp"
#include "challenger.hpp"
#include "gl.cuh"
#include <algorithm>
#include <cstring>
#include <numeric>
#include <string>

// p2_host_simd.cpp: eight Poseidon2 permutations per AVX-512 call (host only; the tree tops of the prover use them too)
bool p2_host_simd_available();
void p2_host_compress8(const uint64_t* pairs, int n, uint64_t* out);
void p2_host_permute8(uint64_t* states);

namespace {

struct Reject : MhError {
  explicit Reject(const std::string& m) : MhError(MH_ERR_INVALID, "proof rejected: " + m) {}
};

// The two streams of a proof, consumed front to back (crates/stark-transcript/src/verifier.rs).
struct Reader {
  HostChallenger ch;
  const u64* f;
  size_t nf, pf = 0;
  const u64* c;
  size_t nc, pc = 0;  // commitments counted in digests
  u64 hint_field() {
    if (pf >= nf) throw Reject("transcript ran out of field elements");
    const u64 v = f[pf++];
    if (v >= GL_P) throw Reject("non-canonical field element");
    return v;
  }
  Digest4 hint_digest() {
    if (pc >= nc) throw Reject("transcript ran out of commitments");
    Digest4 d;
    for (int i = 0; i < 4; i++) {
      d[i] = c[4 * pc + i];
      if (!ch.bytes() && d[i] >= GL_P) throw Reject("non-canonical digest element");  // a byte digest's words are not felts
    }
    pc++;
    return d;
  }
  u64 recv_field() {
    const u64 v = hint_field();
    ch.observe(v);
    return v;
  }
  e2 recv_ef() {
    const u64 a = recv_field();
    return e2{a, recv_field()};
  }
  Digest4 recv_digest() {
    Digest4 d = hint_digest();
    ch.observe_digest(d.data());
    return d;
  }
  void check_pow(int bits) {  // the witness is a transcript field (stark-transcript/src/verifier.rs grind check)
    const u64 w = hint_field();
    if (!ch.check_witness(bits, w)) throw Reject("proof-of-work witness");
  }
};

// the configuration being verified (MH_LMCS_*): set per call, per thread
thread_local int t_hash = 0;
// aligned_len(w, lmcs.alignment()) (util/align.rs:7-13, proof.rs:268): the sponge's rate -- 8, or 17 for Keccak -- and 1 for the
// chaining hasher of the Blake3 LMCS.  (The name dates from the Poseidon2 configuration.)
size_t align8(size_t w) {
  const size_t a = t_hash == MH_LMCS_BLAKE3 ? 1 : (t_hash == MH_LMCS_KECCAK ? 17 : 8);
  return (w + a - 1) / a * a;
}

// Overwrite-mode sponge over whole (already aligned / short) rows: crates/stateful-hasher/src/field_sponge.rs:41-59.
void absorb(u64 st[12], const u64* v, size_t n) {
  for (size_t off = 0; off < n; off += 8) {
    const size_t k = std::min<size_t>(8, n - off);
    for (size_t i = 0; i < k; i++) st[i] = v[off + i];
    for (size_t i = k; i < 8; i++) st[i] = 0;
    alg_permute(t_hash, st);
  }
}
Digest4 compress2(const Digest4& l, const Digest4& r) {
  if (t_hash == MH_LMCS_BLAKE3) {  // blake3(left || right)
    uint8_t msg[64], d[32];
    memcpy(msg, l.data(), 32);
    memcpy(msg + 32, r.data(), 32);
    b3::hash_bytes(msg, 64, d);
    Digest4 o;
    memcpy(o.data(), d, 32);
    return o;
  }
  if (t_hash == MH_LMCS_KECCAK) {
    Digest4 o;
    kk::compress_pair(l.data(), r.data(), o.data());
    return o;
  }
  u64 st[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
  alg_permute(t_hash, st);
  return Digest4{st[0], st[1], st[2], st[3]};
}
// leaf digest of one opened index: the rows of the tree's matrices, each already of its aligned width
Digest4 leaf_digest(const u64* row, const std::vector<size_t>& widths) {
  size_t off = 0;
  if (t_hash == MH_LMCS_BLAKE3) {  // chaining hasher: state := blake3(state || row bytes), zero state first (chaining.rs:32-50)
    Digest4 st{0, 0, 0, 0};
    for (size_t w : widths) {
      std::vector<uint8_t> msg(32 + 8 * w);
      memcpy(msg.data(), st.data(), 32);
      if (w) memcpy(msg.data() + 32, row + off, 8 * w);
      uint8_t d[32];
      b3::hash_bytes(msg.data(), msg.size(), d);
      memcpy(st.data(), d, 32);
      off += w;
    }
    return st;
  }
  if (t_hash == MH_LMCS_KECCAK) {
    u64 st[25] = {0};
    for (size_t w : widths) {
      kk::lmcs_absorb(st, row + off, w);
      off += w;
    }
    return Digest4{st[0], st[1], st[2], st[3]};
  }
  u64 st[12] = {0};
  for (size_t w : widths) {
    absorb(st, row + off, w);
    off += w;
  }
  return Digest4{st[0], st[1], st[2], st[3]};
}

// The Poseidon2 configuration, eight hashes per AVX-512 permutation (p2_host_simd.cpp: the tree tops of the prover use the same code): the leaves of a
// batch opening absorb rows of the same shape in lockstep, the compressions of one tree level are independent of each other.  A Miden proof under
// Poseidon2 verifies in ~1 ms instead of 4.6 (tools/bench_verify.py); the other configurations and CPUs without AVX-512 take the scalar functions above.
bool simd_hashing() { return t_hash == MH_LMCS_POSEIDON2 && p2_host_simd_available(); }

std::vector<Digest4> leaf_digests(const std::vector<std::vector<u64>>& rows, const std::vector<size_t>& widths) {
  std::vector<Digest4> out(rows.size());
  if (!simd_hashing()) {
    for (size_t q = 0; q < rows.size(); q++) out[q] = leaf_digest(rows[q].data(), widths);
    return out;
  }
  for (size_t q0 = 0; q0 < rows.size(); q0 += 8) {
    const size_t n = std::min<size_t>(8, rows.size() - q0);
    u64 st[8][12] = {{0}};
    size_t off = 0;
    for (size_t w : widths) {  // absorb(), eight leaves abreast
      for (size_t o = 0; o < w; o += 8) {
        const size_t k = std::min<size_t>(8, w - o);
        for (size_t j = 0; j < n; j++) {
          for (size_t i = 0; i < k; i++) st[j][i] = gl_canon(rows[q0 + j][off + o + i]);
          for (size_t i = k; i < 8; i++) st[j][i] = 0;
        }
        p2_host_permute8(&st[0][0]);
      }
      off += w;
    }
    for (size_t j = 0; j < n; j++) out[q0 + j] = Digest4{st[j][0], st[j][1], st[j][2], st[j][3]};
  }
  return out;
}
// out[i] = compress2(pairs[i].first, pairs[i].second)
std::vector<Digest4> compress_many(const std::vector<std::pair<Digest4, Digest4>>& pairs) {
  std::vector<Digest4> out(pairs.size());
  if (!simd_hashing()) {
    for (size_t i = 0; i < pairs.size(); i++) out[i] = compress2(pairs[i].first, pairs[i].second);
    return out;
  }
  for (size_t i0 = 0; i0 < pairs.size(); i0 += 8) {
    const int n = (int)std::min<size_t>(8, pairs.size() - i0);
    u64 in[64], o[32];
    for (int j = 0; j < n; j++)
      for (int k = 0; k < 4; k++) {
        in[8 * j + k] = gl_canon(pairs[i0 + j].first[k]);
        in[8 * j + 4 + k] = gl_canon(pairs[i0 + j].second[k]);
      }
    p2_host_compress8(in, n, o);
    for (int j = 0; j < n; j++) out[i0 + j] = Digest4{o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]};
  }
  return out;
}

// lmcs/config.rs:172-211: per sorted unique index the opened rows (one per matrix, already padded), then the
// siblings that cannot be derived, level by level, left to right.  Returns the rows, concatenated per index.
std::vector<std::vector<u64>> open_batch(Reader& rd, const Digest4& root, const std::vector<size_t>& widths,
                                         const std::vector<size_t>& idx, int depth) {
  size_t total = 0;
  for (size_t w : widths) total += w;
  std::vector<std::vector<u64>> rows(idx.size(), std::vector<u64>(total));
  std::vector<std::pair<size_t, Digest4>> level;
  for (size_t q = 0; q < idx.size(); q++)
    for (auto& x : rows[q]) x = rd.hint_field();
  {
    const std::vector<Digest4> leaves = leaf_digests(rows, widths);
    for (size_t q = 0; q < idx.size(); q++) level.push_back({idx[q], leaves[q]});
  }
  for (int d = depth; d > 0; d--) {
    std::vector<size_t> parents;
    std::vector<std::pair<Digest4, Digest4>> pairs;  // (left, right) of every parent of this level, in stream order
    for (size_t i = 0; i < level.size();) {
      const size_t node = level[i].first;
      Digest4 sib;
      size_t used = 1;
      if (i + 1 < level.size() && level[i + 1].first == (node ^ 1)) {
        sib = level[i + 1].second;
        used = 2;
      } else {
        sib = rd.hint_digest();
      }
      parents.push_back(node >> 1);
      pairs.push_back((node & 1) ? std::make_pair(sib, level[i].second) : std::make_pair(level[i].second, sib));
      i += used;
    }
    const std::vector<Digest4> dig = compress_many(pairs);
    std::vector<std::pair<size_t, Digest4>> up;
    for (size_t i = 0; i < parents.size(); i++) up.push_back({parents[i], dig[i]});
    level.swap(up);
  }
  if (level.size() != 1 || level[0].second != root) throw Reject("Merkle root mismatch");
  return rows;
}

int fri_rounds(const mh_pcs_params& p, int log_lde) {
  const int log_max_final = p.log_final_degree + p.log_blowup;
  const int steps = log_lde > log_max_final ? log_lde - log_max_final : 0;
  return (steps + p.log_folding_arity - 1) / p.log_folding_arity;
}

// One folding step on an opened row (fri/fold/mod.rs:113-180, arity2.rs, arity4.rs:46-121, arity8.rs:35-138): the row holds the evaluations on
// the coset s * <w_arity> in bit-reversed order; the result is the interpolant's value at beta.
e2 fold_row(const e2* y, int log_arity, u64 s_inv, e2 beta) {
  const e2 x = e2_mulf(beta, s_inv);
  if (log_arity == 1) {
    const e2 r = e2_add(e2_add(y[0], y[1]), e2_mul(e2_sub(y[0], y[1]), x));
    return e2_mulf(r, gl_inv(2));
  }
  if (log_arity == 3) {
    // fold/arity8.rs as a plain interpolation: c_k = (1/8) sum_m f(s w^m) w^(-mk), result sum_k c_k x^k.
    // The row is bit-reversed: position p holds f(s w^bitrev(p)).
    const u64 wi = gl_inv(gl_two_adic_generator(3));
    e2 acc = e2_make(0);
    for (int k = 7; k >= 0; k--) {
      e2 ck = e2_make(0);
      for (u32 p = 0; p < 8; p++) ck = e2_add(ck, e2_mulf(y[p], gl_pow(wi, (u64)bitrev32(p, 3) * (u64)k)));
      acc = e2_add(e2_mul(acc, x), ck);
    }
    return e2_mulf(acc, gl_inv(8));
  }
  const e2 y0 = y[0], y2 = y[1], y1 = y[2], y3 = y[3];
  const u64 w4 = gl_two_adic_generator(2);
  const e2 s02 = e2_add(y0, y2), d02 = e2_sub(y0, y2), s13 = e2_add(y1, y3), d31 = e2_mulf(e2_sub(y3, y1), w4);
  const e2 c0 = e2_add(s02, s13), c1 = e2_add(d02, d31), c2 = e2_sub(s02, s13), c3 = e2_sub(d02, d31);
  const e2 x2 = e2_mul(x, x), x3 = e2_mul(x2, x);
  return e2_mulf(e2_add(e2_add(c0, e2_mul(c1, x)), e2_add(e2_mul(c2, x2), e2_mul(c3, x3))), gl_inv(4));
}

// Everything an AIR's constraints can read at the out-of-domain point.
struct PointEnv {
  const e2 *main_cur, *main_next, *prep_cur, *prep_next, *aux_cur, *aux_next, *periodic, *randomness, *aux_values;
  const u64* publics;
  e2 is_first, is_last, is_transition;
};
// sum_k alpha^(K-1-k) C_k at the point (constraints/folder.rs:88-105 read backwards = Horner).
e2 fold_constraints(const DagIR& ir, const PointEnv& e, e2 alpha) {
  std::vector<e2> v(ir.nodes.size(), e2_make(0));
  for (size_t i = 0; i < ir.nodes.size(); i++) {
    if (!ir.live[i]) continue;
    const DagNode& nd = ir.nodes[i];
    switch (nd.op) {
      case DOP_CONST: v[i] = e2_make(nd.c); break;
      case DOP_MAIN: v[i] = (nd.b ? e.main_next : e.main_cur)[nd.a]; break;
      case DOP_AUX: v[i] = (nd.b ? e.aux_next : e.aux_cur)[nd.a]; break;
      case DOP_PREP: v[i] = (nd.b ? e.prep_next : e.prep_cur)[nd.a]; break;
      case DOP_PUBLIC: v[i] = e2_make(gl_canon(e.publics[nd.a])); break;
      case DOP_PERIODIC: v[i] = e.periodic[nd.a]; break;
      case DOP_IS_FIRST: v[i] = e.is_first; break;
      case DOP_IS_LAST: v[i] = e.is_last; break;
      case DOP_IS_TRANSITION: v[i] = e.is_transition; break;
      case DOP_RANDOMNESS: v[i] = e.randomness[nd.a]; break;
      case DOP_AUX_VALUE: v[i] = e.aux_values[nd.a]; break;
      case DOP_ADD: v[i] = e2_add(v[nd.a], v[nd.b]); break;
      case DOP_SUB: v[i] = e2_sub(v[nd.a], v[nd.b]); break;
      case DOP_MUL: v[i] = e2_mul(v[nd.a], v[nd.b]); break;
      default: v[i] = e2_neg(v[nd.a]); break;
    }
  }
  e2 acc = e2_make(0);
  for (uint32_t c : ir.cons) acc = e2_add(e2_mul(acc, alpha), v[c]);
  return acc;
}
// Value at y of the polynomial of degree < P that takes the column's values on the subgroup of order P.
e2 periodic_at(const std::vector<u64>& col, e2 y) {
  const size_t P = col.size();
  int logp = 0;
  while (((size_t)1 << logp) < P) logp++;
  const u64 w_inv = gl_inv(gl_two_adic_generator(logp)), p_inv = gl_inv((u64)P);
  e2 acc = e2_make(0);
  for (size_t k = P; k-- > 0;) {  // coefficient k = (1/P) sum_r col[r] w^(-k r), Horner from the top
    u64 s = 0, x = 1;
    const u64 wk = gl_pow(w_inv, k);
    for (size_t r = 0; r < P; r++) {
      s = gl_add(s, gl_mul(col[r] % GL_P, x));
      x = gl_mul(x, wk);
    }
    acc = e2_add(e2_mul(acc, y), e2_make(gl_mul(s, p_inv)));
  }
  return acc;
}

void verify_impl(const mh_pcs_params& pp, const std::vector<DagIR>& airs, const std::vector<int>& lhs, const std::vector<u64>& publics,
                 const u64* prep_root, Reader& rd, mh_external_assertions external, void* external_user, u64 digest[4]) {
  const size_t n_airs = airs.size();
  const int lb = pp.log_blowup, la = pp.log_folding_arity;
  if (lb < 1 || lb > 8 || la < 1 || la > 3 || pp.num_queries < 1) throw Reject("unsupported PCS parameters");
  // sample_bits works on the low 32 bits of a sample (random_coin.masm sample_bits): more PoW bits cannot be checked
  for (int b : {pp.deep_pow_bits, pp.folding_pow_bits, pp.query_pow_bits})
    if (b < 0 || b > 32) throw Reject("proof-of-work bits must be in 0..32");
  if (pp.log_final_degree < 0 || pp.log_final_degree > 32) throw Reject("log_final_degree must be in 0..32");
  // PcsParams::new (pcs/params.rs:62-69): FinalDegreeUnreachable -- the reference's verifier can only be built from valid parameters
  if (pp.log_final_degree + lb < la - 1) throw Reject("final degree unreachable by fixed-arity folding");
  for (size_t i = 0; i < n_airs; i++) {
    if (lhs[i] < 1) throw Reject("trace too small");
    size_t pmax = 0;
    for (auto& c : airs[i].periodic) pmax = std::max(pmax, c.size());
    if (((size_t)1 << lhs[i]) < pmax) throw Reject("trace shorter than a periodic column");
    if (airs[i].num_public != publics.size()) throw Reject("public value count");
  }
  // proof order: ascending height, ties by instance index (order.rs)
  std::vector<int> order(n_airs);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lhs[a] < lhs[b]; });
  const int log_n = lhs[order.back()], L = log_n + lb;
  if (L > 32) throw Reject("LDE order exceeds the field's two-adicity");
  int logD = 0;
  size_t max_rand = 0;
  for (auto& a : airs) {
    logD = std::max(logD, a.log_quotient_degree);
    max_rand = std::max(max_rand, a.num_randomness);
  }
  if (logD > lb) throw Reject("constraint degree too high for the blowup");
  const size_t D = (size_t)1 << logD;
  const u64 g = gl_lde_shift(L), g_inv = gl_inv(g);

  rd.ch.observe((u64)n_airs);
  for (int lh : lhs) rd.ch.observe((u64)lh);
  // ---- commit phase replay (verifier/mod.rs) ----
  const Digest4 main_root = rd.recv_digest();
  std::vector<e2> randomness;
  for (size_t i = 0; i < max_rand; i++) randomness.push_back(rd.ch.sample_ef());
  const Digest4 aux_root = rd.recv_digest();
  std::vector<std::vector<e2>> aux_values(n_airs);  // proof order
  for (size_t j = 0; j < n_airs; j++)
    for (size_t k = 0; k < airs[order[j]].num_aux_values; k++) aux_values[j].push_back(rd.recv_ef());
  const e2 alpha = rd.ch.sample_ef(), beta = rd.ch.sample_ef();
  const Digest4 quot_root = rd.recv_digest();
  e2 z;
  for (;;) {  // domain.rs:539-553
    z = rd.ch.sample_ef();
    if (e2_is_zero(z)) continue;
    if (e2_eq(e2_exp_pow2(z, log_n), e2_make(1))) continue;
    if (e2_eq(e2_exp_pow2(e2_mulf(z, g_inv), L), e2_make(1))) continue;
    break;
  }
  const e2 zs[2] = {z, e2_mulf(z, gl_two_adic_generator(log_n))};
  // commitment groups, aligned widths, tree depths: [preprocessed?, main, aux, quotient] (proof.rs:326-375)
  std::vector<std::vector<size_t>> widths;
  std::vector<Digest4> roots;
  std::vector<int> depths;
  bool any_prep = false;
  for (auto& a : airs) any_prep |= a.preprocessed_width > 0;
  if (any_prep != (prep_root != nullptr)) throw Reject("preprocessed commitment presence does not match the AIRs");
  if (any_prep) {
    widths.emplace_back();
    int dp = 0;
    for (size_t j = 0; j < n_airs; j++)
      if (airs[order[j]].preprocessed_width) {
        widths.back().push_back(align8(airs[order[j]].preprocessed_width));
        dp = std::max(dp, lhs[order[j]] + lb);
      }
    roots.push_back(Digest4{prep_root[0], prep_root[1], prep_root[2], prep_root[3]});
    depths.push_back(dp);
  }
  const size_t g_main = widths.size();
  widths.resize(g_main + 3);
  for (size_t j = 0; j < n_airs; j++) widths[g_main].push_back(align8(airs[order[j]].main_width));
  for (size_t j = 0; j < n_airs; j++) widths[g_main + 1].push_back(align8(2 * airs[order[j]].aux_width));
  widths[g_main + 2].push_back(align8(2 * D));
  roots.push_back(main_root); roots.push_back(aux_root); roots.push_back(quot_root);
  for (int k = 0; k < 3; k++) depths.push_back(L);
  size_t W = 0;
  for (auto& ws : widths)
    for (size_t w : ws) W += w;
  std::vector<e2> ev[2];
  for (int k = 0; k < 2; k++)
    for (size_t i = 0; i < W; i++) ev[k].push_back(rd.recv_ef());
  rd.check_pow(pp.deep_pow_bits);
  const e2 alpha_d = rd.ch.sample_ef(), beta_d = rd.ch.sample_ef();
  e2 fred[2];
  for (int k = 0; k < 2; k++) {
    e2 a = e2_make(0);
    for (size_t i = 0; i < W; i++) a = e2_add(e2_mul(a, alpha_d), ev[k][i]);
    fred[k] = a;
  }
  const int rounds = fri_rounds(pp, L);
  std::vector<Digest4> fri_roots;
  std::vector<e2> fri_betas;
  for (int r = 0; r < rounds; r++) {
    fri_roots.push_back(rd.recv_digest());
    rd.check_pow(pp.folding_pow_bits);
    fri_betas.push_back(rd.ch.sample_ef());
  }
  const size_t fpd = (size_t)1 << std::max(0, L - rounds * la - lb);
  std::vector<e2> final_poly;  // descending degree
  for (size_t i = 0; i < fpd; i++) final_poly.push_back(rd.recv_ef());
  rd.check_pow(pp.query_pow_bits);
  std::vector<size_t> idx;
  for (int i = 0; i < pp.num_queries; i++) idx.push_back(rd.ch.sample_bits(L));
  std::sort(idx.begin(), idx.end());
  idx.erase(std::unique(idx.begin(), idx.end()), idx.end());

  // ---- query phase: trace openings -> DEEP quotient values (deep/verifier.rs) ----
  std::vector<e2> reduced(idx.size(), e2_make(0));
  for (size_t t = 0; t < widths.size(); t++) {
    // a tree shorter than the max domain is opened at the indices' low bits (lmcs/tree_indices.rs:72-84)
    const size_t mask = ((size_t)1 << depths[t]) - 1;
    std::vector<size_t> tidx;
    for (size_t i : idx) tidx.push_back(i & mask);
    std::sort(tidx.begin(), tidx.end());
    tidx.erase(std::unique(tidx.begin(), tidx.end()), tidx.end());
    auto rows = open_batch(rd, roots[t], widths[t], tidx, depths[t]);
    for (size_t q = 0; q < idx.size(); q++) {
      const size_t k = std::lower_bound(tidx.begin(), tidx.end(), idx[q] & mask) - tidx.begin();
      for (u64 v : rows[k]) reduced[q] = e2_add(e2_mul(reduced[q], alpha_d), e2_make(v));
    }
  }
  const u64 wK = gl_two_adic_generator(L);
  std::vector<std::pair<size_t, e2>> cur;  // (index in the current FRI domain, value)
  for (size_t q = 0; q < idx.size(); q++) {
    const e2 x = e2_make(gl_mul(g, gl_pow(wK, idx[q])));
    e2 acc = e2_make(0), bp = e2_make(1);
    for (int k = 0; k < 2; k++) {
      const e2 den = e2_sub(zs[k], x);
      if (e2_is_zero(den)) throw Reject("out-of-domain point on the LDE coset");
      acc = e2_add(acc, e2_mul(e2_mul(bp, e2_sub(fred[k], reduced[q])), e2_inv(den)));
      bp = e2_mul(bp, beta_d);
    }
    cur.push_back({idx[q], acc});
  }
  // ---- FRI (fri/verifier.rs): each round's row must contain the running value and folds to the next one ----
  int logn = L;
  u64 gen_inv = gl_inv(gl_two_adic_generator(L));
  const size_t arity = (size_t)1 << la;
  for (int r = 0; r < rounds; r++) {
    const int logf = logn - la;
    const size_t mask = ((size_t)1 << logf) - 1;
    std::vector<size_t> ridx;
    for (auto& kv : cur) ridx.push_back(kv.first & mask);
    std::sort(ridx.begin(), ridx.end());
    ridx.erase(std::unique(ridx.begin(), ridx.end()), ridx.end());
    auto rows = open_batch(rd, fri_roots[r], {arity * 2}, ridx, logf);
    std::vector<std::pair<size_t, e2>> next;
    for (auto& kv : cur) {
      const size_t row = kv.first & mask;
      const size_t pos = bitrev32((u32)(kv.first >> logf), la);
      const size_t q = std::lower_bound(ridx.begin(), ridx.end(), row) - ridx.begin();
      e2 y[8];
      for (size_t k = 0; k < arity; k++) y[k] = e2{rows[q][2 * k], rows[q][2 * k + 1]};
      if (!e2_eq(y[pos], kv.second)) throw Reject("FRI round " + std::to_string(r) + ": opened row disagrees with the folded value");
      const e2 folded = fold_row(y, la, gl_pow(gen_inv, row), fri_betas[r]);
      if (next.empty() || next.back().first != row) next.push_back({row, folded});
      else if (!e2_eq(next.back().second, folded)) throw Reject("FRI: two queries fold to different values");
    }
    std::sort(next.begin(), next.end(), [](auto& a, auto& b) { return a.first < b.first; });
    next.erase(std::unique(next.begin(), next.end(), [](auto& a, auto& b) { return a.first == b.first; }), next.end());
    cur.swap(next);
    logn = logf;
    gen_inv = gl_exp_pow2(gen_inv, la);
  }
  {
    const u64 gen = gl_two_adic_generator(logn);
    for (auto& kv : cur) {
      const u64 x = gl_pow(gen, kv.first);
      e2 acc = e2_make(0);
      for (e2 c : final_poly) acc = e2_add(e2_mulf(acc, x), c);
      if (!e2_eq(acc, kv.second)) throw Reject("FRI: final polynomial mismatch");
    }
  }
  // ---- constraint identity at z (verifier/mod.rs): sum over AIRs (beta-folded) == Q(z) * Z_H(z) ----
  e2 accumulated = e2_make(0);
  size_t off_prep = 0, off_main = 0;
  if (any_prep)
    for (size_t w : widths[0]) off_main += w;
  size_t off_aux = off_main;
  for (size_t j = 0; j < n_airs; j++) off_aux += widths[g_main][j];
  size_t off_quot = off_aux;
  for (size_t j = 0; j < n_airs; j++) off_quot += widths[g_main + 1][j];
  const e2 X = e2{0, 1};  // the extension's generator: an EF column is f0 + X * f1 of its two base columns
  for (size_t j = 0; j < n_airs; j++) {
    const DagIR& air = airs[order[j]];
    const int lh = lhs[order[j]];
    std::vector<e2> mc(air.main_width), mn(air.main_width), ac(air.aux_width), an(air.aux_width), per;
    std::vector<e2> pc(air.preprocessed_width), pn(air.preprocessed_width);
    for (size_t c = 0; c < air.preprocessed_width; c++) {
      pc[c] = ev[0][off_prep + c];
      pn[c] = ev[1][off_prep + c];
    }
    if (air.preprocessed_width) off_prep += align8(air.preprocessed_width);
    for (size_t c = 0; c < air.main_width; c++) {
      mc[c] = ev[0][off_main + c];
      mn[c] = ev[1][off_main + c];
    }
    for (size_t c = 0; c < air.aux_width; c++) {
      ac[c] = e2_add(ev[0][off_aux + 2 * c], e2_mul(ev[0][off_aux + 2 * c + 1], X));
      an[c] = e2_add(ev[1][off_aux + 2 * c], e2_mul(ev[1][off_aux + 2 * c + 1], X));
    }
    off_main += widths[g_main][j];
    off_aux += widths[g_main + 1][j];
    const e2 y = e2_exp_pow2(z, log_n - lh);  // the point on this instance's own domain
    const e2 van = e2_sub(e2_exp_pow2(y, lh), e2_make(1));
    const u64 wh_inv = gl_inv(gl_two_adic_generator(lh));
    for (auto& col : air.periodic) {
      int logp = 0;
      while (((size_t)1 << logp) < col.size()) logp++;
      per.push_back(periodic_at(col, e2_exp_pow2(z, log_n - logp)));
    }
    PointEnv e{};
    e.main_cur = mc.data(); e.main_next = mn.data(); e.aux_cur = ac.data(); e.aux_next = an.data();
    e.prep_cur = pc.data(); e.prep_next = pn.data();
    e.periodic = per.data(); e.randomness = randomness.data(); e.aux_values = aux_values[j].data();
    e.publics = publics.data();
    e.is_first = e2_mul(van, e2_inv(e2_sub(y, e2_make(1))));        // domain.rs:518-531
    e.is_last = e2_mul(van, e2_inv(e2_sub(y, e2_make(wh_inv))));
    e.is_transition = e2_sub(y, e2_make(wh_inv));
    accumulated = e2_add(e2_mul(accumulated, beta), fold_constraints(air, e, alpha));
  }
  // ---- external assertions (verifier/mod.rs:488-501): Statement::eval_external over the challenges, the aux values
  // back in INSTANCE order and the log heights; every assertion value must be zero ----
  if (external) {
    std::vector<std::vector<u64>> flat(n_airs);
    for (size_t j = 0; j < n_airs; j++)
      for (e2 v : aux_values[j]) { flat[order[j]].push_back(v.c0); flat[order[j]].push_back(v.c1); }
    std::vector<const u64*> ptrs(n_airs);
    std::vector<size_t> cnt(n_airs);
    std::vector<uint8_t> lh8(n_airs);
    for (size_t i = 0; i < n_airs; i++) { ptrs[i] = flat[i].data(); cnt[i] = flat[i].size() / 2; lh8[i] = (uint8_t)lhs[i]; }
    std::vector<u64> rflat;
    for (e2 r : randomness) { rflat.push_back(r.c0); rflat.push_back(r.c1); }
    const size_t cap = 256;
    std::vector<u64> out(2 * cap, 0);
    const int k = external(external_user, rflat.data(), randomness.size(), ptrs.data(), cnt.data(), lh8.data(), (int)n_airs, out.data(), cap);
    if (k < 0 || (size_t)k > cap) throw Reject("external assertions could not be evaluated (ReductionError)");
    for (int a = 0; a < k; a++)
      if (gl_canon(out[2 * a]) || gl_canon(out[2 * a + 1])) throw Reject("external assertion " + std::to_string(a) + " failed");
  }
  {  // reconstruct_quotient (domain.rs:773-794): barycentric recombination of the D chunk openings
    const u64 wD = gl_two_adic_generator(logD);
    const e2 u = e2_exp_pow2(e2_mulf(z, g_inv), log_n);
    e2 num = e2_make(0), den = e2_make(0);
    u64 wt = 1;
    for (size_t t = 0; t < D; t++) {
      const e2 chunk = e2_add(ev[0][off_quot + 2 * t], e2_mul(ev[0][off_quot + 2 * t + 1], X));
      const e2 d = e2_sub(u, e2_make(wt));
      if (e2_is_zero(d)) throw Reject("out-of-domain point on a quotient chunk domain");
      const e2 wgt = e2_mulf(e2_inv(d), wt);
      num = e2_add(num, e2_mul(wgt, chunk));
      den = e2_add(den, wgt);
      wt = gl_mul(wt, wD);
    }
    const e2 qz = e2_mul(num, e2_inv(den));
    const e2 van = e2_sub(e2_exp_pow2(z, log_n), e2_make(1));
    if (!e2_eq(accumulated, e2_mul(qz, van))) throw Reject("constraints do not vanish on the trace domain (quotient identity)");
  }
  if (rd.pf != rd.nf || rd.pc != rd.nc) throw Reject("trailing data in the transcript");
  rd.ch.finalize(digest);  // CanFinalizeDigest: "unconditionally applies a final state transition" (stark-transcript/src/prover.rs:31-35)
}

}  // namespace

static int verify_entry(int lmcs, const mh_pcs_params* params, int n_airs, const uint64_t* const* air_blobs, const size_t* air_blob_words,
                        const uint8_t* log_trace_heights, const uint64_t* public_values, size_t n_public_values,
                        const uint64_t challenger_state[12], const uint64_t* pre_observe, size_t n_pre_observe, const uint64_t* fields,
                        size_t n_fields, const uint64_t* commitments, size_t n_commitments, const uint64_t* preprocessed_root,
                        mh_external_assertions external, void* external_user, uint64_t digest[4], char* err, size_t err_cap) {
  auto fail = [&](int code, const char* msg) {
    if (err && err_cap) {
      strncpy(err, msg, err_cap - 1);
      err[err_cap - 1] = 0;
    }
    return code;
  };
  try {
    MH_REQUIRE(params && air_blobs && air_blob_words && log_trace_heights && challenger_state && digest, "null argument");
    MH_REQUIRE(n_airs > 0 && n_airs <= 256, "need between 1 and 256 AIR instances");
    MH_REQUIRE((public_values || !n_public_values) && (pre_observe || !n_pre_observe) && (fields || !n_fields) &&
                   (commitments || !n_commitments),
               "null array");
    std::vector<DagIR> airs;
    std::vector<int> lhs;
    for (int i = 0; i < n_airs; i++) {
      airs.push_back(dag_parse(air_blobs[i], air_blob_words[i]));
      lhs.push_back(log_trace_heights[i]);
    }
    MH_REQUIRE(lmcs >= MH_LMCS_POSEIDON2 && lmcs <= MH_LMCS_RPX, "unknown LMCS hasher id");
    t_hash = lmcs;
    Reader rd;
    rd.ch.hash = lmcs;
    rd.ch.init_from_state(challenger_state);
    for (size_t i = 0; i < n_pre_observe; i++) rd.ch.observe_framing(pre_observe[i]);
    rd.f = fields; rd.nf = n_fields;
    rd.c = commitments; rd.nc = n_commitments;
    u64 proot[4];
    if (preprocessed_root)
      for (int i = 0; i < 4; i++) proot[i] = rd.ch.bytes() ? preprocessed_root[i] : gl_canon(preprocessed_root[i]);  // byte digests are not felts
    verify_impl(*params, airs, lhs, std::vector<u64>(public_values, public_values + n_public_values), preprocessed_root ? proot : nullptr,
                rd, external, external_user, digest);
    if (err && err_cap) err[0] = 0;
    return MH_OK;
  } catch (const MhError& e) {
    return fail(e.code, e.what());
  } catch (const std::exception& e) {
    return fail(MH_ERR_INTERNAL, e.what());
  }
}

extern "C" {
int mh_verify(const mh_pcs_params* params, int n_airs, const uint64_t* const* air_blobs, const size_t* air_blob_words,
              const uint8_t* log_trace_heights, const uint64_t* public_values, size_t n_public_values, const uint64_t challenger_state[12],
              const uint64_t* pre_observe, size_t n_pre_observe, const uint64_t* fields, size_t n_fields, const uint64_t* commitments,
              size_t n_commitments, const uint64_t* preprocessed_root, uint64_t digest[4], char* err, size_t err_cap) {
  return verify_entry(MH_LMCS_POSEIDON2, params, n_airs, air_blobs, air_blob_words, log_trace_heights, public_values, n_public_values, challenger_state,
                      pre_observe, n_pre_observe, fields, n_fields, commitments, n_commitments, preprocessed_root, nullptr, nullptr, digest,
                      err, err_cap);
}
int mh_verify_ex(const mh_pcs_params* params, int n_airs, const uint64_t* const* air_blobs, const size_t* air_blob_words,
                 const uint8_t* log_trace_heights, const uint64_t* public_values, size_t n_public_values,
                 const uint64_t challenger_state[12], const uint64_t* pre_observe, size_t n_pre_observe, const uint64_t* fields,
                 size_t n_fields, const uint64_t* commitments, size_t n_commitments, const uint64_t* preprocessed_root,
                 mh_external_assertions external, void* external_user, uint64_t digest[4], char* err, size_t err_cap) {
  return verify_entry(MH_LMCS_POSEIDON2, params, n_airs, air_blobs, air_blob_words, log_trace_heights, public_values, n_public_values,
                      challenger_state, pre_observe, n_pre_observe, fields, n_fields, commitments, n_commitments, preprocessed_root,
                      external, external_user, digest, err, err_cap);
}
int mh_verify_lmcs(int lmcs, const mh_pcs_params* params, int n_airs, const uint64_t* const* air_blobs, const size_t* air_blob_words,
                   const uint8_t* log_trace_heights, const uint64_t* public_values, size_t n_public_values,
                   const uint64_t challenger_state[12], const uint64_t* pre_observe, size_t n_pre_observe, const uint64_t* fields,
                   size_t n_fields, const uint64_t* commitments, size_t n_commitments, const uint64_t* preprocessed_root,
                   mh_external_assertions external, void* external_user, uint64_t digest[4], char* err, size_t err_cap) {
  return verify_entry(lmcs, params, n_airs, air_blobs, air_blob_words, log_trace_heights, public_values, n_public_values,
                      challenger_state, pre_observe, n_pre_observe, fields, n_fields, commitments, n_commitments, preprocessed_root,
                      external, external_user, digest, err, err_cap);
}
// The cross-AIR assertion of a LogUp statement without boundary corrections: the committed accumulator finals
// (aux value 0 of every instance that has one) sum to zero.
int mh_external_logup_balance(void* user, const uint64_t* randomness, size_t n_randomness, const uint64_t* const* aux_values,
                              const size_t* n_aux_values, const uint8_t* log_trace_heights, int n_airs, uint64_t* assertions_out,
                              size_t cap) {
  (void)user; (void)randomness; (void)n_randomness; (void)log_trace_heights;
  if (!assertions_out || cap < 1 || n_airs < 0 || (n_airs && (!aux_values || !n_aux_values))) return -1;
  u64 s0 = 0, s1 = 0;
  for (int i = 0; i < n_airs; i++)
    if (n_aux_values[i]) {
      s0 = gl_add(s0, gl_canon(aux_values[i][0]));
      s1 = gl_add(s1, gl_canon(aux_values[i][1]));
    }
  assertions_out[0] = s0;
  assertions_out[1] = s1;
  return 1;
}
// `ChipletMultiAir::eval_external` of the precompile prover's session (precompiles-prover/src/session/prove.rs:243-256): the sum of the
// committed sigmas + `fixed_boundary_correction` (:205-216) -- the verifier's consumes of the session's fixed environment
// (session/fixed.rs): one `EcGroup` tuple per VM-owned curve group and one `UintVal` tuple per fixed uint (the three domain bounds,
// secp256k1's two coefficients), each as the inverse of its encoded denominator under `Challenges::new(alpha, beta, MAX_MESSAGE_WIDTH = 18,
// NUM_BUS_IDS = 21)` (logup/mod.rs; relations.rs:52-91): bus_prefix[b] = alpha + beta^18 (b + 1), message = prefix + sum_i beta^i f_i.
// Every AIR of the session commits exactly ONE sigma (`aux_values[i]` is AIR i's exposed permutation values -- exactly one, its
// sigma: session/prove.rs:243-247): any other shape is refused, as csrc/miden.cpp refuses it for the VM statement.
// `mh_external_precompile_session` is the whole correction -- what the reference's verifier checks; the reduced form the smaller
// test statements need (those that leave the uint store out: the EcGroup part only) is a callback of its own name,
// `mh_external_precompile_session_ec_only`, so that no flag passed by mistake weakens the real one.
static int precompile_session_external(bool fixed_uints, const uint64_t* randomness, size_t n_randomness, const uint64_t* const* aux_values,
                                       const size_t* n_aux_values, int n_airs, uint64_t* assertions_out, size_t cap) {
  if (!assertions_out || cap < 1 || n_airs < 0 || n_randomness < 2 || !randomness || (n_airs && (!aux_values || !n_aux_values))) return -1;
  for (int i = 0; i < n_airs; i++)
    if (n_aux_values[i] != 1 || !aux_values[i]) return -1;
  static const int MAX_MESSAGE_WIDTH = 18, BUS_UINT_VAL = 10, BUS_EC_GROUP = 14;
  static const u64 U256_BOUND_PTR = 1, K1_BASE_BOUND_PTR = 2, K1_SCALAR_BOUND_PTR = 3, K1_A_PTR = 8, K1_B_PTR = 9, K1_GROUP_PTR = 1;
  struct FixedUint {
    u64 ptr, bound_ptr, limbs[8];  // 32-bit limbs, least significant first
  };
  static const FixedUint uints[5] = {
      {U256_BOUND_PTR, U256_BOUND_PTR, {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}},
      {K1_BASE_BOUND_PTR, K1_BASE_BOUND_PTR, {0xFFFFFC2Eu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}},
      {K1_SCALAR_BOUND_PTR, K1_SCALAR_BOUND_PTR, {0xD0364140u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}},
      {K1_A_PTR, K1_BASE_BOUND_PTR, {0, 0, 0, 0, 0, 0, 0, 0}},
      {K1_B_PTR, K1_BASE_BOUND_PTR, {7, 0, 0, 0, 0, 0, 0, 0}}};
  const e2 alpha = e2{gl_canon(randomness[0]), gl_canon(randomness[1])}, beta = e2{gl_canon(randomness[2]), gl_canon(randomness[3])};
  e2 bp[MAX_MESSAGE_WIDTH + 1];
  bp[0] = e2_make(1);
  for (int i = 1; i <= MAX_MESSAGE_WIDTH; i++) bp[i] = e2_mul(bp[i - 1], beta);
  e2 acc = e2_make(0);
  for (int i = 0; i < n_airs; i++) acc = e2_add(acc, e2{gl_canon(aux_values[i][0]), gl_canon(aux_values[i][1])});
  bool zero_denominator = false;
  auto consume = [&](int bus, const u64* fields, int n) {
    e2 d = e2_add(alpha, e2_mulf(bp[MAX_MESSAGE_WIDTH], (u64)(bus + 1)));
    for (int i = 0; i < n; i++) d = e2_add(d, e2_mulf(bp[i], fields[i]));
    if (e2_is_zero(d)) {
      zero_denominator = true;
      return;
    }
    acc = e2_add(acc, e2_inv(d));
  };
  const u64 k1_group[5] = {K1_GROUP_PTR, K1_A_PTR, K1_B_PTR, K1_BASE_BOUND_PTR, K1_SCALAR_BOUND_PTR};
  consume(BUS_EC_GROUP, k1_group, 5);
  if (fixed_uints)
    for (const FixedUint& u : uints) {
      u64 f[10] = {u.ptr, u.bound_ptr};
      for (int j = 0; j < 8; j++) f[2 + j] = u.limbs[j];
      consume(BUS_UINT_VAL, f, 10);
    }
  if (zero_denominator) return -1;  // ReductionError: "fixed ... boundary denominator was zero"
  assertions_out[0] = acc.c0;
  assertions_out[1] = acc.c1;
  return 1;
}
int mh_external_precompile_session(void* user, const uint64_t* randomness, size_t n_randomness, const uint64_t* const* aux_values,
                                   const size_t* n_aux_values, const uint8_t* log_trace_heights, int n_airs, uint64_t* assertions_out,
                                   size_t cap) {
  (void)user; (void)log_trace_heights;
  return precompile_session_external(true, randomness, n_randomness, aux_values, n_aux_values, n_airs, assertions_out, cap);
}
int mh_external_precompile_session_ec_only(void* user, const uint64_t* randomness, size_t n_randomness, const uint64_t* const* aux_values,
                                           const size_t* n_aux_values, const uint8_t* log_trace_heights, int n_airs, uint64_t* assertions_out,
                                           size_t cap) {
  (void)user; (void)log_trace_heights;
  return precompile_session_external(false, randomness, n_randomness, aux_values, n_aux_values, n_airs, assertions_out, cap);
}
}  // extern "C"
