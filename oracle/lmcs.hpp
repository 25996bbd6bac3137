// ORACLE — TEST INFRASTRUCTURE ONLY (see gl.hpp header).
//
// Lifted Matrix Commitment Scheme (LMCS) with the Poseidon2 sponge (and, g_lmcs, the Blake3 chaining hasher), restating
//   crates/lifted-stark/src/lmcs/lifted_tree.rs:202-284 (build_with_alignment),
//   :363-417 (build_leaf_states_upsampled), :427-461 (absorb_matrix),
//   :472-511 (compress_uniform), :326-341 (collect_rows), :155-180 (prove_batch),
//   crates/lifted-stark/src/lmcs/tree_indices.rs:185-240 (MissingSiblingsIter).
// Matrices are row-major with BIT-REVERSED physical rows (exactly the reference's storage),
// sorted by ascending height.  The HIP path uses a different physical layout; parity is on
// roots, digests, opened rows and sibling lists.
#pragma once
#include "poseidon2.hpp"
#include "blake3.hpp"
#include "keccak.hpp"
#include <algorithm>
#include <array>
#include <cstring>
#include <vector>

namespace oracle {

struct Mat {
  const uint64_t* v;  // row-major, physical (bit-reversed) row order
  size_t h, w;
};
typedef std::array<uint64_t, 4> Digest;

struct LmcsTree {
  std::vector<Mat> leaves;
  // digest_layers[d] has 2^d nodes; [0] = root, last = leaf digests in DOMAIN (natural) order.
  std::vector<std::vector<Digest>> layers;
  size_t height() const { return leaves.back().h; }
  Digest root() const { return layers[0][0]; }
};

// The Blake3 LMCS of air/src/config.rs:275-289 (LmcsConfig<Felt, u8, ChainingHasher<Blake3Hasher>,
// CompressionFunctionFromHasher<Blake3Hasher, 2, 32>>): per-leaf state = a 32-byte digest, zero at first
// (StatefulHasher::hash_rows, stateful-hasher/src/lib.rs: State::default()); absorbing a row =
// blake3(state || felts as canonical u64 little-endian bytes) (chaining.rs:32-50; the byte encoding is pinned by
// crates/crypto/src/hash/blake/tests.rs:24-34); node = blake3(left || right).  Lifting duplicates states exactly as for
// the sponge (lifted_tree.rs:363-417 is generic in the hasher).  A Digest holds the 32 bytes as four little-endian u64.
enum { LMCS_POSEIDON2 = 0, LMCS_BLAKE3 = 1, LMCS_KECCAK = 2, LMCS_RPO = 3, LMCS_RPX = 4 };  // RPO / RPX: the sponge LMCS with another permutation
// The STARK configuration the oracle restates (test infrastructure: one setting for the process, orc_set_lmcs): the LMCS
// hasher, and with it the row alignment (Alignable::ALIGNMENT: 8 for the sponge, 1 for the chaining hasher,
// chaining.rs:161-169 -- every aligned width of the protocol follows lmcs.alignment(), proof.rs:268, deep/prover.rs:133)
// and the challenger (duplex sponge / serializing hash challenger, air/src/config.rs:224,291-292).
inline int g_lmcs = LMCS_POSEIDON2;
static inline size_t lmcs_alignment() { return g_lmcs == LMCS_BLAKE3 ? 1 : (g_lmcs == LMCS_KECCAK ? 17 : 8); }
static inline Digest b3_absorb(const Digest& st, const uint64_t* row, size_t w) {
  std::vector<uint8_t> msg(32 + 8 * w);
  memcpy(msg.data(), st.data(), 32);          // little-endian host
  memcpy(msg.data() + 32, row, 8 * w);
  Digest out;
  b3::hash(msg.data(), msg.size(), reinterpret_cast<uint8_t*>(out.data()));
  return out;
}
static inline Digest b3_compress(const Digest& l, const Digest& r) {
  uint8_t msg[64];
  memcpy(msg, l.data(), 32);
  memcpy(msg + 32, r.data(), 32);
  Digest out;
  b3::hash(msg, 64, reinterpret_cast<uint8_t*>(out.data()));
  return out;
}
static inline LmcsTree lmcs_build_b3(const std::vector<Mat>& mats) {
  LmcsTree t;
  t.leaves = mats;
  size_t H = mats.back().h;
  int lgH = log2_strict(H);
  std::vector<Digest> st(H, Digest{0, 0, 0, 0}), scratch(H);
  size_t active = mats.front().h;
  for (const Mat& m : mats) {
    if (m.h > active) {
      size_t f = m.h / active;
      for (size_t i = 0; i < active; i++)
        for (size_t k = 0; k < f; k++) scratch[i * f + k] = st[i];
      std::swap(st, scratch);
    }
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)m.h; r++) st[r] = b3_absorb(st[r], m.v + (size_t)r * m.w, m.w);
    active = m.h;
  }
  std::vector<Digest> cur(H);
  for (size_t i = 0; i < H; i++) cur[i] = st[bitrev((uint32_t)i, lgH)];
  std::vector<std::vector<Digest>> up;
  up.push_back(cur);
  while (up.back().size() > 1) {
    const auto& prev = up.back();
    std::vector<Digest> next(prev.size() / 2);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)next.size(); i++) next[i] = b3_compress(prev[2 * i], prev[2 * i + 1]);
    up.push_back(std::move(next));
  }
  std::reverse(up.begin(), up.end());
  t.layers = std::move(up);
  return t;
}

// The Keccak LMCS of air/src/config.rs:307-353: SerializingStatefulSponge<StatefulSponge<KeccakF, 25, 17, 4>> -- the same
// overwrite-mode sponge as the algebraic one (stateful-hasher/src/field_sponge.rs:41-64 is generic in the item type), over u64
// lanes: felts enter as their canonical u64 (serializing_sponge.rs:60-75), 17 per permutation, digest = lanes 0..3; alignment 17
// (field_sponge.rs:70, serializing_sponge.rs:163-199: lcm(8, 17 * 8) / 8).  Node = PaddingFreeSponge<KeccakF, 25, 17, 4> over
// the 8 lanes of left || right (p3-symmetric: overwrite the first lanes of a zero state, permute once, lanes 0..3).
static inline void keccak_absorb(uint64_t st[25], const uint64_t* in, size_t n) {
  size_t pos = 0;
  while (pos < n) {
    size_t k = std::min<size_t>(17, n - pos);
    for (size_t i = 0; i < k; i++) st[i] = in[pos + i];
    for (size_t i = k; i < 17; i++) st[i] = 0;
    kk::f1600(st);
    pos += k;
  }
}
static inline Digest keccak_compress(const Digest& l, const Digest& r) {
  uint64_t st[25] = {0};
  for (int i = 0; i < 4; i++) {
    st[i] = l[i];
    st[4 + i] = r[i];
  }
  kk::f1600(st);
  return Digest{st[0], st[1], st[2], st[3]};
}
static inline LmcsTree lmcs_build_keccak(const std::vector<Mat>& mats) {
  LmcsTree t;
  t.leaves = mats;
  size_t H = mats.back().h;
  int lgH = log2_strict(H);
  typedef std::array<uint64_t, 25> St;
  std::vector<St> st(H), scratch(H);
  for (auto& s : st) s.fill(0);
  size_t active = mats.front().h;
  for (const Mat& m : mats) {
    if (m.h > active) {
      size_t f = m.h / active;
      for (size_t i = 0; i < active; i++)
        for (size_t k = 0; k < f; k++) scratch[i * f + k] = st[i];
      std::swap(st, scratch);
    }
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)m.h; r++) keccak_absorb(st[r].data(), m.v + (size_t)r * m.w, m.w);
    active = m.h;
  }
  std::vector<Digest> cur(H);
  for (size_t i = 0; i < H; i++) {
    const St& s = st[bitrev((uint32_t)i, lgH)];
    cur[i] = Digest{s[0], s[1], s[2], s[3]};
  }
  std::vector<std::vector<Digest>> up;
  up.push_back(cur);
  while (up.back().size() > 1) {
    const auto& prev = up.back();
    std::vector<Digest> next(prev.size() / 2);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)next.size(); i++) next[i] = keccak_compress(prev[2 * i], prev[2 * i + 1]);
    up.push_back(std::move(next));
  }
  std::reverse(up.begin(), up.end());
  t.layers = std::move(up);
  return t;
}

static inline LmcsTree lmcs_build(const std::vector<Mat>& mats) {
  if (g_lmcs == LMCS_BLAKE3) return lmcs_build_b3(mats);
  if (g_lmcs == LMCS_KECCAK) return lmcs_build_keccak(mats);
  LmcsTree t;
  t.leaves = mats;
  size_t H = mats.back().h;
  int lgH = log2_strict(H);
  // lifted_tree.rs:363-417: per-leaf sponge states carried across matrices; when the height
  // grows, state i is duplicated to slots [i*f, (i+1)*f).
  std::vector<std::array<uint64_t, 12>> st(H), scratch(H);
  for (auto& s : st) s.fill(0);
  size_t active = mats.front().h;
  for (const Mat& m : mats) {
    if (m.h > active) {
      size_t f = m.h / active;
      for (size_t i = 0; i < active; i++)
        for (size_t k = 0; k < f; k++) scratch[i * f + k] = st[i];
      std::swap(st, scratch);
    }
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)m.h; r++) sponge_absorb(st[r].data(), m.v + (size_t)r * m.w, m.w);
    active = m.h;
  }
  // lifted_tree.rs:247-258: digest[i] = squeeze(state[bitrev(i)])
  std::vector<Digest> cur(H);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)H; i++) {
    const auto& s = st[bitrev((uint32_t)i, lgH)];
    cur[i] = Digest{s[0], s[1], s[2], s[3]};
  }
  std::vector<std::vector<Digest>> up;
  up.push_back(cur);
  while (up.back().size() > 1) {
    const auto& prev = up.back();
    std::vector<Digest> next(prev.size() / 2);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)next.size(); i++) compress(prev[2 * i].data(), prev[2 * i + 1].data(), next[i].data());
    up.push_back(std::move(next));
  }
  std::reverse(up.begin(), up.end());
  t.layers = std::move(up);
  return t;
}

// lifted_tree.rs:326-341: rows of every matrix for DOMAIN index `idx`, each padded with zeros to a
// multiple of `alignment` (alignment 1 = unpadded).
static inline std::vector<uint64_t> lmcs_rows(const LmcsTree& t, size_t idx, size_t alignment) {
  size_t H = t.height();
  int lgH = log2_strict(H);
  size_t br = bitrev((uint32_t)idx, lgH);
  std::vector<uint64_t> out;
  for (const Mat& m : t.leaves) {
    int sh = log2_strict(H / m.h);
    const uint64_t* row = m.v + (br >> sh) * m.w;
    out.insert(out.end(), row, row + m.w);
    size_t padded = (m.w + alignment - 1) / alignment * alignment;
    out.insert(out.end(), padded - m.w, 0);
  }
  return out;
}

// tree_indices.rs: sorted+dedup indices at `depth`; missing siblings bottom-up, left-to-right.
// Returns (depth, position) pairs.
static inline std::vector<std::pair<int, size_t>> missing_siblings(std::vector<size_t> idx, int depth) {
  std::sort(idx.begin(), idx.end());
  idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
  std::vector<std::pair<int, size_t>> out;
  std::vector<size_t> cur = idx;
  for (int d = depth; d > 0; d--) {
    std::vector<size_t> next;
    for (size_t i = 0; i < cur.size();) {
      size_t node = cur[i], sib = node ^ 1;
      bool present = (i + 1 < cur.size() && cur[i + 1] == sib);
      if (next.empty() || next.back() != (node >> 1)) next.push_back(node >> 1);
      if (!present) out.push_back({d, sib});
      i += present ? 2 : 1;
    }
    cur = next;
  }
  return out;
}

// lifted_tree.rs:155-180 prove_batch: hinted felts (opened aligned rows per sorted unique index)
// and hinted commitments (missing siblings).
static inline void lmcs_prove_batch(const LmcsTree& t, std::vector<size_t> indices, size_t alignment,
                                    std::vector<uint64_t>& fields, std::vector<Digest>& commitments) {
  std::sort(indices.begin(), indices.end());
  indices.erase(std::unique(indices.begin(), indices.end()), indices.end());
  for (size_t i : indices) {
    auto r = lmcs_rows(t, i, alignment);
    fields.insert(fields.end(), r.begin(), r.end());
  }
  int depth = log2_strict(t.height());
  for (auto& s : missing_siblings(indices, depth)) commitments.push_back(t.layers[s.first][s.second]);
}

}  // namespace oracle
