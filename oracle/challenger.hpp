// ORACLE — TEST INFRASTRUCTURE ONLY (see gl.hpp header).
//
// Duplex challenger + prover/verifier transcripts.
//   * DuplexChallenger<Felt, Poseidon2, 12, 8> is the external crate p3-challenger 0.6.2; its
//     semantics are restated from the reference's in-tree MASM mirror
//     crates/lib/core/asm/stark/random_coin.masm:103-116 (squeeze-only permute, no tag),
//     :128-139 (sample = rate[--output_len]), :151-170 (sample_bits = low bits of low 32 bits),
//     :181-221 (observe: clear outputs, append, duplex when 8 buffered, state[8] += 8),
//     :272-303 (flush: zero rate[k..8), state[8] += k, permute), :944-975 (grind check).
//   * ProverTranscript / VerifierTranscript: crates/stark-transcript/src/prover.rs:116-145,
//     verifier.rs (fields + commitments streams; send = record + observe, hint = record only).
// PARITY UNPINNED: the PoW witness search order of `grind` and `finalize()` are external code with
// no in-tree restatement; this oracle takes the smallest witness; finalize = one unconditional duplexing (the only
// in-tree statement about it: stark-transcript/src/prover.rs:31-35) and state[0..4].
#pragma once
#include "poseidon2.hpp"
#include "lmcs.hpp"
#include <algorithm>
#include <array>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace oracle {

typedef std::array<uint64_t, 4> Digest;

// modes LMCS_BLAKE3 / LMCS_KECCAK: SerializingChallenger64<Felt, HashChallenger<u8, Blake3Hasher | Keccak256Hash, 32>>
// (air/src/config.rs:291-303, 334-353; Keccak256Hash = Keccak-256 with the original 0x01 padding, p3-keccak), both
// external (p3-challenger 0.6.2) and with NO in-tree mirror -- PARITY UNPINNED, restated from the published crate:
//   HashChallenger: observe(byte) clears the output buffer and appends to the input buffer; sampling from an empty output
//     buffer flushes: output = hash(input), input := output (chaining); sample = output.pop() (from the END);
//   SerializingChallenger64: observe(felt) = its canonical u64 as 8 little-endian bytes (a 32-byte digest is observed byte
//     by byte = the same as its four little-endian u64s); sample = u64 from 8 sampled bytes (first sampled = lowest),
//     rejected and redrawn when >= p; sample_bits = low bits of such a u64 without rejection; grind/check_witness =
//     observe(witness), sample_bits(bits) == 0.
// The configuration starts from an empty input buffer and observes RELATION_DIGEST first (config.rs:301-302): here the
// capacity words st[8..12] of the initial state (where the sponge configuration keeps that digest) are observed.
struct Challenger {
  int mode = LMCS_POSEIDON2;
  uint64_t st[12];
  std::vector<uint64_t> in, out;
  std::vector<uint8_t> bin, bout;  // byte mode
  Challenger() : mode(g_lmcs) {
    for (auto& x : st) x = 0;
  }
  void init(const uint64_t init_state[12]) {
    for (int i = 0; i < 12; i++) st[i] = init_state[i];
    if (bytes())
      for (int i = 8; i < 12; i++) observe(st[i]);
  }
  void duplexing() {
    size_t k = in.size();
    if (k) {
      for (size_t i = 0; i < k; i++) st[i] = in[i];
      for (size_t i = k; i < 8; i++) st[i] = 0;
      st[8] = fadd(st[8], (uint64_t)k);
      in.clear();
    }
    alg_permute(st);
    out.assign(st, st + 8);
  }
  bool bytes() const { return mode == LMCS_BLAKE3 || mode == LMCS_KECCAK; }
  void hash_bytes(const uint8_t* p, size_t n, uint8_t d[32]) const {
    if (mode == LMCS_KECCAK) kk::hash256(p, n, 0x01, d);
    else b3::hash(p, n, d);
  }
  void flush_bytes() {
    uint8_t d[32];
    hash_bytes(bin.data(), bin.size(), d);
    bout.assign(d, d + 32);
    bin.assign(d, d + 32);
  }
  uint64_t sample_u64_bytes() {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) {
      if (bout.empty()) flush_bytes();
      v |= (uint64_t)bout.back() << (8 * i);
      bout.pop_back();
    }
    return v;
  }
  void observe(uint64_t x) {
    if (bytes()) {
      bout.clear();
      for (int i = 0; i < 8; i++) bin.push_back((uint8_t)(x >> (8 * i)));
      return;
    }
    out.clear();
    in.push_back(x);
    if (in.size() == 8) duplexing();
  }
  void observe_digest(const Digest& d) {
    for (uint64_t x : d) observe(x);
  }
  uint64_t sample() {
    if (bytes()) {
      for (;;) {
        uint64_t v = sample_u64_bytes();
        if (v < P) return v;
      }
    }
    if (!in.empty() || out.empty()) duplexing();
    uint64_t x = out.back();
    out.pop_back();
    return x;
  }
  E2 sample_ef() {
    uint64_t c0 = sample();
    uint64_t c1 = sample();
    return E2{c0, c1};
  }
  size_t sample_bits(int bits) {
    if (bytes()) return (size_t)(sample_u64_bytes() & (((uint64_t)1 << bits) - 1));
    return (size_t)((sample() & 0xFFFFFFFFULL) & (((uint64_t)1 << bits) - 1));
  }
  bool check_witness(int bits, uint64_t w) {
    if (bits == 0) return w == 0;
    observe(w);
    return sample_bits(bits) == 0;
  }
  // check_witness on a stack copy of the state (no heap traffic): exactly one duplexing happens
  // between observe(w) and the sampled bits, whether the buffer fills up (8) or not.
  bool trial(int bits, uint64_t w) const {
    if (bytes()) {
      std::vector<uint8_t> m(bin);
      for (int i = 0; i < 8; i++) m.push_back((uint8_t)(w >> (8 * i)));
      uint8_t d[32];
      hash_bytes(m.data(), m.size(), d);
      uint64_t v = 0;
      for (int i = 0; i < 8; i++) v |= (uint64_t)d[31 - i] << (8 * i);
      return (v & (((uint64_t)1 << bits) - 1)) == 0;
    }
    uint64_t s[12];
    for (int i = 0; i < 12; i++) s[i] = st[i];
    size_t k = in.size() + 1;
    for (size_t i = 0; i < 8; i++) s[i] = i < in.size() ? in[i] : (i == in.size() ? w : 0);
    s[8] = fadd(s[8], (uint64_t)k);
    alg_permute(s);
    return ((s[7] & 0xFFFFFFFFULL) & (((uint64_t)1 << bits) - 1)) == 0;
  }
  uint64_t grind(int bits) {
    if (bits == 0) return 0;
    // smallest valid witness; windows searched in parallel (OpenMP), result independent of threads
    const uint64_t window = 4096;
    for (uint64_t base = 0;; base += window) {
      uint64_t best = ~0ULL;
#pragma omp parallel for reduction(min : best) schedule(static)
      for (long i = 0; i < (long)window; i++)
        if (trial(bits, base + (uint64_t)i)) best = std::min(best, base + (uint64_t)i);
      if (best != ~0ULL) {
        bool ok = check_witness(bits, best);
        if (!ok) throw std::runtime_error("oracle grind: fast trial disagrees with check_witness");
        return best;
      }
    }
  }
  Digest finalize() {
    if (bytes()) {  // one unconditional state transition (stark-transcript/src/prover.rs:31-35), then the digest
      flush_bytes();
      Digest d;
      memcpy(d.data(), bout.data(), 32);
      return d;
    }
    duplexing();  // unconditional: crates/stark-transcript/src/prover.rs:31-35
    return Digest{st[0], st[1], st[2], st[3]};
  }
};

struct ProverTranscript {
  Challenger ch;
  std::vector<uint64_t> fields;
  std::vector<Digest> commitments;
  void send_field(uint64_t x) {
    fields.push_back(x);
    ch.observe(x);
  }
  void send_ef(E2 x) {
    send_field(x.c0);
    send_field(x.c1);
  }
  void send_commitment(const Digest& d) {
    commitments.push_back(d);
    ch.observe_digest(d);
  }
  void hint_fields(const std::vector<uint64_t>& v) { fields.insert(fields.end(), v.begin(), v.end()); }
  void hint_commitments(const std::vector<Digest>& v) { commitments.insert(commitments.end(), v.begin(), v.end()); }
  uint64_t grind(int bits) {
    uint64_t w = ch.grind(bits);
    fields.push_back(w);
    return w;
  }
};

struct TranscriptError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct VerifierTranscript {
  Challenger ch;
  const uint64_t* f;
  size_t nf, fpos = 0;
  const Digest* c;
  size_t nc, cpos = 0;
  uint64_t next_field() {
    if (fpos >= nf) throw TranscriptError("transcript: out of field elements");
    return f[fpos++];
  }
  const Digest& next_commitment() {
    if (cpos >= nc) throw TranscriptError("transcript: out of commitments");
    return c[cpos++];
  }
  uint64_t receive_field() {
    uint64_t x = next_field();
    if (x >= P) throw TranscriptError("transcript: non-canonical field element");
    ch.observe(x);
    return x;
  }
  E2 receive_ef() {
    uint64_t a = receive_field();
    uint64_t b = receive_field();
    return E2{a, b};
  }
  Digest receive_commitment() {
    Digest d = next_commitment();
    ch.observe_digest(d);
    return d;
  }
  uint64_t hint_field() { return next_field(); }
  Digest hint_commitment() { return next_commitment(); }
  void grind(int bits) {
    uint64_t w = next_field();
    if (!ch.check_witness(bits, w)) throw TranscriptError("transcript: invalid proof-of-work witness");
  }
  bool exhausted() const { return fpos == nf && cpos == nc; }
};

}  // namespace oracle
