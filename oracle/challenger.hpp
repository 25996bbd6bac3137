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
#include <algorithm>
#include <array>
#include <stdexcept>
#include <vector>

namespace oracle {

typedef std::array<uint64_t, 4> Digest;

struct Challenger {
  uint64_t st[12];
  std::vector<uint64_t> in, out;
  Challenger() {
    for (auto& x : st) x = 0;
  }
  void duplexing() {
    size_t k = in.size();
    if (k) {
      for (size_t i = 0; i < k; i++) st[i] = in[i];
      for (size_t i = k; i < 8; i++) st[i] = 0;
      st[8] = fadd(st[8], (uint64_t)k);
      in.clear();
    }
    p2_permute(st);
    out.assign(st, st + 8);
  }
  void observe(uint64_t x) {
    out.clear();
    in.push_back(x);
    if (in.size() == 8) duplexing();
  }
  void observe_digest(const Digest& d) {
    for (uint64_t x : d) observe(x);
  }
  uint64_t sample() {
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
  size_t sample_bits(int bits) { return (size_t)((sample() & 0xFFFFFFFFULL) & (((uint64_t)1 << bits) - 1)); }
  bool check_witness(int bits, uint64_t w) {
    if (bits == 0) return w == 0;
    observe(w);
    return sample_bits(bits) == 0;
  }
  // check_witness on a stack copy of the state (no heap traffic): exactly one duplexing happens
  // between observe(w) and the sampled bits, whether the buffer fills up (8) or not.
  bool trial(int bits, uint64_t w) const {
    uint64_t s[12];
    for (int i = 0; i < 12; i++) s[i] = st[i];
    size_t k = in.size() + 1;
    for (size_t i = 0; i < 8; i++) s[i] = i < in.size() ? in[i] : (i == in.size() ? w : 0);
    s[8] = fadd(s[8], (uint64_t)k);
    p2_permute(s);
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
