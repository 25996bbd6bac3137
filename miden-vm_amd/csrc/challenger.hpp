// Host-side Fiat-Shamir: duplex challenger over an algebraic permutation (Poseidon2; RPO / RPX) and the prover transcript.
//
// In the reference these live on the host as well: p3-challenger 0.6.2 `DuplexChallenger`
// (semantics mirrored in-tree by crates/lib/core/asm/stark/random_coin.masm:103-303, :944-975) and
// `ProverTranscript` (crates/stark-transcript/src/prover.rs:116-145).  Only `grind` is moved to the
// GPU (fri.hip k_grind): the device searches a window of witnesses in parallel and the host
// replays the smallest hit, so the transcript is a pure function of the inputs.
#pragma once
#include "rescue.cuh"
#include <array>
#include <vector>

typedef std::array<u64, 4> Digest4;

struct HostChallenger {
  int hash = 0;  // MH_LMCS_* id of the algebraic configuration: which permutation the sponge uses (0 Poseidon2, 3 RPO, 4 RPX)
  u64 st[12];
  std::vector<u64> in, out;
  HostChallenger() {
    for (auto& x : st) x = 0;
  }
  void duplexing() {
    size_t k = in.size();
    if (k) {
      for (size_t i = 0; i < k; i++) st[i] = in[i];
      for (size_t i = k; i < 8; i++) st[i] = 0;
      st[8] = gl_add(st[8], (u64)k);
      in.clear();
    }
    alg_permute(hash, st);
    out.assign(st, st + 8);
  }
  void observe(u64 x) {
    out.clear();
    in.push_back(gl_canon(x));
    if (in.size() == 8) duplexing();
  }
  void observe_digest(const u64 d[4]) {
    for (int i = 0; i < 4; i++) observe(d[i]);
  }
  u64 sample() {
    if (!in.empty() || out.empty()) duplexing();
    u64 x = out.back();
    out.pop_back();
    return x;
  }
  e2 sample_ef() {
    u64 c0 = sample();
    u64 c1 = sample();
    return e2{c0, c1};
  }
  size_t sample_bits(int bits) { return (size_t)((sample() & 0xFFFFFFFFULL) & (((u64)1 << bits) - 1)); }
  bool check_witness(int bits, u64 w) {
    if (bits == 0) return w == 0;
    observe(w);
    return sample_bits(bits) == 0;
  }
};

struct HostTranscript {
  HostChallenger ch;
  std::vector<u64> fields;
  std::vector<Digest4> commitments;
  void send_field(u64 x) {
    fields.push_back(x);
    ch.observe(x);
  }
  void send_ef(e2 x) {
    send_field(x.c0);
    send_field(x.c1);
  }
  void send_commitment(const u64 d[4]) {
    commitments.push_back(Digest4{d[0], d[1], d[2], d[3]});
    ch.observe_digest(d);
  }
  void hint_fields(const std::vector<u64>& v) { fields.insert(fields.end(), v.begin(), v.end()); }
  void hint_commitments(const std::vector<u64>& flat) {
    for (size_t i = 0; i + 4 <= flat.size(); i += 4) commitments.push_back(Digest4{flat[i], flat[i + 1], flat[i + 2], flat[i + 3]});
  }
};
