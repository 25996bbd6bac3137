// Host-side Fiat-Shamir: duplex challenger over an algebraic permutation (Poseidon2; RPO / RPX) and the prover transcript.
//
// In the reference these live on the host as well: p3-challenger 0.6.2 `DuplexChallenger`
// (semantics mirrored in-tree by crates/lib/core/asm/stark/random_coin.masm:103-303, :944-975) and
// `ProverTranscript` (crates/stark-transcript/src/prover.rs:116-145).  Only `grind` is moved to the
// GPU (fri.hip k_grind): the device searches a window of witnesses in parallel and the host
// replays the smallest hit, so the transcript is a pure function of the inputs.
#pragma once
#include "rescue.cuh"
#include "blake3.cuh"
#include "keccak.cuh"
#include <array>
#include <cstring>
#include <vector>

typedef std::array<u64, 4> Digest4;

// Byte configurations (hash = MH_LMCS_BLAKE3 = 1 / MH_LMCS_KECCAK = 2; air/src/config.rs:291-303, 334-353):
// SerializingChallenger64<Felt, HashChallenger<u8, Blake3Hasher | Keccak256Hash, 32>> -- p3-challenger 0.6.2, external, no in-tree
// mirror (parity unpinned), restated from the published crate:
//   observe(felt)  = clear the output buffer, append the felt's canonical u64 as 8 little-endian bytes to the input buffer
//                    (a 32-byte digest is observed byte by byte = its four little-endian u64s);
//   flush          = output := hash(input); input := output (the chaining value);   sample byte = output.pop() (from the END),
//                    flushing first when the output buffer is empty;
//   sample felt    = u64 from 8 sampled bytes (first sampled = lowest byte), redrawn while >= p;   sample_bits = the low bits of
//                    such a u64 (no redraw);   check_witness = observe(witness), sample_bits(bits) == 0.
// The configuration observes RELATION_DIGEST before anything else (config.rs:301-302): init_from_state() observes the capacity
// words st[8..12] of the state the caller hands over (where the sponge configurations keep that digest).
struct HostChallenger {
  int hash = 0;  // MH_LMCS_* id of the configuration: 0 Poseidon2, 3 RPO, 4 RPX (duplex sponge); 1 Blake3, 2 Keccak (byte hash)
  u64 st[12];
  std::vector<u64> in, out;
  std::vector<uint8_t> bin, bout;
  HostChallenger() {
    for (auto& x : st) x = 0;
  }
  bool bytes() const { return hash == 1 || hash == 2; }
  void init_from_state(const u64 state[12]) {
    for (int i = 0; i < 12; i++) st[i] = gl_canon(state[i]);
    if (bytes())
      for (int i = 8; i < 12; i++) observe(st[i]);
  }
  void hash_bytes(const uint8_t* p, size_t n, uint8_t d[32]) const {
    if (hash == 2) kk::hash256(p, n, 0x01, d);
    else b3::hash_bytes(p, n, d);
  }
  void flush_bytes() {
    uint8_t d[32];
    hash_bytes(bin.data(), bin.size(), d);
    bout.assign(d, d + 32);
    bin.assign(d, d + 32);
  }
  u64 sample_u64_bytes() {
    u64 v = 0;
    for (int i = 0; i < 8; i++) {
      if (bout.empty()) flush_bytes();
      v |= (u64)bout.back() << (8 * i);
      bout.pop_back();
    }
    return v;
  }
  // the transcript digest (CanFinalizeDigest): one unconditional state transition, then 4 felts / 32 bytes
  void finalize(u64 digest[4]) {
    if (bytes()) {
      flush_bytes();
      memcpy(digest, bout.data(), 32);
      return;
    }
    duplexing();
    for (int i = 0; i < 4; i++) digest[i] = st[i];
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
    if (bytes()) {
      bout.clear();
      const u64 v = gl_canon(x);
      for (int i = 0; i < 8; i++) bin.push_back((uint8_t)(v >> (8 * i)));
      return;
    }
    out.clear();
    in.push_back(gl_canon(x));
    if (in.size() == 8) duplexing();
  }
  // A word of the statement framing observed before the proof starts (protocol parameters, public values, a setup
  // commitment: prover/mod.rs:252-286).  Felts are canonical by contract; a byte configuration's setup commitment is four
  // words that are NOT felts and goes in as its raw bytes, exactly as the reference observes the 32-byte digest.
  void observe_framing(u64 x) {
    if (bytes()) {
      bout.clear();
      for (int i = 0; i < 8; i++) bin.push_back((uint8_t)(x >> (8 * i)));
      return;
    }
    observe(x);
  }
  void observe_digest(const u64 d[4]) {
    if (bytes()) {  // 32 raw bytes: the words of a byte digest are not field elements
      bout.clear();
      for (int i = 0; i < 4; i++)
        for (int k = 0; k < 8; k++) bin.push_back((uint8_t)(d[i] >> (8 * k)));
      return;
    }
    for (int i = 0; i < 4; i++) observe(d[i]);
  }
  u64 sample() {
    if (bytes()) {
      for (;;) {
        const u64 v = sample_u64_bytes();
        if (v < GL_P) return v;
      }
    }
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
  size_t sample_bits(int bits) {
    if (bytes()) return (size_t)(sample_u64_bytes() & (((u64)1 << bits) - 1));
    return (size_t)((sample() & 0xFFFFFFFFULL) & (((u64)1 << bits) - 1));
  }
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
