// ORACLE — TEST INFRASTRUCTURE ONLY (see gl.hpp header).
//
// BLAKE3 (unkeyed hash, 32-byte output) written from the published specification (BLAKE3 paper, section 2: compression
// function 2.2, chunk chaining 2.4, tree 2.5).  The reference reaches it through the external crates `blake3 1.8` /
// `p3-blake3 0.6` (crates/crypto/src/hash/blake/mod.rs:16,47-49; Cargo.toml:128,175), used by the Blake3 STARK
// configuration air/src/config.rs:275-305.  Pinned by tests/golden/blake3.json: digests of the official test pattern
// (byte i = i mod 251) at the lengths of the official test_vectors.json, produced with an independent implementation
// (the BLAKE3 team's C code as shipped in LLVM, tests/golden/make_blake3_golden.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace oracle {
namespace b3 {

static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const int PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static inline void g(uint32_t* s, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
  s[a] = s[a] + s[b] + mx;
  s[d] = rotr(s[d] ^ s[a], 16);
  s[c] = s[c] + s[d];
  s[b] = rotr(s[b] ^ s[c], 12);
  s[a] = s[a] + s[b] + my;
  s[d] = rotr(s[d] ^ s[a], 8);
  s[c] = s[c] + s[d];
  s[b] = rotr(s[b] ^ s[c], 7);
}
// out[16]: the full output of the compression function (first 8 words = the new chaining value)
static inline void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len, uint32_t flags,
                            uint32_t out[16]) {
  uint32_t s[16], m[16], t[16];
  for (int i = 0; i < 8; i++) s[i] = cv[i];
  for (int i = 0; i < 4; i++) s[8 + i] = IV[i];
  s[12] = (uint32_t)counter;
  s[13] = (uint32_t)(counter >> 32);
  s[14] = block_len;
  s[15] = flags;
  for (int i = 0; i < 16; i++) m[i] = block[i];
  for (int r = 0; r < 7; r++) {
    g(s, 0, 4, 8, 12, m[0], m[1]);
    g(s, 1, 5, 9, 13, m[2], m[3]);
    g(s, 2, 6, 10, 14, m[4], m[5]);
    g(s, 3, 7, 11, 15, m[6], m[7]);
    g(s, 0, 5, 10, 15, m[8], m[9]);
    g(s, 1, 6, 11, 12, m[10], m[11]);
    g(s, 2, 7, 8, 13, m[12], m[13]);
    g(s, 3, 4, 9, 14, m[14], m[15]);
    if (r < 6) {
      for (int i = 0; i < 16; i++) t[i] = m[PERM[i]];
      for (int i = 0; i < 16; i++) m[i] = t[i];
    }
  }
  for (int i = 0; i < 8; i++) {
    out[i] = s[i] ^ s[i + 8];
    out[i + 8] = s[i + 8] ^ cv[i];
  }
}
static inline void words_from_bytes(const uint8_t* p, size_t n, uint32_t w[16]) {  // n <= 64, zero padded, little endian
  uint8_t b[64];
  memset(b, 0, 64);
  if (n) memcpy(b, p, n);
  for (int i = 0; i < 16; i++) w[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
}
// A node whose compression is deferred until it is known whether it is the root
struct Output {
  uint32_t cv[8], block[16];
  uint64_t counter;
  uint32_t block_len, flags;
  void chaining_value(uint32_t out8[8]) const {
    uint32_t o[16];
    compress(cv, block, counter, block_len, flags, o);
    for (int i = 0; i < 8; i++) out8[i] = o[i];
  }
  void root_bytes(uint8_t out32[32]) const {
    uint32_t o[16];
    compress(cv, block, 0, block_len, flags | ROOT, o);
    for (int i = 0; i < 8; i++)
      for (int k = 0; k < 4; k++) out32[4 * i + k] = (uint8_t)(o[i] >> (8 * k));
  }
};
static inline Output chunk_output(const uint8_t* p, size_t n, uint64_t chunk_index) {  // n <= 1024; n == 0 only for the empty input
  uint32_t cv[8];
  for (int i = 0; i < 8; i++) cv[i] = IV[i];
  size_t n_blocks = n ? (n + 63) / 64 : 1;
  Output o;
  for (size_t b = 0; b < n_blocks; b++) {
    const size_t off = 64 * b, len = n - off < 64 ? n - off : 64;
    uint32_t w[16];
    words_from_bytes(p + off, len, w);
    uint32_t flags = (b == 0 ? CHUNK_START : 0) | (b + 1 == n_blocks ? CHUNK_END : 0);
    if (b + 1 == n_blocks) {
      for (int i = 0; i < 8; i++) o.cv[i] = cv[i];
      for (int i = 0; i < 16; i++) o.block[i] = w[i];
      o.counter = chunk_index;
      o.block_len = (uint32_t)len;
      o.flags = flags;
    } else {
      uint32_t out[16];
      compress(cv, w, chunk_index, 64, flags, out);
      for (int i = 0; i < 8; i++) cv[i] = out[i];
    }
  }
  return o;
}
static inline Output parent_output(const uint32_t l[8], const uint32_t r[8]) {
  Output o;
  for (int i = 0; i < 8; i++) {
    o.cv[i] = IV[i];
    o.block[i] = l[i];
    o.block[8 + i] = r[i];
  }
  o.counter = 0;
  o.block_len = 64;
  o.flags = PARENT;
  return o;
}
static inline void hash(const uint8_t* p, size_t n, uint8_t out32[32]) {
  std::vector<std::vector<uint32_t>> stack;  // chaining values of completed subtrees, left to right
  size_t n_chunks = n ? (n + 1023) / 1024 : 1;
  for (size_t c = 0; c + 1 < n_chunks; c++) {
    uint32_t cv[8];
    chunk_output(p + 1024 * c, 1024, c).chaining_value(cv);
    uint64_t total = c + 1;  // chunks completed so far: merge one parent per trailing zero bit
    std::vector<uint32_t> cur(cv, cv + 8);
    while ((total & 1) == 0) {
      uint32_t pcv[8];
      parent_output(stack.back().data(), cur.data()).chaining_value(pcv);
      stack.pop_back();
      cur.assign(pcv, pcv + 8);
      total >>= 1;
    }
    stack.push_back(cur);
  }
  const size_t last_off = 1024 * (n_chunks - 1);
  Output o = chunk_output(p + last_off, n - last_off, n_chunks - 1);
  while (!stack.empty()) {
    uint32_t cv[8];
    o.chaining_value(cv);
    o = parent_output(stack.back().data(), cv);
    stack.pop_back();
  }
  o.root_bytes(out32);
}

}  // namespace b3
}  // namespace oracle
