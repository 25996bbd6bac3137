// BLAKE3 (unkeyed, 32-byte output) for the Blake3 LMCS of the reference's default STARK configuration
// (air/src/config.rs:275-305: LmcsConfig<Felt, u8, ChainingHasher<Blake3Hasher>, CompressionFunctionFromHasher<Blake3Hasher, 2, 32>>;
// ProvingOptions::default() = HashFunction::Blake3_256, prover/src/proving_options.rs:42-46).  Written from the published
// specification (compression function, chunk chaining, tree); host + device.  32-bit adds, xors and rotates: one
// v_alignbit per rotate, no multiplier anywhere -- on this GPU the byte hash is far cheaper than the algebraic one.
//
//   leaf of the LMCS   = chain over the matrices of  H(state || row felts as 8 LE bytes each),  state = 32 zero bytes at first
//                        (crates/stateful-hasher/src/chaining.rs:32-50; alignment 1, :161-169)
//   node of the tree   = H(left || right)   (p3-symmetric CompressionFunctionFromHasher: hash of the concatenation)
// A digest travels as four u64 = the 32 bytes little-endian, so trees, openings and transcripts keep one container.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#define B3_HD __host__ __device__ __forceinline__
#else
#define B3_HD inline
#endif

namespace b3 {

enum : uint32_t { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };
static constexpr int MAX_STACK = 8;   // device streams: chaining values of completed subtrees, messages up to 2^8 chunks = 256 KiB
static constexpr int HOST_STACK = 54;  // host streams: any message the specification allows (2^64 bytes = 2^54 chunks)

B3_HD uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
B3_HD uint32_t iv(int i) {
  constexpr uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
  return IV[i];
}
#define B3_G(a, b, c, d, mx, my) \
  do {                           \
    a = a + b + (mx);            \
    d = b3::rotr(d ^ a, 16);     \
    c = c + d;                   \
    b = b3::rotr(b ^ c, 12);     \
    a = a + b + (my);            \
    d = b3::rotr(d ^ a, 8);      \
    c = c + d;                   \
    b = b3::rotr(b ^ c, 7);      \
  } while (0)

// cv <- first 8 words of compress(cv, m, counter, block_len, flags).  The message schedule of round r is the r-th power
// of the permutation [2,6,3,10,7,0,4,13,1,11,12,5,9,14,15,8]: written out, so that every index is a compile-time constant.
B3_HD void compress(uint32_t cv[8], const uint32_t m[16], uint64_t counter, uint32_t block_len, uint32_t flags) {
  constexpr int S[7][16] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},  {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8},
                            {3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1},  {10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6},
                            {12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4},  {9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7},
                            {11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13}};
  uint32_t s0 = cv[0], s1 = cv[1], s2 = cv[2], s3 = cv[3], s4 = cv[4], s5 = cv[5], s6 = cv[6], s7 = cv[7];
  uint32_t s8 = iv(0), s9 = iv(1), s10 = iv(2), s11 = iv(3);
  uint32_t s12 = (uint32_t)counter, s13 = (uint32_t)(counter >> 32), s14 = block_len, s15 = flags;
#pragma unroll
  for (int r = 0; r < 7; r++) {
    B3_G(s0, s4, s8, s12, m[S[r][0]], m[S[r][1]]);
    B3_G(s1, s5, s9, s13, m[S[r][2]], m[S[r][3]]);
    B3_G(s2, s6, s10, s14, m[S[r][4]], m[S[r][5]]);
    B3_G(s3, s7, s11, s15, m[S[r][6]], m[S[r][7]]);
    B3_G(s0, s5, s10, s15, m[S[r][8]], m[S[r][9]]);
    B3_G(s1, s6, s11, s12, m[S[r][10]], m[S[r][11]]);
    B3_G(s2, s7, s8, s13, m[S[r][12]], m[S[r][13]]);
    B3_G(s3, s4, s9, s14, m[S[r][14]], m[S[r][15]]);
  }
  cv[0] = s0 ^ s8; cv[1] = s1 ^ s9; cv[2] = s2 ^ s10; cv[3] = s3 ^ s11;
  cv[4] = s4 ^ s12; cv[5] = s5 ^ s13; cv[6] = s6 ^ s14; cv[7] = s7 ^ s15;
}

// H(left || right): one parent-less 64-byte message = one block that is chunk start, chunk end and root
B3_HD void compress_pair(const uint32_t l[8], const uint32_t r[8], uint32_t out[8]) {
  uint32_t m[16];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    m[i] = l[i];
    m[8 + i] = r[i];
    out[i] = iv(i);
  }
  compress(out, m, 0, 64, CHUNK_START | CHUNK_END | ROOT);
}
B3_HD void parent_cv(const uint32_t l[8], const uint32_t r[8], uint32_t out[8], uint32_t extra_flags) {
  uint32_t m[16];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    m[i] = l[i];
    m[8 + i] = r[i];
    out[i] = iv(i);
  }
  compress(out, m, 0, 64, PARENT | extra_flags);
}

// Streaming hasher over 64-byte blocks handed in as 16 little-endian words.  The caller says which block is the last one
// of the message (and its length in bytes); chunk boundaries (16 blocks) and the tree above them are handled here.
template <int STACK>
struct StreamT {
  uint32_t cv[8];
  uint32_t stack[STACK][8];
  int sp;
  uint32_t chunk, blk_in_chunk;
  B3_HD void init() {
#pragma unroll
    for (int i = 0; i < 8; i++) cv[i] = iv(i);
    sp = 0;
    chunk = 0;
    blk_in_chunk = 0;
  }
  // a full block that is NOT the last of the message
  B3_HD void block(const uint32_t m[16]) {
    const uint32_t flags = (blk_in_chunk == 0 ? CHUNK_START : 0u) | (blk_in_chunk == 15 ? CHUNK_END : 0u);
    compress(cv, m, chunk, 64, flags);
    if (++blk_in_chunk == 16) {  // chunk complete: one parent per trailing zero bit of the number of chunks so far
      uint32_t cur[8];
#pragma unroll
      for (int i = 0; i < 8; i++) cur[i] = cv[i];
      uint32_t total = chunk + 1;
      while ((total & 1) == 0) {
        uint32_t p[8];
        parent_cv(stack[--sp], cur, p, 0);
#pragma unroll
        for (int i = 0; i < 8; i++) cur[i] = p[i];
        total >>= 1;
      }
#pragma unroll
      for (int i = 0; i < 8; i++) {
        stack[sp][i] = cur[i];
        cv[i] = iv(i);
      }
      sp++;
      chunk++;
      blk_in_chunk = 0;
    }
  }
  // the last block (len bytes, 1..64; 0 only for the empty message): out = the 32-byte digest as 8 words
  B3_HD void finish(const uint32_t m[16], uint32_t len, uint32_t out[8]) {
    const uint32_t flags = (blk_in_chunk == 0 ? CHUNK_START : 0u) | CHUNK_END;
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = cv[i];
    if (sp == 0) {
      compress(out, m, chunk, len, flags | ROOT);
      return;
    }
    compress(out, m, chunk, len, flags);
    while (sp > 1) {
      uint32_t p[8];
      parent_cv(stack[--sp], out, p, 0);
#pragma unroll
      for (int i = 0; i < 8; i++) out[i] = p[i];
    }
    uint32_t p[8];
    parent_cv(stack[0], out, p, ROOT);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = p[i];
    sp = 0;
  }
};
using Stream = StreamT<MAX_STACK>;      // kernels (rows of at most 256 KiB: checked where leaves are hashed) and kernel arguments
using HostStream = StreamT<HOST_STACK>;  // host: transcripts, verifier leaves, mh_blake3 -- no length limit

// plain byte-string hash (host side: the cap of a sharded tree, the challenger, the verifier, tests); any length
inline void hash_bytes(const uint8_t* p, size_t n, uint8_t out32[32]) {
  HostStream s;
  s.init();
  size_t off = 0;
  uint32_t m[16];
  auto load = [&](size_t o, size_t len) {
    uint8_t b[64] = {0};
    for (size_t i = 0; i < len; i++) b[i] = p[o + i];
    for (int i = 0; i < 16; i++) m[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
  };
  while (n - off > 64) {
    load(off, 64);
    s.block(m);
    off += 64;
  }
  load(off, n - off);
  uint32_t o[8];
  s.finish(m, (uint32_t)(n - off), o);
  for (int i = 0; i < 8; i++)
    for (int k = 0; k < 4; k++) out32[4 * i + k] = (uint8_t)(o[i] >> (8 * k));
}

}  // namespace b3
