// ORACLE — TEST INFRASTRUCTURE ONLY (see gl.hpp header).
//
// Keccak-f[1600] and the Keccak / SHA-3 sponge hash, written from FIPS 202 (sections 3.2-3.4: theta, rho, pi, chi, iota; 4: the
// sponge; rate 136 bytes for a 256-bit digest).  The reference reaches them through external crates (p3-keccak 0.6: KeccakF on
// [u64; 25], Keccak256Hash = the original Keccak padding 0x01; air/src/config.rs:307-353).  Pinned by tests/test_keccak.py
// against Python's hashlib.sha3_256 (the same permutation and sponge with the SHA-3 domain byte 0x06) and against the
// well-known first lanes of Keccak-f on the zero state.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace oracle {
namespace kk {

static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
                                0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                                0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
                                0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                                0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
                                0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
// rho offsets r[x][y] (FIPS 202 table 2), lane index x + 5y
static const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};

static inline uint64_t rotl(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

// a[x + 5y]
static inline void f1600(uint64_t a[25]) {
  for (int round = 0; round < 24; round++) {
    uint64_t c[5], d[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
    for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
    // rho + pi: B[y, 2x + 3y] = rot(A[x, y], r[x, y])
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], RHO[x + 5 * y]);
    for (int y = 0; y < 5; y++)
      for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= RC[round];
  }
}

// sponge hash with rate 136 and a 32-byte digest; pad = 0x01 (Keccak-256) or 0x06 (SHA3-256)
static inline void hash256(const uint8_t* p, size_t n, uint8_t pad, uint8_t out32[32]) {
  uint64_t st[25];
  memset(st, 0, sizeof st);
  const size_t rate = 136;
  size_t off = 0;
  auto xor_block = [&](const uint8_t* b) {
    for (size_t i = 0; i < rate / 8; i++) {
      uint64_t w = 0;
      for (int k = 0; k < 8; k++) w |= (uint64_t)b[8 * i + k] << (8 * k);
      st[i] ^= w;
    }
  };
  while (n - off >= rate) {
    xor_block(p + off);
    f1600(st);
    off += rate;
  }
  uint8_t last[136];
  memset(last, 0, rate);
  if (n - off) memcpy(last, p + off, n - off);
  last[n - off] ^= pad;
  last[rate - 1] ^= 0x80;
  xor_block(last);
  f1600(st);
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 8; k++) out32[8 * i + k] = (uint8_t)(st[i] >> (8 * k));
}

}  // namespace kk
}  // namespace oracle
