// Keccak-f[1600] for the Keccak LMCS of the reference (air/src/config.rs:307-353: SerializingStatefulSponge<StatefulSponge<KeccakF,
// 25, 17, 4>> leaves, PaddingFreeSponge<KeccakF, 25, 17, 4> nodes): the overwrite-mode sponge of the algebraic configuration
// over 64-bit lanes -- 17 felts (as canonical u64) per permutation, digest = lanes 0..3, row alignment 17.  Written from FIPS 202
// (theta, rho, pi, chi, iota); host + device; every lane index is a compile-time constant so that the state lives in registers.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#define KK_HD __host__ __device__ __forceinline__
#else
#define KK_HD inline
#endif

namespace kk {

KK_HD uint64_t rotl(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
KK_HD uint64_t rc(int r) {
  constexpr uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
                               0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                               0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
                               0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                               0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
                               0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  return RC[r];
}
// one round on a[x + 5y]
KK_HD void round(uint64_t a[25], uint64_t round_constant) {
  constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  uint64_t c[5], b[25];
#pragma unroll
  for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
  for (int x = 0; x < 5; x++) {
    const uint64_t d = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
#pragma unroll
    for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
  }
#pragma unroll
  for (int x = 0; x < 5; x++)
#pragma unroll
    for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], RHO[x + 5 * y]);
#pragma unroll
  for (int y = 0; y < 5; y++)
#pragma unroll
    for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
  a[0] ^= round_constant;
}
KK_HD void f1600(uint64_t a[25]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int r = 0; r < 24; r++) round(a, rc(r));
}
// PaddingFreeSponge over left || right (8 lanes): overwrite the first lanes of a zero state, one permutation, lanes 0..3
KK_HD void compress_pair(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
  uint64_t a[25];
#pragma unroll
  for (int i = 0; i < 25; i++) a[i] = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    a[i] = l[i];
    a[4 + i] = r[i];
  }
  f1600(a);
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = a[i];
}

// Keccak sponge hash, rate 136 bytes, 32-byte digest; pad = 0x01 (Keccak-256: the challenger's Keccak256Hash) or 0x06 (SHA3-256).
// XOR-absorbing, FIPS 202 section 4 (unlike the overwrite-mode LMCS sponge above).  Host side (challenger, verifier).
struct Sponge256 {
  uint64_t st[25];
  uint8_t buf[136];
  uint32_t fill;
  KK_HD void init() {
    for (int i = 0; i < 25; i++) st[i] = 0;
    fill = 0;
  }
  KK_HD void absorb_block(const uint8_t* b) {
    for (int i = 0; i < 17; i++) {
      uint64_t w = 0;
      for (int k = 0; k < 8; k++) w |= (uint64_t)b[8 * i + k] << (8 * k);
      st[i] ^= w;
    }
    f1600(st);
  }
  KK_HD void update(const uint8_t* p, size_t n) {
    for (size_t i = 0; i < n; i++) {
      buf[fill++] = p[i];
      if (fill == 136) {
        absorb_block(buf);
        fill = 0;
      }
    }
  }
  KK_HD void finish(uint8_t pad, uint8_t out32[32]) {
    for (uint32_t i = fill; i < 136; i++) buf[i] = 0;
    buf[fill] ^= pad;
    buf[135] ^= 0x80;
    absorb_block(buf);
    for (int i = 0; i < 4; i++)
      for (int k = 0; k < 8; k++) out32[8 * i + k] = (uint8_t)(st[i] >> (8 * k));
  }
};
KK_HD void hash256(const uint8_t* p, size_t n, uint8_t pad, uint8_t out32[32]) {
  Sponge256 s;
  s.init();
  s.update(p, n);
  s.finish(pad, out32);
}
// the overwrite-mode sponge of the Keccak LMCS over whole rows (field_sponge.rs:41-64 with WIDTH 25, RATE 17)
KK_HD void lmcs_absorb(uint64_t st[25], const uint64_t* v, size_t n) {
  for (size_t off = 0; off < n; off += 17) {
    const size_t k = n - off < 17 ? n - off : 17;
    for (size_t i = 0; i < k; i++) st[i] = v[off + i];
    for (size_t i = k; i < 17; i++) st[i] = 0;
    f1600(st);
  }
}

}  // namespace kk
