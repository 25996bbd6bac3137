// K9 / K10 / K11 / K14: FRI folding, per-round commitments and proof-of-work grinding, gfx950.
//
// Replaces crates/lifted-stark/src/pcs/fri/prover.rs:93-242 (FriPolys::new),
// pcs/fri/fold/arity4.rs:46-121, arity2.rs and arity8.rs:35-138 (fold_evals), and the PoW search of
// crates/stark-transcript/src/prover.rs:140-144 (p3 `grind`, external).
//
// Layout: a FRI layer of n = Nl*C points is stored coset-major like every LDE here:
// slot j*Nl + r  <->  natural index i = r*C + j  (C = 2^cbits cosets; cbits = log_blowup until the
// layer gets shorter than 4 rows per coset, then 0).  The arity-`a` coset of natural index i0 is
// {i0 + m*n/a} = rows r0 + m*Nl/a of the SAME coset j, so a fold reads `a` unit-stride streams and
// writes one; the reference's bit-reversed row [y0,y2,y1,y3] is rebuilt only in the leaf hash and
// in query openings.
// s_inv for row i0 is w_n^(-i0) (fri/prover.rs:117-142: the coset shift is deliberately ignored).
#include <cstring>
#include "gl.cuh"
#include "../../include/midenhip.h"
#include "kernels.hpp"
#include "blake3.cuh"
#include "keccak.cuh"
#include "poseidon2_fast.cuh"
#define RESCUE_FAST 1  // S-boxes through p2f_mulN (poseidon2_fast.cuh is included above)
#include "rescue.cuh"

__device__ __forceinline__ e2 ld_e2(const u64* p, size_t idx) {
  const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(p + 2 * idx);
  return e2{v.x, v.y};
}
__device__ __forceinline__ void st_e2(u64* p, size_t idx, e2 v) { *reinterpret_cast<ulonglong2*>(p + 2 * idx) = make_ulonglong2(v.c0, v.c1); }

// reference row order inside a leaf: position p holds y_{bitrev(p)}
__device__ __forceinline__ u32 fri_row_pos(u32 p, int log_arity) { return bitrev32(p, log_arity); }

// ---- leaf digests: one permutation per leaf for arity 4 (8 felts = the rate), ---------------------
// sponge over 2*arity felts in general.
__global__ __launch_bounds__(256) void k_fri_leaf_hash(const u64* __restrict__ ev, int log_rows /* Nl */, int cbits, int log_arity,
                                                       u64* __restrict__ digests) {
  const int log_q = log_rows - log_arity;  // rows per coset after grouping
  const size_t leaves = (size_t)1 << (log_q + cbits);
  const size_t s = blockIdx.x * (size_t)256 + threadIdx.x;
  if (s >= leaves) return;
  const size_t j = s >> log_q, r0 = s & (((size_t)1 << log_q) - 1);
  const u32 arity = 1u << log_arity;
  u64 st[12];
#pragma unroll
  for (int i = 0; i < 12; i++) st[i] = 0;
  // absorb in chunks of 4 EF = 8 felts
  for (u32 p0 = 0; p0 < arity; p0 += 4) {
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
      e2 v = e2_make(0);
      if (p0 + k < arity) v = ld_e2(ev, (j << log_rows) + r0 + ((size_t)fri_row_pos(p0 + k, log_arity) << log_q));
      st[2 * k] = (p0 + k < arity) ? v.c0 : 0;
      st[2 * k + 1] = (p0 + k < arity) ? v.c1 : 0;
    }
    p2f_permute(st);
  }
  ulonglong2* o = reinterpret_cast<ulonglong2*>(digests + 4 * s);
  o[0] = make_ulonglong2(st[0], st[1]);
  o[1] = make_ulonglong2(st[2], st[3]);
}

// Blake3 LMCS (air/src/config.rs:275-289): leaf = blake3(32 zero bytes || the row's 2 * arity felts, 8 LE bytes each)
__global__ __launch_bounds__(256) void k_fri_leaf_hash_b3(const u64* __restrict__ ev, int log_rows, int cbits, int log_arity,
                                                          u64* __restrict__ digests) {
  const int log_q = log_rows - log_arity;
  const size_t leaves = (size_t)1 << (log_q + cbits);
  const size_t s = blockIdx.x * (size_t)256 + threadIdx.x;
  if (s >= leaves) return;
  const size_t j = s >> log_q, r0 = s & (((size_t)1 << log_q) - 1);
  const u32 arity = 1u << log_arity;
  const u32 total = 32 + 16 * arity, n_blocks = (total + 63) / 64;  // <= 160 bytes: one chunk
  b3::Stream h;
  h.init();
  uint32_t m[16], out[8];
#pragma unroll 1
  for (u32 b = 0; b < n_blocks; b++) {
    // words [16b, 16b + 16): words 0..7 = the zero state, then four words per EF element (position p)
    const int p0 = (int)(4 * b) - 2;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int pp = p0 + k;
      e2 v = e2_make(0);
      if (pp >= 0 && (u32)pp < arity) v = ld_e2(ev, (j << log_rows) + r0 + ((size_t)fri_row_pos((u32)pp, log_arity) << log_q));
      m[4 * k] = (uint32_t)v.c0; m[4 * k + 1] = (uint32_t)(v.c0 >> 32);
      m[4 * k + 2] = (uint32_t)v.c1; m[4 * k + 3] = (uint32_t)(v.c1 >> 32);
    }
    if (b + 1 < n_blocks) h.block(m);
    else h.finish(m, total - 64 * b, out);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) digests[4 * s + i] = (u64)out[2 * i] | ((u64)out[2 * i + 1] << 32);
}

// RPO / RPX: k_fri_leaf_hash with the Rescue permutations
__global__ __launch_bounds__(256) void k_fri_leaf_hash_alg(const u64* __restrict__ ev, int log_rows, int cbits, int log_arity,
                                                           u64* __restrict__ digests, int lmcs) {
  const int log_q = log_rows - log_arity;
  const size_t leaves = (size_t)1 << (log_q + cbits);
  const size_t s = blockIdx.x * (size_t)256 + threadIdx.x;
  if (s >= leaves) return;
  const size_t j = s >> log_q, r0 = s & (((size_t)1 << log_q) - 1);
  const u32 arity = 1u << log_arity;
  u64 st[12];
#pragma unroll
  for (int i = 0; i < 12; i++) st[i] = 0;
#pragma unroll 1
  for (u32 p0 = 0; p0 < arity; p0 += 4) {
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
      e2 v = e2_make(0);
      if (p0 + k < arity) v = ld_e2(ev, (j << log_rows) + r0 + ((size_t)fri_row_pos(p0 + k, log_arity) << log_q));
      st[2 * k] = (p0 + k < arity) ? v.c0 : 0;
      st[2 * k + 1] = (p0 + k < arity) ? v.c1 : 0;
    }
    alg_permute(lmcs, st);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) digests[4 * s + i] = st[i];
}

// Keccak LMCS: the sponge over the row's 2 * arity <= 16 felts = one permutation of (felts, zeros)
__global__ __launch_bounds__(256) void k_fri_leaf_hash_kk(const u64* __restrict__ ev, int log_rows, int cbits, int log_arity,
                                                          u64* __restrict__ digests) {
  const int log_q = log_rows - log_arity;
  const size_t leaves = (size_t)1 << (log_q + cbits);
  const size_t s = blockIdx.x * (size_t)256 + threadIdx.x;
  if (s >= leaves) return;
  const size_t j = s >> log_q, r0 = s & (((size_t)1 << log_q) - 1);
  const u32 arity = 1u << log_arity;
  uint64_t st[25];
#pragma unroll
  for (int i = 0; i < 25; i++) st[i] = 0;
#pragma unroll
  for (u32 p = 0; p < 8; p++) {
    if (p < arity) {
      const e2 v = ld_e2(ev, (j << log_rows) + r0 + ((size_t)fri_row_pos(p, log_arity) << log_q));
      st[2 * p] = v.c0;
      st[2 * p + 1] = v.c1;
    }
  }
  kk::f1600(st);
#pragma unroll
  for (int i = 0; i < 4; i++) digests[4 * s + i] = st[i];
}

// ---- fold ----------------------------------------------------------------------------------------
struct FoldArgs {
  const u64* ev;
  u64* out;
  int log_rows, cbits, log_arity;
  const u64* tw_inv;     // w_Nl^(-k), k < Nl/2   (Nl = rows per coset of the INPUT layer)
  const u64* coset_inv;  // [C] w_n^(-j)
  e2 beta;
  u64 w4, w8_inv, inv_arity;
};
__global__ __launch_bounds__(256) void k_fri_fold(FoldArgs a) {
  const int log_q = a.log_rows - a.log_arity;
  const size_t total = (size_t)1 << (log_q + a.cbits);
  const size_t s = blockIdx.x * (size_t)256 + threadIdx.x;
  if (s >= total) return;
  const size_t j = s >> log_q, r0 = s & (((size_t)1 << log_q) - 1);
  // s_inv = w_n^(-(r0*C + j)) = w_Nl^(-r0) * w_n^(-j);  r0 < Nl/arity <= Nl/2
  const u64 s_inv = gl_mul(a.log_rows ? a.tw_inv[r0] : 1, a.coset_inv[j]);
  const e2 x = e2_mulf(a.beta, s_inv);
  const size_t base = (j << a.log_rows) + r0;
  e2 res;
  if (a.log_arity == 1) {
    e2 y0 = ld_e2(a.ev, base), y1 = ld_e2(a.ev, base + ((size_t)1 << log_q));
    res = e2_add(e2_add(y0, y1), e2_mul(e2_sub(y0, y1), x));
  } else if (a.log_arity == 3) {
    // fold/arity8.rs: the interpolant of the 8 values on s*<w_8> at beta.  The values are read in NATURAL order
    // (y_m = f(s w_8^m) = row r0 + m*Nl/8 of this coset), so the inverse DFT runs as three DIF stages and leaves
    // 8*c_k at position bitrev(k); sum_k c_k x^k is then Horner over the bit-reversed positions.
    e2 y[8];
#pragma unroll
    for (int m = 0; m < 8; m++) y[m] = ld_e2(a.ev, base + ((size_t)m << log_q));
    const u64 wi1 = a.w8_inv, wi2 = gl_mul(wi1, wi1), wi3 = gl_mul(wi2, wi1);  // w_8^-1, w_8^-2 = w_4^-1, w_8^-3
#pragma unroll
    for (int m = 0; m < 4; m++) {  // span 4, twiddle w_8^-m on the difference
      const e2 u = y[m], v = y[m + 4];
      y[m] = e2_add(u, v);
      const e2 d = e2_sub(u, v);
      y[m + 4] = m == 0 ? d : e2_mulf(d, m == 1 ? wi1 : (m == 2 ? wi2 : wi3));
    }
#pragma unroll
    for (int h = 0; h < 8; h += 4)
#pragma unroll
      for (int m = 0; m < 2; m++) {  // span 2, twiddle w_4^-m
        const e2 u = y[h + m], v = y[h + m + 2];
        y[h + m] = e2_add(u, v);
        const e2 d = e2_sub(u, v);
        y[h + m + 2] = m == 0 ? d : e2_mulf(d, wi2);
      }
#pragma unroll
    for (int h = 0; h < 8; h += 2) {
      const e2 u = y[h], v = y[h + 1];
      y[h] = e2_add(u, v);
      y[h + 1] = e2_sub(u, v);
    }
    // position p holds 8*c_{bitrev3(p)}: c0..c7 = y[0], y[4], y[2], y[6], y[1], y[5], y[3], y[7]
    res = y[7];
    res = e2_add(e2_mul(res, x), y[3]);
    res = e2_add(e2_mul(res, x), y[5]);
    res = e2_add(e2_mul(res, x), y[1]);
    res = e2_add(e2_mul(res, x), y[6]);
    res = e2_add(e2_mul(res, x), y[2]);
    res = e2_add(e2_mul(res, x), y[4]);
    res = e2_add(e2_mul(res, x), y[0]);
  } else {
    e2 y0 = ld_e2(a.ev, base), y1 = ld_e2(a.ev, base + ((size_t)1 << log_q)), y2 = ld_e2(a.ev, base + ((size_t)2 << log_q)),
       y3 = ld_e2(a.ev, base + ((size_t)3 << log_q));
    e2 s02 = e2_add(y0, y2), d02 = e2_sub(y0, y2), s13 = e2_add(y1, y3), d31w = e2_mulf(e2_sub(y3, y1), a.w4);
    e2 c0 = e2_add(s02, s13), c1 = e2_add(d02, d31w), c2 = e2_sub(s02, s13), c3 = e2_sub(d02, d31w);
    e2 x2 = e2_sqr(x), x3 = e2_mul(x2, x);
    res = e2_add(e2_add(c0, e2_mul(c1, x)), e2_add(e2_mul(c2, x2), e2_mul(c3, x3)));
  }
  st_e2(a.out, s, e2_mulf(res, a.inv_arity));
}

// coset-major [C][Nl] -> natural order [Nl*C]
__global__ void k_fri_to_natural(const u64* ev, u64* out, int log_rows, int cbits) {
  const size_t total = (size_t)1 << (log_rows + cbits);
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t j = i & (((size_t)1 << cbits) - 1), r = i >> cbits;
  st_e2(out, i, ld_e2(ev, (j << log_rows) + r));
}

void fri_leaf_hash(mh_ctx* c, const u64* ev, int log_rows, int cbits, int log_arity, u64* digests) {
  const size_t leaves = (size_t)1 << (log_rows - log_arity + cbits);
  ProfScope ps(c, "fri_leaf_hash", (double)leaves * (16.0 * (1 << log_arity) + 32.0));
  if (c->lmcs == MH_LMCS_RPO || c->lmcs == MH_LMCS_RPX)
    MH_LAUNCH(k_fri_leaf_hash_alg, dim3((unsigned)((leaves + 255) / 256)), dim3(256), 0, c->stream, ev, log_rows, cbits, log_arity, digests,
                       c->lmcs);
  else if (c->lmcs == MH_LMCS_KECCAK)
    MH_LAUNCH(k_fri_leaf_hash_kk, dim3((unsigned)((leaves + 255) / 256)), dim3(256), 0, c->stream, ev, log_rows, cbits, log_arity, digests);
  else if (c->lmcs == MH_LMCS_BLAKE3)
    MH_LAUNCH(k_fri_leaf_hash_b3, dim3((unsigned)((leaves + 255) / 256)), dim3(256), 0, c->stream, ev, log_rows, cbits, log_arity, digests);
  else
    MH_LAUNCH(k_fri_leaf_hash, dim3((unsigned)((leaves + 255) / 256)), dim3(256), 0, c->stream, ev, log_rows, cbits, log_arity,
                       digests);
}

void fri_fold(mh_ctx* c, const u64* ev, int log_rows, int cbits, int cbits_global, size_t coset0, int log_arity, e2 beta, u64* out) {
  MH_REQUIRE(log_arity >= 1 && log_arity <= 3, "FRI folding arity must be 2, 4 or 8");
  MH_REQUIRE(log_rows >= log_arity, "internal: FRI layer too short for a coset-major fold");
  const int logn = log_rows + cbits_global;  // size of the whole layer
  const size_t C = (size_t)1 << cbits;       // cosets stored here
  std::vector<u64> ci(C);
  const u64 wn_inv = gl_inv(gl_two_adic_generator(logn));
  u64 x = gl_pow(wn_inv, coset0);
  for (size_t j = 0; j < C; j++) {
    ci[j] = x;
    x = gl_mul(x, wn_inv);
  }
  DevBuf d(C * 8);
  c->h2d(d.p, ci.data(), C * 8);
  FoldArgs a{};
  a.ev = ev; a.out = out; a.log_rows = log_rows; a.cbits = cbits; a.log_arity = log_arity;
  a.tw_inv = log_rows ? c->twiddles(log_rows, true) : nullptr;
  a.coset_inv = d.u();
  a.beta = beta;
  a.w4 = gl_two_adic_generator(2);
  a.w8_inv = gl_inv(gl_two_adic_generator(3));
  a.inv_arity = gl_inv((u64)1 << log_arity);
  const size_t total = (size_t)1 << (log_rows + cbits - log_arity);
  {
    ProfScope ps(c, "fri_fold", (double)total * 16.0 * ((1 << log_arity) + 1));
    MH_LAUNCH(k_fri_fold, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, a);
  }
  HIP_CHECK(hipStreamSynchronize(c->stream));
}

void fri_to_natural(mh_ctx* c, const u64* ev, int log_rows, int cbits, u64* out) {
  const size_t total = (size_t)1 << (log_rows + cbits);
  MH_LAUNCH(k_fri_to_natural, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, ev, out, log_rows, cbits);
}

// ---- proof-of-work grinding ------------------------------------------------------------------------
// Device mirror of DuplexChallenger::check_witness on a snapshot of the host challenger: lane i tries
// witness base + i and the smallest passing witness of the window wins (deterministic result).
struct GrindArgs {
  u64 st[12];
  u64 in[8];
  int n_in, bits;
  u64 base;
  unsigned long long* best;
};
__global__ __launch_bounds__(256) void k_grind(GrindArgs a) {
  const u64 w = a.base + blockIdx.x * (u64)256 + threadIdx.x;
  if (w >= GL_P) return;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = a.st[i];
  // observe(w): append to the input buffer; the buffer then holds n_in + 1 <= 8 elements and either
  // way exactly one duplexing happens before the sample (at 8 inside observe, otherwise in sample).
  const int k = a.n_in + 1;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 v = (i < a.n_in) ? a.in[i] : (i == a.n_in ? w : 0);
    s[i] = (i < k) ? v : 0;
  }
  s[8] = gl_add(s[8], (u64)k);
  p2f_permute(s);
  const u64 x = s[7];  // first sample after a duplexing = rate[7]
  if (((x & 0xFFFFFFFFULL) & (((u64)1 << a.bits) - 1)) == 0) atomicMin(a.best, (unsigned long long)w);
}

__global__ __launch_bounds__(256) void k_grind_alg(GrindArgs a, int lmcs) {
  const u64 w = a.base + blockIdx.x * (u64)256 + threadIdx.x;
  if (w >= GL_P) return;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = a.st[i];
  const int k = a.n_in + 1;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 v = (i < a.n_in) ? a.in[i] : (i == a.n_in ? w : 0);
    s[i] = (i < k) ? v : 0;
  }
  s[8] = gl_add(s[8], (u64)k);
  alg_permute(lmcs, s);
  if (((s[7] & 0xFFFFFFFFULL) & (((u64)1 << a.bits) - 1)) == 0) atomicMin(a.best, (unsigned long long)w);
}

// ---- PoW search for the byte challengers (Blake3 / Keccak configurations) ----
// check_witness = observe(w): input := prefix || w (8 LE bytes); sample_bits: output = hash(input), the sampled u64 is built from
// the digest's LAST eight bytes, last byte lowest (HashChallenger pops from the end) = bswap64 of digest bytes 24..31.
// The host hashes every block that lies wholly inside the prefix once; a trial costs the last one or two blocks.
struct GrindB3Args {
  b3::Stream pre;       // after the full 64-byte blocks of the prefix
  uint8_t tail[64];     // the rest of the prefix (tail_len < 64 bytes)
  uint32_t tail_len;
  int bits;
  u64 base;
  unsigned long long* best;
};
__global__ __launch_bounds__(256) void k_grind_b3(GrindB3Args a) {
  const u64 w = a.base + blockIdx.x * (u64)256 + threadIdx.x;
  if (w >= GL_P) return;
  uint8_t buf[128];
#pragma unroll 1
  for (int i = 0; i < 128; i++) buf[i] = 0;
  for (uint32_t i = 0; i < a.tail_len; i++) buf[i] = a.tail[i];
  for (int k = 0; k < 8; k++) buf[a.tail_len + k] = (uint8_t)(w >> (8 * k));
  const uint32_t n = a.tail_len + 8;
  b3::Stream h = a.pre;
  uint32_t m[16], out[8];
  auto words = [&](const uint8_t* p) {
    for (int i = 0; i < 16; i++) m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
  };
  if (n <= 64) {
    words(buf);
    h.finish(m, n, out);
  } else {
    words(buf);
    h.block(m);
    words(buf + 64);
    h.finish(m, n - 64, out);
  }
  const u64 x = (u64)out[6] | ((u64)out[7] << 32);
  const u64 v = __builtin_bswap64(x);
  if ((v & (((u64)1 << a.bits) - 1)) == 0) atomicMin(a.best, (unsigned long long)w);
}
struct GrindKkArgs {
  u64 st[25];           // after the full 136-byte blocks of the prefix
  uint8_t tail[136];
  uint32_t tail_len;    // < 136
  int bits;
  u64 base;
  unsigned long long* best;
};
__global__ __launch_bounds__(256) void k_grind_kk(GrindKkArgs a) {
  const u64 w = a.base + blockIdx.x * (u64)256 + threadIdx.x;
  if (w >= GL_P) return;
  kk::Sponge256 s;
  for (int i = 0; i < 25; i++) s.st[i] = a.st[i];
  for (uint32_t i = 0; i < a.tail_len; i++) s.buf[i] = a.tail[i];
  s.fill = a.tail_len;
  uint8_t wb[8], d[32];
  for (int k = 0; k < 8; k++) wb[k] = (uint8_t)(w >> (8 * k));
  s.update(wb, 8);
  s.finish(0x01, d);
  u64 v = 0;
  for (int i = 0; i < 8; i++) v |= (u64)d[31 - i] << (8 * i);
  if ((v & (((u64)1 << a.bits) - 1)) == 0) atomicMin(a.best, (unsigned long long)w);
}
u64 fri_grind_bytes(mh_ctx* c, int lmcs, const std::vector<uint8_t>& prefix, int bits) {
  MH_REQUIRE(bits > 0 && bits <= 32 && (lmcs == MH_LMCS_BLAKE3 || lmcs == MH_LMCS_KECCAK), "bad byte-challenger grind request");
  DevBuf best(8);
  GrindB3Args ab{};
  GrindKkArgs ak{};
  if (lmcs == MH_LMCS_BLAKE3) {
    // the device stream (a kernel argument) keeps MAX_STACK chaining values: fewer than 2^MAX_STACK chunks of 1 KiB
    MH_REQUIRE(prefix.size() + 8 <= ((size_t)1024 << b3::MAX_STACK) - 1024,
               "Blake3 PoW search: more than 255 KiB observed since the last sample (device stream stack)");
    ab.pre.init();
    const size_t k = prefix.size() / 64;
    for (size_t b = 0; b < k; b++) {
      uint32_t m[16];
      for (int i = 0; i < 16; i++) {
        const uint8_t* p = prefix.data() + 64 * b + 4 * i;
        m[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
      }
      ab.pre.block(m);
    }
    ab.tail_len = (uint32_t)(prefix.size() - 64 * k);
    memcpy(ab.tail, prefix.data() + 64 * k, ab.tail_len);
    ab.bits = bits;
    ab.best = (unsigned long long*)best.p;
  } else {
    kk::Sponge256 s;
    s.init();
    const size_t k = prefix.size() / 136;
    s.update(prefix.data(), 136 * k);  // full blocks only: fill stays 0
    for (int i = 0; i < 25; i++) ak.st[i] = s.st[i];
    ak.tail_len = (uint32_t)(prefix.size() - 136 * k);
    memcpy(ak.tail, prefix.data() + 136 * k, ak.tail_len);
    ak.bits = bits;
    ak.best = (unsigned long long*)best.p;
  }
  u64 window = (u64)1 << (bits + 2);
  if (window < 65536) window = 65536;
  ProfScope ps(c, "grind", 0);
  for (u64 base = 0;; base += window) {
    unsigned long long init = ~0ULL;
    c->h2d(best.p, &init, 8);
    if (lmcs == MH_LMCS_BLAKE3) {
      ab.base = base;
      MH_LAUNCH(k_grind_b3, dim3((unsigned)(window / 256)), dim3(256), 0, c->stream, ab);
    } else {
      ak.base = base;
      MH_LAUNCH(k_grind_kk, dim3((unsigned)(window / 256)), dim3(256), 0, c->stream, ak);
    }
    unsigned long long got = 0;
    c->d2h(&got, best.p, 8);
    if (got != ~0ULL) return (u64)got;
  }
}

// Returns the smallest witness >= 0 accepted by `check_witness` for the given challenger snapshot.
u64 fri_grind(mh_ctx* c, const u64 st[12], const u64* in, int n_in, int bits) {
  MH_REQUIRE(bits > 0 && bits <= 32 && n_in >= 0 && n_in < 8, "bad grind request");
  DevBuf best(8);
  GrindArgs a{};
  for (int i = 0; i < 12; i++) a.st[i] = st[i];
  for (int i = 0; i < n_in; i++) a.in[i] = in[i];
  a.n_in = n_in; a.bits = bits;
  a.best = (unsigned long long*)best.p;
  // window ~ 4x the expected number of trials, at least one full wave of the chip
  u64 window = (u64)1 << (bits + 2);
  if (window < 65536) window = 65536;
  ProfScope ps(c, "grind", 0);
  for (u64 base = 0;; base += window) {
    unsigned long long init = ~0ULL;
    c->h2d(best.p, &init, 8);
    a.base = base;
    if (c->lmcs == MH_LMCS_RPO || c->lmcs == MH_LMCS_RPX)
      MH_LAUNCH(k_grind_alg, dim3((unsigned)(window / 256)), dim3(256), 0, c->stream, a, c->lmcs);
    else
      MH_LAUNCH(k_grind, dim3((unsigned)(window / 256)), dim3(256), 0, c->stream, a);
    unsigned long long got = 0;
    c->d2h(&got, best.p, 8);
    if (got != ~0ULL) return (u64)got;
  }
}
