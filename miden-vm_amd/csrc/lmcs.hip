// K2 / K3 / K10 / K13: Lifted Matrix Commitment Scheme with the Poseidon2 sponge, gfx950.
//
// Replaces crates/lifted-stark/src/lmcs/lifted_tree.rs:
//   :363-461 build_leaf_states_upsampled / absorb_matrix  -> k_leaf_absorb
//   :247-258 squeeze + bit-reverse                        -> folded into the layouts below
//   :472-511 compress_uniform                             -> k_compress
//   :155-180, :326-341 prove_batch / collect_rows         -> lmcs_open
// Sponge: crates/stateful-hasher/src/field_sponge.rs:41-64 (overwrite mode, rate 8, zero-pad the
// trailing partial chunk, state carried across matrices).  Compression: perm([L|R|0000])[0..4]
// (air/src/config.rs:213-220).
//
// Layout (MI355X-first, differs from the reference's bit-reversed row-major storage):
//   leaf slot q = j*N + r  <->  domain (natural) index i = r*B + j   (B = 2^log_blowup cosets)
//   Tree pairs (2p, 2p+1) in natural order = cosets (2j', 2j'+1) at equal r: for the first
//   log_blowup levels both children streams are unit-stride; afterwards the layer is in plain
//   natural order and children are adjacent (64 contiguous bytes per lane).
//   A lifted (shorter) matrix contributes row r mod N_m of coset j (i mod B*N_m).
// One sponge per lane, state in VGPRs; Poseidon2 is integer-ALU bound.
#include "ctx.hpp"
#include "../../include/midenhip.h"
#include "kernels.hpp"
#include "poseidon2_fast.cuh"
#include "poseidon2_lanes.cuh"
#include "poseidon2_quad.cuh"
#include "blake3.cuh"
#include "keccak.cuh"
#define RESCUE_FAST 1  // S-boxes through p2f_mulN (poseidon2_fast.cuh is included above)
#include "rescue.cuh"
#include "gl.cuh"
#include <algorithm>
#include <cstring>

static constexpr int LEAF_MAX_MATS = 8;
static constexpr int LEAF_THREADS = 256;

__global__ __launch_bounds__(256) void k_permute_soa(u64* st, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 s[12];
#pragma unroll
  for (int j = 0; j < 12; j++) s[j] = st[j * n + i];
  p2f_permute(s);
#pragma unroll
  for (int j = 0; j < 12; j++) st[j * n + i] = s[j];
}

void poseidon2_permute_device(mh_ctx* c, u64* states_soa, size_t n) {
  if (!n) return;
  MH_LAUNCH(k_permute_soa, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, states_soa, n);
}

// Register-only permutation chain: the VALU ceiling the hash kernels are measured against (bench.py).
__global__ __launch_bounds__(256) void k_perm_rate(u64* out, int iters, u64 seed) {
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_canon(seed * (i + 1) + threadIdx.x + blockIdx.x * 131);
#pragma unroll 1
  for (int i = 0; i < iters; i++) p2f_permute(s);
  u64 x = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) x ^= s[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
double poseidon2_register_rate(mh_ctx* c) {
  const int blocks = 2048, iters = 64;  // 8 waves per SIMD x 64 permutations: ~12 ms per launch
  DevBuf out((size_t)blocks * 256 * 8);
  hipEvent_t a = c->get_event(), b = c->get_event();
  MH_LAUNCH(k_perm_rate, dim3(blocks), dim3(256), 0, c->stream, out.u(), iters, 1ULL);  // warm-up (clocks)
  double best = 0;
  for (int rep = 0; rep < 3; rep++) {
    HIP_CHECK(hipEventRecord(a, c->stream));
    MH_LAUNCH(k_perm_rate, dim3(blocks), dim3(256), 0, c->stream, out.u(), iters, 12345ULL + rep);
    HIP_CHECK(hipEventRecord(b, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    best = std::max(best, (double)blocks * 256 * iters / (ms * 1e-3));
  }
  c->event_pool.push_back(a);
  c->event_pool.push_back(b);
  return best;
}

struct LeafMat {
  const u64* data;  // [width][B][N_m]
  u32 width;
};
struct LeafArgs {
  LeafMat m[LEAF_MAX_MATS];
  int n_mats;
  int log_blowup;
  int log_n;             // height (per coset) of this group
  const u64* state_in;   // [12][B * 2^log_n_prev] or null (zero state)
  int log_n_prev;
  u64* state_out;        // [12][B * 2^log_n] or null
  u64* digest_out;       // [B * 2^log_n][4] or null
  size_t q_begin, q_count;  // the leaves this launch hashes; q_count = 0: all of them
};

__global__ __launch_bounds__(LEAF_THREADS) void k_leaf_absorb(LeafArgs a) {
  const size_t leaves = (size_t)1 << (a.log_n + a.log_blowup);
  const size_t q = a.q_begin + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (q >= (a.q_count ? a.q_begin + a.q_count : leaves)) return;
  const size_t j = q >> a.log_n, r = q & (((size_t)1 << a.log_n) - 1);
  u64 s[12];
  if (a.state_in) {
    const size_t leaves_prev = (size_t)1 << (a.log_n_prev + a.log_blowup);
    const size_t qi = (j << a.log_n_prev) + (r & (((size_t)1 << a.log_n_prev) - 1));
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = a.state_in[i * leaves_prev + qi];
  } else {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
  }
  const size_t col_stride = leaves;  // B * N
#pragma unroll 1
  for (int mi = 0; mi < a.n_mats; mi++) {
    const u64* base = a.m[mi].data + q;
    const u32 w = a.m[mi].width;
#pragma unroll 1
    for (u32 c0 = 0; c0 < w; c0 += 8) {
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] = (c0 + k < w) ? base[(size_t)(c0 + k) * col_stride] : 0;
      p2f_permute(s);
    }
  }
  if (a.digest_out) {
    ulonglong2* o = reinterpret_cast<ulonglong2*>(a.digest_out + 4 * q);
    o[0] = make_ulonglong2(s[0], s[1]);
    o[1] = make_ulonglong2(s[2], s[3]);
  } else {
#pragma unroll
    for (int i = 0; i < 12; i++) a.state_out[i * leaves + q] = s[i];
  }
}

// ---- Blake3 LMCS (mh_ctx_set_lmcs(MH_LMCS_BLAKE3)): same launch structure, the state is the 32-byte digest itself ----
// One thread per leaf; per matrix one hash of  state || row  (chaining.rs:32-50).  Block 0 = the state and the first four
// felts, every later block eight felts (coalesced column reads, as in k_leaf_absorb); state_in / state_out / digest_out
// are all [leaves][4] u64 (a digest = its 32 bytes little-endian).
__global__ __launch_bounds__(LEAF_THREADS) void k_leaf_absorb_b3(LeafArgs a) {
  const size_t leaves = (size_t)1 << (a.log_n + a.log_blowup);
  const size_t q = a.q_begin + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (q >= (a.q_count ? a.q_begin + a.q_count : leaves)) return;
  const size_t j = q >> a.log_n, r = q & (((size_t)1 << a.log_n) - 1);
  uint32_t st[8];
  if (a.state_in) {
    const size_t qi = (j << a.log_n_prev) + (r & (((size_t)1 << a.log_n_prev) - 1));
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u64 v = a.state_in[4 * qi + i];
      st[2 * i] = (uint32_t)v;
      st[2 * i + 1] = (uint32_t)(v >> 32);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] = 0;
  }
  const size_t col_stride = leaves;
#pragma unroll 1
  for (int mi = 0; mi < a.n_mats; mi++) {
    const u64* base = a.m[mi].data + q;
    const u32 w = a.m[mi].width;
    const u32 total = 32 + 8 * w;                 // message bytes
    const u32 n_blocks = (total + 63) / 64;
    b3::Stream h;
    h.init();
    uint32_t m[16];
#pragma unroll 1
    for (u32 b = 0; b < n_blocks; b++) {
      // words [16b, 16b + 16) of the message: words 0..7 = the state, then two words per felt
      const int f0 = (int)(8 * b) - 4;  // first felt of this block (block 0: felts 0..3 sit in words 8..15)
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int f = f0 + k;
        u64 v = 0;
        if (f >= 0 && (u32)f < w) v = base[(size_t)f * col_stride];
        m[2 * k] = (uint32_t)v;
        m[2 * k + 1] = (uint32_t)(v >> 32);
      }
      if (b == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) m[k] = st[k];
      }
      if (b + 1 < n_blocks) h.block(m);
      else h.finish(m, total - 64 * b, st);
    }
  }
  u64* o = (a.digest_out ? a.digest_out : a.state_out) + 4 * q;
#pragma unroll
  for (int i = 0; i < 4; i++) o[i] = (u64)st[2 * i] | ((u64)st[2 * i + 1] << 32);
}
__global__ __launch_bounds__(256) void k_compress_b3(const u64* __restrict__ in, u64* __restrict__ out, size_t n_out, int log_n_coset) {
  const size_t q = blockIdx.x * (size_t)256 + threadIdx.x;
  if (q >= n_out) return;
  size_t l, rgt;
  if (log_n_coset >= 0) {
    const size_t N = (size_t)1 << log_n_coset;
    const size_t jp = q >> log_n_coset, r = q & (N - 1);
    l = ((2 * jp) << log_n_coset) + r;
    rgt = l + N;
  } else {
    l = 2 * q;
    rgt = l + 1;
  }
  uint32_t a[8], b[8], o[8];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const u64 x = in[4 * l + i], y = in[4 * rgt + i];
    a[2 * i] = (uint32_t)x; a[2 * i + 1] = (uint32_t)(x >> 32);
    b[2 * i] = (uint32_t)y; b[2 * i + 1] = (uint32_t)(y >> 32);
  }
  b3::compress_pair(a, b, o);
#pragma unroll
  for (int i = 0; i < 4; i++) out[4 * q + i] = (u64)o[2 * i] | ((u64)o[2 * i + 1] << 32);
}

// The top of a Blake3 tree in ONE launch: levels d_start .. 0 (<= 256 nodes each, natural order) by one workgroup, a barrier
// between levels.  A node is one 64-byte block (~0.7 k instructions): at these sizes a level is all launch latency, 8-9 launches
// per tree, 10 trees per proof.  Layer d starts at digest (2^(L+1) - 2^(d+1)) of the node array (lmcs_alloc_layers).
__global__ __launch_bounds__(256) void k_compress_b3_top(u64* nodes, int log_height, int d_start) {
  const size_t two_l1 = (size_t)2 << log_height;
  for (int d = d_start; d >= 0; d--) {
    const size_t n_out = (size_t)1 << d;
    const u64* in = nodes + 4 * (two_l1 - ((size_t)2 << (d + 1)));
    u64* out = nodes + 4 * (two_l1 - ((size_t)2 << d));
    const size_t q = threadIdx.x;
    if (q < n_out) {
      uint32_t a[8], b[8], o[8];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const u64 x = in[8 * q + i], y = in[8 * q + 4 + i];
        a[2 * i] = (uint32_t)x; a[2 * i + 1] = (uint32_t)(x >> 32);
        b[2 * i] = (uint32_t)y; b[2 * i + 1] = (uint32_t)(y >> 32);
      }
      b3::compress_pair(a, b, o);
#pragma unroll
      for (int i = 0; i < 4; i++) out[4 * q + i] = (u64)o[2 * i] | ((u64)o[2 * i + 1] << 32);
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ---- RPO / RPX (MH_LMCS_RPO, MH_LMCS_RPX): the sponge and the compression of the Poseidon2 LMCS with the Rescue permutations
// (rescue.cuh; plain arithmetic, one state per lane -- supported, not tuned) ----
__global__ __launch_bounds__(LEAF_THREADS) void k_leaf_absorb_alg(LeafArgs a, int lmcs) {
  const size_t leaves = (size_t)1 << (a.log_n + a.log_blowup);
  const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (q >= leaves) return;
  const size_t j = q >> a.log_n, r = q & (((size_t)1 << a.log_n) - 1);
  u64 s[12];
  if (a.state_in) {
    const size_t leaves_prev = (size_t)1 << (a.log_n_prev + a.log_blowup);
    const size_t qi = (j << a.log_n_prev) + (r & (((size_t)1 << a.log_n_prev) - 1));
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = a.state_in[i * leaves_prev + qi];
  } else {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
  }
  const size_t col_stride = leaves;
#pragma unroll 1
  for (int mi = 0; mi < a.n_mats; mi++) {
    const u64* base = a.m[mi].data + q;
    const u32 w = a.m[mi].width;
#pragma unroll 1
    for (u32 c0 = 0; c0 < w; c0 += 8) {
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] = (c0 + k < w) ? base[(size_t)(c0 + k) * col_stride] : 0;
      alg_permute(lmcs, s);
    }
  }
  if (a.digest_out) {
#pragma unroll
    for (int i = 0; i < 4; i++) a.digest_out[4 * q + i] = s[i];
  } else {
#pragma unroll
    for (int i = 0; i < 12; i++) a.state_out[i * leaves + q] = s[i];
  }
}
__global__ __launch_bounds__(256) void k_compress_alg(const u64* __restrict__ in, u64* __restrict__ out, size_t n_out, int log_n_coset,
                                                      int lmcs) {
  const size_t q = blockIdx.x * (size_t)256 + threadIdx.x;
  if (q >= n_out) return;
  size_t l, rgt;
  if (log_n_coset >= 0) {
    const size_t N = (size_t)1 << log_n_coset;
    const size_t jp = q >> log_n_coset, r = q & (N - 1);
    l = ((2 * jp) << log_n_coset) + r;
    rgt = l + N;
  } else {
    l = 2 * q;
    rgt = l + 1;
  }
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    s[i] = in[4 * l + i];
    s[4 + i] = in[4 * rgt + i];
    s[8 + i] = 0;
  }
  alg_permute(lmcs, s);
#pragma unroll
  for (int i = 0; i < 4; i++) out[4 * q + i] = s[i];
}

// ---- Keccak LMCS (MH_LMCS_KECCAK): the sponge of k_leaf_absorb with Keccak-f[1600], 25 lanes, rate 17 ----
__global__ __launch_bounds__(LEAF_THREADS) void k_leaf_absorb_kk(LeafArgs a) {
  const size_t leaves = (size_t)1 << (a.log_n + a.log_blowup);
  const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (q >= leaves) return;
  const size_t j = q >> a.log_n, r = q & (((size_t)1 << a.log_n) - 1);
  uint64_t s[25];
  if (a.state_in) {
    const size_t leaves_prev = (size_t)1 << (a.log_n_prev + a.log_blowup);
    const size_t qi = (j << a.log_n_prev) + (r & (((size_t)1 << a.log_n_prev) - 1));
#pragma unroll
    for (int i = 0; i < 25; i++) s[i] = a.state_in[i * leaves_prev + qi];
  } else {
#pragma unroll
    for (int i = 0; i < 25; i++) s[i] = 0;
  }
  const size_t col_stride = leaves;
#pragma unroll 1
  for (int mi = 0; mi < a.n_mats; mi++) {
    const u64* base = a.m[mi].data + q;
    const u32 w = a.m[mi].width;
#pragma unroll 1
    for (u32 c0 = 0; c0 < w; c0 += 17) {
#pragma unroll
      for (int k = 0; k < 17; k++) s[k] = (c0 + k < w) ? base[(size_t)(c0 + k) * col_stride] : 0;
      kk::f1600(s);
    }
  }
  if (a.digest_out) {
#pragma unroll
    for (int i = 0; i < 4; i++) a.digest_out[4 * q + i] = s[i];
  } else {
#pragma unroll
    for (int i = 0; i < 25; i++) a.state_out[i * leaves + q] = s[i];
  }
}
__global__ __launch_bounds__(256) void k_compress_kk(const u64* __restrict__ in, u64* __restrict__ out, size_t n_out, int log_n_coset) {
  const size_t q = blockIdx.x * (size_t)256 + threadIdx.x;
  if (q >= n_out) return;
  size_t l, rgt;
  if (log_n_coset >= 0) {
    const size_t N = (size_t)1 << log_n_coset;
    const size_t jp = q >> log_n_coset, r = q & (N - 1);
    l = ((2 * jp) << log_n_coset) + r;
    rgt = l + N;
  } else {
    l = 2 * q;
    rgt = l + 1;
  }
  uint64_t a[4], b[4], o[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    a[i] = in[4 * l + i];
    b[i] = in[4 * rgt + i];
  }
  kk::compress_pair(a, b, o);
#pragma unroll
  for (int i = 0; i < 4; i++) out[4 * q + i] = o[i];
}

// out[q] = compress(in[left(q)], in[left(q) + sib]) ; coset phase: left = (2j')*N + r, sib = N;
// natural phase: left = 2q, sib = 1.
__global__ __launch_bounds__(256) void k_compress(const u64* __restrict__ in, u64* __restrict__ out, size_t n_out, int log_n_coset) {
  const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (q >= n_out) return;
  size_t l, rgt;
  if (log_n_coset >= 0) {
    size_t N = (size_t)1 << log_n_coset;
    size_t jp = q >> log_n_coset, r = q & (N - 1);
    l = ((2 * jp) << log_n_coset) + r;
    rgt = l + N;
  } else {
    l = 2 * q;
    rgt = l + 1;
  }
  const ulonglong2* pl = reinterpret_cast<const ulonglong2*>(in + 4 * l);
  const ulonglong2* pr = reinterpret_cast<const ulonglong2*>(in + 4 * rgt);
  ulonglong2 l0 = pl[0], l1 = pl[1], r0 = pr[0], r1 = pr[1];
  u64 s[12] = {l0.x, l0.y, l1.x, l1.y, r0.x, r0.y, r1.x, r1.y, 0, 0, 0, 0};
  p2f_permute(s);
  ulonglong2* o = reinterpret_cast<ulonglong2*>(out + 4 * q);
  o[0] = make_ulonglong2(s[0], s[1]);
  o[1] = make_ulonglong2(s[2], s[3]);
}

// Same compression with one state element per lane (16 lanes per node): the low-latency form for the
// small layers near the root (poseidon2_lanes.cuh).
static size_t compress_lanes_max_nodes() {
  static const size_t v = [] {
    const char* e = getenv("MH_LANES_MAX_NODES");  // experiments
    return e ? (size_t)atol(e) : (size_t)8192;
  }();
  return v;
}
__global__ __launch_bounds__(256) void k_compress_lanes(const u64* __restrict__ in, u64* __restrict__ out, size_t n_out, int log_n_coset) {
  const size_t node = (blockIdx.x * (size_t)256 + threadIdx.x) >> 4;
  const int g = threadIdx.x & 15;
  const bool active = node < n_out;
  const size_t q = active ? node : 0;  // idle groups recompute node 0 (keeps every wave converged)
  size_t l, rgt;
  if (log_n_coset >= 0) {
    const size_t N = (size_t)1 << log_n_coset;
    const size_t jp = q >> log_n_coset, r = q & (N - 1);
    l = ((2 * jp) << log_n_coset) + r;
    rgt = l + N;
  } else {
    l = 2 * q;
    rgt = l + 1;
  }
  // lanes 0-3: the left digest, 4-7: the right one, the rest zeros -- ONE masked load (two branches were two memory latencies in a row)
  const u64* src = in + 4 * (g < 4 ? l : rgt) + (g & 3);
  u64 s = g < 8 ? *src : 0;
  s = p2l_permute(s);
  if (active && g < 4) out[4 * q + g] = s;
}

// The same compression with four lanes per node (three state elements per lane, poseidon2_quad.cuh): 16 nodes per wave, for the
// levels between the 16-lane form (latency-optimal below ~2^13 nodes) and the state-per-lane form (throughput-optimal above ~2^15).
static size_t compress_quad_min_nodes() {
  static const size_t v = [] {
    const char* e = getenv("MH_QUAD_MIN_NODES");  // experiments (measured: 2^14 .. 2^15 is where it pays, profiles/scripts/exp_quad.sh)
    return e ? (size_t)atol(e) : (size_t)16384;
  }();
  return v;
}
static size_t compress_quad_max_nodes() {
  static const size_t v = [] {
    const char* e = getenv("MH_QUAD_MAX_NODES");  // experiments; 0 disables the form
    return e ? (size_t)atol(e) : (size_t)32768;
  }();
  return v;
}
__global__ __launch_bounds__(256) void k_compress_quad(const u64* __restrict__ in, u64* __restrict__ out, size_t n_out, int log_n_coset) {
  const size_t node = (blockIdx.x * (size_t)256 + threadIdx.x) >> 2;
  const int j = threadIdx.x & 3;
  const bool active = node < n_out;
  const size_t q = active ? node : 0;  // idle quads recompute node 0 (keeps every wave converged)
  size_t l, rgt;
  if (log_n_coset >= 0) {
    const size_t N = (size_t)1 << log_n_coset;
    const size_t jp = q >> log_n_coset, r = q & (N - 1);
    l = ((2 * jp) << log_n_coset) + r;
    rgt = l + N;
  } else {
    l = 2 * q;
    rgt = l + 1;
  }
  u64 s[3] = {in[4 * l + j], in[4 * rgt + j], 0};  // elements j (left digest), 4 + j (right digest), 8 + j (capacity)
  p2q_permute(s);
  if (active) out[4 * q + j] = s[0];
}

void lmcs_alloc_layers(mh_tree* t, int log_height) {
  t->log_height = log_height;
  // layers: depth L (H nodes) first, then L-1, ..., 0
  t->layer_off.assign(log_height + 1, 0);
  size_t off = 0;
  for (int d = log_height; d >= 0; d--) {
    t->layer_off[d] = off;
    off += (size_t)1 << d;
  }
  t->nodes.alloc(off * 32);
}
u64* lmcs_leaf_layer(mh_tree* t) { return t->nodes.u() + 4 * t->layer_off[t->log_height]; }

// The top levels of a tree are finished on the host: the level of 2^k nodes comes back in the copy that fetched the root anyway, the
// host compresses it with eight nodes per AVX-512 permutation (p2_host_simd.cpp; ~2 us per eight nodes, scalar ~2 us per node), and a
// level of a few hundred nodes costs the device a lone-wave permutation (17.6 us with a state spread over 16 lanes) whatever its
// size.  The nodes go back into the layer buffer (openings read sibling digests from it) with an asynchronous copy from a page-locked
// buffer of the context.
static constexpr int LMCS_HOST_TOP_MAX = 10;
static int lmcs_host_top_levels(int lmcs) {
  static const int forced = [] {
    const char* e = getenv("MH_HOST_TOP");  // experiments: 0 (off) .. 10, every hasher
    const int x = e ? atoi(e) : -1;
    return x > LMCS_HOST_TOP_MAX ? LMCS_HOST_TOP_MAX : x;
  }();
  if (forced >= 0) return forced;
  return lmcs == MH_LMCS_POSEIDON2 && p2_host_simd_available() ? 6 : 5;  // scalar host compressions: five levels (31 nodes)
}

void lmcs_compress_layers(mh_ctx* c, mh_tree* t) {
  t->lmcs = c->lmcs;
  const int lb = t->log_blowup;
  const size_t H = (size_t)1 << t->log_height;
  // levels [0, host_top) on the host: only natural-order levels (above the coset phase); Blake3 has its one-launch top
  const int want = lmcs_host_top_levels(c->lmcs);
  const int host_top = (c->lmcs != MH_LMCS_BLAKE3 && t->log_height - lb >= want && t->log_height > want) ? want : 0;
  {
    ProfScope ps(c, "lmcs_compress", 96.0 * (double)H);
    for (int d = t->log_height - 1; d >= host_top; d--) {
      size_t n_out = (size_t)1 << d;
      int cbits_child = (d + 1) - (t->log_height - lb);  // coset bits of the child layer
      int log_n_coset = cbits_child > 0 ? (t->log_height - lb) : -1;
      if (c->lmcs == MH_LMCS_RPO || c->lmcs == MH_LMCS_RPX)
        MH_LAUNCH(k_compress_alg, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, c->stream,
                           t->nodes.u() + 4 * t->layer_off[d + 1], t->nodes.u() + 4 * t->layer_off[d], n_out, log_n_coset, c->lmcs);
      else if (c->lmcs == MH_LMCS_KECCAK)
        MH_LAUNCH(k_compress_kk, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, c->stream,
                           t->nodes.u() + 4 * t->layer_off[d + 1], t->nodes.u() + 4 * t->layer_off[d], n_out, log_n_coset);
      else if (c->lmcs == MH_LMCS_BLAKE3 && n_out <= 256 && log_n_coset < 0) {
        MH_LAUNCH(k_compress_b3_top, dim3(1), dim3(256), 0, c->stream, t->nodes.u(), t->log_height, d);
        break;  // that launch went down to the root
      } else if (c->lmcs == MH_LMCS_BLAKE3)
        MH_LAUNCH(k_compress_b3, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, c->stream,
                           t->nodes.u() + 4 * t->layer_off[d + 1], t->nodes.u() + 4 * t->layer_off[d], n_out, log_n_coset);
      else if (n_out >= compress_quad_min_nodes() && n_out <= compress_quad_max_nodes())
        MH_LAUNCH(k_compress_quad, dim3((unsigned)((n_out * 4 + 255) / 256)), dim3(256), 0, c->stream,
                           t->nodes.u() + 4 * t->layer_off[d + 1], t->nodes.u() + 4 * t->layer_off[d], n_out, log_n_coset);
      else if (n_out <= compress_lanes_max_nodes())
        MH_LAUNCH(k_compress_lanes, dim3((unsigned)((n_out * 16 + 255) / 256)), dim3(256), 0, c->stream,
                           t->nodes.u() + 4 * t->layer_off[d + 1], t->nodes.u() + 4 * t->layer_off[d], n_out, log_n_coset);
      else
        MH_LAUNCH(k_compress, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, c->stream,
                           t->nodes.u() + 4 * t->layer_off[d + 1], t->nodes.u() + 4 * t->layer_off[d], n_out, log_n_coset);
    }
  }
  if (!host_top) {
    c->d2h(t->root, t->nodes.u() + 4 * t->layer_off[0], 32);
    return;
  }
  // page-locked: [level read back: 2^host_top nodes][the host_top layers above it: 2^host_top - 1 nodes]
  const size_t n_level = (size_t)1 << host_top;
  if (!c->pinned_top) HIP_CHECK(hipHostMalloc(&c->pinned_top, ((size_t)64 << LMCS_HOST_TOP_MAX), hipHostMallocDefault));
  u64* level = static_cast<u64*>(c->pinned_top);
  u64* top = level + 4 * n_level;
  c->d2h(level, t->nodes.u() + 4 * t->layer_off[host_top], 32 * n_level);
  // layers host_top - 1 .. 0 lie one after the other at the end of the node buffer (lmcs_alloc_layers): the same order in `top`
  const u64* child = level;
  size_t off = 0;
  for (int d = host_top - 1; d >= 0; d--) {
    lmcs_host_compress_level(c->lmcs, child, (size_t)1 << d, top + off);
    child = top + off;
    off += (size_t)4 << d;
  }
  memcpy(t->root, top + off - 4, 32);
  // the buffer is written again only after the next tree's blocking read-back on this stream: this copy has completed by then
  HIP_CHECK(hipMemcpyAsync(t->nodes.u() + 4 * t->layer_off[host_top - 1], top, off * 8, hipMemcpyHostToDevice, c->stream));
}

void lmcs_build_tree(mh_ctx* c, mh_tree* t) {
  MH_REQUIRE(!t->mats.empty(), "cannot commit empty batch");
  lmcs_alloc_layers(t, t->mats.back().log_n + t->log_blowup);
  lmcs_hash_leaves(c, t->mats, t->log_blowup, lmcs_leaf_layer(t));
  lmcs_compress_layers(c, t);
}

// One launch over the leaves [q_begin, q_begin + q_count): every matrix has the same height (no state chained between launches) and
// the hasher's leaf kernel takes a range (Poseidon2, Blake3).
bool lmcs_leaves_rangeable(mh_ctx* c, const std::vector<LdeMatrix>& mats) {
  if (mats.empty() || mats.size() > (size_t)LEAF_MAX_MATS) return false;
  if (c->lmcs != MH_LMCS_POSEIDON2 && c->lmcs != MH_LMCS_BLAKE3) return false;
  for (const LdeMatrix& m : mats)
    if (m.log_n != mats[0].log_n || m.log_cosets != mats[0].log_cosets) return false;
  return true;
}
void lmcs_hash_leaves_range(mh_ctx* c, const std::vector<LdeMatrix>& mats, int lb, u64* digests, size_t q_begin, size_t q_count) {
  MH_REQUIRE(lmcs_leaves_rangeable(c, mats), "internal: leaves cannot be hashed by ranges");
  LeafArgs a{};
  double bytes = 32.0 * (double)q_count;
  for (const LdeMatrix& m : mats) {
    MH_REQUIRE(c->lmcs != MH_LMCS_BLAKE3 || m.width <= (((size_t)1024 << b3::MAX_STACK) - 32) / 8,
               "matrix too wide for the Blake3 LMCS (a row must fit 256 KiB)");
    a.m[a.n_mats].data = m.lde.u();
    a.m[a.n_mats].width = (u32)m.width;
    bytes += (double)m.width * 8.0 * (double)q_count;
    a.n_mats++;
  }
  a.log_blowup = lb;
  a.log_n = mats[0].log_n;
  a.digest_out = digests;
  a.q_begin = q_begin;
  a.q_count = q_count;
  MH_REQUIRE(q_count > 0 && q_begin + q_count <= ((size_t)1 << (a.log_n + lb)), "internal: leaf range out of bounds");
  ProfScope ps(c, "lmcs_leaf_absorb", bytes);
  const dim3 grid((unsigned)((q_count + LEAF_THREADS - 1) / LEAF_THREADS));
  if (c->lmcs == MH_LMCS_BLAKE3) MH_LAUNCH(k_leaf_absorb_b3, grid, dim3(LEAF_THREADS), 0, c->stream, a);
  else MH_LAUNCH(k_leaf_absorb, grid, dim3(LEAF_THREADS), 0, c->stream, a);
}

// Leaf digests of a group of LDE matrices holding 2^lb cosets each (all cosets, or one rank's share
// of them in the coset-sharded commit): digest slot j*N + r, N = tallest height.
void lmcs_hash_leaves(mh_ctx* c, const std::vector<LdeMatrix>& mats, int lb, u64* digests) {
  MH_REQUIRE(!mats.empty(), "cannot commit empty batch");
  for (size_t i = 1; i < mats.size(); i++)
    MH_REQUIRE(mats[i - 1].log_n <= mats[i].log_n, "matrices must be sorted by ascending height");
  // one launch per (height group, <=8 matrices) chained through a state buffer
  DevBuf st_a, st_b;
  const u64* state_in = nullptr;
  int log_n_prev = 0;
  size_t i = 0;
  const size_t nm = mats.size();
  while (i < nm) {
    int ln = mats[i].log_n;
    LeafArgs a{};
    a.n_mats = 0;
    double bytes = 0;
    while (i < nm && mats[i].log_n == ln && a.n_mats < LEAF_MAX_MATS) {
      // a row is one BLAKE3 message of 32 + 8 * width bytes; b3::Stream keeps at most 2^MAX_STACK chunks of 1 KiB on its stack
      MH_REQUIRE(c->lmcs != MH_LMCS_BLAKE3 || mats[i].width <= (((size_t)1024 << b3::MAX_STACK) - 32) / 8,
                 "matrix too wide for the Blake3 LMCS (a row must fit 256 KiB)");
      a.m[a.n_mats].data = mats[i].lde.u();
      a.m[a.n_mats].width = (u32)mats[i].width;
      bytes += (double)mats[i].width * 8.0 * (double)((size_t)1 << (ln + lb));
      a.n_mats++;
      i++;
    }
    a.log_blowup = lb;
    a.log_n = ln;
    a.state_in = state_in;
    a.log_n_prev = log_n_prev;
    const bool last = (i == nm);
    const size_t leaves = (size_t)1 << (ln + lb);
    DevBuf& outbuf = (state_in == st_a.u()) ? st_b : st_a;
    if (last) {
      a.digest_out = digests;
      a.state_out = nullptr;
      bytes += 32.0 * leaves;
    } else {
      outbuf.alloc(leaves * (c->lmcs == MH_LMCS_KECCAK ? 25 : 12) * 8);  // sponge state per leaf (Blake3: the first 4 words of it)
      a.state_out = outbuf.u();
      a.digest_out = nullptr;
      bytes += 96.0 * leaves;
    }
    if (state_in) bytes += 96.0 * leaves;
    {
      ProfScope ps(c, "lmcs_leaf_absorb", bytes);
      if (c->lmcs == MH_LMCS_RPO || c->lmcs == MH_LMCS_RPX)
        MH_LAUNCH(k_leaf_absorb_alg, dim3((unsigned)((leaves + LEAF_THREADS - 1) / LEAF_THREADS)), dim3(LEAF_THREADS), 0, c->stream, a,
                           c->lmcs);
      else if (c->lmcs == MH_LMCS_KECCAK)
        MH_LAUNCH(k_leaf_absorb_kk, dim3((unsigned)((leaves + LEAF_THREADS - 1) / LEAF_THREADS)), dim3(LEAF_THREADS), 0, c->stream, a);
      else if (c->lmcs == MH_LMCS_BLAKE3)
        MH_LAUNCH(k_leaf_absorb_b3, dim3((unsigned)((leaves + LEAF_THREADS - 1) / LEAF_THREADS)), dim3(LEAF_THREADS), 0, c->stream, a);
      else
        MH_LAUNCH(k_leaf_absorb, dim3((unsigned)((leaves + LEAF_THREADS - 1) / LEAF_THREADS)), dim3(LEAF_THREADS), 0,
                           c->stream, a);
    }
    state_in = a.state_out;
    log_n_prev = ln;
  }
}

// tree_indices.rs:185-240 (MissingSiblingsIter): bottom-up, left-to-right.
std::vector<std::pair<int, size_t>> lmcs_missing_siblings(const std::vector<size_t>& idx, int depth) {
  std::vector<std::pair<int, size_t>> out;
  std::vector<size_t> cur(idx);
  for (int d = depth; d > 0; d--) {
    std::vector<size_t> next;
    next.reserve(cur.size());
    for (size_t i = 0; i < cur.size();) {
      size_t node = cur[i], sib = node ^ 1;
      bool present = (i + 1 < cur.size() && cur[i + 1] == sib);
      if (next.empty() || next.back() != (node >> 1)) next.push_back(node >> 1);
      if (!present) out.emplace_back(d, sib);
      i += present ? 2 : 1;
    }
    cur.swap(next);
  }
  return out;
}

// gather list: out[k] = src[k] dereferenced
__global__ void k_gather(const u64* const* __restrict__ ptrs, u64* __restrict__ out, size_t n) {
  size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (k < n) out[k] = ptrs[k] ? *ptrs[k] : 0;
}

// The gather list of one tree's opening, appended to `ptrs` (one device pointer per opened word, null = not stored on this rank /
// filled from the host-side cap): first the rows, then the missing sibling digests bottom-up (tree_indices.rs:185-240).
OpenPlan lmcs_open_plan(const mh_tree* t, const std::vector<size_t>& idx, size_t alignment, const Dist* dist,
                        std::vector<const u64*>& ptrs) {
  // Sharded proofs: every opened value lives on exactly one rank (rows: the rank that stores the coset;
  // tree nodes: the rank whose row range covers them; the cap: known to all).  Each rank gathers what
  // it owns, leaves zeros elsewhere, and one all-reduce (sum) assembles the identical answer everywhere.
  OpenPlan plan;
  plan.first = ptrs.size();
  const bool distributed = dist && dist->on();
  const bool i_contribute = !distributed || t->shard_logG > 0 || dist->rank == 0;  // unsharded tree in a sharded proof: rank 0
  const int lb = t->log_blowup, G = t->shard_logG;
  const int full_height = t->log_height + G;
  const size_t Bm = ((size_t)1 << lb) - 1;
  for (size_t i : idx) {
    MH_REQUIRE(i < ((size_t)1 << full_height), "opening index out of range");
    size_t j = i & Bm, r = i >> lb;
    if (t->fri_log_rows >= 0) {
      // FRI round tree: the leaf row is the arity-coset of EF values, bit-reversed inside the row
      // (fri/prover.rs:117-142), flattened [c0, c1]; FRI trees are unaligned (build_tree).
      const int log_q = t->fri_log_rows - t->fri_log_arity;
      const bool mine = i_contribute && j >= t->fri_coset0 && j < t->fri_coset0 + ((size_t)1 << t->fri_log_cosets);
      for (u32 p = 0; p < (1u << t->fri_log_arity); p++) {
        size_t e = ((j - t->fri_coset0) << t->fri_log_rows) + r + ((size_t)bitrev32(p, t->fri_log_arity) << log_q);
        ptrs.push_back(mine ? t->fri_layer.u() + 2 * e : nullptr);
        ptrs.push_back(mine ? t->fri_layer.u() + 2 * e + 1 : nullptr);
      }
      continue;
    }
    for (const LdeMatrix& m : t->mats) {
      size_t N = (size_t)1 << m.log_n;
      size_t rm = r & (N - 1);
      const bool mine = i_contribute && j >= m.coset0 && j < m.coset0 + ((size_t)1 << m.log_cosets);
      for (size_t cidx = 0; cidx < m.width; cidx++)
        ptrs.push_back(mine ? m.lde.u() + (((cidx << m.log_cosets) + (j - m.coset0)) << m.log_n) + rm : nullptr);
      size_t padded = (m.width + alignment - 1) / alignment * alignment;
      for (size_t k = m.width; k < padded; k++) ptrs.push_back(nullptr);
    }
  }
  plan.n_fields = ptrs.size() - plan.first;
  auto sib = lmcs_missing_siblings(idx, full_height);
  for (auto& s : sib) {
    const int d = s.first;
    const size_t p = s.second;
    if (G > 0 && d <= G) {
      plan.cap_fill.emplace_back(ptrs.size() - plan.first, t->cap.data() + 4 * ((((size_t)1) << d) - 1 + p));
      for (int k = 0; k < 4; k++) ptrs.push_back(nullptr);
      continue;
    }
    const bool mine = i_contribute && (G == 0 || (p >> (d - G)) == (size_t)t->shard_rank);
    const size_t pl = G ? (p & ((((size_t)1) << (d - G)) - 1)) : p;
    const u64* q = t->nodes.u() + 4 * (t->layer_off[d - G] + t->node_slot(d - G, pl));
    for (int k = 0; k < 4; k++) ptrs.push_back(mine ? q + k : nullptr);
  }
  plan.n = ptrs.size() - plan.first;
  return plan;
}

// ONE gather (and, in a sharded proof, one all-reduce) for the gather lists of any number of trees: host[i] = *ptrs[i] (0 for null).
// The query phase opens ten trees: per tree this was a pageable upload of the list, a kernel and a blocking read-back.
void lmcs_open_run(mh_ctx* c, const std::vector<const u64*>& ptrs, const Dist* dist, std::vector<u64>& host) {
  const size_t n = ptrs.size();
  host.assign(n, 0);
  if (!n) return;
  DevBuf dptrs(n * 8), dout(n * 8);
  c->h2d(dptrs.p, ptrs.data(), n * 8);  // through the page-locked ring: a plain DMA
  MH_LAUNCH(k_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const u64* const*)dptrs.p, dout.u(), n);
  if (dist && dist->on()) dist->all_reduce_sum(c, dout.u(), n);
  c->d2h(host.data(), dout.p, n * 8);
}

// A tree's slice of the gathered words -> its opened rows and sibling digests (cap digests from the host copy of the cap).
void lmcs_open_take(const OpenPlan& plan, std::vector<u64>& host, std::vector<u64>& fields, std::vector<u64>& commitments) {
  for (auto& cf : plan.cap_fill)
    for (int k = 0; k < 4; k++) host[plan.first + cf.first + k] = cf.second[k];
  fields.assign(host.begin() + plan.first, host.begin() + plan.first + plan.n_fields);
  commitments.assign(host.begin() + plan.first + plan.n_fields, host.begin() + plan.first + plan.n);
}

void lmcs_open(mh_ctx* c, const mh_tree* t, const std::vector<size_t>& idx, size_t alignment, std::vector<u64>& fields,
               std::vector<u64>& commitments, const Dist* dist) {
  std::vector<const u64*> ptrs;
  const OpenPlan plan = lmcs_open_plan(t, idx, alignment, dist, ptrs);
  std::vector<u64> host;
  lmcs_open_run(c, ptrs, dist, host);
  lmcs_open_take(plan, host, fields, commitments);
}

// [2^lbl][2^log_rows] digests -> [G][2^lbl][rows/G]: the block of every destination rank contiguous
__global__ void k_repack_digests(const ulonglong2* __restrict__ in, ulonglong2* __restrict__ out, int lbl, int log_rows, int logG) {
  const size_t total = (size_t)1 << (lbl + log_rows);
  const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (q >= total) return;
  const int log_rl = log_rows - logG;
  const size_t jl = q >> log_rows, r = q & (((size_t)1 << log_rows) - 1);
  const size_t d = r >> log_rl, rr = r & (((size_t)1 << log_rl) - 1);
  const size_t o = (((d << lbl) + jl) << log_rl) + rr;
  out[2 * o] = in[2 * q];
  out[2 * o + 1] = in[2 * q + 1];
}

// One 2-to-1 node on the HOST under hasher `lmcs` (MH_LMCS_*): pair = left || right (8 words), out = 4 words.  Byte
// digests (Blake3, Keccak) are never canonicalised: their words are not field elements.
void lmcs_host_compress(int lmcs, const u64* pair, u64* out) {
  u64 st[12] = {0};
  if (lmcs == MH_LMCS_KECCAK) {
    kk::compress_pair(pair, pair + 4, st);
  } else if (lmcs == MH_LMCS_BLAKE3) {
    uint8_t dg[32];
    b3::hash_bytes(reinterpret_cast<const uint8_t*>(pair), 64, dg);  // little-endian host: a digest's u64s are its bytes
    memcpy(st, dg, 32);
  } else {
    for (int k = 0; k < 8; k++) st[k] = pair[k];
    alg_permute(lmcs, st);
  }
  memcpy(out, st, 32);
}

void lmcs_host_compress_level(int lmcs, const u64* children, size_t n_out, u64* out) {
  size_t q = 0;
  if (lmcs == MH_LMCS_POSEIDON2 && p2_host_simd_available()) {
    for (; q < n_out; q += 8) {
      const int n = (int)std::min<size_t>(8, n_out - q);
      u64 pairs[64];
      for (int j = 0; j < 8 * n; j++) pairs[j] = gl_canon(children[8 * q + j]);
      p2_host_compress8(pairs, n, out + 4 * q);
    }
    return;
  }
  for (; q < n_out; q++) lmcs_host_compress(lmcs, children + 8 * q, out + 4 * q);
}

void lmcs_build_sharded(mh_ctx* c, mh_tree* t, const Dist& dist, const u64* local_digests, int log_rows) {
  const int G = dist.logG, lb = t->log_blowup, lbl = lb - G;
  MH_REQUIRE(G > 0 && lbl >= 0 && log_rows >= G, "internal: bad sharded tree shape");
  const size_t local = (size_t)1 << (lbl + log_rows);
  DevBuf packed(local * 32);
  MH_LAUNCH(k_repack_digests, dim3((unsigned)((local + 255) / 256)), dim3(256), 0, c->stream,
                     (const ulonglong2*)local_digests, (ulonglong2*)packed.p, lbl, log_rows, G);
  lmcs_alloc_layers(t, log_rows - G + lb);
  dist.all_to_all(c, packed.p, lmcs_leaf_layer(t), (local >> G) * 32);
  lmcs_compress_layers(c, t);  // subroot in t->root and at layer 0 of t->nodes
  DevBuf all((size_t)32 << G);
  dist.all_gather(c, t->nodes.u() + 4 * t->layer_off[0], all.p, 32);
  const size_t Gn = (size_t)1 << G;
  t->cap.assign(4 * (2 * Gn - 1), 0);
  HIP_CHECK(hipMemcpyAsync(t->cap.data() + 4 * (Gn - 1), all.p, 32 * Gn, hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipStreamSynchronize(c->stream));
  for (int d = G - 1; d >= 0; d--)
    for (size_t p = 0; p < ((size_t)1 << d); p++) {
      const u64* l = t->cap.data() + 4 * ((((size_t)2) << d) - 1 + 2 * p);
      lmcs_host_compress(c->lmcs, l, t->cap.data() + 4 * ((((size_t)1) << d) - 1 + p));
    }
  memcpy(t->root, t->cap.data(), 32);
  t->shard_logG = G;
  t->shard_rank = dist.rank;
}
