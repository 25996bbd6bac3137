// K1 / K6: radix-2 NTT over Goldilocks and the coset low-degree extension, gfx950.
//
// Replaces p3-dft 0.6.2 `Radix2DitParallel::coset_lde_batch` as called by the reference at
// crates/lifted-stark/src/prover/commit.rs:173 (trace LDE) and `coset_lde_batch_with_transform`
// at crates/lifted-stark/src/prover/quotient.rs:186 (quotient chunks).
//
// MI355X design (NOT the reference's row-major/bit-reversed layout):
//   * every column is a contiguous vector -> all global accesses are unit-stride across lanes;
//   * the LDE of a column is stored coset-major: out[(col*B + j)*N + r] = f(shift * w_K^j * w_H^r),
//     i.e. natural index i = r*B + j of the reference's (virtual) natural order.  "Next trace row"
//     is r+1 in the same coset, Merkle siblings are (j, j^1) at equal r, and FRI cosets are
//     r + N/4 strides -- all unit-stride for the kernels downstream.
//   * iNTT = DIF (natural in, bit-reversed out), coset NTT = DIT (bit-reversed in, natural out):
//     no bit-reversal permutation pass ever touches HBM.
//   * each pass stages a 2^12-element tile (32 KB + padding) in LDS and runs up to 12 butterfly stages
//     on it as radix-16 rounds in registers; strided passes move >=128-byte contiguous segments.
// Roofline: algorithmic bytes per column = (1 + B) * N * 8 (read trace once, write LDE once), two
// passes move (4 + 4B) * N * 8; measured, the passes are VALU-issue bound (no 64-bit multiplier on
// CDNA4: ~23 VALU instructions per element-stage), see DESIGN.md section 3.
#include "ctx.hpp"
#include "gl.cuh"
#include "kernels.hpp"
#include "poseidon2_fast.cuh"
#include <utility>

// Field multiplications of the passes (table twiddles, coset scale): 0 = the compiler-scheduled C form (~23 VALU),
// 1 = the volatile interleaved asm form of the hash kernels (pins the schedule: table loads are no longer hoisted, VGPRs
// 119 -> 150+; LDE 11.5 -> 12.7 ms), 2 = the same 13-instruction product as NON-volatile statements that carry their own
// wait states (108-111 VGPRs, LDE 11.12 -> 10.87 ms per 2^20-row proof).  Default 2.
#ifndef NTT_ASM_MUL
#define NTT_ASM_MUL 2
#endif
#if P2F_ASM && NTT_ASM_MUL == 1
#define NTT_MUL1 p2f_mul
#elif P2F_ASM && NTT_ASM_MUL == 2
#define NTT_MUL1 p2f_mul_nv  // non-volatile statements: the compiler keeps its freedom to hoist loads and interleave
#else
#define NTT_MUL1 p2f_mul_c
#endif

// LDS tile of a pass: 2^12 elements (34 KB with padding, 256 threads, 4 workgroups per CU) up to 2^20-point
// transforms; 2^14 elements (136 KB, 1024 threads, one workgroup per CU -- the same 4 waves per SIMD) beyond, so
// that 2^21 and 2^22 still take two passes over HBM instead of three.
static constexpr int NTT_TILE_LOG = 12;
static constexpr int NTT_TILE_LOG_BIG = 14;
static constexpr int NTT_THREADS = 256;
// (measured, LDE per proof: 2^22 three-AIR shape 69.4 -> 63.2 ms with the big tile; at 2^24 the 10-stage strided
// pass it implies -- 128 B segments, 1024 rows -- loses to three small-tile passes, 289.7 vs 259.6 ms)
static int ntt_tile_log(int log_n) {
  static const int big_max = [] {
    const char* e = getenv("MH_NTT_BIG_MAX");  // experiments: the largest log_n that uses the 2^14 tile
    return e ? atoi(e) : 22;
  }();
  return log_n > 20 && log_n <= big_max ? NTT_TILE_LOG_BIG : NTT_TILE_LOG;
}

// Layout of a coefficient vector between the inverse and the forward transform (internal to this file and to its two callers):
// bit-reversed order, and -- when the contiguous pass works on full 2^12 tiles -- every tile ROTATED: element l of the tile sits at
// position (l & 15) << 8 | l >> 4.  The last round of the inverse leaves thread q with elements 16 q .. 16 q + 15 and the first round
// of the forward transform wants exactly those: with the rotation the inverse's final stores are coalesced (position e * 256 + q;
// they were 8-byte stores 128 B apart) and the forward's tile load (position tid + 256 j) IS round one's register set -- no LDS round
// trip, no barrier before the first butterflies.  The scale tables of the first forward pass are built in the same order.
static bool ntt_coef_rot(int log_n) {
  static const int on = [] { const char* e = getenv("MH_NTT_ROT"); return e ? atoi(e) : 1; }();
  return on && log_n >= NTT_TILE_LOG && ntt_tile_log(log_n) == NTT_TILE_LOG;
}

// ---------------------------------------------------------------------------------------------
// twiddle table: tw[k] = w^k, k < n_half, given w^(2^i) in pw[]
struct PowTable {
  u64 pw[32];
};
__global__ void k_fill_powers(u64* out, size_t n, PowTable t, u64 scale) {
  size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (k >= n) return;
  u64 r = scale;
  size_t e = k;
#pragma unroll 1
  for (int i = 0; e; i++, e >>= 1)
    if (e & 1) r = gl_mul(r, t.pw[i]);
  out[k] = r;
}

static void fill_powers(mh_ctx* c, u64* out, size_t n, u64 base, u64 scale) {
  PowTable t;
  u64 b = base;
  for (int i = 0; i < 32; i++) {
    t.pw[i] = b;
    b = gl_sqr(b);
  }
  if (n == 0) return;
  MH_LAUNCH(k_fill_powers, dim3((n + 255) / 256), dim3(256), 0, c->stream, out, n, t, scale);
}

const u64* mh_ctx::twiddles(int log_n, bool inverse) {
  auto& m = inverse ? tw_inv : tw_fwd;
  auto it = m.find(log_n);
  if (it != m.end()) return it->second.u();
  size_t half = log_n ? ((size_t)1 << (log_n - 1)) : 1;
  DevBuf b(half * 8);
  u64 w = gl_two_adic_generator(log_n);
  if (inverse) w = gl_inv(w);
  fill_powers(this, b.u(), half, w, 1);
  const u64* p = b.u();
  m[log_n] = std::move(b);
  return p;
}

// ---------------------------------------------------------------------------------------------
// row-major [n][w] -> column-major [w][n]   (32x32 LDS tile, +1 padding)
__global__ __launch_bounds__(256) void k_transpose_rm_to_cm(const u64* __restrict__ in, u64* __restrict__ out, size_t n,
                                                           size_t w) {
  __shared__ u64 tile[32][33];
  size_t r0 = (size_t)blockIdx.x * 32, c0 = (size_t)blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    size_t r = r0 + k, cc = c0 + tx;
    if (r < n && cc < w) tile[k][tx] = in[r * w + cc];
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    size_t cc = c0 + k, r = r0 + tx;
    if (r < n && cc < w) out[cc * n + r] = gl_canon(tile[tx][k]);
  }
}

void launch_transpose_rm_to_cm(mh_ctx* c, const u64* in, u64* out, size_t n, size_t w, hipStream_t stream) {
  dim3 grid((unsigned)((n + 31) / 32), (unsigned)((w + 31) / 32));
  MH_LAUNCH(k_transpose_rm_to_cm, grid, dim3(256), 0, stream ? stream : c->stream, in, out, n, w);
}

// ---------------------------------------------------------------------------------------------
struct NttPassArgs {
  const u64* src;
  u64* dst;
  size_t src_col_stride, dst_col_stride, dst_z_stride;  // in elements
  int log_n, s_lo, r_bits, cb;                          // stages s_lo .. s_lo+r_bits-1, tile = 2^(r_bits+cb)
  int dif;                                              // 1 = DIF (a+b,(a-b)w), descending; 0 = DIT
  int canon_out;                                        // store canonical values (last pass of a transform whose output leaves the NTT)
  const u64* tw_round[4];                               // per round of this pass: plane [2^G - 1][2^s] of W^rev(e), W = w_{2^(s+G)}^gm
  const u64* scale_lo;                                  // optional: multiply on load by
  const u64* scale_hi;                                  //   scale_lo[z][k & m] * scale_hi[z][k >> lb], k = bitrev(pos)
  int lb;
  size_t scale_lo_z, scale_hi_z;
  u32 n_z;                                              // output cosets produced per workgroup (first pass of a coset LDE), else 1
  const u64* scale_full;                                // optional, instead of scale_lo/hi: the whole product table [z][pos] (one load, one product)
  int col_fastest;                                      // grid = (columns, tiles): neighbouring workgroups share a tile's twiddle / scale slices
  size_t src_z_stride;                                  // 0: every coset reads the same source (first pass); else the source of coset z is src + z * src_z_stride
  int rot;                                              // contiguous pass on rotated coefficient tiles (ntt_coef_rot): inverse = its stores, forward = its loads
  int direct_first;                                     // forward strided pass, cb = 4, r_bits a multiple of 4 and >= 8: the first round loads from HBM
  u32 group_cols, group_z;                              // group_cols > 0: columns come in groups with their OWN coset shifts (quotient chunks):
                                                        //   the scale row of (column, coset z) is (column / group_cols) * group_z + z
  const u64* step;                                      // MODE 1 (first pass, cosets in geometric progression): [pos] = (base_{z+1} / base_z)^k, k = bitrev(pos)
};

// ---------------------------------------------------------------------------------------------
// Radix-16 passes.  A thread keeps 2^G (G <= 4) elements in registers and runs G butterfly stages on
// them between two LDS round trips (a 12-stage pass = 3 round trips instead of 12):
//   DIT round over stages s..s+G-1:  x_e *= W^rev(e), W = w_{2^(s+G)}^gm (one table load each), then a
//     2^G-point DFT whose twiddles are powers of w_16 = 2^156 (w_16^-1 = 2^36): SHIFTS, not products
//     (2^96 = -1, 2^64 = 2^32 - 1 mod p);
//   DIF round: the transpose (DFT first, then y_e *= W^rev(e)).
// Arithmetic is lazy: values are any representative < 2^64; a +- t needs only t canonical.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ u64 ntt_canon(u64 t) {  // t >= p  <=>  t + eps carries out of 64 bits
  const u64 u = t + GL_EPS;
  return u < t ? u : t;
}
__device__ __forceinline__ u64 ntt_add(u64 a, u64 t) {  // t canonical
  u64 s = a + t;
  s += (s < t) ? GL_EPS : 0;
  return s;
}
__device__ __forceinline__ u64 ntt_sub(u64 a, u64 t) {  // t canonical
  u64 d = a - t;
  d -= (a < t) ? GL_EPS : 0;
  return d;
}
// x * 2^S for a compile-time S in [0, 96); any x < 2^64, result some representative.
__device__ __forceinline__ u64 ntt_mul_pow2(u64 x, int S) {
  const int q = S >> 5, r = S & 31;
  const u64 y01 = x << r;                              // low 64 bits of the 96-bit x * 2^r
  const u32 y2 = r ? (u32)(hi32(x) >> (32 - r)) : 0u;  // its top limb (< 2^r)
  if (q == 0) {  // y01 + y2 * 2^64
    u64 v = (u64)y2 * 0xFFFFFFFFu + y01;
    v += (v < y01) ? GL_EPS : 0;
    return v;
  }
  if (q == 1) {  // y0 * 2^32 + y1 * 2^64 + y2 * 2^96 = (y0 << 32) + y1 * eps - y2
    const u64 a0 = (u64)lo32(y01) << 32;
    u64 v = (u64)hi32(y01) * 0xFFFFFFFFu + a0;
    v += (v < a0) ? GL_EPS : 0;
    u64 w = v - y2;
    w -= (v < (u64)y2) ? GL_EPS : 0;
    return w;
  }
  // q == 2: y0 * 2^64 + y1 * 2^96 + y2 * 2^128 = y0 * eps - (y1 + y2 * 2^32)
  const u64 v = (u64)lo32(y01) * 0xFFFFFFFFu;
  const u64 b = ((u64)y2 << 32) | hi32(y01);
  u64 w = v - b;
  w -= (v < b) ? GL_EPS : 0;
  return w;
}
// (a, b) <- (a + 2^E b, a - 2^E b), E in [0, 192)   [DIT]
__device__ __forceinline__ void ntt_bfly_dit(u64& a, u64& b, int E) {
  const u64 t = ntt_canon(E % 96 ? ntt_mul_pow2(b, E % 96) : b);
  const u64 s = ntt_add(a, t), d = ntt_sub(a, t);
  if (E >= 96) { a = d; b = s; } else { a = s; b = d; }
}
// (a, b) <- (a + b, 2^E (a - b))   [DIF]
__device__ __forceinline__ void ntt_bfly_dif(u64& a, u64& b, int E) {
  const u64 t = ntt_canon(b);
  const u64 s = ntt_add(a, t);
  const u64 d = E >= 96 ? ntt_sub(t, ntt_canon(a)) : ntt_sub(a, t);  // sign of 2^96 = -1 folded into the difference
  a = s;
  b = E % 96 ? ntt_mul_pow2(d, E % 96) : d;
}
// ---- the forward (DIT) register DFT with hand-placed carry chains (NTT_ASM_BFLY, default 1) ----------------------------------
// The C butterflies above cost hipcc ~14 VALU instructions without a shift and ~30 with one (every carry a v_cmp_lt_u64 +
// v_cndmask pair, every 32 -> 64-bit composition a v_mov / v_or); a radix-16 round spent 870 of its ~1130 VALU instructions
// there.  Here a stage's 2^(G-1) butterflies run as ONE stage-interleaved sequence of single-instruction asm statements, the way
// p2f_mulN interleaves products (carries live in SGPR pairs, N >= 4 independent chains give every carry its 2 wait states):
//   shifted butterfly:  X = b * 2^(E mod 96) laid out as lo64 + h0 * 2^64 + h1 * 2^96   (3-4 instructions, prelude in C)
//                       r = lo + h0 * eps - h1  (the 6-instruction tail of the field multiplication: 1 mad + carry fix-ups)
//   every butterfly:    t = canonical(r or b) (4), a + t (4), a - t (4); E >= 96 (2^96 = -1) swaps the two outputs.
// 12 VALU for a plain butterfly, 21-22 for a shifted one: ~545 per radix-16 round instead of 870.
#ifndef NTT_ASM_BFLY
#define NTT_ASM_BFLY 1
#endif
#define NTT_A asm volatile
// carry-outs nobody reads rotate through fixed scratch SGPR pairs (hipcc separates asm statements that share a register by s_nop)
#define NTT_DEAD(i, TXT_PRE, TXT_POST, ...)                                          \
  do {                                                                              \
    switch ((i) & 3) {                                                              \
      case 0: NTT_A(TXT_PRE "s[84:85]" TXT_POST : __VA_ARGS__ : "s84", "s85"); break; \
      case 1: NTT_A(TXT_PRE "s[86:87]" TXT_POST : __VA_ARGS__ : "s86", "s87"); break; \
      case 2: NTT_A(TXT_PRE "s[88:89]" TXT_POST : __VA_ARGS__ : "s88", "s89"); break; \
      default: NTT_A(TXT_PRE "s[90:91]" TXT_POST : __VA_ARGS__ : "s90", "s91"); break; \
    }                                                                               \
  } while (0)
// r[i] = lo[i] + h0[i] * 2^64 + h1[i] * 2^96 (mod p), any representative  [i < N, stage-interleaved]
template <int N>
__device__ __forceinline__ void ntt_reduce128N(u64 (&r)[N], const u64 (&lo)[N], const u32 (&h0)[N], const u32 (&h1)[N]) {
  u64 t[N], c1[N], bb[N], bw[N], c3[N], k3[N];
  u32 rl[N], rh[N];
#pragma unroll
  for (int i = 0; i < N; i++) NTT_A("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(t[i]), "=s"(c1[i]) : "v"(h0[i]), "v"(lo[i]));
#pragma unroll
  for (int i = 0; i < N; i++) NTT_A("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(rl[i]), "=s"(bb[i]) : "v"(lo32(t[i])), "v"(h1[i]), "s"(c1[i]));
#pragma unroll
  for (int i = 0; i < N; i++) NTT_DEAD(i, "v_addc_co_u32_e64 %0, ", ", %1, 0, %2", "=v"(rh[i]) : "v"(hi32(t[i])), "s"(c1[i]));
#pragma unroll
  for (int i = 0; i < N; i++) NTT_A("v_subb_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rh[i]), "=s"(bw[i]) : "0"(rh[i]), "s"(bb[i]));
#pragma unroll
  for (int i = 0; i < N; i++) NTT_A("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rl[i]), "=s"(c3[i]) : "0"(rl[i]), "s"(bw[i]));
#pragma unroll
  for (int i = 0; i < N; i++) k3[i] = bw[i] & ~c3[i];  // scalar unit
#pragma unroll
  for (int i = 0; i < N; i++) {
    NTT_DEAD(i, "v_subb_co_u32_e64 %0, ", ", %1, 0, %2", "=v"(rh[i]) : "0"(rh[i]), "s"(k3[i]));
    r[i] = ((u64)rh[i] << 32) | rl[i];
  }
}
// (a[i], t[i]) <- (a[i] + t[i], a[i] - t[i]) for any representatives a[i], t[i] < 2^64; results are representatives < 2^64
template <int N>
__device__ __forceinline__ void ntt_bfly_tailN(u64 (&a)[N], u64 (&t)[N]) {
  u32 u0[N], u1[N], t0[N], t1[N], s0[N], s1[N], d0[N], d1[N];
  u64 c0[N], c1[N], ca[N], cb[N], bs[N], ks[N], b0[N], b1[N], cd[N], kd[N];
  // t <- canonical(t):  t >= p  <=>  t + eps carries out of 64 bits, and then t - p = t + eps (mod 2^64) = (t.lo - 1, 0)
#pragma unroll
  for (int i = 0; i < N; i++) NTT_A("v_add_co_u32_e64 %0, %1, %2, -1" : "=v"(u0[i]), "=s"(c0[i]) : "v"(lo32(t[i])));
#pragma unroll
  for (int i = 0; i < N; i++) NTT_A("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(u1[i]), "=s"(c1[i]) : "v"(hi32(t[i])), "s"(c0[i]));
#pragma unroll
  for (int i = 0; i < N; i++) NTT_A("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(t0[i]) : "v"(lo32(t[i])), "v"(u0[i]), "s"(c1[i]));
#pragma unroll
  for (int i = 0; i < N; i++) NTT_A("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(t1[i]) : "v"(hi32(t[i])), "v"(u1[i]), "s"(c1[i]));
  // s = a + t: a carry out of 64 bits is worth eps = 2^32 - 1: lo -= 1 (borrow bs), hi += 1 - bs; cannot carry again (t < p)
#pragma unroll
  for (int i = 0; i < N; i++) NTT_A("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(s0[i]), "=s"(ca[i]) : "v"(lo32(a[i])), "v"(t0[i]));
#pragma unroll
  for (int i = 0; i < N; i++) NTT_A("v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(s1[i]), "=s"(cb[i]) : "v"(hi32(a[i])), "v"(t1[i]), "s"(ca[i]));
  // d = a - t: a borrow is worth -eps: lo += 1 (carry cd), hi -= 1 - cd; cannot borrow again (t < p)
#pragma unroll
  for (int i = 0; i < N; i++) NTT_A("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(d0[i]), "=s"(b0[i]) : "v"(lo32(a[i])), "v"(t0[i]));
#pragma unroll
  for (int i = 0; i < N; i++) NTT_A("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(d1[i]), "=s"(b1[i]) : "v"(hi32(a[i])), "v"(t1[i]), "s"(b0[i]));
#pragma unroll
  for (int i = 0; i < N; i++) NTT_A("v_subb_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(s0[i]), "=s"(bs[i]) : "0"(s0[i]), "s"(cb[i]));
#pragma unroll
  for (int i = 0; i < N; i++) NTT_A("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(d0[i]), "=s"(cd[i]) : "0"(d0[i]), "s"(b1[i]));
#pragma unroll
  for (int i = 0; i < N; i++) {
    ks[i] = cb[i] & ~bs[i];  // scalar unit
    kd[i] = b1[i] & ~cd[i];
  }
#pragma unroll
  for (int i = 0; i < N; i++) NTT_DEAD(i, "v_addc_co_u32_e64 %0, ", ", %1, 0, %2", "=v"(s1[i]) : "0"(s1[i]), "s"(ks[i]));
#pragma unroll
  for (int i = 0; i < N; i++) {
    NTT_DEAD(i + 2, "v_subb_co_u32_e64 %0, ", ", %1, 0, %2", "=v"(d1[i]) : "0"(d1[i]), "s"(kd[i]));
    a[i] = ((u64)s1[i] << 32) | s0[i];
    t[i] = ((u64)d1[i] << 32) | d0[i];
  }
}
// b * 2^S (S = 32 q + r in (0, 96)) as lo + h0 * 2^64 + h1 * 2^96:  y = b << r = (y0, y1, y2), y2 < 2^r;
//   q = 0: (y0, y1 | y2, 0);  q = 1: (0, y0 | y1, y2);  q = 2: y0 * 2^64 + y1 * 2^96 + y2 * 2^128, and 2^128 = -2^32 mod p:
//   p - y2 * 2^32 = (1, ~y2) as 32-bit words (y2 < 2^31), so (1, ~y2 | y0, y1).
template <int S>
__device__ __forceinline__ void ntt_shift_parts(u64 b, u64& lo, u32& h0, u32& h1) {
  constexpr int q = S >> 5, r = S & 31;
  const u32 b0 = lo32(b), b1 = hi32(b);
  const u32 y0 = r ? (b0 << r) : b0;
  const u32 y1 = r ? __builtin_amdgcn_alignbit(b1, b0, 32 - r) : b1;
  const u32 y2 = r ? (b1 >> (32 - r)) : 0u;
  if constexpr (q == 0) {
    lo = ((u64)y1 << 32) | y0; h0 = y2; h1 = 0;
  } else if constexpr (q == 1) {
    lo = (u64)y0 << 32; h0 = y1; h1 = y2;
  } else {
    lo = ((u64)(~y2) << 32) | 1u; h0 = y0; h1 = y1;
  }
}
// exponent of 2 in the twiddle of butterfly (e, e | 1 << m) of the forward DIT (position bit m = stage m; w_16 = 2^156)
__host__ __device__ constexpr int ntt_fwd_exp(int m, int e) { return (156 * (8 >> m) * (e & ((1 << m) - 1))) % 192; }
__host__ __device__ constexpr int ntt_bfly_lo(int m, int k) { return ((k >> m) << (m + 1)) | (k & ((1 << m) - 1)); }  // k-th index without bit m
__host__ __device__ constexpr bool ntt_bfly_shifted(int m, int k) { return (ntt_fwd_exp(m, ntt_bfly_lo(m, k)) % 96) != 0; }
__host__ __device__ constexpr int ntt_shift_slot(int m, int k) {  // shifted butterflies of stage m in front of butterfly k
  int n = 0;
  for (int i = 0; i < k; i++) n += ntt_bfly_shifted(m, i);
  return n;
}
// Butterfly K (and the following ones) of stage M: operands into the interleaved arrays / results back into x.
template <int H, int M, int K>
struct NttBfly {
  static constexpr int E = ntt_bfly_lo(M, K), EX = ntt_fwd_exp(M, E), SLOT = ntt_shift_slot(M, K);
  static constexpr bool SH = (EX % 96) != 0;
  __device__ __forceinline__ static void pre(const u64* x, u64* a, u64* t, u64* lo, u32* h0, u32* h1) {
    a[K] = x[E];
    if constexpr (SH) ntt_shift_parts<EX % 96>(x[E | (1 << M)], lo[SLOT], h0[SLOT], h1[SLOT]);
    else t[K] = x[E | (1 << M)];
    if constexpr (K + 1 < H) NttBfly<H, M, K + 1>::pre(x, a, t, lo, h0, h1);
  }
  __device__ __forceinline__ static void mid(u64* t, const u64* r) {
    if constexpr (SH) t[K] = r[SLOT];
    if constexpr (K + 1 < H) NttBfly<H, M, K + 1>::mid(t, r);
  }
  __device__ __forceinline__ static void post(u64* x, const u64* a, const u64* t) {
    if constexpr (EX >= 96) {  // 2^96 = -1: the twiddle was negated, the two outputs trade places
      x[E] = t[K];
      x[E | (1 << M)] = a[K];
    } else {
      x[E] = a[K];
      x[E | (1 << M)] = t[K];
    }
    if constexpr (K + 1 < H) NttBfly<H, M, K + 1>::post(x, a, t);
  }
};
// Interleave at most NTT_ILV chains at a time: every live carry is an SGPR pair, eight butterflies at once (24+ pairs next to the
// kernel's own scalars) made hipcc spill SGPRs into VGPR lanes (v_writelane / v_readlane in the round loop).
#ifndef NTT_ILV
#define NTT_ILV 4
#endif
template <int N>
__device__ __forceinline__ void ntt_tail_split(u64 (&a)[N], u64 (&t)[N]) {
  if constexpr (N <= NTT_ILV) {
    ntt_bfly_tailN<N>(a, t);
  } else {
    constexpr int A = NTT_ILV, B = N - NTT_ILV;
    u64 a0[A], t0[A], a1[B], t1[B];
#pragma unroll
    for (int i = 0; i < A; i++) { a0[i] = a[i]; t0[i] = t[i]; }
#pragma unroll
    for (int i = 0; i < B; i++) { a1[i] = a[A + i]; t1[i] = t[A + i]; }
    ntt_bfly_tailN<A>(a0, t0);
    ntt_tail_split<B>(a1, t1);
#pragma unroll
    for (int i = 0; i < A; i++) { a[i] = a0[i]; t[i] = t0[i]; }
#pragma unroll
    for (int i = 0; i < B; i++) { a[A + i] = a1[i]; t[A + i] = t1[i]; }
  }
}
template <int N>
__device__ __forceinline__ void ntt_reduce_split(u64 (&r)[N], const u64 (&lo)[N], const u32 (&h0)[N], const u32 (&h1)[N]) {
  if constexpr (N <= NTT_ILV) {
    ntt_reduce128N<N>(r, lo, h0, h1);
  } else {
    constexpr int A = (N >= 2 * NTT_ILV || N - NTT_ILV >= 3) ? NTT_ILV : N - 3, B = N - A;  // never leave a tail of fewer than 3 chains
    u64 r0[A], lo0[A], r1[B], lo1[B];
    u32 p0[A], q0[A], p1[B], q1[B];
#pragma unroll
    for (int i = 0; i < A; i++) { lo0[i] = lo[i]; p0[i] = h0[i]; q0[i] = h1[i]; }
#pragma unroll
    for (int i = 0; i < B; i++) { lo1[i] = lo[A + i]; p1[i] = h0[A + i]; q1[i] = h1[A + i]; }
    ntt_reduce128N<A>(r0, lo0, p0, q0);
    ntt_reduce_split<B>(r1, lo1, p1, q1);
#pragma unroll
    for (int i = 0; i < A; i++) r[i] = r0[i];
#pragma unroll
    for (int i = 0; i < B; i++) r[A + i] = r1[i];
  }
}
template <int G, int M>
__device__ __forceinline__ void ntt_fwd_stage(u64 (&x)[1 << G]) {
  constexpr int H = 1 << (G - 1);
  constexpr int NS = ntt_shift_slot(M, H);  // shifted butterflies in this stage: 0, H/2, 3H/4, 7H/8
  u64 a[H], t[H];
  u64 lo[NS ? NS : 1], r[NS ? NS : 1];
  u32 h0[NS ? NS : 1], h1[NS ? NS : 1];
  NttBfly<H, M, 0>::pre(x, a, t, lo, h0, h1);
  if constexpr (NS > 0) {
    ntt_reduce_split<NS>(r, lo, h0, h1);
    NttBfly<H, M, 0>::mid(t, r);
  }
  ntt_tail_split<H>(a, t);
  NttBfly<H, M, 0>::post(x, a, t);
}
template <int G, int M = 0>
__device__ __forceinline__ void ntt_fwd_dft_asm(u64 (&x)[1 << G]) {
  if constexpr (M < G) {
    ntt_fwd_stage<G, M>(x);
    ntt_fwd_dft_asm<G, M + 1>(x);
  }
}

// 2^G-point DFT on registers; position bit m is stage m.  DIT: bit-reversed in, natural out; DIF: the transpose.
template <int G, bool INV>
__device__ __forceinline__ void ntt_dft_regs(u64 (&x)[1 << G], bool dif) {
#if P2F_ASM && NTT_ASM_BFLY
  if constexpr (!INV && G >= 3) {  // the forward passes (8/9 of an LDE); 4 or 8 butterflies per stage keep every carry 2 wait states away
    ntt_fwd_dft_asm<G>(x);
    return;
  }
#endif
  constexpr int W16 = INV ? 36 : 156;  // log2 of w_16 in the transform direction
  if (!dif) {
#pragma unroll
    for (int m = 0; m < G; m++)
#pragma unroll
      for (int e = 0; e < (1 << G); e++)
        if (!(e & (1 << m))) ntt_bfly_dit(x[e], x[e | (1 << m)], (W16 * (8 >> m) * (e & ((1 << m) - 1))) % 192);
  } else {
#pragma unroll
    for (int m = G - 1; m >= 0; m--)
#pragma unroll
      for (int e = 0; e < (1 << G); e++)
        if (!(e & (1 << m))) ntt_bfly_dif(x[e], x[e | (1 << m)], (W16 * (8 >> m) * (e & ((1 << m) - 1))) % 192);
  }
}
// x[e] *= tw[(e - 1) << s | gm] for e = OFF .. OFF + K - 1: the table twiddles of a round, multiplied in interleaved
// groups of <= 5 products (p2f_mulN); consecutive lanes = consecutive gm: coalesced loads
template <int K, int OFF>
__device__ __forceinline__ void ntt_tw_mul(u64* x, const u64* __restrict__ tw, int s, u32 gm) {
  if constexpr (K > 0) {
    constexpr int C = K >= 5 ? 5 : K;
    u64 a[C], b[C];
#pragma unroll
    for (int i = 0; i < C; i++) {
      a[i] = x[OFF + i];
      b[i] = (tw + ((size_t)(OFF + i - 1) << s))[gm];  // wave-uniform row pointer + the thread's column
    }
#if P2F_ASM && NTT_ASM_MUL == 1
    p2f_mulN<C>(a, a, b);
#else
#pragma unroll
    for (int i = 0; i < C; i++) a[i] = NTT_MUL1(a[i], b[i]);
#endif
#pragma unroll
    for (int i = 0; i < C; i++) x[OFF + i] = a[i];
    ntt_tw_mul<K - C, OFF + C>(x, tw, s, gm);
  }
}
#else
template <int K, int OFF>
__device__ void ntt_tw_mul(u64* x, const u64* __restrict__ tw, int s, u32 gm);
template <int G, bool INV>
__device__ void ntt_dft_regs(u64 (&x)[1 << G], bool dif);
__device__ u64 ntt_canon(u64 t);
__device__ u64 p2f_mul_c(u64 a, u64 b);
#endif

#ifndef NTT16_OCC
#define NTT16_OCC
#endif
// LDS placement of tile element l.  NTT_SWZ = 1 (default, the 2^12 tile): XOR swizzle l ^ ((l >> 4) & 31) -- conflict-free for the
// three access shapes of a pass (element stride 1 / 16 / 256 between a thread's sixteen values, consecutive lanes = consecutive
// low bits: every group of 32 lanes covers all 32 bank pairs) and for the linear tile load, in EXACTLY 32 KB: five workgroups share
// a CU's 160 KB (the padded tile, 34 KB, allowed four).  The 2^14 tile (one workgroup per CU anyway) keeps one pad element per 16.
#ifndef NTT_SWZ
#define NTT_SWZ 1
#endif
static size_t ntt_lds_bytes(int tile_log) {
  if (NTT_SWZ && tile_log == NTT_TILE_LOG) return ((size_t)1 << tile_log) * 8;
  return (((size_t)1 << tile_log) + ((size_t)1 << (tile_log - 4))) * 8;
}
template <bool SWZ>
__device__ __forceinline__ u32 ntt_pad(u32 l) {
  return SWZ ? (l ^ ((l >> 4) & 31u)) : (l + (l >> 4));
}

// B0: the LDS bit position of the element index e in this round (= st + cb) when it is 0, 4 or 8 -- every round of the 2^20 plan --
// else -1 (decided at run time).  With B0 known at compile time the LDS byte address of element e is the thread's pre-scaled
// address XOR a LITERAL (B0 = 0, 4: one v_xor, no scalar registers), or, for B0 = 8, one of two pre-computed addresses plus an
// immediate offset in the ds instruction (the swizzle touches bit 4 only: slot = (p0 ^ ((e & 1) << 4)) + (e << 8)): no VALU at all.
// (Streaming stores for the output of a pass -- __builtin_nontemporal_store -- were measured in round 4: the forward passes do not
// change, the inverse transform gets 0.4 ms slower per 51 columns: its output is the next kernel's input and came from the cache.)
#define NTT_STORE(p, v) (*(p) = (v))
template <int G, bool INV, bool SWZ, int B0>
__device__ __forceinline__ void ntt_round(const NttPassArgs& a, u64* lds, int st, u32 tile_n, size_t lo0, u64* dst_direct,
                                          size_t gbase, const u64* __restrict__ tw, const u64* __restrict__ src_direct = nullptr,
                                          const u64* xin = nullptr) {
  const int b0 = B0 >= 0 ? B0 : st + a.cb, s = a.s_lo + st;
  const u32 cb_mask = (1u << a.cb) - 1;
  constexpr bool FAST = SWZ && B0 >= 0;
  for (u32 q = threadIdx.x; q < (tile_n >> G); q += blockDim.x) {
    const u32 low = q & ((1u << b0) - 1);
    const u32 l0 = ((q >> b0) << (b0 + G)) | low;
    const u32 gm = (((l0 >> a.cb) & ((1u << st) - 1)) << a.s_lo) | ((u32)lo0 << a.cb) | (l0 & cb_mask);
    // LDS slot of element e: the swizzle is linear over XOR and l0, e << b0 have no bit in common, so
    // slot(l0 | e << b0) = slot(l0) ^ slot(e << b0)
    const u32 p0 = ntt_pad<SWZ>(l0);
    char* const ldsb = reinterpret_cast<char*>(lds);
    const u32 pb = p0 * 8u, pb1 = (p0 ^ 16u) * 8u;  // byte addresses (pb1: B0 = 8, odd e)
    auto slot = [&](int e) -> u64* {
      if constexpr (FAST && B0 == 8) return reinterpret_cast<u64*>(ldsb + ((e & 1) ? pb1 : pb) + ((u32)e << 11));
      else if constexpr (FAST) return reinterpret_cast<u64*>(ldsb + (pb ^ (ntt_pad<true>((u32)e << B0) * 8u)));
      else return lds + (SWZ ? (p0 ^ ntt_pad<true>((u32)e << b0)) : ntt_pad<false>(l0 | ((u32)e << b0)));
    };
    u64 x[1 << G];
    if (xin) {  // the thread's registers already hold this round's values (first forward pass on rotated tiles)
#pragma unroll
      for (int e = 0; e < (1 << G); e++) x[e] = xin[e];
    } else if (src_direct) {
      // the first round of a strided pass takes its sixteen values straight from HBM: they are the elements l0 | e << b0 of the tile,
      // rows (l0 >> cb) + e of 2^cb consecutive elements each -- the same 128-byte segments the staged tile load moved, without the
      // LDS round trip and without the barrier (every wave starts its butterflies when ITS loads have landed)
      const u32 six = (u32)(gbase | ((size_t)(l0 >> a.cb) << a.s_lo) | (l0 & cb_mask));
#pragma unroll
      for (int e = 0; e < (1 << G); e++) x[e] = (src_direct + ((size_t)e << s))[six];
    } else {
#pragma unroll
    for (int e = 0; e < (1 << G); e++) x[e] = *slot(e);
    }
    if (!INV && s > 0) ntt_tw_mul<(1 << G) - 1, 1>(x, tw, s, gm);
    ntt_dft_regs<G, INV>(x, INV);
    if (INV && s > 0) ntt_tw_mul<(1 << G) - 1, 1>(x, tw, s, gm);
    if (INV && dst_direct && a.rot && b0 == 0 && G == 4) {
      // rotated coefficient tile: element 16 q + e of the tile goes to position e * 256 + q
      const u32 rix = (u32)gbase + (l0 >> 4);
#pragma unroll
      for (int e = 0; e < (1 << G); e++) NTT_STORE((dst_direct + ((size_t)e << 8)) + rix, a.canon_out ? ntt_canon(x[e]) : x[e]);
    } else if (dst_direct) {
      // global index of element e: b0 >= cb, so e << b0 lands above the tile's contiguous bits: index = index(l0) + (e << (st + s_lo)).
      // A wave-uniform base pointer per element + one 32-bit per-thread index: the scalar-base form of global_store, no address VALU.
      const u32 tix = (u32)(gbase | ((size_t)(l0 >> a.cb) << a.s_lo) | (l0 & cb_mask));
      if (a.canon_out) {  // decided once per round, not per element
#pragma unroll
        for (int e = 0; e < (1 << G); e++) NTT_STORE((dst_direct + ((size_t)e << s)) + tix, ntt_canon(x[e]));
      } else {
#pragma unroll
        for (int e = 0; e < (1 << G); e++) NTT_STORE((dst_direct + ((size_t)e << s)) + tix, x[e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < (1 << G); e++) *slot(e) = x[e];
    }
  }
}

// INV: the inverse transform, always run as DIF (natural in, bit-reversed out); forward = DIT.
// MODE 1 = the first pass of a coset LDE whose coset shifts are a geometric progression (they are: shift * w^j): the thread keeps its
// sixteen SCALED coefficients in registers over the coset loop and moves them from coset z to z + 1 with one product by step[pos]
// (an 8 MB table at 2^20 that every column and every coset of a tile shares) -- instead of re-reading the tile and one row of the
// [z][pos] table (64 MB at 2^20, blowup 8) per coset, which was 0.8 ms of the 2.5 ms of this pass on 51 columns (round 4: skipping the
// table load made the pass that much faster, skipping the tile re-read changed nothing: it hits L2).
template <bool INV, int THREADS, int MODE = 0>
__global__ __launch_bounds__(THREADS) NTT16_OCC void k_ntt16_pass(NttPassArgs a) {
  extern __shared__ u64 lds[];
  constexpr bool SWZ = NTT_SWZ && THREADS == NTT_THREADS;  // the 2^12 tile
  const int tile_log = a.r_bits + a.cb;
  const u32 tile_n = 1u << tile_log;
  const u32 cb_mask = (1u << a.cb) - 1;
  const u32 tile = a.col_fastest ? blockIdx.y : blockIdx.x;
  const u32 col_id = a.col_fastest ? blockIdx.x : blockIdx.y;
  const int lo_bits = a.s_lo - a.cb;
  const size_t lo0 = tile & ((1u << lo_bits) - 1);
  const size_t hi = tile >> lo_bits;
  const size_t gbase = (hi << (a.s_lo + a.r_bits)) | (lo0 << a.cb);
  const u64* src0 = a.src + (size_t)col_id * a.src_col_stride;
  // The first pass of a coset LDE reads one coefficient tile and produces it on every output coset: the workgroup
  // loops over the cosets itself (n_z > 1), so the tile comes from HBM once and from this XCD's L2 afterwards --
  // with the cosets spread over grid.z the same tile was fetched by up to n_z workgroups on different XCDs.
  u64 raw[MODE == 1 ? 16 : 1];
  for (u32 z = 0; z < a.n_z; z++) {
  const u32 zc = blockIdx.z * a.n_z + z;
  const u32 srow = a.group_cols ? (col_id / a.group_cols) * a.group_z + zc : zc;  // row of the scale tables
  u64* dst = a.dst + (size_t)col_id * a.dst_col_stride + (size_t)zc * a.dst_z_stride;
  const u64* src = src0 + (size_t)zc * a.src_z_stride;
  if (z) __syncthreads();  // the previous coset's last round still reads the tile

  constexpr int LOG_T = THREADS == 256 ? 8 : 10;
  // no staging of the tile when the first round executed can load its own values in whole segments: the forward strided pass of
  // whole radix-16 rounds (pass 1 of the 2^20 plan: element bit 4 of the tile, 128-byte segments) and every full-tile pass of the
  // inverse transform (DIF starts with the top four bits: element stride 256 = the very pattern of the staged load)
  const bool direct_first = MODE == 0 && THREADS == NTT_THREADS && a.direct_first && tile_n == 16u * THREADS;
  if (direct_first) {
  } else if (tile_n == 16u * THREADS && a.cb <= LOG_T) {
    // full tile: element i of this thread is l = tid + i * THREADS; THREADS >= 2^cb, so i * THREADS lands above the contiguous bits:
    // global index = index(tid) + (i << (LOG_T - cb + s_lo)), LDS slot = slot(tid) ^ slot(i * THREADS) (linear swizzle)
    const u32 tid = threadIdx.x;
    const u32 g0 = (u32)(gbase | ((size_t)(tid >> a.cb) << a.s_lo) | (tid & cb_mask));
    const int shg = LOG_T - a.cb + a.s_lo;
    const u32 p0 = ntt_pad<SWZ>(tid);
    if constexpr (MODE == 1) {
      if (z == 0) {
#pragma unroll
        for (int j = 0; j < 16; j++) raw[j] = (src + ((size_t)j << shg))[g0];
        if (a.scale_full) {
          const u64* sf = a.scale_full + ((size_t)srow << a.log_n);
#pragma unroll
          for (int c = 0; c < 16; c += 4) {
            u64 t[4];
#pragma unroll
            for (int j = 0; j < 4; j++) t[j] = (sf + ((size_t)(c + j) << shg))[g0];
#pragma unroll
            for (int j = 0; j < 4; j++) raw[c + j] = NTT_MUL1(raw[c + j], t[j]);
          }
        } else {  // few columns (quotient chunks): the first coset from the two-level tables, no [z][pos] table is built for them
#pragma unroll 4
          for (int j = 0; j < 16; j++) {
            const u32 k = bitrev32(a.rot ? (g0 - tid) + ((tid << 4) | (u32)j) : g0 + ((u32)j << shg), a.log_n);
            const u64 sc = NTT_MUL1(a.scale_lo[srow * a.scale_lo_z + (k & ((1u << a.lb) - 1))], a.scale_hi[srow * a.scale_hi_z + (k >> a.lb)]);
            raw[j] = NTT_MUL1(raw[j], sc);
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < 16; c += 8) {
          u64 t[8];
#pragma unroll
          for (int j = 0; j < 8; j++) t[j] = (a.step + ((size_t)(c + j) << shg))[g0];
#pragma unroll
          for (int j = 0; j < 8; j++) raw[c + j] = NTT_MUL1(raw[c + j], t[j]);
        }
      }
      if (!a.rot) {
#pragma unroll
        for (int j = 0; j < 16; j++) lds[p0 ^ ntt_pad<true>((u32)j * THREADS)] = raw[j];
      }
    } else
    if (THREADS == NTT_THREADS && !a.scale_lo && !a.scale_full) {  // (the 1024-thread kernel has 128 VGPRs: it would spill)
      // no scale (every pass but the first of a coset LDE): all sixteen loads in flight -- the strided passes are short of bytes in
      // flight, not of issue slots (pass 1 of the 2^20 plan: 3.3 TB/s with four at a time)
      u64 v[16];
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = (src + ((size_t)j << shg))[g0];
#pragma unroll
      for (int j = 0; j < 16; j++)
        lds[SWZ ? (p0 ^ ntt_pad<true>((u32)j * THREADS)) : ntt_pad<false>(tid + (u32)j * THREADS)] = v[j];
    } else
    // four elements at a time: all sixteen in flight (plus their scale-table loads) cost 200 VGPRs
#pragma unroll 1
    for (int c = 0; c < 16; c += 4) {
      u64 v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = (src + ((size_t)(c + j) << shg))[g0];  // wave-uniform base + the thread's index
      if (a.scale_full) {
        const u64* sf = a.scale_full + ((size_t)srow << a.log_n);
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = NTT_MUL1(v[j], (sf + ((size_t)(c + j) << shg))[g0]);
      } else if (a.scale_lo) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const u32 k = bitrev32(a.rot ? (g0 - tid) + ((tid << 4) | (u32)(c + j)) : g0 + ((u32)(c + j) << shg), a.log_n);
          const u64 sc = NTT_MUL1(a.scale_lo[srow * a.scale_lo_z + (k & ((1u << a.lb) - 1))], a.scale_hi[srow * a.scale_hi_z + (k >> a.lb)]);
          v[j] = NTT_MUL1(v[j], sc);
        }
      }
      if (a.rot) {  // position tid + 256 (c + j) holds element 16 tid + c + j of the tile
#pragma unroll
        for (int j = 0; j < 4; j++) lds[ntt_pad<SWZ>((tid << 4) | (u32)(c + j))] = v[j];
      } else
#pragma unroll
      for (int j = 0; j < 4; j++)
        lds[SWZ ? (p0 ^ ntt_pad<true>((u32)(c + j) * THREADS)) : ntt_pad<false>(tid + (u32)(c + j) * THREADS)] = v[j];
    }
  } else {
    for (u32 l = threadIdx.x; l < tile_n; l += THREADS) {
      size_t g = gbase | ((size_t)(l >> a.cb) << a.s_lo) | (l & cb_mask);
      u64 v = src[g];
      if (a.scale_full) {
        v = NTT_MUL1(v, a.scale_full[((size_t)srow << a.log_n) + g]);
      } else if (a.scale_lo) {
        u32 k = bitrev32((u32)g, a.log_n);
        u64 sc = NTT_MUL1(a.scale_lo[srow * a.scale_lo_z + (k & ((1u << a.lb) - 1))], a.scale_hi[srow * a.scale_hi_z + (k >> a.lb)]);
        v = NTT_MUL1(v, sc);
      }
      lds[ntt_pad<SWZ>(l)] = v;
    }
  }
  const bool regs_first = MODE == 1 && a.rot;  // round one runs on `raw` itself
  if (!direct_first && !regs_first) __syncthreads();
  // stage groups: as many radix-16 rounds as fit, the remainder (1..3 stages) in one smaller round.
  // DIT ascends (small group first keeps the last, directly-stored round wide); DIF descends.
  const int rem = a.r_bits & 3, n16 = a.r_bits >> 2;
  const int n_rounds = n16 + (rem ? 1 : 0);
  if (n_rounds == 0) {  // a 1-point transform: only the scaling above
    if (threadIdx.x == 0) dst[gbase] = ntt_canon(lds[0]);
    continue;
  }
  for (int i = 0; i < n_rounds; i++) {
    // round i of a DIT pass covers [st, st+g); a DIF pass runs the same rounds in reverse order
    const int ri = INV ? n_rounds - 1 - i : i;
    int st, g;
    if (rem) {
      st = ri == 0 ? 0 : rem + 4 * (ri - 1);
      g = ri == 0 ? rem : 4;
    } else {
      st = 4 * ri;
      g = 4;
    }
    u64* direct = (i == n_rounds - 1) ? dst : nullptr;
    const int b0 = st + a.cb;
#define NTT_ROUND(GG, BB) ntt_round<GG, INV, SWZ, BB>(a, lds, st, tile_n, lo0, direct, gbase, a.tw_round[ri])
    if (direct_first && i == 0) ntt_round<4, INV, SWZ, INV ? 8 : 4>(a, lds, st, tile_n, lo0, direct, gbase, a.tw_round[ri], src);
    else if (regs_first && i == 0) ntt_round<4, INV, SWZ, 0>(a, lds, st, tile_n, lo0, direct, gbase, a.tw_round[ri], nullptr, raw);
    else if (g == 4 && b0 == 0) NTT_ROUND(4, 0);
    else if (g == 4 && b0 == 4) NTT_ROUND(4, 4);
    else if (g == 4 && b0 == 8) NTT_ROUND(4, 8);
    else if (g == 4) NTT_ROUND(4, -1);
    else if (g == 3) NTT_ROUND(3, -1);
    else if (g == 2) NTT_ROUND(2, -1);
    else NTT_ROUND(1, -1);
#undef NTT_ROUND
    if (!direct) __syncthreads();
  }
  }  // cosets
}

struct PassPlan {
  int s_lo, r_bits, cb;
};
// stages [0, log_n) split into one contiguous pass (low stages) + strided passes, ascending order.
static std::vector<PassPlan> plan_passes(int log_n) {
  std::vector<PassPlan> p;
  const int T = ntt_tile_log(log_n);
  int c = log_n < T ? log_n : T;
  p.push_back({0, c, 0});
  int rem = log_n - c;
  if (rem > 0) {
    const int max_r = T - 4;  // keep >= 16 consecutive elements (128 B) per segment
    // balanced strided passes.  (Unbalanced 8 + remainder needs fewer radix-16 rounds but measured slower at
    // 2^22 / 2^24 -- 106.5 vs 102.8 ms and 517 vs 420 ms of LDE per proof: its last pass strides by 8 MB.)
    int np = (rem + max_r - 1) / max_r;
    int s = c;
    for (int i = 0; i < np; i++) {
      int r = rem / np + (i < rem % np ? 1 : 0);
      p.push_back({s, r, T - r});
      s += r;
    }
  }
  return p;
}

// Rounds of a pass, in ascending stage order (mirrors k_ntt16_pass): (first local stage, number of stages).
static std::vector<std::pair<int, int>> pass_rounds(int r_bits) {
  std::vector<std::pair<int, int>> r;
  const int rem = r_bits & 3;
  int st = 0;
  if (rem) {
    r.push_back({0, rem});
    st = rem;
  }
  for (; st < r_bits; st += 4) r.push_back({st, 4});
  return r;
}
// Twiddle planes of every round of every pass of a 2^log_n transform, laid out in the order the lanes read
// them: plane(s, G)[e - 1][gm] = w^((gm * rev_G(e)) << (log_n - s - G)), gm < 2^s.  ~N entries in total, built
// once per (size, direction).  (Indexing one w^k table instead scatters a wave's loads over one 64 B sector
// per lane; at 2^24 that table no longer fits the caches and the passes turned memory bound.)
__global__ void k_fill_round_plane(u64* out, int s, int G, int log_n, PowTable t) {
  const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (idx >= ((size_t)((1 << G) - 1) << s)) return;
  const u32 e = (u32)(idx >> s) + 1;
  const size_t gm = idx & (((size_t)1 << s) - 1);
  const u32 rho = bitrev32(e, G);
  size_t ex = (gm * rho) << (log_n - s - G);
  u64 r = 1;
#pragma unroll 1
  for (int i = 0; ex; i++, ex >>= 1)
    if (ex & 1) r = gl_mul(r, t.pw[i]);
  out[idx] = r;
}
struct NttPlanes {
  const u64* base;
  std::vector<size_t> off;  // [pass * 4 + round]
};
static NttPlanes ntt_planes(mh_ctx* c, int log_n, bool inverse, const std::vector<PassPlan>& plan) {
  const std::string key = "nttp:" + std::to_string(log_n) + (inverse ? ":i" : ":f");
  auto it = c->tables.find(key);
  if (it == c->tables.end()) {
    std::vector<size_t> off(plan.size() * 4, 0);
    size_t total = 1;
    for (size_t i = 0; i < plan.size(); i++) {
      auto rounds = pass_rounds(plan[i].r_bits);
      for (size_t k = 0; k < rounds.size(); k++) {
        const int s = plan[i].s_lo + rounds[k].first, G = rounds[k].second;
        off[i * 4 + k] = total;
        if (s > 0) total += (size_t)((1 << G) - 1) << s;
      }
    }
    DevBuf b(total * 8);
    PowTable t;
    u64 w = gl_two_adic_generator(log_n);
    if (inverse) w = gl_inv(w);
    for (int i = 0; i < 32; i++) {
      t.pw[i] = w;
      w = gl_sqr(w);
    }
    for (size_t i = 0; i < plan.size(); i++) {
      auto rounds = pass_rounds(plan[i].r_bits);
      for (size_t k = 0; k < rounds.size(); k++) {
        const int s = plan[i].s_lo + rounds[k].first, G = rounds[k].second;
        if (s == 0) continue;
        const size_t cnt = (size_t)((1 << G) - 1) << s;
        MH_LAUNCH(k_fill_round_plane, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, b.u() + off[i * 4 + k], s, G,
                           log_n, t);
      }
    }
    c->table_index[key] = off;
    it = c->tables.emplace(key, std::move(b)).first;
  }
  return NttPlanes{it->second.u(), c->table_index[key]};
}

static void launch_pass(mh_ctx* c, NttPassArgs a, size_t n_cols, size_t n_z, bool one_coset_per_workgroup = false) {
  size_t tiles = (size_t)1 << (a.log_n - a.r_bits - a.cb);
  // the workgroup loops over the output cosets; they are spread over grid.z only as far as it takes to fill the chip
  // (a quotient chunk is 2 columns: 512 workgroups at 2^20 otherwise)
  MH_REQUIRE(n_z >= 1 && (n_z & (n_z - 1)) == 0, "internal: the number of output cosets of a pass must be a power of two");
  size_t zsplit = 1;
  while (zsplit < n_z && (one_coset_per_workgroup || tiles * n_cols * zsplit < 4096)) zsplit *= 2;
  a.n_z = (u32)(n_z / zsplit);
  static const int colfast = [] { const char* e = getenv("MH_NTT_COLFAST"); return e ? atoi(e) : 1; }();
  a.col_fastest = (colfast && tiles <= 65535) ? 1 : 0;
  dim3 grid(a.col_fastest ? (unsigned)n_cols : (unsigned)tiles, a.col_fastest ? (unsigned)tiles : (unsigned)n_cols, (unsigned)zsplit);
  const int T = ntt_tile_log(a.log_n);
  const size_t lds = ntt_lds_bytes(T);
  if (T == NTT_TILE_LOG) {
    if (a.dif) MH_LAUNCH((k_ntt16_pass<true, NTT_THREADS>), grid, dim3(NTT_THREADS), lds, c->stream, a);
    else if (a.step && a.n_z > 1 && NTT_SWZ && a.r_bits + a.cb == NTT_TILE_LOG && a.cb == 0)
      MH_LAUNCH((k_ntt16_pass<false, NTT_THREADS, 1>), grid, dim3(NTT_THREADS), lds, c->stream, a);
    else MH_LAUNCH((k_ntt16_pass<false, NTT_THREADS>), grid, dim3(NTT_THREADS), lds, c->stream, a);
  } else {
    if (!c->ntt_big_lds_attr) {  // > 64 KB of dynamic LDS must be requested per kernel AND per device: a flag of the ctx, not of the process
      HIP_CHECK(hipFuncSetAttribute((const void*)k_ntt16_pass<true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      HIP_CHECK(hipFuncSetAttribute((const void*)k_ntt16_pass<false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      c->ntt_big_lds_attr = true;
    }
    if (a.dif) MH_LAUNCH((k_ntt16_pass<true, 1024>), grid, dim3(1024), lds, c->stream, a);
    else MH_LAUNCH((k_ntt16_pass<false, 1024>), grid, dim3(1024), lds, c->stream, a);
  }
}

// In-place inverse DFT (unscaled: result = N * coefficients) of `n_cols` contiguous columns of
// length 2^log_n: natural-order evaluations in, BIT-REVERSED coefficients out.
void ntt_inverse_dif_inplace(mh_ctx* c, u64* cols, size_t n_cols, int log_n) { ntt_inverse_dif(c, cols, cols, n_cols, log_n); }

// The same, reading `src` and leaving the result in `dst` (the first pass moves the data: no copy is needed before an
// in-place transform of a buffer that must stay intact).  src == dst: in place.
void ntt_inverse_dif(mh_ctx* c, const u64* src, u64* dst, size_t n_cols, int log_n) {
  if (log_n == 0) {
    if (src != dst) HIP_CHECK(hipMemcpyAsync(dst, src, n_cols * 8, hipMemcpyDeviceToDevice, c->stream));
    return;
  }
  auto plan = plan_passes(log_n);
  const NttPlanes tw = ntt_planes(c, log_n, true, plan);
  for (int i = (int)plan.size() - 1; i >= 0; i--) {
    NttPassArgs a{};
    a.src = (i == (int)plan.size() - 1) ? src : dst; a.dst = dst;
    a.src_col_stride = a.dst_col_stride = (size_t)1 << log_n;
    a.dst_z_stride = 0;
    a.log_n = log_n; a.s_lo = plan[i].s_lo; a.r_bits = plan[i].r_bits; a.cb = plan[i].cb;
    a.dif = 1; a.scale_lo = nullptr; a.scale_hi = nullptr;
    for (int k = 0; k < 4; k++) a.tw_round[k] = tw.base + tw.off[i * 4 + k];
    a.canon_out = 0;  // the coefficients only feed the forward passes' multiplications
    static const int direct = [] { const char* e = getenv("MH_NTT_DIRECT"); return e ? atoi(e) : 1; }();
    a.direct_first = direct && NTT_SWZ && a.r_bits >= 4 && a.r_bits + a.cb == NTT_TILE_LOG;
    a.rot = i == 0 && ntt_coef_rot(log_n);
    launch_pass(c, a, n_cols, 1);
  }
}

// Coset tables: for each output coset z (base s_z): lo[z][x] = s_z^x (x < 2^lb),
// hi[z][y] = s_z^(y * 2^lb) * post_scale (y < 2^(log_n - lb)).
struct CosetTables {
  const u64* lo;
  const u64* hi;
  int lb;
};
// Built once per (size, set of cosets) and kept in the context: a proof re-uses the same ten sets every time
// (main/aux LDE, one per quotient chunk), 160 tiny launches per proof otherwise.
static CosetTables coset_tables(mh_ctx* c, int log_n, const std::vector<u64>& bases, u64 post_scale) {
  CosetTables t;
  t.lb = (log_n + 1) / 2;
  const size_t nlo = (size_t)1 << t.lb, nhi = (size_t)1 << (log_n - t.lb);
  u64 h = 0xcbf29ce484222325ULL ^ (u64)log_n;
  for (u64 b : bases) h = (h ^ b) * 0x100000001b3ULL + (h >> 29);
  h = (h ^ post_scale) * 0x100000001b3ULL;
  const std::string key = "coset:" + std::to_string(log_n) + ":" + std::to_string(bases.size()) + ":" + std::to_string(h);
  auto it = c->tables.find(key);
  if (it == c->tables.end()) {
    DevBuf b((bases.size() * (nlo + nhi)) * 8);
    for (size_t z = 0; z < bases.size(); z++) {
      fill_powers(c, b.u() + z * nlo, nlo, bases[z], 1);
      fill_powers(c, b.u() + bases.size() * nlo + z * nhi, nhi, gl_exp_pow2(bases[z], t.lb), post_scale);
    }
    // keep the exact key material so that a hash collision cannot hand back another set's table
    std::vector<size_t> ident(bases.begin(), bases.end());
    ident.push_back((size_t)post_scale);
    c->table_index[key] = ident;
    it = c->tables.emplace(key, std::move(b)).first;
  } else {
    const std::vector<size_t>& ident = c->table_index[key];
    bool same = ident.size() == bases.size() + 1 && ident.back() == (size_t)post_scale;
    for (size_t z = 0; same && z < bases.size(); z++) same = ident[z] == (size_t)bases[z];
    MH_REQUIRE(same, "internal: coset table key collision");
  }
  t.lo = it->second.u();
  t.hi = it->second.u() + bases.size() * nlo;
  return t;
}

// The whole coset-scale table [z][pos] = lo[z][k & m] * hi[z][k >> lb], k = bitrev(pos): one load and one product per element in the
// first forward pass instead of two loads and two products.  8 B per LDE element (64 MB at 2^20 rows, blowup 8): worth it for the
// many-column trace LDEs only.  Measured at 2^20 x (51 + 16) columns (round 4, ms of LDE per proof): two-level tables 8.70, full table
// 8.22, full table + column-fastest grid (the 256 KB slice of a tile is shared by neighbouring workgroups in L2) 8.08; the
// column-fastest grid alone 8.72.  MH_NTT_FULLSCALE=0 / MH_NTT_COLFAST=0 restore the round-3 path.
__global__ void k_fill_scale_full(u64* out, const u64* lo, const u64* hi, int log_n, int lb, size_t lo_z, size_t hi_z, int rot) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t n = (size_t)1 << log_n;
  const size_t z = blockIdx.y;
  if (i >= n) return;
  const u32 il = rot ? (((u32)i & ~4095u) | (((u32)i & 255u) << 4) | (((u32)i >> 8) & 15u)) : (u32)i;  // position -> element (ntt_coef_rot)
  const u32 k = bitrev32(il, log_n);
  out[z * n + i] = gl_mul(lo[z * lo_z + (k & ((1u << lb) - 1))], hi[z * hi_z + (k >> lb)]);
}
static const u64* coset_scale_full(mh_ctx* c, int log_n, const std::vector<u64>& bases, u64 post_scale, const CosetTables& t) {
  const int rot = ntt_coef_rot(log_n) ? 1 : 0;
  u64 h = 0xcbf29ce484222325ULL ^ (u64)log_n;
  for (u64 b : bases) h = (h ^ b) * 0x100000001b3ULL + (h >> 29);
  h = (h ^ post_scale) * 0x100000001b3ULL;
  const std::string key = "cosetfull:" + std::to_string(log_n) + ":" + std::to_string(bases.size()) + ":" + std::to_string(h) + (rot ? ":rot" : "");
  auto it = c->tables.find(key);
  if (it == c->tables.end()) {
    const size_t n = (size_t)1 << log_n;
    DevBuf b(bases.size() * n * 8);
    MH_LAUNCH(k_fill_scale_full, dim3((unsigned)((n + 255) / 256), (unsigned)bases.size()), dim3(256), 0, c->stream, b.u(), t.lo, t.hi, log_n,
              t.lb, (size_t)1 << t.lb, (size_t)1 << (log_n - t.lb), rot);
    it = c->tables.emplace(key, std::move(b)).first;
  }
  return it->second.u();
}

// Forward coset evaluation: `coef_br` holds, per column, N*coefficients in bit-reversed order
// (output of ntt_inverse_dif_inplace).  For every output coset z, out[(col*n_z + z)*N + r] =
// sum_k c_k * bases[z]^k * w_N^(r k)   (1/N folded into the table).
// out_col_stride != 0: the cosets are a GROUP of a wider coset-major matrix -- `out` points at the group's first coset of column 0,
// consecutive columns lie out_col_stride apart (e.g. 8 N for blowup 8 while bases.size() = 2).
// group_cols != 0: the columns come in n_cols / group_cols groups, each with its OWN coset shifts (the chunks of the quotient, whose
// coefficients belong to different input cosets): bases = [group][z], all groups in one pair of launches instead of one pair each
// (a quotient chunk is 2 columns: 512 workgroups per launch at 2^20 rows, 43-55 us per column against 36-44 in a 16-column launch).
void ntt_forward_cosets(mh_ctx* c, const u64* coef_br, size_t n_cols, int log_n, const std::vector<u64>& bases,
                        u64* out, size_t out_col_stride, size_t group_cols) {
  size_t N = (size_t)1 << log_n;
  const size_t n_groups = group_cols ? n_cols / group_cols : 1;
  MH_REQUIRE(n_groups >= 1 && (!group_cols || n_cols % group_cols == 0) && bases.size() % n_groups == 0, "internal: coset groups");
  size_t nz = bases.size() / n_groups;
  const bool grouped = out_col_stride != 0 && out_col_stride != nz * N;
  u64 n_inv = gl_inv((u64)N % GL_P);
  const CosetTables t = coset_tables(c, log_n, bases, n_inv);
  auto plan = plan_passes(log_n);
  const NttPlanes tw = ntt_planes(c, log_n, false, plan);
  // $MH_NTT_COLGROUP = k > 0: the passes run column group by column group (k columns x nz cosets through ALL passes before the next
  // group), so that what pass i wrote is still in the 256 MB Infinity Cache when pass i + 1 reads it (2^20 rows, blowup 8: 64 MB per
  // column -- k = 3 keeps a group at 192 MB).  0 = every pass over all columns (the inter-pass matrix, 3.4 GB for 51 columns, goes
  // through HBM).
  static const int colgroup = [] { const char* e = getenv("MH_NTT_COLGROUP"); return e ? atoi(e) : 0; }();
  static thread_local bool inside = false;  // this call is one column group of an outer call
  if (colgroup > 0 && !group_cols && !grouped && plan.size() > 1 && n_cols > (size_t)colgroup) {
    if (!inside) {
      inside = true;
      try {
        for (size_t c0 = 0; c0 < n_cols; c0 += (size_t)colgroup) {
          const size_t k = std::min<size_t>((size_t)colgroup, n_cols - c0);
          ntt_forward_cosets(c, coef_br + c0 * N, k, log_n, bases, out + c0 * nz * N, out_col_stride, group_cols);
        }
      } catch (...) { inside = false; throw; }
      inside = false;
      return;
    }
  }
  for (size_t i = 0; i < plan.size(); i++) {
    NttPassArgs a{};
    a.dst = out;
    a.dst_col_stride = grouped ? out_col_stride : nz * N;
    a.dst_z_stride = N;
    if (i == 0) {
      a.src = coef_br;
      a.src_col_stride = N;
      a.scale_lo = t.lo; a.scale_hi = t.hi; a.lb = t.lb;
      a.scale_lo_z = (size_t)1 << t.lb;
      a.scale_hi_z = (size_t)1 << (log_n - t.lb);
      static const int fullscale = [] { const char* e = getenv("MH_NTT_FULLSCALE"); return e ? atoi(e) : 1; }();
      a.group_cols = (u32)group_cols; a.group_z = (u32)nz;
      a.rot = ntt_coef_rot(log_n);
      if (fullscale && !group_cols && (n_cols >= 16 || inside) && log_n >= 12 && log_n <= 24) a.scale_full = coset_scale_full(c, log_n, bases, n_inv, t);
      // the geometric stepping of the scaled coefficients (registers over the coset loop, one shared 8 MB step table) is off by default:
      // since the strided pass stopped staging through LDS (first-round register loads) the full [z][pos] table is the faster form
      // again (2^20 x 51 + 8 EF: lde 7.77 vs 8.25 ms, proof 46.0 vs 46.25 ms, three alternating runs on one box; gpurun_out/nttexp.txt)
      static const int geo = [] { const char* e = getenv("MH_NTT_STEP"); return e ? atoi(e) : 0; }();
      if (geo && nz > 1 && log_n >= NTT_TILE_LOG && log_n <= 24) {  // bases[z + 1] = bases[z] * ratio for every z?
        const u64 ratio = gl_mul(bases[1], gl_inv(bases[0]));
        bool geometric = true;
        for (size_t g = 0; g < n_groups; g++)
          for (size_t z = 0; z + 1 < nz; z++) geometric = geometric && gl_mul(bases[g * nz + z], ratio) == bases[g * nz + z + 1];
        if (geometric) {
          const std::vector<u64> rb{ratio};
          a.step = coset_scale_full(c, log_n, rb, 1, coset_tables(c, log_n, rb, 1));
        }
      }
    } else {
      // in place inside each coset block: fold z into the source stride
      a.src = out;
      a.src_col_stride = N;  // with grid.y = n_cols*nz and z-stride 0 (see below)
      a.scale_lo = nullptr; a.scale_hi = nullptr;
    }
    a.log_n = log_n; a.s_lo = plan[i].s_lo; a.r_bits = plan[i].r_bits; a.cb = plan[i].cb;
    a.dif = 0;
    for (int k = 0; k < 4; k++) a.tw_round[k] = tw.base + tw.off[i * 4 + k];
    a.canon_out = i + 1 == plan.size();
    static const int direct = [] { const char* e = getenv("MH_NTT_DIRECT"); return e ? atoi(e) : 1; }();
    a.direct_first = direct && i > 0 && NTT_SWZ && a.cb == 4 && a.r_bits >= 8 && a.r_bits % 4 == 0 && a.r_bits + a.cb == NTT_TILE_LOG;
    if (i == 0) {
      launch_pass(c, a, n_cols, nz);
    } else if (grouped) {  // in place, one workgroup per (tile, column, coset): the columns of the group are not contiguous
      a.src_col_stride = out_col_stride;
      a.src_z_stride = N;
      launch_pass(c, a, n_cols, nz, true);
    } else {
      static const int zloop = [] { const char* e = getenv("MH_NTT_ZLOOP"); return e ? atoi(e) : 1; }();
      if (zloop && nz > 1) {  // the workgroup walks the cosets of its (tile, column): 1/nz of the workgroup launches
        a.src_col_stride = nz * N;
        a.src_z_stride = N;
        launch_pass(c, a, n_cols, nz);
      } else {
        a.dst_col_stride = N;
        a.dst_z_stride = 0;
        launch_pass(c, a, n_cols * nz, 1);
      }
    }
  }
}

// LDE of column-major columns: evaluations on a*H (natural) -> evaluations on b_z*H for all z.
void lde_columns(mh_ctx* c, const u64* cols_in, size_t n_cols, int log_n, u64 in_shift, const std::vector<u64>& out_shifts,
                 u64* out, u64* scratch /* n_cols * N */) {
  {
    ProfScope ps(c, "lde_intt", 16.0 * (double)n_cols * (double)((size_t)1 << log_n));  // nested in "lde": the part every rank of a sharded proof repeats
    ntt_inverse_dif(c, cols_in, scratch, n_cols, log_n);
  }
  u64 a_inv = gl_inv(in_shift);
  std::vector<u64> bases;
  for (u64 b : out_shifts) bases.push_back(gl_mul(b, a_inv));
  ntt_forward_cosets(c, scratch, n_cols, log_n, bases, out);
}

// The two halves of lde_columns, for a caller that pipelines the forward transforms by groups of output cosets (commit_traces):
// coefficients once, then each group into its place of the full coset-major matrix (`out` = first coset of the group, column 0).
void lde_coefficients(mh_ctx* c, const u64* cols_in, size_t n_cols, int log_n, u64* coef_br) {
  ProfScope ps(c, "lde_intt", 16.0 * (double)n_cols * (double)((size_t)1 << log_n));
  ntt_inverse_dif(c, cols_in, coef_br, n_cols, log_n);
}
void lde_forward_group(mh_ctx* c, const u64* coef_br, size_t n_cols, int log_n, u64 in_shift, const std::vector<u64>& group_shifts,
                       u64* out, size_t out_col_stride) {
  const u64 a_inv = gl_inv(in_shift);
  std::vector<u64> bases;
  for (u64 b : group_shifts) bases.push_back(gl_mul(b, a_inv));
  ntt_forward_cosets(c, coef_br, n_cols, log_n, bases, out, out_col_stride);
}
