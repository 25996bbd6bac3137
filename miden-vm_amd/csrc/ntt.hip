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
//   * each pass stages a 2^12-element tile (32 KB) in LDS and runs up to 12 butterfly stages on
//     it; strided passes move >=128-byte contiguous segments.
// Roofline: HBM-bound; algorithmic bytes per column = (1 + B) * N * 8 (read trace once, write
// LDE once); this first implementation moves (4 + 4B) * N * 8.
#include "ctx.hpp"
#include "gl.cuh"
#include "kernels.hpp"

static constexpr int NTT_TILE_LOG = 12;
static constexpr int NTT_THREADS = 256;

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
  hipLaunchKernelGGL(k_fill_powers, dim3((n + 255) / 256), dim3(256), 0, c->stream, out, n, t, scale);
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

void launch_transpose_rm_to_cm(mh_ctx* c, const u64* in, u64* out, size_t n, size_t w) {
  dim3 grid((unsigned)((n + 31) / 32), (unsigned)((w + 31) / 32));
  hipLaunchKernelGGL(k_transpose_rm_to_cm, grid, dim3(256), 0, c->stream, in, out, n, w);
}

// ---------------------------------------------------------------------------------------------
struct NttPassArgs {
  const u64* src;
  u64* dst;
  size_t src_col_stride, dst_col_stride, dst_z_stride;  // in elements
  int log_n, s_lo, r_bits, cb;                          // stages s_lo .. s_lo+r_bits-1, tile = 2^(r_bits+cb)
  int dif;                                              // 1 = DIF (a+b,(a-b)w), descending; 0 = DIT
  const u64* tw;                                        // w^k (k < N/2) for the transform direction
  const u64* scale_lo;                                  // optional: multiply on load by
  const u64* scale_hi;                                  //   scale_lo[z][k & m] * scale_hi[z][k >> lb], k = bitrev(pos)
  int lb;
  size_t scale_lo_z, scale_hi_z;
};

__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass(NttPassArgs a) {
  __shared__ u64 lds[1 << NTT_TILE_LOG];
  const int tile_log = a.r_bits + a.cb;
  const u32 tile_n = 1u << tile_log;
  const u32 cb_mask = (1u << a.cb) - 1;
  const u32 tile = blockIdx.x;
  const int lo_bits = a.s_lo - a.cb;
  const size_t lo0 = tile & ((1u << lo_bits) - 1);
  const size_t hi = tile >> lo_bits;
  const size_t gbase = (hi << (a.s_lo + a.r_bits)) | (lo0 << a.cb);
  const u64* src = a.src + (size_t)blockIdx.y * a.src_col_stride;
  u64* dst = a.dst + (size_t)blockIdx.y * a.dst_col_stride + (size_t)blockIdx.z * a.dst_z_stride;

  for (u32 l = threadIdx.x; l < tile_n; l += NTT_THREADS) {
    size_t g = gbase | ((size_t)(l >> a.cb) << a.s_lo) | (l & cb_mask);
    u64 v = src[g];
    if (a.scale_lo) {
      u32 k = bitrev32((u32)g, a.log_n);
      u64 s = gl_mul(a.scale_lo[blockIdx.z * a.scale_lo_z + (k & ((1u << a.lb) - 1))],
                     a.scale_hi[blockIdx.z * a.scale_hi_z + (k >> a.lb)]);
      v = gl_mul(v, s);
    }
    lds[l] = v;
  }
  __syncthreads();

  // Butterflies: two stages at a time on 4 LDS elements held in registers (one radix-4 step =
  // 4 twiddle multiplications, like two radix-2 stages, but half the LDS traffic, half the barriers
  // and one index computation per 4 butterflies); a lone radix-2 stage when r_bits is odd.
  // Global index of local l modulo 2^s:  gm(l, st) = ((l >> cb) mod 2^st) << s_lo | lo0 << cb | l mod 2^cb.
  auto single_stage = [&](int st) {
    const int b = st + a.cb, s = a.s_lo + st, tw_shift = a.log_n - s - 1;
    for (u32 q = threadIdx.x; q < (tile_n >> 1); q += NTT_THREADS) {
      const u32 low = q & ((1u << b) - 1);
      const u32 l0 = ((q >> b) << (b + 1)) | low, l1 = l0 | (1u << b);
      const size_t gm = (((size_t)((l0 >> a.cb) & ((1u << st) - 1))) << a.s_lo) | (lo0 << a.cb) | (l0 & cb_mask);
      const u64 w = a.tw[gm << tw_shift];
      const u64 x = lds[l0], y = lds[l1];
      if (a.dif) {
        lds[l0] = gl_add(x, y);
        lds[l1] = gl_mul(gl_sub(x, y), w);
      } else {
        const u64 wy = gl_mul(y, w);
        lds[l0] = gl_add(x, wy);
        lds[l1] = gl_sub(x, wy);
      }
    }
    __syncthreads();
  };
  auto double_stage = [&](int st) {  // local stages st (low) and st + 1 (high)
    const int b = st + a.cb, s = a.s_lo + st;
    const int sh_hi = a.log_n - s - 2;  // table shift of the higher stage (twiddle order 2^(s+2))
    const size_t quarter = (size_t)1 << (a.log_n - 2);
    for (u32 q = threadIdx.x; q < (tile_n >> 2); q += NTT_THREADS) {
      const u32 low = q & ((1u << b) - 1);
      const u32 l00 = ((q >> b) << (b + 2)) | low;
      const u32 l01 = l00 | (1u << b), l10 = l00 | (2u << b), l11 = l00 | (3u << b);
      const size_t k = (((size_t)((l00 >> a.cb) & ((1u << st) - 1))) << a.s_lo) | (lo0 << a.cb) | (l00 & cb_mask);
      // higher stage: w_{2^(s+2)}^k for (l00,l10), times w_4 for (l01,l11); lower stage: its square
      const u64 wh0 = a.tw[k << sh_hi], wh1 = a.tw[(k << sh_hi) + quarter];
      const u64 wl = a.tw[k << (sh_hi + 1)];
      u64 x0 = lds[l00], x1 = lds[l01], x2 = lds[l10], x3 = lds[l11];
      if (a.dif) {  // high stage first, then low
        u64 t0 = gl_add(x0, x2), t2 = gl_mul(gl_sub(x0, x2), wh0);
        u64 t1 = gl_add(x1, x3), t3 = gl_mul(gl_sub(x1, x3), wh1);
        x0 = gl_add(t0, t1); x1 = gl_mul(gl_sub(t0, t1), wl);
        x2 = gl_add(t2, t3); x3 = gl_mul(gl_sub(t2, t3), wl);
      } else {  // low stage first, then high
        u64 m1 = gl_mul(x1, wl), m3 = gl_mul(x3, wl);
        u64 t0 = gl_add(x0, m1), t1 = gl_sub(x0, m1), t2 = gl_add(x2, m3), t3 = gl_sub(x2, m3);
        u64 n2 = gl_mul(t2, wh0), n3 = gl_mul(t3, wh1);
        x0 = gl_add(t0, n2); x2 = gl_sub(t0, n2);
        x1 = gl_add(t1, n3); x3 = gl_sub(t1, n3);
      }
      lds[l00] = x0; lds[l01] = x1; lds[l10] = x2; lds[l11] = x3;
    }
    __syncthreads();
  };
  if (a.dif) {
    int st = a.r_bits - 1;
    if (a.r_bits & 1) single_stage(st--);
    for (; st >= 1; st -= 2) double_stage(st - 1);
  } else {
    int st = 0;
    for (; st + 1 < a.r_bits; st += 2) double_stage(st);
    if (st < a.r_bits) single_stage(st);
  }

  for (u32 l = threadIdx.x; l < tile_n; l += NTT_THREADS) {
    size_t g = gbase | ((size_t)(l >> a.cb) << a.s_lo) | (l & cb_mask);
    dst[g] = lds[l];
  }
}

struct PassPlan {
  int s_lo, r_bits, cb;
};
// stages [0, log_n) split into one contiguous pass (low stages) + strided passes, ascending order.
static std::vector<PassPlan> plan_passes(int log_n) {
  std::vector<PassPlan> p;
  int c = log_n < NTT_TILE_LOG ? log_n : NTT_TILE_LOG;
  p.push_back({0, c, 0});
  int rem = log_n - c;
  if (rem > 0) {
    const int max_r = NTT_TILE_LOG - 4;  // keep >= 16 consecutive elements (128 B) per segment
    int np = (rem + max_r - 1) / max_r;
    int s = c;
    for (int i = 0; i < np; i++) {
      int r = rem / np + (i < rem % np ? 1 : 0);
      p.push_back({s, r, NTT_TILE_LOG - r});
      s += r;
    }
  }
  return p;
}

static void launch_pass(mh_ctx* c, NttPassArgs a, size_t n_cols, size_t n_z) {
  size_t tiles = (size_t)1 << (a.log_n - a.r_bits - a.cb);
  dim3 grid((unsigned)tiles, (unsigned)n_cols, (unsigned)n_z);
  hipLaunchKernelGGL(k_ntt_pass, grid, dim3(NTT_THREADS), 0, c->stream, a);
}

// In-place inverse DFT (unscaled: result = N * coefficients) of `n_cols` contiguous columns of
// length 2^log_n: natural-order evaluations in, BIT-REVERSED coefficients out.
void ntt_inverse_dif_inplace(mh_ctx* c, u64* cols, size_t n_cols, int log_n) {
  if (log_n == 0) return;
  auto plan = plan_passes(log_n);
  const u64* tw = c->twiddles(log_n, true);
  for (int i = (int)plan.size() - 1; i >= 0; i--) {
    NttPassArgs a{};
    a.src = cols; a.dst = cols;
    a.src_col_stride = a.dst_col_stride = (size_t)1 << log_n;
    a.dst_z_stride = 0;
    a.log_n = log_n; a.s_lo = plan[i].s_lo; a.r_bits = plan[i].r_bits; a.cb = plan[i].cb;
    a.dif = 1; a.tw = tw; a.scale_lo = nullptr; a.scale_hi = nullptr;
    launch_pass(c, a, n_cols, 1);
  }
}

// Coset tables: for each output coset z (base s_z): lo[z][x] = s_z^x (x < 2^lb),
// hi[z][y] = s_z^(y * 2^lb) * post_scale (y < 2^(log_n - lb)).
struct CosetTables {
  DevBuf lo, hi;
  int lb;
};
static CosetTables make_coset_tables(mh_ctx* c, int log_n, const std::vector<u64>& bases, u64 post_scale) {
  CosetTables t;
  t.lb = (log_n + 1) / 2;
  size_t nlo = (size_t)1 << t.lb, nhi = (size_t)1 << (log_n - t.lb);
  t.lo.alloc(bases.size() * nlo * 8);
  t.hi.alloc(bases.size() * nhi * 8);
  for (size_t z = 0; z < bases.size(); z++) {
    fill_powers(c, t.lo.u() + z * nlo, nlo, bases[z], 1);
    fill_powers(c, t.hi.u() + z * nhi, nhi, gl_exp_pow2(bases[z], t.lb), post_scale);
  }
  return t;
}

// Forward coset evaluation: `coef_br` holds, per column, N*coefficients in bit-reversed order
// (output of ntt_inverse_dif_inplace).  For every output coset z, out[(col*n_z + z)*N + r] =
// sum_k c_k * bases[z]^k * w_N^(r k)   (1/N folded into the table).
void ntt_forward_cosets(mh_ctx* c, const u64* coef_br, size_t n_cols, int log_n, const std::vector<u64>& bases,
                        u64* out) {
  size_t N = (size_t)1 << log_n;
  size_t nz = bases.size();
  u64 n_inv = gl_inv((u64)N % GL_P);
  CosetTables t = make_coset_tables(c, log_n, bases, n_inv);
  auto plan = plan_passes(log_n);
  const u64* tw = log_n ? c->twiddles(log_n, false) : nullptr;
  for (size_t i = 0; i < plan.size(); i++) {
    NttPassArgs a{};
    a.dst = out;
    a.dst_col_stride = nz * N;
    a.dst_z_stride = N;
    if (i == 0) {
      a.src = coef_br;
      a.src_col_stride = N;
      a.scale_lo = t.lo.u(); a.scale_hi = t.hi.u(); a.lb = t.lb;
      a.scale_lo_z = (size_t)1 << t.lb;
      a.scale_hi_z = (size_t)1 << (log_n - t.lb);
    } else {
      // in place inside each coset block: fold z into the source stride
      a.src = out;
      a.src_col_stride = N;  // with grid.y = n_cols*nz and z-stride 0 (see below)
      a.scale_lo = nullptr; a.scale_hi = nullptr;
    }
    a.log_n = log_n; a.s_lo = plan[i].s_lo; a.r_bits = plan[i].r_bits; a.cb = plan[i].cb;
    a.dif = 0; a.tw = tw;
    if (i == 0) {
      launch_pass(c, a, n_cols, nz);
    } else {
      a.dst_col_stride = N;
      a.dst_z_stride = 0;
      launch_pass(c, a, n_cols * nz, 1);
    }
  }
  // the tables must outlive the kernels that read them
  HIP_CHECK(hipStreamSynchronize(c->stream));
}

// LDE of column-major columns: evaluations on a*H (natural) -> evaluations on b_z*H for all z.
void lde_columns(mh_ctx* c, const u64* cols_in, size_t n_cols, int log_n, u64 in_shift, const std::vector<u64>& out_shifts,
                 u64* out, u64* scratch /* n_cols * N */) {
  size_t N = (size_t)1 << log_n;
  HIP_CHECK(hipMemcpyAsync(scratch, cols_in, n_cols * N * 8, hipMemcpyDeviceToDevice, c->stream));
  ntt_inverse_dif_inplace(c, scratch, n_cols, log_n);
  u64 a_inv = gl_inv(in_shift);
  std::vector<u64> bases;
  for (u64 b : out_shifts) bases.push_back(gl_mul(b, a_inv));
  ntt_forward_cosets(c, scratch, n_cols, log_n, bases, out);
}
