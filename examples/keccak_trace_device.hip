// A GPU trace generator on the other side of the boundary ("keep matrices on device"): the main traces of the
// precompile prover's Keccak round chiplet and of its byte-pair table are BUILT in HBM and handed to the prover with
// mh_trace_from_device -- they never cross PCIe.  Not part of libmidenhip (trace building is the client's job, as in the reference:
// precompiles-prover/src/hash/keccak/round/mod.rs:725-806 `generate_trace_from_states_inner`, primitives/byte_pair_lut.rs:294-303
// `generate_trace`); it uses the public C ABI only.
//
// The chiplet is a three-address machine `c = ROL(a OP b, s)`: one wave runs the 3200 program rows of ONE permutation (address space in
// LDS), lane c writes column c of its 34-column band and counts the byte-pair / Range16 requests of every active row into the table's
// multiplicity matrix (64-bit atomics in HBM).  The
// 128-slot round program (op, rotation, back-offsets, provide multiplicity) is passed in by the caller: the kernel is the machine,
// not the Keccak schedule.
//
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libkeccak_trace_device.so keccak_trace_device.hip   (examples/Makefile)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;

enum { OP_NOP = 0, OP_XOR = 1, OP_ANDNOT = 2, OP_ROL = 3, OP_XORROL = 4 };
enum { ROUND_PERIOD = 128, NUM_ROUNDS = 24, PERM_CYCLE = 25 * 128, IP_BOUNDARY = 25, LANE_WIDTH = 34, NUM_LANES = 2 };
enum { COL_IP = 0, COL_A = 1, COL_B = 9, COL_R = 17, COL_ROT = 25, COL_ACT = 33 };

struct Slot { int op, shift, back_a, back_b, mult; };
struct Program { Slot s[ROUND_PERIOD]; };

__device__ __forceinline__ u64 rol64(u64 x, int s) { return s ? (x << s) | (x >> (64 - s)) : x; }

// One WAVE per permutation: the machine's address space (3225 words: the initial state, RC[r] at 25 + 128 r, the value written by
// program row r at 25 + r) lives in LDS, every lane reads the two operands of the row (broadcast), lane c writes column c of the 34-column
// band (one contiguous 272-byte store per row) and raises the request that belongs to its column: lanes 1-8 the byte-pair request of byte
// c - 1, lanes 25-32 the Range16 request of limb c - 25 (64-bit atomics on the table's multiplicity matrix in HBM).
// memory_out: [n_perms][MEM_WORDS] (the address spaces, for the caller's output extraction); trace: row-major [height][68]; counts:
// row-major [65536][3] = the table's (andnot, xor, range16) multiplicities.
enum { MEM_WORDS = IP_BOUNDARY + PERM_CYCLE };
__global__ __launch_bounds__(64) void k_keccak_round_trace(Program prog, const u64* __restrict__ states, const u64* __restrict__ rcs,
                                                           u64* __restrict__ memory_out, int n_perms, int perms_per_lane,
                                                           u64* __restrict__ trace, u64* __restrict__ counts) {
  __shared__ u64 mem[MEM_WORDS];
  const int p = blockIdx.x, c = threadIdx.x;
  for (int i = c; i < MEM_WORDS; i += 64) mem[i] = 0;
  __syncthreads();
  if (c < 25) mem[c] = states[(size_t)p * 25 + c];
  if (c < NUM_ROUNDS) mem[IP_BOUNDARY + c * ROUND_PERIOD] = rcs[c];
  __syncthreads();
  const int lane = p / perms_per_lane, p_in_lane = p - lane * perms_per_lane;
  u64* row = trace + ((size_t)p_in_lane * PERM_CYCLE) * (LANE_WIDTH * NUM_LANES) + lane * LANE_WIDTH;
  const int active = NUM_ROUNDS * ROUND_PERIOD;
  for (int r = 0; r < PERM_CYCLE; r++, row += LANE_WIDTH * NUM_LANES) {
    const Slot sl = prog.s[r & (ROUND_PERIOD - 1)];
    const bool act = r < active;
    const u64 a = sl.op != OP_NOP ? mem[IP_BOUNDARY + r - sl.back_a] : 0;
    const u64 b = (sl.op == OP_XOR || sl.op == OP_ANDNOT || sl.op == OP_XORROL) ? mem[IP_BOUNDARY + r - sl.back_b] : 0;
    const u64 rv = (sl.op == OP_XOR || sl.op == OP_XORROL) ? (a ^ b) : (sl.op == OP_ANDNOT ? (~a & b) : a);
    const bool rot = sl.op == OP_ROL || sl.op == OP_XORROL;
    __syncthreads();  // every lane has read its operands before the row's result lands in the address space
    if (c == 0 && act && sl.mult > 0) mem[IP_BOUNDARY + r] = rot ? rol64(rv, sl.shift) : rv;
    u64 val = 0;
    if (c >= COL_A && c < COL_A + 8) {
      const u64 ab = (a >> (8 * (c - COL_A))) & 0xff, bb = (b >> (8 * (c - COL_A))) & 0xff;
      val = ab;
      if (act && sl.op != OP_NOP) atomicAdd(&counts[((ab << 8) | bb) * 3 + (sl.op == OP_ANDNOT ? 0 : 1)], 1ull);
    } else if (c >= COL_B && c < COL_B + 8) {
      val = (b >> (8 * (c - COL_B))) & 0xff;
    } else if (c >= COL_R && c < COL_R + 8) {
      val = (rv >> (8 * (c - COL_R))) & 0xff;
    } else if (c >= COL_ROT && c < COL_ROT + 8) {
      if (rot) {  // the limbs of (r_half + 2^32) k for the reduced shift (rol_decompose: the half-swap takes rotations >= 32)
        const u64 k = 1ull << (sl.shift >= 32 ? sl.shift - 32 : sl.shift);
        const int i = c - COL_ROT;
        const u64 half = i < 4 ? (rv & 0xffffffffull) : (rv >> 32);
        val = (((half + (1ull << 32)) * k) >> (16 * (i & 3))) & 0xffff;
        if (act) atomicAdd(&counts[(((val & 0xff) << 8) | (val >> 8)) * 3 + 2], 1ull);  // Range16: w = a + 256 b, table row (a << 8) | b
      }
    } else if (c == COL_ACT) {
      val = act ? 1 : 0;
    }
    if (c >= COL_A && c <= COL_ACT) row[c] = val;
    __syncthreads();
  }
  for (int i = c; i < MEM_WORDS; i += 64) memory_out[(size_t)p * MEM_WORDS + i] = mem[i];
}

// ip = 25 + lane's first permutation * 3200 + row, on every row of both lanes (pads included: the pointer chain is ungated)
__global__ void k_keccak_round_ip(u64* trace, size_t height, int perms_per_lane) {
  const size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (r >= height) return;
  for (int lane = 0; lane < NUM_LANES; lane++)
    trace[r * (LANE_WIDTH * NUM_LANES) + lane * LANE_WIDTH + COL_IP] = IP_BOUNDARY + (u64)lane * perms_per_lane * PERM_CYCLE + r;
}

// buffers of one call: freed again when the call fails half way
struct KtBuffers {
  u64 *mem = nullptr, *tr = nullptr, *cnt = nullptr, *st = nullptr, *rc = nullptr;
  bool keep = false;
  ~KtBuffers() {
    (void)hipFree(st);
    (void)hipFree(rc);
    if (!keep) { (void)hipFree(mem); (void)hipFree(tr); (void)hipFree(cnt); }
  }
};
#define KT_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

extern "C" {
// states: host [n_perms][25]; rcs: host [24]; program: host [128][5] ints (op, shift, back_a, back_b, mult).
// -> *trace_dev: device row-major [2^*log_n][68] (the chiplet's main trace), *counts_dev: device row-major [65536][3] (the table's main
// trace), *memory_dev: device [n_perms][3225] (the machines' address spaces: outputs of permutation p at words 25 + 23 * 128 + 103 ...).
// The caller frees the three buffers with kt_free.  Returns 0 or a hipError_t.
int kt_keccak_round_trace(const uint64_t* states, int n_perms, const uint64_t* rcs, const int* program, uint64_t** trace_dev, int* log_n,
                          uint64_t** counts_dev, uint64_t** memory_dev) {
  if (n_perms < 1) return -1;
  const int ppl = (n_perms + NUM_LANES - 1) / NUM_LANES;
  size_t height = 2;
  int lg = 1;
  while (height < (size_t)ppl * PERM_CYCLE) { height <<= 1; lg++; }
  Program prog;
  for (int i = 0; i < ROUND_PERIOD; i++) prog.s[i] = Slot{program[5 * i], program[5 * i + 1], program[5 * i + 2], program[5 * i + 3], program[5 * i + 4]};
  KtBuffers b;
  KT_CHECK(hipMalloc(&b.mem, (size_t)MEM_WORDS * n_perms * 8));
  KT_CHECK(hipMalloc(&b.tr, height * LANE_WIDTH * NUM_LANES * 8));
  KT_CHECK(hipMalloc(&b.cnt, (size_t)65536 * 3 * 8));
  KT_CHECK(hipMalloc(&b.st, (size_t)25 * n_perms * 8));
  KT_CHECK(hipMalloc(&b.rc, (size_t)NUM_ROUNDS * 8));
  KT_CHECK(hipMemset(b.tr, 0, height * LANE_WIDTH * NUM_LANES * 8));
  KT_CHECK(hipMemset(b.cnt, 0, (size_t)65536 * 3 * 8));
  // the only host data: 25 lanes per permutation and the round constants
  KT_CHECK(hipMemcpy(b.st, states, (size_t)25 * n_perms * 8, hipMemcpyHostToDevice));
  KT_CHECK(hipMemcpy(b.rc, rcs, (size_t)NUM_ROUNDS * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_keccak_round_ip, dim3((unsigned)((height + 255) / 256)), dim3(256), 0, 0, b.tr, height, ppl);
  hipLaunchKernelGGL(k_keccak_round_trace, dim3((unsigned)n_perms), dim3(64), 0, 0, prog, b.st, b.rc, b.mem, n_perms, ppl, b.tr, b.cnt);
  KT_CHECK(hipGetLastError());
  KT_CHECK(hipDeviceSynchronize());
  b.keep = true;
  *trace_dev = (uint64_t*)b.tr; *log_n = lg; *counts_dev = (uint64_t*)b.cnt; *memory_dev = (uint64_t*)b.mem;
  return 0;
}
int kt_download(uint64_t* dst, const uint64_t* src_dev, size_t words) { return (int)hipMemcpy(dst, src_dev, words * 8, hipMemcpyDeviceToHost); }
void kt_free(uint64_t* p) { (void)hipFree(p); }
}
