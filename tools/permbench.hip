// Register-only Poseidon2 permutation rate for one build variant of poseidon2_fast.cuh (-DP2F_ASM=0/1, -DP2F_GROUP=n),
// checked against the plain permutation (poseidon2.cuh) on the same inputs, with the shader clock measured by
// s_memtime so that rates convert to cycles.  Build: make -C tools   (permbench_c, permbench_a3, _a4, _a6)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../miden-vm_amd/csrc/poseidon2_fast.cuh"

template <bool FAST>
__global__ __launch_bounds__(256) void k_perm(u64* out, int iters, u64 seed, u64* cyc) {
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = seed * (i + 1) + threadIdx.x * 0x9E3779B97F4A7C15ULL + blockIdx.x * 131;  // any u64
  const u64 t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < iters; i++) {
    if (FAST) p2f_permute(s);
    else {
#pragma unroll
      for (int k = 0; k < 12; k++) s[k] = gl_canon(s[k]);
      p2_permute(s);
    }
  }
  const u64 t1 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 12; i++) out[(size_t)i * gridDim.x * 256 + blockIdx.x * 256 + threadIdx.x] = gl_canon(s[i]);
  if (cyc && threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// chained multiplications: 12 independent chains per lane (the S-box layer's shape)
__global__ __launch_bounds__(256) void k_mulchain(u64* out, int iters, u64 seed) {
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = seed * (i + 1) + threadIdx.x * 0x9E3779B97F4A7C15ULL + blockIdx.x * 131;
#pragma unroll 1
  for (int i = 0; i < iters; i++) p2f_sbox12(s);
  u64 x = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) x ^= s[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_mulserial(u64* out, int iters, u64 seed) {
  u64 x = seed + threadIdx.x * 0x9E3779B97F4A7C15ULL + blockIdx.x * 131;
#pragma unroll 1
  for (int i = 0; i < iters; i++) x = p2f_sbox(x) + 1;
  out[blockIdx.x * 256 + threadIdx.x] = x;
}

int main() {
  const int blocks = 2048, iters = 64;
  u64 *out, *cyc;
  hipMalloc(&out, (size_t)12 * blocks * 256 * 8);
  hipMalloc(&cyc, blocks * 8);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float ms;
  printf("variant: P2F_ASM=%d P2F_GROUP=%d\n", P2F_ASM, P2F_GROUP);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k_perm<true>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345ULL + rep, cyc);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    std::vector<u64> hc(blocks);
    hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (u64 v : hc) avg += (double)v;
    avg /= blocks;
    printf("permute: %8.3f ms  %.3f Gperm/s   s_memtime ticks per block %.0f (= %.0f per permutation per wave)\n", ms,
           (double)blocks * 256 * iters / ms / 1e6, avg, avg / iters);
  }
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k_mulchain, dim3(blocks), dim3(256), 0, 0, out, 2000, 777ULL);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("S-box layer (12 x^7): %8.3f ms  %.3f Tmul/s  %.3f ns per mul per wave per SIMD\n", ms, (double)blocks * 256 * 2000 * 48 / ms / 1e9,
           ms * 1e6 / ((double)blocks * 4 * 2000 * 48 / 1024));
  }
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k_mulserial, dim3(blocks), dim3(256), 0, 0, out, 20000, 777ULL);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("serial x^7 chain:     %8.3f ms  %.3f Tmul/s  %.3f ns per mul per wave per SIMD\n", ms, (double)blocks * 256 * 20000 * 4 / ms / 1e9,
           ms * 1e6 / ((double)blocks * 4 * 20000 * 4 / 1024));
  }
  {  // parity with the plain permutation on arbitrary (non-canonical) 64-bit inputs
    const size_t n = (size_t)12 * blocks * 256;
    std::vector<u64> ha(n), hb(n);
    hipLaunchKernelGGL(k_perm<false>, dim3(blocks), dim3(256), 0, 0, out, 3, 0xFFFFFFFF00000000ULL, nullptr);
    hipMemcpy(ha.data(), out, n * 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_perm<true>, dim3(blocks), dim3(256), 0, 0, out, 3, 0xFFFFFFFF00000000ULL, nullptr);
    hipMemcpy(hb.data(), out, n * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) bad += ha[i] != hb[i];
    printf("fast vs plain permutation mismatches: %zu of %zu\n", bad, n);
  }
  return 0;
}
