// Lone-wave latency of one Poseidon2 permutation in its three forms (what a small tree level costs): one state per lane
// (poseidon2_fast.cuh), one element per lane in 16-lane groups (poseidon2_lanes.cuh), in s_memtime ticks (100 MHz) and
// shader cycles; first (cold constants / instruction cache) and steady-state iterations apart.  Build: make -C tools lanebench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../miden-vm_amd/csrc/poseidon2_lanes.cuh"
#ifdef HAVE_QUAD
#include "../miden-vm_amd/csrc/poseidon2_quad.cuh"
#endif

template <int MODE>
__global__ __launch_bounds__(64) void k_lat(u64* out, int iters, u64 seed, u64* ticks) {
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_canon(seed * (i + 1) + threadIdx.x * 0x9E3779B97F4A7C15ULL);
  u64 t[3];
  t[0] = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < 2; rep++) {
#pragma unroll 1
    for (int i = 0; i < (rep ? iters : 1); i++) {
      if (MODE == 0) p2f_permute(s);
      if (MODE == 1) s[0] = p2l_permute(s[0]);
#ifdef HAVE_QUAD
      if (MODE == 2) p2q_permute(s);
#endif
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    t[rep + 1] = __builtin_amdgcn_s_memtime();
  }
#pragma unroll
  for (int i = 0; i < 12; i++) out[i * 64 + threadIdx.x] = s[i];
  if (threadIdx.x == 0) {
    ticks[0] = t[1] - t[0];
    ticks[1] = t[2] - t[1];
  }
}

int main() {
  u64 *out, *ticks;
  hipMalloc(&out, 12 * 64 * 8);
  hipMalloc(&ticks, 16);
  const int iters = 32;
  const char* names[3] = {"one state per lane (p2f)", "16 lanes per state (p2l)", "4 lanes per state (p2q)"};
  for (int mode = 0; mode < 3; mode++) {
#ifndef HAVE_QUAD
    if (mode == 2) break;
#endif
    for (int rep = 0; rep < 2; rep++) {
      hipEvent_t a, b;
      hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(k_lat<0>, dim3(1), dim3(64), 0, 0, out, iters, 99ULL, ticks);
      if (mode == 1) hipLaunchKernelGGL(k_lat<1>, dim3(1), dim3(64), 0, 0, out, iters, 99ULL, ticks);
      if (mode == 2) hipLaunchKernelGGL(k_lat<2>, dim3(1), dim3(64), 0, 0, out, iters, 99ULL, ticks);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      u64 h[2];
      hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
      printf("%-28s launch %d: first permutation %6.2f us, steady %6.2f us each (%d chained), kernel %7.1f us\n", names[mode], rep,
             h[0] / 100.0, h[1] / 100.0 / iters, iters, ms * 1e3);
    }
  }
  return 0;
}
