// Microbenchmarks that set the integer-ALU roofline used in DESIGN.md: throughput of
// v_mad_u64_u32, of the Goldilocks mul/add device functions and of the register-resident
// Poseidon2 permutation on gfx950, plus a plain HBM copy.  Build: make -C tools
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../miden-vm_amd/csrc/poseidon2_fast.cuh"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_mad(u64* out, int iters, u64 seed) {
  u64 a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = a ^ b, d = a + 77;
  u32 x = (u32)seed | 1, y = (u32)(seed >> 7) | 1;
#pragma unroll 1
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      a = (u64)(u32)a * x + b;
      b = (u64)(u32)b * y + c;
      c = (u64)(u32)c * x + d;
      d = (u64)(u32)d * y + a;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d;
}
__global__ __launch_bounds__(256) void k_glmul(u64* out, int iters, u64 seed) {
  u64 a = gl_canon(seed + threadIdx.x), b = gl_canon(seed * 3 + threadIdx.x), c = gl_canon(a ^ b), d = gl_canon(a + 77);
#pragma unroll 1
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      a = gl_mul(a, b); b = gl_mul(b, c); c = gl_mul(c, d); d = gl_mul(d, a);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d;
}
__global__ __launch_bounds__(256) void k_gladd(u64* out, int iters, u64 seed) {
  u64 a = gl_canon(seed + threadIdx.x), b = gl_canon(seed * 3 + threadIdx.x), c = gl_canon(a ^ b), d = gl_canon(a + 77);
#pragma unroll 1
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      a = gl_add(a, b); b = gl_sub(b, c); c = gl_add(c, d); d = gl_sub(d, a);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d;
}
__global__ __launch_bounds__(256) void k_perm(u64* out, int iters, u64 seed) {
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_canon(seed * (i + 1) + threadIdx.x + blockIdx.x * 131);
#pragma unroll 1
  for (int i = 0; i < iters; i++) p2_permute(s);
  u64 x = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) x ^= s[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_permf(u64* out, int iters, u64 seed) {
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
__global__ __launch_bounds__(256) void k_copy(const ulonglong2* __restrict__ in, ulonglong2* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

template <class F>
static float time_ms(F f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; i++) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const int blocks = 256 * 8, iters = 2000;
  u64* out;
  CK(hipMalloc(&out, blocks * 256 * 8));
  double lanes = (double)blocks * 256;
  float ms;
  ms = time_ms([&] { hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(256), 0, 0, out, iters, 12345ULL); }, 3);
  printf("v_mad_u64_u32      : %8.3f ms  %.3f Tops/s (lane-ops)\n", ms, lanes * iters * 32 / ms / 1e9);
  ms = time_ms([&] { hipLaunchKernelGGL(k_glmul, dim3(blocks), dim3(256), 0, 0, out, iters, 12345ULL); }, 3);
  printf("gl_mul             : %8.3f ms  %.3f Tmul/s\n", ms, lanes * iters * 32 / ms / 1e9);
  ms = time_ms([&] { hipLaunchKernelGGL(k_gladd, dim3(blocks), dim3(256), 0, 0, out, iters, 12345ULL); }, 3);
  printf("gl_add/sub         : %8.3f ms  %.3f Tadd/s\n", ms, lanes * iters * 32 / ms / 1e9);
  const int piters = 64;
  ms = time_ms([&] { hipLaunchKernelGGL(k_perm, dim3(blocks), dim3(256), 0, 0, out, piters, 12345ULL); }, 3);
  printf("poseidon2 permute  : %8.3f ms  %.3f Gperm/s\n", ms, lanes * piters / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL(k_permf, dim3(blocks), dim3(256), 0, 0, out, piters, 12345ULL); }, 3);
  printf("poseidon2 fast     : %8.3f ms  %.3f Gperm/s\n", ms, lanes * piters / ms / 1e6);
  {
    // same seeds through both permutations must agree (sanity; the parity tests are the real check)
    std::vector<u64> ha(blocks * 256), hb(blocks * 256);
    hipLaunchKernelGGL(k_perm, dim3(blocks), dim3(256), 0, 0, out, 3, 777ULL);
    hipMemcpy(ha.data(), out, ha.size() * 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_permf, dim3(blocks), dim3(256), 0, 0, out, 3, 777ULL);
    hipMemcpy(hb.data(), out, hb.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < ha.size(); i++) bad += ha[i] != hb[i];
    printf("fast vs reference permutation mismatches: %zu of %zu\n", bad, ha.size());
  }
  size_t n = (size_t)1 << 27;  // 2 GiB each way
  ulonglong2 *a, *b;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
  CK(hipMemset(a, 1, n * 16));
  ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, a, b, n); }, 5);
  printf("copy 2+2 GiB       : %8.3f ms  %.1f GB/s (read+write)\n", ms, 2.0 * n * 16 / ms / 1e6);
  return 0;
}
