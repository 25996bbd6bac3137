// H2D of a row-major 2^20 x 51 trace (428 MB) from page-locked memory, by column groups -- which mechanism moves a
// [rows][8..27 columns] window fastest, and does it run under a VALU-bound kernel?
//   (a) one bulk hipMemcpyAsync of everything                         (the round-2 path)
//   (b) hipMemcpy2DAsync of a column window (width = 8*gw bytes, pitch = 8*w) into a packed device buffer
//   (c) a kernel that reads the host buffer directly (zero-copy) and writes the window column-major
// each alone and next to a long register-only integer kernel on another stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;

__global__ void k_spin(u64* out, int iters) {  // VALU-bound filler: 64-bit multiply-adds
  u64 a = threadIdx.x + 1, b = blockIdx.x + 3;
  for (int i = 0; i < iters; i++) { a = a * b + 0x9E3779B97F4A7C15ull; b = b * a + 1; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b;
}
// window [c0, c0+gw) of a row-major [n][w] host matrix -> column-major [gw][n] device matrix
__global__ __launch_bounds__(256) void k_pull(const u64* __restrict__ host, u64* __restrict__ out, size_t n, int w, int c0, int gw) {
  __shared__ u64 tile[32][33];
  // a workgroup handles 32 rows x 32 columns of the window per step; grid-stride over row blocks
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (size_t rb = blockIdx.x; rb < n / 32; rb += gridDim.x) {
    for (int cb = 0; cb < gw; cb += 32) {
      for (int k = ty; k < 32; k += 8) {
        const int c = cb + tx;
        if (c < gw) tile[k][tx] = __builtin_nontemporal_load(host + (rb * 32 + k) * (size_t)w + c0 + c);
      }
      __syncthreads();
      for (int k = ty; k < 32; k += 8) {
        const int c = cb + k;
        if (c < gw) out[(size_t)c * n + rb * 32 + tx] = tile[tx][k];
      }
      __syncthreads();
    }
  }
}
static double ms_between(hipEvent_t a, hipEvent_t b) { float m; CK(hipEventElapsedTime(&m, a, b)); return m; }

int main() {
  const size_t n = 1 << 20; const int w = 51;
  u64 *host, *dev, *dev2, *spin;
  CK(hipHostMalloc((void**)&host, n * w * 8, hipHostMallocDefault));
  for (size_t i = 0; i < n * w; i++) host[i] = i * 0x9E3779B97F4A7C15ull;
  CK(hipMalloc((void**)&dev, n * w * 8)); CK(hipMalloc((void**)&dev2, n * w * 8)); CK(hipMalloc((void**)&spin, 4096 * 256 * 8));
  hipStream_t s_copy, s_comp; CK(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_comp, hipStreamNonBlocking));
  hipEvent_t e0, e1, c0e, c1e; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&c0e)); CK(hipEventCreate(&c1e));
  // calibrate the filler: ~20 ms
  int iters = 20000;
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(c0e, s_comp)); hipLaunchKernelGGL(k_spin, dim3(4096), dim3(256), 0, s_comp, spin, iters); CK(hipEventRecord(c1e, s_comp));
    CK(hipStreamSynchronize(s_comp));
    double m = ms_between(c0e, c1e); if (rep == 0) iters = (int)(iters * 20.0 / m);
    else printf("filler kernel alone: %.2f ms\n", m);
  }
  auto run = [&](const char* name, double bytes, auto&& enqueue) {
    for (int with = 0; with < 2; with++) {
      double best = 1e9, comp = 0;
      for (int rep = 0; rep < 3; rep++) {
        if (with) { CK(hipEventRecord(c0e, s_comp)); hipLaunchKernelGGL(k_spin, dim3(4096), dim3(256), 0, s_comp, spin, iters); CK(hipEventRecord(c1e, s_comp)); }
        CK(hipEventRecord(e0, s_copy)); enqueue(); CK(hipEventRecord(e1, s_copy));
        CK(hipStreamSynchronize(s_copy)); CK(hipStreamSynchronize(s_comp));
        double m = ms_between(e0, e1); if (m < best) best = m;
        if (with) comp = ms_between(c0e, c1e);
      }
      printf("%-44s %s  %7.2f ms  %6.1f GB/s", name, with ? "under filler" : "alone       ", best, bytes / best / 1e6);
      if (with) printf("   (filler %.2f ms)", comp);
      printf("\n");
    }
  };
  run("(a) bulk hipMemcpyAsync 428 MB", n * w * 8.0, [&] { CK(hipMemcpyAsync(dev, host, n * w * 8, hipMemcpyHostToDevice, s_copy)); });
  for (int gw : {8, 16, 27, 51}) {
    char nm[96];
    snprintf(nm, sizeof nm, "(b) hipMemcpy2DAsync window of %d columns", gw);
    run(nm, n * gw * 8.0, [&] { CK(hipMemcpy2DAsync(dev, gw * 8, host, w * 8, gw * 8, n, hipMemcpyHostToDevice, s_copy)); });
    for (int blocks : {64, 256, 1024}) {
      snprintf(nm, sizeof nm, "(c) zero-copy pull kernel, %d cols, %d WGs", gw, blocks);
      run(nm, n * gw * 8.0, [&] { hipLaunchKernelGGL(k_pull, dim3(blocks), dim3(256), 0, s_copy, host, dev2, n, w, 0, gw); });
    }
  }
  // row chunks: contiguous 1/8 of the rows at a time (what a row-chunked pipeline would issue)
  run("(d) 8 x hipMemcpyAsync of 1/8 of the rows", n * w * 8.0, [&] { for (int k = 0; k < 8; k++) CK(hipMemcpyAsync(dev + k * (n / 8) * w, host + k * (n / 8) * w, n / 8 * w * 8, hipMemcpyHostToDevice, s_copy)); });
  // correctness of the pull kernel
  hipLaunchKernelGGL(k_pull, dim3(256), dim3(256), 0, s_copy, host, dev2, n, w, 8, 16);
  CK(hipStreamSynchronize(s_copy));
  std::vector<u64> chk(16 * n);
  CK(hipMemcpy(chk.data(), dev2, 16 * n * 8, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (int c = 0; c < 16; c++) for (size_t r = 0; r < n; r += 4097) if (chk[c * n + r] != host[r * w + 8 + c]) bad++;
  printf("pull kernel check: %zu mismatches\n", bad);
  return bad != 0;
}
