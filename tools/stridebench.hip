// Strided-tile copy microbenchmark for the second pass of the 2^20 NTT plan: a workgroup of 256 threads
// moves one 2^12-element tile made of 256 segments of 16 consecutive u64 (128 B) whose starts are `stride` elements apart --
// the access shape of pass 1 -- and writes it back the same way.  Question: is the 3.3 TB/s of that pass the price of 128-B
// segments as such, or of the POWER-OF-TWO stride (channel / bank aliasing)?  Usage: stridebench   (prints GB/s per shape)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// tile t of column c: rows h = 0..255 at (c * col_stride + h * row_stride + t * seg) .. + seg;  seg = 16 << seg_shift elements
template <int SEG>  // elements per segment: 16, 32, 64
__global__ __launch_bounds__(256) void k_copy(const u64* __restrict__ src, u64* __restrict__ dst, size_t row_stride, size_t col_stride,
                                             int tiles_per_col, size_t dst_row_stride, size_t dst_col_stride) {
  const int t = blockIdx.x % tiles_per_col, c = blockIdx.x / tiles_per_col;
  const u64* s = src + (size_t)c * col_stride + (size_t)t * SEG;
  u64* d = dst + (size_t)c * dst_col_stride + (size_t)t * SEG;
  const int lane = threadIdx.x % SEG, row0 = threadIdx.x / SEG;
  constexpr int ROWS_PER_IT = 256 / SEG, ITS = 4096 / 256;
  u64 v[ITS];
#pragma unroll
  for (int i = 0; i < ITS; i++) v[i] = s[(size_t)(row0 + i * ROWS_PER_IT) * row_stride + lane];
#pragma unroll
  for (int i = 0; i < ITS; i++) d[(size_t)(row0 + i * ROWS_PER_IT) * dst_row_stride + lane] = v[i] + 1;
}

__global__ __launch_bounds__(256) void k_plain(const u64* __restrict__ src, u64* __restrict__ dst) {
  const size_t base = (size_t)blockIdx.x * 4096 + threadIdx.x;
  u64 v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = src[base + i * 256];
#pragma unroll
  for (int i = 0; i < 16; i++) dst[base + i * 256] = v[i] + 1;
}

__global__ void k_init(u64* a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = i * 0x9e3779b97f4a7c15ull;
}

template <int SEG>
static void run(const char* name, u64* a, u64* b, size_t row_stride, size_t dst_row_stride, int n_cols, size_t col_elems) {
  const int rows = 4096 / SEG;                   // rows per tile
  const int tiles_per_col = (int)((1u << 20) / 4096);  // 256 tiles of 2^12 elements per 2^20 column
  (void)rows;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dim3 grid(tiles_per_col * n_cols);
  if (row_stride == 0) {
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; r++) k_plain<<<grid, 256>>>(a, b);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s %8.3f ms/launch  %7.1f GB/s (read + write)\n", name, ms / 10, 2.0 * 8 * (double)(1u << 20) * n_cols * 10 / ms * 1e-6);
    return;
  }
  for (int w = 0; w < 2; w++) k_copy<SEG><<<grid, 256>>>(a, b, row_stride, col_elems, tiles_per_col, dst_row_stride, col_elems);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int r = 0; r < reps; r++) k_copy<SEG><<<grid, 256>>>(a, b, row_stride, col_elems, tiles_per_col, dst_row_stride, col_elems);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  double bytes = 2.0 * 8 * (double)(1u << 20) * n_cols * reps;
  // spot check: the last element of the last tile of the last column
  {
    const int SEGS_PER_ROW = 4096 / SEG, ROWS = 4096 / SEG, t = tiles_per_col - 1, c = n_cols - 1;
    size_t so = (size_t)c * col_elems + (size_t)(t / SEGS_PER_ROW) * ROWS * row_stride + (size_t)(t % SEGS_PER_ROW) * SEG + (size_t)(ROWS - 1) * row_stride + SEG - 1;
    size_t dof = (size_t)c * col_elems + (size_t)(t / SEGS_PER_ROW) * ROWS * dst_row_stride + (size_t)(t % SEGS_PER_ROW) * SEG + (size_t)(ROWS - 1) * dst_row_stride + SEG - 1;
    u64 x, y;
    CK(hipMemcpy(&x, a + so, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&y, b + dof, 8, hipMemcpyDeviceToHost));
    if (y != x + 1) printf("  CHECK FAILED %llx %llx\n", x, y);
  }
  printf("%-64s %8.3f ms/launch  %7.1f GB/s (read + write)\n", name, ms / reps, bytes / ms * 1e-6);
}

int main() {
  const int n_cols = 408;  // 51 columns x 8 cosets
  // room for the padded variants: rows up to 4096 + 64 elements apart, 256 rows
  const size_t col_elems = (size_t)256 * (4096 + 64) + 4096;
  u64 *a, *b;
  CK(hipMalloc(&a, col_elems * n_cols * 8));
  CK(hipMalloc(&b, col_elems * n_cols * 8));
  k_init<<<4096, 256>>>(a, col_elems * n_cols);
  CK(hipMemset(b, 0, col_elems * n_cols * 8));
  CK(hipDeviceSynchronize());
  // NOTE: with row_stride = 4096 a "row" of 4096 elements holds 256 segments of 16: tile t takes segment t of every row.
  run<16>("seg 128 B, stride 32 KB -> same (pass 1 today)", a, b, 4096, 4096, n_cols, col_elems);
  run<16>("seg 128 B, stride 32 KB + 128 B -> same", a, b, 4096 + 16, 4096 + 16, n_cols, col_elems);
  run<16>("seg 128 B, stride 32 KB + 256 B -> same", a, b, 4096 + 32, 4096 + 32, n_cols, col_elems);
  run<16>("seg 128 B, stride 32 KB + 512 B -> same", a, b, 4096 + 64, 4096 + 64, n_cols, col_elems);
  run<16>("seg 128 B, read stride 32 KB + 256 B, write stride 32 KB", a, b, 4096 + 32, 4096, n_cols, col_elems);
  run<16>("seg 128 B, read stride 32 KB, write stride 32 KB + 256 B", a, b, 4096, 4096 + 32, n_cols, col_elems);
  run<32>("seg 256 B, stride 32 KB (128 rows x 32: half the tiles/col)", a, b, 4096, 4096, n_cols, col_elems);
  run<64>("seg 512 B, stride 32 KB", a, b, 4096, 4096, n_cols, col_elems);
  run<16>("plain contiguous copy of the same bytes (k_plain)", a, b, 0, 0, n_cols, col_elems);
  return 0;
}
