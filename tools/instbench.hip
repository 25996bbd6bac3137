// Per-instruction VALU issue cost on gfx950 (wave64), used to design gl.cuh / poseidon2.cuh.
// Each kernel issues 8 independent copies x 8 of one instruction per loop iteration from
// 2048 blocks x 256 threads (8 waves per SIMD) and reports cycles per wave-instruction per SIMD
// assuming 1024 SIMDs at the clock measured by the v_add_u32 row (nominal 2 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
typedef uint64_t u64;

#define REP8(X) X X X X X X X X
#define KERNEL(name, BODY)                                                        \
  __global__ __launch_bounds__(256) void name(u64* out, int iters) {             \
    u32 a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 9;           \
    u32 b0 = a0 + 11, b1 = a1 + 13, b2 = a2 + 17, b3 = a3 + 19;                   \
    u64 q0 = a0 * 0x10001ull + 5, q1 = q0 * 3, q2 = q0 ^ 0x1234567, q3 = q0 + 99; \
    u64 r0 = q0 + 1, r1 = q1 + 2, r2 = q2 + 3, r3 = q3 + 4;                       \
    _Pragma("unroll 1") for (int i = 0; i < iters; i++) { REP8(BODY) }            \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 + q0 + q1 + q2 + q3 + r0 + r1 + r2 + r3; \
  }

// 8 instructions per BODY
KERNEL(k_add_u32,
  asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7\n"
               "v_add_u32 %1, %1, %0\n v_add_u32 %3, %3, %2\n v_add_u32 %5, %5, %4\n v_add_u32 %7, %7, %6\n"
               : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));)
KERNEL(k_add3_u32,
  asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %2, %2, %3, %4\n v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %6, %6, %7, %0\n"
               "v_add3_u32 %1, %1, %0, %3\n v_add3_u32 %3, %3, %2, %5\n v_add3_u32 %5, %5, %4, %7\n v_add3_u32 %7, %7, %6, %1\n"
               : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));)
KERNEL(k_addco_addc,  // 4 pairs of (v_add_co_u32, v_addc_co_u32) through vcc
  asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc\n v_add_co_u32 %4, vcc, %4, %5\n v_addc_co_u32 %6, vcc, %6, %7, vcc\n"
               "v_add_co_u32 %1, vcc, %1, %0\n v_addc_co_u32 %3, vcc, %3, %2, vcc\n v_add_co_u32 %5, vcc, %5, %4\n v_addc_co_u32 %7, vcc, %7, %6, vcc\n"
               : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3) :: "vcc");)
KERNEL(k_addco_only,
  asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_add_co_u32 %2, vcc, %2, %3\n v_add_co_u32 %4, vcc, %4, %5\n v_add_co_u32 %6, vcc, %6, %7\n"
               "v_add_co_u32 %1, vcc, %1, %0\n v_add_co_u32 %3, vcc, %3, %2\n v_add_co_u32 %5, vcc, %5, %4\n v_add_co_u32 %7, vcc, %7, %6\n"
               : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3) :: "vcc");)
KERNEL(k_lshl_add_u64,
  asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %6, %6, 0, %7\n"
               "v_lshl_add_u64 %1, %1, 0, %0\n v_lshl_add_u64 %3, %3, 0, %2\n v_lshl_add_u64 %5, %5, 0, %4\n v_lshl_add_u64 %7, %7, 0, %6\n"
               : "+v"(q0), "+v"(r0), "+v"(q1), "+v"(r1), "+v"(q2), "+v"(r2), "+v"(q3), "+v"(r3));)
KERNEL(k_cmp_u64_cndmask,  // 4 x (v_cmp_lt_u64 + v_cndmask)
  asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %8, %8, %9, vcc\n v_cmp_lt_u64 vcc, %2, %3\n v_cndmask_b32 %10, %10, %11, vcc\n"
               "v_cmp_lt_u64 vcc, %4, %5\n v_cndmask_b32 %9, %9, %8, vcc\n v_cmp_lt_u64 vcc, %6, %7\n v_cndmask_b32 %11, %11, %10, vcc\n"
               : "+v"(q0), "+v"(r0), "+v"(q1), "+v"(r1), "+v"(q2), "+v"(r2), "+v"(q3), "+v"(r3), "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1) :: "vcc");)
KERNEL(k_cmp_u64,
  asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %2, %3\n v_cmp_lt_u64 vcc, %4, %5\n v_cmp_lt_u64 vcc, %6, %7\n"
               "v_cmp_lt_u64 vcc, %1, %0\n v_cmp_lt_u64 vcc, %3, %2\n v_cmp_lt_u64 vcc, %5, %4\n v_cmp_lt_u64 vcc, %7, %6\n"
               : "+v"(q0), "+v"(r0), "+v"(q1), "+v"(r1), "+v"(q2), "+v"(r2), "+v"(q3), "+v"(r3) :: "vcc");)
KERNEL(k_cmp_u32,
  asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %4, %5\n v_cmp_lt_u32 vcc, %6, %7\n"
               "v_cmp_lt_u32 vcc, %1, %0\n v_cmp_lt_u32 vcc, %3, %2\n v_cmp_lt_u32 vcc, %5, %4\n v_cmp_lt_u32 vcc, %7, %6\n"
               : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3) :: "vcc");)
KERNEL(k_cndmask,
  asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n"
               "v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %3, %3, %2, vcc\n v_cndmask_b32 %5, %5, %4, vcc\n v_cndmask_b32 %7, %7, %6, vcc\n"
               : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3) :: "vcc");)
KERNEL(k_mad_u64_u32,
  asm volatile("v_mad_u64_u32 %0, s[4:5], %8, %9, %0\n v_mad_u64_u32 %1, s[4:5], %9, %10, %1\n v_mad_u64_u32 %2, s[4:5], %10, %11, %2\n v_mad_u64_u32 %3, s[4:5], %11, %8, %3\n"
               "v_mad_u64_u32 %4, s[4:5], %8, %10, %4\n v_mad_u64_u32 %5, s[4:5], %9, %11, %5\n v_mad_u64_u32 %6, s[4:5], %8, %8, %6\n v_mad_u64_u32 %7, s[4:5], %9, %9, %7\n"
               : "+v"(q0), "+v"(r0), "+v"(q1), "+v"(r1), "+v"(q2), "+v"(r2), "+v"(q3), "+v"(r3) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "s4", "s5");)
KERNEL(k_mul_lo_u32,
  asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %6, %6, %7\n"
               "v_mul_lo_u32 %1, %1, %0\n v_mul_lo_u32 %3, %3, %2\n v_mul_lo_u32 %5, %5, %4\n v_mul_lo_u32 %7, %7, %6\n"
               : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));)
KERNEL(k_mul_hi_u32,
  asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %4, %4, %5\n v_mul_hi_u32 %6, %6, %7\n"
               "v_mul_hi_u32 %1, %1, %0\n v_mul_hi_u32 %3, %3, %2\n v_mul_hi_u32 %5, %5, %4\n v_mul_hi_u32 %7, %7, %6\n"
               : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));)
KERNEL(k_mad_u32_u24,
  asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %2, %2, %3, %4\n v_mad_u32_u24 %4, %4, %5, %6\n v_mad_u32_u24 %6, %6, %7, %0\n"
               "v_mad_u32_u24 %1, %1, %0, %3\n v_mad_u32_u24 %3, %3, %2, %5\n v_mad_u32_u24 %5, %5, %4, %7\n v_mad_u32_u24 %7, %7, %6, %1\n"
               : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));)
KERNEL(k_alignbit,
  asm volatile("v_alignbit_b32 %0, %0, %1, 1\n v_alignbit_b32 %2, %2, %3, 1\n v_alignbit_b32 %4, %4, %5, 1\n v_alignbit_b32 %6, %6, %7, 1\n"
               "v_alignbit_b32 %1, %1, %0, 1\n v_alignbit_b32 %3, %3, %2, 1\n v_alignbit_b32 %5, %5, %4, 1\n v_alignbit_b32 %7, %7, %6, 1\n"
               : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3));)
KERNEL(k_fma_f64,
  asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %4, %4, %5, %6\n v_fma_f64 %6, %6, %7, %0\n"
               "v_fma_f64 %1, %1, %0, %3\n v_fma_f64 %3, %3, %2, %5\n v_fma_f64 %5, %5, %4, %7\n v_fma_f64 %7, %7, %6, %1\n"
               : "+v"(q0), "+v"(r0), "+v"(q1), "+v"(r1), "+v"(q2), "+v"(r2), "+v"(q3), "+v"(r3));)
KERNEL(k_subb_sgprcarry,  // carry chain through explicit SGPR pairs (as the compiler emits)
  asm volatile("v_sub_co_u32 %0, s[4:5], %0, %1\n v_subb_co_u32 %2, s[4:5], %2, %3, s[4:5]\n v_sub_co_u32 %4, s[6:7], %4, %5\n v_subb_co_u32 %6, s[6:7], %6, %7, s[6:7]\n"
               "v_sub_co_u32 %1, s[4:5], %1, %0\n v_subb_co_u32 %3, s[4:5], %3, %2, s[4:5]\n v_sub_co_u32 %5, s[6:7], %5, %4\n v_subb_co_u32 %7, s[6:7], %7, %6, s[6:7]\n"
               : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3) :: "s4", "s5", "s6", "s7");)

template <class K>
static double run(K k, const char* name, u64* out, double base_cycles_per_inst, double* ms_out) {
  const int blocks = 2048, iters = 4000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  *ms_out = ms;
  double wave_insts = (double)blocks * 4 * iters * 64;  // 4 waves per block, 64 instructions/iter
  double ns_per_inst_per_simd = ms * 1e6 / (wave_insts / 1024.0);
  printf("%-22s %8.3f ms  %6.3f ns per wave-instruction per SIMD", name, ms, ns_per_inst_per_simd);
  if (base_cycles_per_inst > 0) printf("  = %5.2f cycles (v_add_u32 := 2)", ns_per_inst_per_simd / base_cycles_per_inst);
  printf("\n");
  return ns_per_inst_per_simd;
}

int main() {
  u64* out;
  hipMalloc(&out, 2048 * 256 * 8);
  double ms;
  double base = run(k_add_u32, "v_add_u32", out, 0, &ms) / 2.0;  // ns per cycle
  printf("implied clock %.2f GHz\n", 1.0 / base);
#define R(k, n) run(k, n, out, base, &ms)
  R(k_add_u32, "v_add_u32"); R(k_add3_u32, "v_add3_u32"); R(k_addco_only, "v_add_co_u32"); R(k_addco_addc, "add_co+addc_co (vcc)");
  R(k_subb_sgprcarry, "sub_co+subb_co (sgpr)"); R(k_lshl_add_u64, "v_lshl_add_u64"); R(k_cmp_u32, "v_cmp_lt_u32");
  R(k_cmp_u64, "v_cmp_lt_u64"); R(k_cmp_u64_cndmask, "cmp_u64+cndmask"); R(k_cndmask, "v_cndmask_b32");
  R(k_mad_u64_u32, "v_mad_u64_u32"); R(k_mul_lo_u32, "v_mul_lo_u32"); R(k_mul_hi_u32, "v_mul_hi_u32"); R(k_mad_u32_u24, "v_mad_u32_u24");
  R(k_alignbit, "v_alignbit_b32"); R(k_fma_f64, "v_fma_f64");
  return 0;
}
