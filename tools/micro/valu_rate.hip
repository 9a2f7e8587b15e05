// Issue rate of VALU instruction kinds on one SIMD (gfx950): cycles per wave64 instruction with 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/micro/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void rate_kernel(float* out, unsigned long long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f + 0.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0) {        // v_add_f32
      asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n"
                   "v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == 1) { // v_exp_f32
      asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                   "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == 2) { // v_rcp_f32
      asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                   "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == 3) { // v_pk_mul_f32 (2 fp32 per lane)
      double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
      asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n"
                   "v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3"
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
      a0 = (float)d0; a1 = (float)d1; a2 = (float)d2; a3 = (float)d3;
    } else if constexpr (KIND == 4) { // v_exp_f16
      asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n"
                   "v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == 5) { // SiLU as the kernels do it: mul, exp, add, rcp, mul on 8 values
      asm volatile(
          "v_mul_f32 %8, 0xbfb8aa3b, %0\n v_mul_f32 %9, 0xbfb8aa3b, %1\n v_mul_f32 %10, 0xbfb8aa3b, %2\n v_mul_f32 %11, 0xbfb8aa3b, %3\n"
          "v_exp_f32 %8, %8\n v_exp_f32 %9, %9\n v_exp_f32 %10, %10\n v_exp_f32 %11, %11\n"
          "v_add_f32 %8, 1.0, %8\n v_add_f32 %9, 1.0, %9\n v_add_f32 %10, 1.0, %10\n v_add_f32 %11, 1.0, %11\n"
          "v_rcp_f32 %8, %8\n v_rcp_f32 %9, %9\n v_rcp_f32 %10, %10\n v_rcp_f32 %11, %11\n"
          "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %10\n v_mul_f32 %3, %3, %11"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_iter) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 1024 * sizeof(unsigned long long));
  const int iters = 4096;
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 64 * 4 * waves_per_simd;      // one workgroup per CU, 4 SIMDs
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wall-clock cycles per instruction per SIMD at 2.4 GHz (the shader clock may be lower: compare kinds, not absolutes)
    const double inst_per_simd = (double)iters * per_iter * waves_per_simd;
    printf("%-28s %d wave(s)/SIMD: %.2f ns per wave-instruction per SIMD  (%.1f cycles at 2.4 GHz)\n", name, waves_per_simd,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("v_add_f32", 8);
  run<1>("v_exp_f32", 8);
  run<2>("v_rcp_f32", 8);
  run<3>("v_pk_mul_f32", 8);
  run<4>("v_exp_f16", 8);
  run<5>("SiLU x4 (20 instr)", 20);
  return 0;
}
