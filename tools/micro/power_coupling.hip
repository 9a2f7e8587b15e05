// Do the matrix pipes and the HBM stream of an MI355X share a budget?  A forward costs about its MFMA time PLUS its HBM time
// (DESIGN.md section 9).  Two readings: (a) structural - a workgroup loads, multiplies and stores in turn and nothing else fits on its
// CU - or (b) physical - the chip is power-limited, so HBM traffic next to MFMA work lowers the clock the MFMAs run at.  This probe runs
// a register-only MFMA burner (bf16 16x16x32, non-trivial operands, no memory) and a float4 copy kernel, each ALONE and BOTH AT ONCE
// (two streams; both kernels are small enough to be co-resident on every CU), and reports TFLOP/s, TB/s and the shader clock read inside
// the burner (s_memtime ticks per s_memrealtime tick).  If (b), the burner loses rate / clock as soon as the copy runs beside it.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/power_coupling tools/micro/power_coupling.hip && tools/micro/power_coupling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

// 8 independent accumulator chains per wave; operands differ per lane and change every iteration (a constant operand draws less power)
__global__ void __launch_bounds__(256) mfma_burn(float* out, unsigned long long* clk, int iters, unsigned seed) {
  const unsigned t = threadIdx.x + blockIdx.x * 256u;
  u32x4_t a = {0x3f803f80u ^ (t * 2654435761u & 0x007f007fu), 0x3fa03f90u ^ (t & 0x003f003fu), 0xbf803f80u ^ (seed & 0x007f007fu), 0x3f00bf00u ^ (t >> 3 & 0x007f007fu)};
  u32x4_t b = {0x3f903f70u ^ (t * 40503u & 0x007f007fu), 0xbf603f50u, 0x3f403f30u ^ (seed >> 7 & 0x007f007fu), 0x3f20bf10u};
  f32x4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  unsigned long long t0 = __builtin_readcyclecounter();          // s_memtime: shader-clock ticks
  unsigned long long w0 = wall_clock64();                        // s_memrealtime: constant-rate ticks
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[i], 0, 0, 0);
    a.x = a.x * 1664525u + 1013904223u; a.x = (a.x & 0x807f807fu) | 0x3f003f00u;      // new mantissas / signs, exponents kept near 1
    b.y = b.y * 22695477u + 1u; b.y = (b.y & 0x807f807fu) | 0x3f003f00u;
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned long long w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[t] = s;
  if (threadIdx.x == 0 && clk != nullptr) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

__global__ void __launch_bounds__(256) stream_copy(const float4* __restrict__ src, float4* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  int wall_khz = 0; (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  printf("%s, %d CUs, wall clock %d kHz\n", p.name, ncu, wall_khz);
  const long nbytes = 4L << 30;
  float4 *src, *dst; (void)hipMalloc(&src, nbytes); (void)hipMalloc(&dst, nbytes);
  (void)hipMemset(src, 1, nbytes); (void)hipMemset(dst, 0, nbytes);
  const int burn_blocks = ncu * 4;                     // 16 waves per CU: 4 per SIMD
  float* out; (void)hipMalloc(&out, (size_t)burn_blocks * 256 * 4);
  unsigned long long* clk; (void)hipMalloc(&clk, (size_t)burn_blocks * 16);
  std::vector<unsigned long long> h(burn_blocks * 2);
  hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
  hipEvent_t a0, a1, b0, b1; (void)hipEventCreate(&a0); (void)hipEventCreate(&a1); (void)hipEventCreate(&b0); (void)hipEventCreate(&b1);
  const int iters = 200000;                            // ~45 ms of MFMA work per wave
  const double burn_flop = (double)burn_blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 32;
  const int copy_reps = 12;
  auto run = [&](bool burn, bool copy, const char* tag) {
    (void)hipDeviceSynchronize();
    if (burn) { (void)hipEventRecord(a0, s1); hipLaunchKernelGGL(mfma_burn, dim3(burn_blocks), dim3(256), 0, s1, out, clk, iters, 12345u); (void)hipEventRecord(a1, s1); }
    if (copy) {
      (void)hipEventRecord(b0, s2);
      for (int r = 0; r < copy_reps; ++r) hipLaunchKernelGGL(stream_copy, dim3(ncu * 8), dim3(256), 0, s2, src, dst, nbytes / 16);
      (void)hipEventRecord(b1, s2);
    }
    (void)hipDeviceSynchronize();
    float mb = 0, mc = 0;
    if (burn) (void)hipEventElapsedTime(&mb, a0, a1);
    if (copy) (void)hipEventElapsedTime(&mc, b0, b1);
    double mhz = 0;
    if (burn) {
      (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
      double r = 0; for (int i = 0; i < burn_blocks; ++i) r += (double)h[2 * i] / (double)h[2 * i + 1];
      mhz = r / burn_blocks * wall_khz / 1e3;
    }
    printf("%-34s", tag);
    if (burn) printf("  MFMA %7.1f TFLOP/s (%.1f ms, shader clock %.0f MHz)", burn_flop / (mb * 1e-3) / 1e12, mb, mhz);
    if (copy) printf("  copy %5.2f TB/s read+write (%.1f ms)", 2.0 * nbytes * copy_reps / (mc * 1e-3) / 1e12, mc);
    printf("\n");
  };
  run(true, false, "warm-up");
  run(true, false, "MFMA burner alone");
  run(false, true, "copy alone");
  run(true, true, "both at once");
  run(true, false, "MFMA burner alone (again)");
  run(false, true, "copy alone (again)");
  run(true, true, "both at once (again)");
  return 0;
}
