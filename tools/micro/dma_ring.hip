// L2/HBM -> LDS request stream of one CU (global_load_lds_dwordx4), every CU of the chip streaming at once: bytes per second against
// the BYTES IN FLIGHT (ring depth x stage size) and the ROW WIDTH of the staged image (128-byte rows = full cache lines, 64-byte rows =
// half lines, what a K step of 32 bf16 stages).  Answers (round 4): is the ~54 GB/s per CU that the round-2 loads-only loop measured a
// latency x bytes-in-flight product (then a deeper ring lifts it) or a request-rate limit of the texture-addresser path (then it does not)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/dma_ring tools/micro/dma_ring.hip && tools/micro/dma_ring
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// ROWB: bytes per staged row; STAGE_KIB: bytes of one stage; DEPTH: stages kept in flight; SRC: 0 = a 256-row weight-like matrix shared by
// every workgroup (row stride 4608 B: L2 hits), 1 = activation-like rows private to the workgroup, streamed from HBM (row stride 512 B, every
// byte read once), 2 = half the loads of each kind (what a GEMM K step stages); BAR: raw s_barrier per stage (couples the 8 waves).
template <int ROWB, int STAGE_KIB, int DEPTH, int SRC, bool BAR>
__global__ void __launch_bounds__(512) dma_kernel(const unsigned char* w, const unsigned char* x, int nsteps, long x_rows_per_wg, int* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = STAGE_KIB * 1024;
  constexpr int LPS = STAGE / 8192;               // 16-byte loads per thread per stage
  constexpr int NSLOT = DEPTH + 1;
  constexpr int GPR = ROWB / 16;                  // granules per row
  constexpr int RPP = 512 / GPR;                  // rows per pass of the workgroup
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = tid / GPR, g = tid % GPR;
  const unsigned char* wsrc[LPS];
  const unsigned char* xsrc[LPS];
#pragma unroll
  for (int i = 0; i < LPS; ++i) {
    const int row = r0 + i * RPP;
    wsrc[i] = w + (long)(row & 255) * 4608 + g * 16 + (row >> 8) * 2304;
    xsrc[i] = x + ((long)blockIdx.x * x_rows_per_wg + row) * 512 + g * 16;
  }
  int wk = 0;            // byte offset of the k chunk inside a weight row
  long xk = 0;           // byte offset of the chunk inside the private rows
  int xc = 0;
  for (int s = 0; s < nsteps; ++s) {
    const int slot = s % NSLOT;
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
      const bool from_w = SRC == 0 || (SRC == 2 && (i & 1));
      const unsigned char* src = from_w ? wsrc[i] + wk : xsrc[i] + xk;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(smem + slot * STAGE + i * 8192 + wave * 1024), 16, 0, 0);
    }
    wk += ROWB; if (wk >= 2304) wk = 0;
    xk += ROWB; xc += ROWB;
    if (xc == 512) { xc = 0; xk += (long)(LPS * RPP) * 512 - 512; }     // next block of private rows
    if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS * (DEPTH - 1)) : "memory");
    if constexpr (BAR) asm volatile("s_barrier" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink != nullptr && smem[tid * 16] == 123 && nsteps < 0) sink[tid] = 1;
}

template <int ROWB, int STAGE_KIB, int DEPTH, int SRC, bool BAR>
static void run(const unsigned char* w, const unsigned char* x, long xbytes, int ncu) {
  constexpr int STAGE = STAGE_KIB * 1024;
  const int smem = (DEPTH + 1) * STAGE;
  if (smem > 160 * 1024) return;
  auto k = dma_kernel<ROWB, STAGE_KIB, DEPTH, SRC, BAR>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  constexpr int LPS = STAGE / 8192, RPP = 512 / (ROWB / 16);
  const long rows_per_block = (long)LPS * RPP;                      // private rows consumed per 512 / ROWB stages
  const int nsteps = 256;
  const long x_rows_per_wg = rows_per_block * ((long)nsteps * ROWB / 512 + 1);
  if (SRC != 0 && (long)ncu * x_rows_per_wg * 512 > xbytes) { printf("skip (x too small)\n"); return; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(ncu), dim3(512), smem, 0, w, x, nsteps, x_rows_per_wg, (int*)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = (double)ncu * nsteps * STAGE;
  printf("rows %3d B  stage %2d KiB  depth %d (%3d KiB in flight)  src %d  barrier %d : %7.1f GB/s per CU  %6.2f TB/s chip  %.3f us per 64 KiB\n",
         ROWB, STAGE_KIB, DEPTH, DEPTH * STAGE_KIB, SRC, (int)BAR, bytes / ncu / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12,
         best * 1e3 / nsteps * (65536.0 / STAGE));
  hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int SRC, bool BAR>
static void sweep(const unsigned char* w, const unsigned char* x, long xbytes, int ncu) {
  run<128, 64, 1, SRC, BAR>(w, x, xbytes, ncu);
  run<128, 32, 1, SRC, BAR>(w, x, xbytes, ncu);
  run<128, 32, 2, SRC, BAR>(w, x, xbytes, ncu);
  run<128, 32, 3, SRC, BAR>(w, x, xbytes, ncu);
  run<128, 32, 4, SRC, BAR>(w, x, xbytes, ncu);
  run<128, 16, 2, SRC, BAR>(w, x, xbytes, ncu);
  run<128, 16, 4, SRC, BAR>(w, x, xbytes, ncu);
  run<128, 16, 6, SRC, BAR>(w, x, xbytes, ncu);
  run<128, 16, 8, SRC, BAR>(w, x, xbytes, ncu);
  run<64, 32, 1, SRC, BAR>(w, x, xbytes, ncu);
  run<64, 32, 2, SRC, BAR>(w, x, xbytes, ncu);
  run<64, 32, 3, SRC, BAR>(w, x, xbytes, ncu);
  run<64, 32, 4, SRC, BAR>(w, x, xbytes, ncu);
  run<64, 16, 4, SRC, BAR>(w, x, xbytes, ncu);
  run<64, 16, 8, SRC, BAR>(w, x, xbytes, ncu);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  printf("%s, %d CUs\n", p.name, ncu);
  unsigned char *w, *x;
  const long wbytes = 4 << 20, xbytes = 6L << 30;
  hipMalloc(&w, wbytes); hipMalloc(&x, xbytes);
  hipMemset(w, 1, wbytes); hipMemset(x, 2, xbytes);
  hipDeviceSynchronize();
  printf("== weight-like source (shared, L2 hits) ==\n");      sweep<0, true>(w, x, xbytes, ncu);
  printf("== activation-like source (private, HBM stream) ==\n"); sweep<1, true>(w, x, xbytes, ncu);
  printf("== half / half (a GEMM K step) ==\n");                sweep<2, true>(w, x, xbytes, ncu);
  printf("== half / half, no barrier ==\n");                    sweep<2, false>(w, x, xbytes, ncu);
  hipFree(w); hipFree(x);
  return 0;
}
