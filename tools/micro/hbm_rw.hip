// What HBM rate can kernels on an MI355X actually reach, by direction?  The roofline lines of bench.py price bytes at the guide's 6.29 TB/s
// ("float4 copy"); a conv epilogue writes as much as the K loop reads, and the HBM-bound 1x1 layers run at 3.4 - 4.6 TB/s.  This probe
// measures read-only, write-only and copy streams (16 bytes per lane, grid-stride) over grid sizes and loads in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/hbm_rw tools/micro/hbm_rw.hip && tools/micro/hbm_rw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int U>
__global__ void __launch_bounds__(256) k_read(const u32x4_t* __restrict__ src, unsigned* sink, long n) {
  u32x4_t acc = {0u, 0u, 0u, 0u};
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u32x4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = 1;
}

template <int U>
__global__ void __launch_bounds__(256) k_write(u32x4_t* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * 256;
  const u32x4_t v = {threadIdx.x, blockIdx.x, 3u, 4u};
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) dst[i + u * stride] = v;
  }
}

template <int U>
__global__ void __launch_bounds__(256) k_copy(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u32x4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) dst[i + u * stride] = v[u];
  }
}

template <typename F>
static float best_ms(F&& launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  return best;
}

template <int U>
static void sweep(const u32x4_t* src, u32x4_t* dst, unsigned* sink, long n, int ncu) {
  for (int bpc : {2, 4, 8, 16}) {
    const int grid = ncu * bpc;
    const double gb = (double)n * 16 / 1e9;
    const float r = best_ms([&] { hipLaunchKernelGGL(k_read<U>, dim3(grid), dim3(256), 0, 0, src, sink, n); });
    const float w = best_ms([&] { hipLaunchKernelGGL(k_write<U>, dim3(grid), dim3(256), 0, 0, dst, n); });
    const float c = best_ms([&] { hipLaunchKernelGGL(k_copy<U>, dim3(grid), dim3(256), 0, 0, src, dst, n); });
    printf("%2d x 16 B in flight per lane, %2d blocks of 256 per CU:  read %5.2f TB/s   write %5.2f TB/s   copy %5.2f TB/s (read + write bytes)\n",
           U, bpc, gb / r, gb / w, 2 * gb / c);
  }
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  printf("%s, %d CUs\n", p.name, ncu);
  const long nbytes = 4L << 30, n = nbytes / 16;
  u32x4_t *src, *dst; unsigned* sink;
  (void)hipMalloc(&src, nbytes); (void)hipMalloc(&dst, nbytes); (void)hipMalloc(&sink, 4096);
  (void)hipMemset(src, 0x5a, nbytes); (void)hipMemset(dst, 0, nbytes);
  (void)hipDeviceSynchronize();
  sweep<1>(src, dst, sink, n, ncu);
  sweep<4>(src, dst, sink, n, ncu);
  sweep<8>(src, dst, sink, n, ncu);
  return 0;
}
