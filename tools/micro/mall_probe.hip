// Does the 256-MiB Infinity Cache (MALL) serve a CONSUMER kernel what a PRODUCER kernel just wrote?  Decides whether running the
// byte-heavy prefix of the network sub-batch by sub-batch (Model.depth_first) can take tensor reads off HBM.
// For a working set of S bytes:  write(S) then read(S) [timed: the read];  read(S) twice [timed: the second];  copy a->b then
// copy b->c [timed: the second: reads what the first wrote while writing as much again].
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mall_probe tools/micro/mall_probe.hip && tools/micro/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__global__ void __launch_bounds__(256) k_read(const u32x4_t* __restrict__ src, unsigned* sink, long n) {
  u32x4_t acc = {0u, 0u, 0u, 0u};
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    u32x4_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = 1;
}

__global__ void __launch_bounds__(256) k_write(u32x4_t* __restrict__ dst, long n, unsigned tag) {
  const long stride = (long)gridDim.x * 256;
  const u32x4_t v = {threadIdx.x, blockIdx.x, tag, 4u};
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
#pragma unroll
    for (int u = 0; u < 4; ++u) dst[i + u * stride] = v;
  }
}

__global__ void __launch_bounds__(256) k_copy(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    u32x4_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u) dst[i + u * stride] = v[u];
  }
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount * 8;
  printf("%s, %d CUs, l2CacheSize %d\n", p.name, p.multiProcessorCount, p.l2CacheSize);
  const long cap = 3L << 30;
  u32x4_t *a, *b, *c, *junk; unsigned* sink;
  (void)hipMalloc(&a, cap); (void)hipMalloc(&b, cap); (void)hipMalloc(&c, cap); (void)hipMalloc(&junk, cap); (void)hipMalloc(&sink, 4096);
  (void)hipMemset(a, 0x5a, cap); (void)hipMemset(b, 0, cap); (void)hipMemset(c, 0, cap); (void)hipMemset(junk, 1, cap);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const long sizes_mb[] = {8, 16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 2048};
  printf("%8s %14s %14s %14s %16s\n", "S (MB)", "read-after-wr", "read-after-rd", "read-cold", "copy-after-copy");
  for (long mb : sizes_mb) {
    const long n = mb * (1L << 20) / 16;
    const double gb = (double)n * 16 / 1e9;
    float best[4] = {1e9f, 1e9f, 1e9f, 1e9f};
    for (int rep = 0; rep < 4; ++rep) {
      float ms;
      // (0) producer write, consumer read
      hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, junk, sink, cap / 16);        // flush: 3 GB of other traffic
      hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n, (unsigned)rep);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, b, sink, n);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best[0]) best[0] = ms;
      // (1) read, read again
      hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, junk, sink, cap / 16);
      hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, sink, n);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, sink, n);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best[1]) best[1] = ms;
      // (2) cold read (after the flush)
      hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, junk, sink, cap / 16);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, sink, n);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best[2]) best[2] = ms;
      // (3) copy a -> b, then copy b -> c (timed)
      hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, junk, sink, cap / 16);
      hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, b, c, n);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best[3]) best[3] = ms;
    }
    printf("%8ld %9.2f TB/s %9.2f TB/s %9.2f TB/s %9.2f TB/s (r+w)\n", mb, gb / best[0], gb / best[1], gb / best[2], 2 * gb / best[3]);
  }
  return 0;
}
