// Round 6: the hand-scheduled K loop of the 256 x 256 x 64 GEMM tile as ONE inline-asm block per wave group,
// as a REAL GEMM (C = A . B^T, bf16 operands, fp32 result) so that every schedule variant is checked for races / stale tiles
// against a plain reference before it is timed.  The asm text comes from tools/micro/gen_kloop.py (register map and schedule
// there); this file is the harness: operands, the two timing modes and the check.
//   python tools/micro/gen_kloop.py > tools/micro/kloop_variants.inc
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/kloop_asm tools/micro/kloop_asm.hip && tools/micro/kloop_asm
// Kill criterion: all three of MFMAs + fragment reads + LDS-DMA <= 1.25 us per K step on an L2-hit source
// (the no-scheduling loop of tools/micro/dma_ring.hip: 1.48); above 1.35: stop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <algorithm>
#include "kloop_variants.inc"

typedef __attribute__((address_space(3))) void lds_void_t;

struct KArgs {
  const uint16_t* A;      // [M][lda] bf16
  const uint16_t* B;      // [N][ldb] bf16
  float* C;               // [M][N] fp32 (nullptr: timing run, nothing stored)
  unsigned long long* timers;   // per wave: {100-MHz ticks, shader cycles} of the asm block
  int M, N, lda, ldb;
  int npairs;             // K steps / 2
  unsigned kwrap;         // bytes: the K offset wraps to 0 here (timing runs re-walk an L2-sized K range); 0xffffffff: never
  int same_tile;          // 1: every workgroup computes tile (0, 0) - every request of the chip hits the same 2 x 256 rows (L2 hits)
  int tilesN;
};

#define KLOOP_CLOBBERS                                                                                                     \
  "memory", "scc",                                                                                                         \
  "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
  "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63", \
  "a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95", \
  "a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127", \
  "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
  "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
  "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"

#define KLOOP_STMT(text_)                                                                                                  \
  asm volatile(text_                                                                                                       \
               : [sofa] "+s"(sofa), [sofb] "+s"(sofb), [cnt] "+s"(cnt), [m0s] "=&s"(m0s)                                   \
               : [voa0] "v"(voa[0]), [voa1] "v"(voa[1]), [voa2] "v"(voa[2]), [voa3] "v"(voa[3]),                           \
                 [vob0] "v"(vob[0]), [vob1] "v"(vob[1]), [vob2] "v"(vob[2]), [vob3] "v"(vob[3]),                           \
                 [ra00] "v"(ra[0][0]), [ra01] "v"(ra[0][1]), [ra10] "v"(ra[1][0]), [ra11] "v"(ra[1][1]),                   \
                 [rb00] "v"(rb[0][0]), [rb01] "v"(rb[0][1]), [rb10] "v"(rb[1][0]), [rb11] "v"(rb[1][1]),                   \
                 [srda] "s"(srdA), [srdb] "s"(srdB), [wb] "s"(wb), [kwrap] "s"(kwrap)                                      \
               : KLOOP_CLOBBERS)

struct KState {
  uint32_t voa[4], vob[4], ra[2][2], rb[2][2];
  int m0, n0, wm, wn, lane, wave;
};

__device__ __forceinline__ void kloop_setup(const KArgs& a, unsigned char* smem, KState& s) {
  const int bid = blockIdx.x;
  int tm = 0, tn = 0;
  if (!a.same_tile) {
    const int nb = gridDim.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, slot = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;   // consecutive tiles on one XCD
    tm = logical / a.tilesN; tn = logical - tm * a.tilesN;
  }
  s.m0 = tm * 256; s.n0 = tn * 256;
  const int tid = threadIdx.x;
  s.lane = tid & 63;
  s.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  s.wm = s.wave >> 2; s.wn = s.wave & 3;
  const int rs = tid >> 3, g = (tid & 7) ^ (rs & 7);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s.voa[i] = (uint32_t)(((long)(s.m0 + rs + 64 * i) * a.lda + g * 8) * 2);
    s.vob[i] = (uint32_t)(((long)(s.n0 + rs + 64 * i) * a.ldb + g * 8) * 2);
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const int lrow = s.lane & 15, lgrp = s.lane >> 4;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint32_t rd = (uint32_t)(lrow * 128 + (((h * 4 + lgrp) ^ (lrow & 7)) << 4));
      s.ra[h][c] = lds0 + c * 65536u + s.wm * 16384u + rd;
      s.rb[h][c] = lds0 + c * 65536u + 32768u + s.wn * 8192u + rd;
    }
}

__device__ __forceinline__ void kloop_store(const KArgs& a, const KState& s) {
  if (a.C == nullptr) return;
  float v[128];
#define RD(n_) asm volatile("v_accvgpr_read_b32 %0, a" #n_ : "=v"(v[n_]));
  RD(0) RD(1) RD(2) RD(3) RD(4) RD(5) RD(6) RD(7) RD(8) RD(9) RD(10) RD(11) RD(12) RD(13) RD(14) RD(15)
  RD(16) RD(17) RD(18) RD(19) RD(20) RD(21) RD(22) RD(23) RD(24) RD(25) RD(26) RD(27) RD(28) RD(29) RD(30) RD(31)
  RD(32) RD(33) RD(34) RD(35) RD(36) RD(37) RD(38) RD(39) RD(40) RD(41) RD(42) RD(43) RD(44) RD(45) RD(46) RD(47)
  RD(48) RD(49) RD(50) RD(51) RD(52) RD(53) RD(54) RD(55) RD(56) RD(57) RD(58) RD(59) RD(60) RD(61) RD(62) RD(63)
  RD(64) RD(65) RD(66) RD(67) RD(68) RD(69) RD(70) RD(71) RD(72) RD(73) RD(74) RD(75) RD(76) RD(77) RD(78) RD(79)
  RD(80) RD(81) RD(82) RD(83) RD(84) RD(85) RD(86) RD(87) RD(88) RD(89) RD(90) RD(91) RD(92) RD(93) RD(94) RD(95)
  RD(96) RD(97) RD(98) RD(99) RD(100) RD(101) RD(102) RD(103) RD(104) RD(105) RD(106) RD(107) RD(108) RD(109) RD(110) RD(111)
  RD(112) RD(113) RD(114) RD(115) RD(116) RD(117) RD(118) RD(119) RD(120) RD(121) RD(122) RD(123) RD(124) RD(125) RD(126) RD(127)
#undef RD
  const int lrow = s.lane & 15, lgrp = s.lane >> 4;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = s.m0 + s.wm * 128 + i * 16 + lgrp * 4 + e, n = s.n0 + s.wn * 64 + j * 16 + lrow;
        if (m < a.M && n < a.N) a.C[(long)m * a.N + n] = v[4 * (4 * i + j) + e];
      }
}

#define KLOOP_KERNEL(n_)                                                                                                   \
  __global__ void __launch_bounds__(512) kloop_kernel_##n_(const KArgs a) {                                                \
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                                                   \
    KState s;                                                                                                              \
    kloop_setup(a, smem, s);                                                                                               \
    uint32_t voa[4] = {s.voa[0], s.voa[1], s.voa[2], s.voa[3]}, vob[4] = {s.vob[0], s.vob[1], s.vob[2], s.vob[3]};         \
    uint32_t ra[2][2] = {{s.ra[0][0], s.ra[0][1]}, {s.ra[1][0], s.ra[1][1]}}, rb[2][2] = {{s.rb[0][0], s.rb[0][1]}, {s.rb[1][0], s.rb[1][1]}}; \
    const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.A), 0, (int)((long)a.M * a.lda * 2), 0x00020000); \
    const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.B), 0, (int)((long)a.N * a.ldb * 2), 0x00020000); \
    uint32_t sofa = 0, sofb = 0, cnt = (uint32_t)a.npairs, m0s;                                                            \
    const uint32_t wb = (uint32_t)(uintptr_t)(lds_void_t*)smem + (uint32_t)s.wave * 1024u;                                 \
    const uint32_t kwrap = a.kwrap;                                                                                        \
    const unsigned long long t0 = wall_clock64(), c0 = clock64();                                                          \
    asm volatile("" ::: "memory");                                                                                         \
    if (s.wm == 0) { KLOOP_STMT(KLOOP_ASM_##n_##_G0); } else { KLOOP_STMT(KLOOP_ASM_##n_##_G1); }                          \
    asm volatile("" ::: "memory");                                                                                         \
    const unsigned long long t1 = wall_clock64(), c1 = clock64();                                                          \
    if (a.timers != nullptr && s.lane == 0) {                                                                              \
      a.timers[((long)blockIdx.x * 8 + s.wave) * 2] = t1 - t0;                                                             \
      a.timers[((long)blockIdx.x * 8 + s.wave) * 2 + 1] = c1 - c0;                                                         \
    }                                                                                                                      \
    kloop_store(a, s);                                                                                                     \
  }
KLOOP_FOR_EACH(KLOOP_KERNEL)

typedef void (*kloop_fn)(const KArgs);
#define KLOOP_PTR(n_) kloop_kernel_##n_,
static kloop_fn kloop_table[KLOOP_NVARIANTS] = {KLOOP_FOR_EACH(KLOOP_PTR)};

// ------------------------------------------------------------------------------------------------------------------------
__global__ void fill_bf16(uint16_t* p, long n, unsigned seed) {   // uniform [-1, 1), full-range mantissas, mixed signs
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    unsigned h = ((unsigned)i + seed) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float f = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    unsigned u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u);
    p[i] = (uint16_t)(u >> 16);
  }
}

__global__ void ref_gemm(const uint16_t* A, const uint16_t* B, float* C, int M, int N, int K, int lda, int ldb) {
  const int n = blockIdx.x * 16 + (threadIdx.x & 15), m = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (m >= M || n >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k)
    s = fmaf(__uint_as_float((unsigned)A[(long)m * lda + k] << 16), __uint_as_float((unsigned)B[(long)n * ldb + k] << 16), s);
  C[(long)m * N + n] = s;
}

static bool check(int v, const uint16_t* A, const uint16_t* B, float* C, float* Cref, int M, int N, int K) {
  const int lda = K + 128, ldb = K + 128;
  KArgs a{A, B, C, nullptr, M, N, lda, ldb, K / 128, 0xffffffffu, 0, N / 256};
  hipMemset(C, 0xff, (size_t)M * N * 4);
  hipLaunchKernelGGL(kloop_table[v], dim3((M / 256) * (N / 256)), dim3(512), 131072, 0, a);
  hipLaunchKernelGGL(ref_gemm, dim3(N / 16, M / 16), dim3(256), 0, 0, A, B, Cref, M, N, K, lda, ldb);
  if (hipDeviceSynchronize() != hipSuccess) { printf("  %-14s launch failed: %s\n", kloop_names[v], hipGetErrorString(hipGetLastError())); return false; }
  std::vector<float> h((size_t)M * N), r((size_t)M * N);
  hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(r.data(), Cref, r.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0; long bad = 0;
  for (size_t i = 0; i < h.size(); ++i) {
    const double e = std::fabs((double)h[i] - r[i]);
    if (!(e <= 1e-3 * (1.0 + std::sqrt((double)K)))) ++bad;
    if (e > maxerr || e != e) maxerr = e;
  }
  printf("  %-14s M=%d N=%d K=%d: max |err| = %.3e, %ld bad of %zu  %s\n", kloop_names[v], M, N, K, maxerr, bad, h.size(), bad ? "FAIL" : "ok");
  return bad == 0;
}

static void timeit(int v, const uint16_t* A, const uint16_t* B, unsigned long long* timers, int ncu, int same_tile, int nk, int kspan) {
  // timing: ncu workgroups (one per CU), nk K steps each over a K range of kspan elements that wraps (L2-resident operands)
  const int T = same_tile ? 1 : 16;      // same_tile 0: a 4096 x 4096 output = 256 tiles
  const int M = T * 256, N = T * 256, lda = kspan + 128, ldb = kspan + 128;
  KArgs a{A, B, nullptr, timers, M, N, lda, ldb, nk / 2, (unsigned)kspan * 2u, same_tile, T};
  double best_us = 1e30, best_mhz = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(kloop_table[v], dim3(same_tile ? ncu : T * T), dim3(512), 131072, 0, a);
    if (hipDeviceSynchronize() != hipSuccess) { printf("  %-14s launch failed\n", kloop_names[v]); return; }
    const int nw = (same_tile ? ncu : T * T) * 8;
    std::vector<unsigned long long> t((size_t)nw * 2);
    hipMemcpy(t.data(), timers, t.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> us(nw);
    double cyc = 0;
    for (int i = 0; i < nw; ++i) { us[i] = t[2 * i] / 100.0; cyc += (double)t[2 * i + 1]; }
    std::sort(us.begin(), us.end());
    const double med = us[nw / 2];
    if (rep > 0 && med < best_us) { best_us = med; double tt = 0; for (int i = 0; i < nw; ++i) tt += t[2 * i] / 100.0; best_mhz = cyc / tt; }
  }
  const double per = best_us / nk;
  printf("  %-14s %s: %.3f us per 256x256x64 step = %7.1f TFLOP/s chip-equivalent (256 CUs), shader clock %.0f MHz, %.0f cycles per step\n", kloop_names[v],
         same_tile ? "one tile, L2 hits " : "4096^2 GEMM tiles", per, 2.0 * 256 * 256 * 64 * 256 / (per * 1e-6) / 1e12, best_mhz, per * best_mhz);
}

int main(int argc, char** argv) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  printf("%s, %d CUs\n", p.name, ncu);
  for (int v = 0; v < KLOOP_NVARIANTS; ++v)
    hipFuncSetAttribute(reinterpret_cast<const void*>(kloop_table[v]), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int KMAX = 4096;
  const long nA = 4096L * (KMAX + 128);
  uint16_t *A, *B; float *C, *Cref; unsigned long long* timers;
  hipMalloc(&A, nA * 2); hipMalloc(&B, nA * 2); hipMalloc(&C, 4096L * 4096 * 4); hipMalloc(&Cref, 4096L * 4096 * 4);
  hipMalloc(&timers, 4096 * 8 * 2 * 8);
  hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, A, nA, 12345u);
  hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, B, nA, 98765u);
  hipDeviceSynchronize();
  const bool only_time = argc > 1 && argv[1][0] == 't';
  bool okv[KLOOP_NVARIANTS];
  printf("== check against a plain fp32 reference (the ablated variants are wrong by construction) ==\n");
  for (int v = 0; v < KLOOP_NVARIANTS; ++v) {
    okv[v] = true;
    if (only_time) continue;
    okv[v] = check(v, A, B, C, Cref, 512, 512, 512);
    if (okv[v]) okv[v] = check(v, A, B, C, Cref, 4096, 4096, 1024);
  }
  printf("== time per K step (in-kernel 100-MHz clock around the asm block, median wave, best of 4) ==\n");
  for (int v = 0; v < KLOOP_NVARIANTS; ++v) {
    timeit(v, A, B, timers, ncu, 1, 512, 2048);
    timeit(v, A, B, timers, ncu, 0, 512, 2048);
  }
  return 0;
}
