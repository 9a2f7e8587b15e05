// The 8-wave, register-double-buffered 256 x 256 implicit-GEMM kernel of round 4 (see the comment on the kernel).  It is bit-identical
// to conv_gemm_kernel and was measured SLOWER than the 16-wave kernel on every layer shape of the forward (profiles/r04_gemm_experiments.md),
// so it is compiled into the PROBE build only (tools/build_probes.sh, -DCFT_PROBES: variants 90 / 91 / 190 / 290 / 1690 of
// cft_set_conv_variant); the product library holds the stub at the end of this file.
#include "../conv_common.h"

#ifdef CFT_PROBES

// ------------------------------------------------------------------------------------ 8-wave kernel (round 4)
// 256 x 256 tile, EIGHT waves (2 x 4), wave tile 128 x 64, for the wide 16-bit layers (N >= 256, Cin % 64 == 0).  What it changes
// against conv_gemm_kernel<256,256,4,4> (16 waves of 64 x 64, [6 reads, wait, 8 MFMAs] x 4 and a __syncthreads() per K step):
//   * REGISTER-DOUBLE-BUFFERED FRAGMENTS: 256 VGPRs per wave (16 waves have 128) hold two fragment sets of one 32-wide k half
//     each; the reads of the next half run under the MFMAs of the current one, so a wave has matrix work ready the moment it
//     leaves the barrier, and 12 fragment reads serve 32 MFMAs (0.375 per MFMA against 0.5 for 64 x 64 wave tiles);
//   * a staging buffer is free as soon as every wave has READ it - half a K step before its second half is multiplied - so the
//     64 KiB of step s + 2 are requested in the MIDDLE of step s, right behind the wait that retires step s + 1: the request
//     stream never pauses for a phase of the loop, and a K step has ONE raw s_barrier (64 MFMAs per wave between barriers);
//   * staging by buffer_load_dwordx4 ... lds with the scalar walk in the SGPR offset: no address arithmetic per request.
// The K step stays 64 wide = FULL 128-byte cache lines per staged row.  The first form of this kernel staged 32-wide sub-steps
// (64-byte rows) through a four-slot ring with three sub-steps in flight; it was bit-identical and 3-8 % SLOWER than the 16-wave
// kernel: the L2 -> LDS stream is bound by the request rate of the texture-addresser path, not by latency - half-line rows halve
// its throughput (tools/micro/dma_ring.hip: 114 GB/s per CU with 128-byte rows at >= 64 KiB in flight, 62 GB/s with 64-byte rows,
// independent of the depth; profiles/r04_gemm_experiments.md).
// LDS image, swizzle, products, k order per accumulator and the epilogue are those of conv_gemm_kernel: bit-identical to it
// (tests/test_gpu_ops.py: variant 91 against 900).
template <typename T, bool CHUNK, int ABLATE = 0>      // CHUNK: chunk-major K walk (3x3, Cin >= 256), else tap-major - as conv_gemm_kernel picks it
__global__ void __launch_bounds__(512) conv_gemm_ring_kernel(const ConvParams p) {
  static_assert(sizeof(T) == 2, "16-bit operand types only");
  constexpr int BM = 256, BN = 256, GE = 8, ES = 2, BK = 64;
  constexpr int STAGE_A = BM * 128, STAGE = (BM + BN) * 128;     // one K step: 32 KiB + 32 KiB
  constexpr int WM = 128, WN = 64, MT = 8, NT = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int nb = gridDim.x, bid = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, slot = bid >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  const int tm = logical / p.tilesN, tn = logical - tm * p.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int rs = tid >> 3;                                             // staging row of this thread inside a 64-row pass
  const int g = (tid & 7) ^ (rs & 7);                                  // k-granule it fetches (source-side swizzle: slot ^ (row & 7))

  int a_off[4];
  uint32_t a_mask[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + rs + i * 64;
    a_off[i] = 0;
    a_mask[i] = 0;
    if (m < p.M) {
      const int t = fast_div(m, p.wo_mul, p.wo_sh);
      const int wo = m - t * p.Wo;
      const int b = fast_div(t, p.ho_mul, p.ho_sh);
      const int ho = t - b * p.Ho;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      a_off[i] = ((b * p.H + hi0) * p.W + wi0) * p.ldx + p.xoff;
      uint32_t wbits = 0, mk = 0;
      for (int kw = 0; kw < p.KS; ++kw) wbits |= ((unsigned)(wi0 + kw) < (unsigned)p.W ? 1u : 0u) << kw;
      for (int kh = 0; kh < p.KS; ++kh)
        if ((unsigned)(hi0 + kh) < (unsigned)p.H) mk |= wbits << (kh * p.KS);
      a_mask[i] = mk;
    }
  }
  // Staging by buffer_load_dwordx4 ... lds: the address is SRD base + a CONSTANT per-thread VGPR offset + an SGPR offset that carries the
  // scalar walk, so a K step's eight requests cost no address arithmetic at all; a masked granule (tap outside the image, row beyond M
  // or N, step beyond K) is fetched at the out-of-range offset 2^31, for which a buffer load returns - and the DMA writes - zeros.
  // Only voffset is range-checked, so the input SRD starts `abias` bytes BELOW the tensor (the most negative tap of a border pixel) and
  // every voffset carries + abias: in-image taps of border pixels then have voffset >= 0.
  constexpr uint32_t OOB = 0x80000000u;
  const long abias = ((long)p.W + 1) * p.ldx * ES;
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.x) - abias, 0, (int)(p.x_bytes + abias), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  uint32_t voffA[4], voffB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    voffA[i] = (uint32_t)(((long)a_off[i] + g * GE) * ES + abias);
    const int n = n0 + rs + i * 64;
    voffB[i] = (n < p.N) ? (uint32_t)(((long)n * p.Kpad + g * GE) * ES) : OOB;
  }
  // scalar walk of the ISSUE pointer over the 64-wide K steps (as the UNIK path of conv_gemm_kernel); all uniform, branch-free
  const int tap_step = (p.ldx - p.Cin) * ES;
  const int row_step = (p.W - p.KS) * p.ldx * ES;
  constexpr bool chunk_major = CHUNK;
  const int cm_tap = p.ldx * ES, cm_tapb = p.Cin * ES;                                     // chunk-major: next tap of the same chunk
  const int cm_chunk = (BK - 3 * p.W * p.ldx) * ES, cm_chunkb = (BK - 9 * p.Cin) * ES;     //              after nine taps the next chunk
  int u_tap = 0, u_kw = 0, u_ci = 0, u_offa = 0, u_offb = 0, si = 0;
  const int nk = p.Kpad / BK;

#define RING_ISSUE(sl_)                                                                                     \
  {                                                                                                         \
    const bool live_ = si < nk;                                                                             \
    const uint32_t tapbit_ = live_ ? (1u << u_tap) : 0u;                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
      const uint32_t vo_ = (a_mask[i] & tapbit_) ? voffA[i] : OOB;                                          \
      if constexpr (!(ABLATE & 1))                                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_void_t*)(smem + (sl_) * STAGE + i * 8192 + wave * 1024), 16, vo_, u_offa, 0, 0); \
    }                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
      const uint32_t vo_ = live_ ? voffB[i] : OOB;                                                          \
      if constexpr (!(ABLATE & 1))                                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_void_t*)(smem + (sl_) * STAGE + STAGE_A + i * 8192 + wave * 1024), 16, vo_, u_offb, 0, 0); \
    }                                                                                                       \
    ++si;                                                                                                   \
    if (chunk_major) {                                                                                      \
      const int t1_ = u_tap + 1, k1_ = u_kw + 1;                                                            \
      const bool roww_ = k1_ == 3, nextc_ = t1_ == 9;                                                       \
      u_offa += cm_tap + (roww_ ? row_step : 0) + (nextc_ ? cm_chunk : 0);                                  \
      u_offb += cm_tapb + (nextc_ ? cm_chunkb : 0);                                                         \
      u_kw = roww_ ? 0 : k1_; u_tap = nextc_ ? 0 : t1_;                                                     \
    } else {                                                                                                \
      const int c1_ = u_ci + BK;                                                                            \
      const bool wrap_ = c1_ == p.Cin;                                                                      \
      const int k1_ = u_kw + 1;                                                                             \
      const bool roww_ = wrap_ && k1_ == p.KS;                                                              \
      u_offa += BK * ES + (wrap_ ? tap_step : 0) + (roww_ ? row_step : 0);                                  \
      u_offb += BK * ES;                                                                                    \
      u_ci = wrap_ ? 0 : c1_;                                                                               \
      u_tap += wrap_ ? 1 : 0;                                                                               \
      u_kw = wrap_ ? (roww_ ? 0 : k1_) : u_kw;                                                              \
    }                                                                                                       \
  }

  // fragment reads: lane l -> row l & 15 of a 16-row MFMA tile, k-granule (half * 4 + (l >> 4)) of the K step in slot ^ (row & 7):
  // the second half flips bit 6 of the byte offset
  const int rdoff = (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
  const unsigned char* rdA0 = smem + (wm * WM) * 128 + rdoff;
  const unsigned char* rdB0 = smem + STAGE_A + (wn * WN) * 128 + rdoff;
  const unsigned char* rdA1 = smem + (wm * WM) * 128 + (rdoff ^ 64);
  const unsigned char* rdB1 = smem + STAGE_A + (wn * WN) * 128 + (rdoff ^ 64);
  gran_t fa0[MT], fb0[NT], fa1[MT], fb1[NT];
#define RING_READ(fa_, fb_, sl_, h_)                                                                        \
  {                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) fa_[i] = *reinterpret_cast<const gran_t*>(((h_) ? rdA1 : rdA0) + (sl_) * STAGE + i * 2048); \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) fb_[j] = *reinterpret_cast<const gran_t*>(((h_) ? rdB1 : rdB0) + (sl_) * STAGE + j * 2048); \
  }
#define RING_MMA(fa_, fb_)                                                                                  \
  _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                            \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                        \
      if constexpr (ABLATE & 2) { asm volatile("" ::"v"(fa_[i]), "v"(fb_[j])); }                            \
      else acc[i][j] = mma_granule<T>(fa_[i], fb_[j], acc[i][j]);                                           \
    }
// One K step s (buffer c_ = s & 1; fa0 / fb0 hold its first half): read its second half under the MFMAs of the first; then
// [every wave has read buffer c_; the DMA of step s + 1 has landed in every wave] -> barrier -> request step s + 2 into buffer c_ ->
// read the first half of step s + 1 under the MFMAs of the second half of step s.
#define RING_STEP(c_)                                                                                       \
  {                                                                                                         \
    RING_READ(fa1, fb1, c_, 1)                                                                              \
    RING_MMA(fa0, fb0)                                                                                      \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                \
    RING_ISSUE(c_)                                                                                          \
    RING_READ(fa0, fb0, (c_) ^ 1, 0)                                                                        \
    RING_MMA(fa1, fb1)                                                                                      \
  }

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bias_v[NT];
  conv_load_bias<WN>(p, n0, wn, lane, bias_v);

  RING_ISSUE(0) RING_ISSUE(1)
  asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");           // step 0 has landed in every wave (step 1: 8 requests per thread in flight)
  RING_READ(fa0, fb0, 0, 0)
  const int npairs = nk >> 1;
  for (int sp = 0; sp < npairs; ++sp) {
    RING_STEP(0)
    RING_STEP(1)
  }
  if (nk & 1) RING_STEP(0)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the zero steps requested past K still land in LDS; the strips alias the buffers
#undef RING_ISSUE
#undef RING_READ
#undef RING_MMA
#undef RING_STEP

  if constexpr (ABLATE & 16) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  // The epilogue's dozen launch parameters are re-read from the kernel-argument segment HERE (the pointer is laundered through an
  // empty asm so that the loads cannot be hoisted): held in SGPRs across the K loop they push the loop's own scalars - buffer
  // descriptors, walk offsets - into VGPRs, and hipcc then wraps every buffer_load in a waterfall loop.
#if defined(__HIP_DEVICE_COMPILE__)
  const __attribute__((address_space(4))) ConvParams* kp =
      (const __attribute__((address_space(4))) ConvParams*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  const ConvParams pe = *kp;
#else
  const ConvParams& pe = p;
#endif
  conv_epilogue<T, WM, WN>(pe, acc, smem, m0, n0, wm, wn, wave, lane, bias_v);
}


// ------------------------------------------------------------------------------------ host
template <typename T, bool CHUNK, int ABLATE>
static int launch_ring_c(const ConvParams& q, int grid, hipStream_t stream) {
  constexpr int smem_bytes = 2 * (256 + 256) * 128;
  cft_allow_lds<&conv_gemm_ring_kernel<T, CHUNK, ABLATE>>(smem_bytes);
  hipLaunchKernelGGL((conv_gemm_ring_kernel<T, CHUNK, ABLATE>), dim3(grid), dim3(512), smem_bytes, stream, q);
  return cft_check_launch("conv_gemm_ring_kernel");
}

template <typename T, int ABLATE>
static int launch_ring_t(const ConvParams& p, hipStream_t stream) {
  ConvParams q = p;
  const int tilesM = (p.M + 255) / 256;
  q.tilesN = (p.N + 255) / 256;
  const bool chunk_major = p.KS == 3 && p.Cin >= 256 && p.K == 9 * p.Cin;      // the rule of conv_gemm_kernel
  if (chunk_major) return launch_ring_c<T, true, ABLATE>(q, tilesM * q.tilesN, stream);
  return launch_ring_c<T, false, ABLATE>(q, tilesM * q.tilesN, stream);
}

template <typename T>
static int launch_ring_a(const ConvParams& p, int ablate, hipStream_t stream) {
  switch (ablate) {
    case 0: return launch_ring_t<T, 0>(p, stream);
    case 1: return launch_ring_t<T, 1>(p, stream);
    case 2: return launch_ring_t<T, 2>(p, stream);
    case 16: return launch_ring_t<T, 16>(p, stream);
    default: break;
  }
  cft_set_error("conv_ring_launch: unknown probe");
  return CFT_EINVAL;
}

int conv_ring_launch(const ConvParams& p, int dtype, int ablate, hipStream_t stream) {
  if (dtype == CFT_BF16) return launch_ring_a<uint16_t>(p, ablate, stream);
  if (dtype == CFT_F16) return launch_ring_a<f16_t>(p, ablate, stream);
  cft_set_error("conv_gemm_ring_kernel: 16-bit operand types only");
  return CFT_EINVAL;
}

#else   // product build: the kernel is not part of the library

int conv_ring_launch(const ConvParams&, int, int, hipStream_t) {
  cft_set_error("conv_gemm_ring_kernel is compiled into the probe build only (tools/build_probes.sh)");
  return CFT_EINVAL;
}

#endif
