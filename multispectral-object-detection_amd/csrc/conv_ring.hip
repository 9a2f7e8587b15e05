// Ring-staged implicit-GEMM kernel for the wide 16-bit layers (see the comment on the kernel); launched by conv_gemm.hip's dispatch.
#include "conv_common.h"

// ------------------------------------------------------------------------------------ ring kernel (round 4)
// 256 x 256 tile, EIGHT waves (2 x 4), wave tile 128 x 64, for the wide 16-bit layers (N >= 256, Cin % 64 == 0).  What it changes
// against conv_gemm_kernel<256,256,4,4> (16 waves, two 64-KiB buffers, one __syncthreads() per 64-wide K step):
//   * the K walk is staged in SUB-STEPS of 32 (64-byte rows: 16 KiB of A + 16 KiB of B) through a ring of FOUR 32-KiB slots;
//   * the fragments of sub-step j + 1 are read into a second register set while the MFMAs of sub-step j run from the first
//     (256 VGPRs per wave leave room for it; 16 waves at 128 do not), so a wave has matrix work ready the moment it leaves
//     the barrier, and a slot is free as soon as every wave has READ it - one sub-step before it is multiplied;
//   * so the LDS-DMA requests of sub-step j + 4 are issued at sub-step j: three sub-steps (96 KiB per CU) stay in flight
//     across the barriers behind a counted s_waitcnt vmcnt(8) - never 0 inside the loop - against one 64-KiB step for the
//     double-buffered kernel (profiles/r02_gemm_experiments.md: the L2 -> LDS stream is latency x bytes-in-flight bound);
//   * one raw s_barrier per sub-step (no stagger between wave groups: round 2's eight barriers per K tile cost more than
//     the stagger returned), 12 fragment reads per 32 MFMAs (0.375 per MFMA against 0.5 for 64 x 64 wave tiles).
// Sub-steps past K are staged from the zero page into the slot that is free anyway, so every vmcnt count is uniform to the end.
// LDS image: row-major 64-byte rows, granule slot XOR ((row >> 3) & 1) << 1 (conflict-free for ds_read_b128's four 16-lane
// groups), applied on the source side of the DMA like the 128-byte-row swizzle.  Same products, same k order per accumulator,
// same epilogue as conv_gemm_kernel: bit-identical to it (tests/test_gpu_ops.py: variant 90 against 900).
template <typename T, bool CHUNK, int ABLATE = 0>      // CHUNK: chunk-major K walk (3x3, Cin >= 256), else tap-major - as conv_gemm_kernel picks it
__global__ void __launch_bounds__(512) conv_gemm_ring_kernel(const ConvParams p) {
  static_assert(sizeof(T) == 2, "16-bit operand types only");
  constexpr int BM = 256, BN = 256, GE = 8, ES = 2, BK = 64;
  constexpr int STAGE_A = BM * 64, STAGE = (BM + BN) * 64;     // one sub-step: 16 KiB + 16 KiB
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
  const int rs = wave * 16 + (lane >> 2);                              // staging row of this thread inside a 128-row pass
  const int g = (lane & 3) ^ (((lane >> 5) & 1) << 1);                 // k-granule it fetches: slot ^ ((row >> 3) & 1) << 1

  int a_off[2];
  uint32_t a_mask[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + rs + i * 128;
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
  // scalar walk, so a sub-step's four requests cost no address arithmetic at all; a masked granule (tap outside the image, row beyond M
  // or N, sub-step beyond K) is fetched at the out-of-range offset 2^31, for which a buffer load returns - and the DMA writes - zeros.
  // Only voffset is range-checked, so the input SRD starts `abias` bytes BELOW the tensor (the most negative tap of a border pixel) and
  // every voffset carries + abias: in-image taps of border pixels then have voffset >= 0.
  constexpr uint32_t OOB = 0x80000000u;
  const long abias = ((long)p.W + 1) * p.ldx * ES;
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.x) - abias, 0, (int)(p.x_bytes + abias), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  uint32_t voffA[2], voffB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    voffA[i] = (uint32_t)(((long)a_off[i] + g * GE) * ES + abias);
    const int n = n0 + rs + i * 128;
    voffB[i] = (n < p.N) ? (uint32_t)(((long)n * p.Kpad + g * GE) * ES) : OOB;
  }
  // scalar walk of the ISSUE pointer over 64-wide K steps (as the UNIK path of conv_gemm_kernel), two sub-steps per step; all uniform
  const int tap_step = (p.ldx - p.Cin) * ES;
  const int row_step = (p.W - p.KS) * p.ldx * ES;
  constexpr bool chunk_major = CHUNK;
  const int cm_tap = p.ldx * ES, cm_tapb = p.Cin * ES;                                     // chunk-major: next tap of the same chunk
  const int cm_chunk = (BK - 3 * p.W * p.ldx) * ES, cm_chunkb = (BK - 9 * p.Cin) * ES;     //              after nine taps the next chunk
  int u_tap = 0, u_kw = 0, u_ci = 0, u_offa = 0, u_offb = 0, u_half = 0, si = 0;
  const int NS = p.Kpad / 32;

#define RING_ISSUE(sl_)                                                                                     \
  {                                                                                                         \
    const bool live_ = si < NS;                                                                             \
    const uint32_t tapbit_ = live_ ? (1u << u_tap) : 0u;                                                    \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                         \
      const uint32_t vo_ = (a_mask[i] & tapbit_) ? voffA[i] : OOB;                                          \
      if constexpr (!(ABLATE & 1))                                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_void_t*)(smem + (sl_) * STAGE + i * 8192 + wave * 1024), 16, vo_, u_offa + u_half, 0, 0); \
    }                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                         \
      const uint32_t vo_ = live_ ? voffB[i] : OOB;                                                          \
      if constexpr (!(ABLATE & 1))                                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_void_t*)(smem + (sl_) * STAGE + STAGE_A + i * 8192 + wave * 1024), 16, vo_, u_offb + u_half, 0, 0); \
    }                                                                                                       \
    ++si;                                                                                                   \
    const bool adv_ = u_half != 0;                 /* second half of a 64-wide step: advance the walk */        \
    u_half ^= 64;                                                                                           \
    if (chunk_major) {                                                                                      \
      const int t1_ = u_tap + 1, k1_ = u_kw + 1;                                                            \
      const bool roww_ = k1_ == 3, nextc_ = t1_ == 9;                                                       \
      const int da_ = cm_tap + (roww_ ? row_step : 0) + (nextc_ ? cm_chunk : 0);                            \
      const int db_ = cm_tapb + (nextc_ ? cm_chunkb : 0);                                                   \
      u_offa += adv_ ? da_ : 0; u_offb += adv_ ? db_ : 0;                                                   \
      u_kw = adv_ ? (roww_ ? 0 : k1_) : u_kw; u_tap = adv_ ? (nextc_ ? 0 : t1_) : u_tap;                    \
    } else {                                                                                                \
      const int c1_ = u_ci + BK;                                                                            \
      const bool wrap_ = c1_ == p.Cin;                                                                      \
      const int k1_ = u_kw + 1;                                                                             \
      const bool roww_ = wrap_ && k1_ == p.KS;                                                              \
      const int da_ = BK * ES + (wrap_ ? tap_step : 0) + (roww_ ? row_step : 0);                            \
      u_offa += adv_ ? da_ : 0; u_offb += adv_ ? BK * ES : 0;                                               \
      u_ci = adv_ ? (wrap_ ? 0 : c1_) : u_ci;                                                               \
      u_tap += (adv_ && wrap_) ? 1 : 0;                                                                     \
      u_kw = adv_ ? (wrap_ ? (roww_ ? 0 : k1_) : u_kw) : u_kw;                                              \
    }                                                                                                       \
  }

  // fragment reads: lane l -> row l & 15 of a 16-row MFMA tile, k-granule l >> 4 of the sub-step
  const int rdoff = (lane & 15) * 64 + (((lane >> 4) ^ (((lane >> 3) & 1) << 1)) << 4);
  const unsigned char* rdA = smem + (wm * WM) * 64 + rdoff;
  const unsigned char* rdB = smem + STAGE_A + (wn * WN) * 64 + rdoff;
  gran_t fa0[MT], fb0[NT], fa1[MT], fb1[NT];
#define RING_READ(fa_, fb_, sl_)                                                                            \
  {                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) fa_[i] = *reinterpret_cast<const gran_t*>(rdA + (sl_) * STAGE + i * 1024); \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) fb_[j] = *reinterpret_cast<const gran_t*>(rdB + (sl_) * STAGE + j * 1024); \
  }
#define RING_MMA(fa_, fb_)                                                                                  \
  _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                            \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                        \
      if constexpr (ABLATE & 2) { asm volatile("" ::"v"(fa_[i]), "v"(fb_[j])); }                            \
      else acc[i][j] = mma_granule<T>(fa_[i], fb_[j], acc[i][j]);                                           \
    }
// One sub-step: [the DMA of sub-step j + 1 has landed in every wave; every wave has read sub-step j] -> barrier -> issue
// sub-step j + 4 into the slot sub-step j occupied -> read the fragments of j + 1 -> multiply the fragments of j.
#define RING_SUBSTEP(c_, fcur_a, fcur_b, fnxt_a, fnxt_b)                                                    \
  {                                                                                                         \
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                \
    RING_ISSUE(c_)                                                                                          \
    RING_READ(fnxt_a, fnxt_b, ((c_) + 1) & 3)                                                               \
    RING_MMA(fcur_a, fcur_b)                                                                                \
  }

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bias_v[NT];
  conv_load_bias<WN>(p, n0, wn, lane, bias_v);

  RING_ISSUE(0) RING_ISSUE(1) RING_ISSUE(2) RING_ISSUE(3)
  asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
  RING_READ(fa0, fb0, 0)
  const int ngroups = NS >> 2;
  for (int gq = 0; gq < ngroups; ++gq) {
    RING_SUBSTEP(0, fa0, fb0, fa1, fb1)
    RING_SUBSTEP(1, fa1, fb1, fa0, fb0)
    RING_SUBSTEP(2, fa0, fb0, fa1, fb1)
    RING_SUBSTEP(3, fa1, fb1, fa0, fb0)
  }
  if (NS & 2) {      // Kpad is a multiple of 64: an odd number of K steps leaves two sub-steps
    RING_SUBSTEP(0, fa0, fb0, fa1, fb1)
    RING_SUBSTEP(1, fa1, fb1, fa0, fb0)
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // zero-page sub-steps staged past K still land in LDS; the strips alias the ring
#undef RING_ISSUE
#undef RING_READ
#undef RING_MMA
#undef RING_SUBSTEP

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
  constexpr int smem_bytes = 4 * (256 + 256) * 64;
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
#ifdef CFT_PROBES
    case 1: return launch_ring_t<T, 1>(p, stream);
    case 2: return launch_ring_t<T, 2>(p, stream);
    case 16: return launch_ring_t<T, 16>(p, stream);
#endif
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
