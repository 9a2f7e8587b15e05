// The wide layers' implicit-GEMM kernel with a HAND-SCHEDULED K loop (round 6): 256 x 256 tile, EIGHT waves (2 x 4) of 128 x 64,
// the whole main loop ONE inline-asm statement written by tools/gen_conv_asm.py (conv_gemm_asm.inc; register map, schedule and
// measurements in that file's header).  Same GEMM as conv_gemm_kernel<T, 256, 256, 4, 4, GLDS, 0, UNIK> (conv_gemm.hip) - the
// reference's Conv.forward / nn.Linear (models/common.py:36-50, :511, :532-538) - with the same fetches, the same LDS image, the
// same fragment reads and the same k order per accumulator: results are bit-identical to that kernel
// (tests/test_gpu_ops.py::test_asm_gemm_kernel_is_bit_identical).  What differs is WHEN things are issued:
//   * fragments are double-buffered in registers (2 x 12 ds_read_b128 per wave and K step, each read one half step ahead of its MFMAs);
//   * a staging buffer is free once every wave has READ it (half a step before its second half is multiplied), so the requests of
//     step t + 2 go out in the second half of step t, one request per four MFMAs, the two waves of a SIMD two MFMAs apart;
//   * one s_barrier per K step, placed where every wave has 32 MFMAs queued behind it;
//   * the K walk (tap / channel-chunk order, byte offsets, tap validity bit) is a TABLE in the kernel-argument segment, read with
//     s_load_dwordx4 one step ahead: the loop has no address arithmetic beyond three VALU operations per masked request.
// hipcc does not keep any of this when it is written in HIP (profiles/r04_gemm_experiments.md, csrc/probes/conv_ring.hip); in asm
// the loop runs at 2 140-2 200 cycles per 256 x 256 x 64 step against the 2 048 the matrix pipe needs (profiles/r06_kloop_microbench.md).
#include "conv_common.h"
#include <utility>
#include "conv_gemm_asm.inc"

#ifndef CONV_ASM_STRIPS
#define CONV_ASM_STRIPS 2       // strips in flight per wave in the epilogue
#endif
constexpr int ASM_MAXE = 236;                 // table entries: K steps (rounded up to even) + 3
struct ConvAsmParams {
  ConvParams p;
  int npairs;                                 // K steps / 2 (an odd count is rounded up: the extra step stages zeros)
  int masked;
  uint32_t table[ASM_MAXE][4];                // per K step: {A byte offset of (tap, chunk) in the pixel neighbourhood, B byte offset in a weight row, 1 << tap, 0}
};
static_assert(sizeof(ConvAsmParams) <= 4096, "kernel-argument segment");

#define CONV_ASM_CLOBBERS                                                                                                  \
  "memory", "scc", "vcc", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99",                                          \
  "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
  "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63", \
  "a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95", \
  "a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127", \
  "v27","v28","v29","v30","v31",                                                                                           \
  "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
  "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
  "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"

#define CONV_ASM_STMT(text_)                                                                                               \
  asm volatile(text_                                                                                                       \
               : [toff] "+s"(toff), [cnt] "+s"(cnt), [m0s] "=&s"(m0s)                                                      \
               : [voa0] "v"(voa[0]), [voa1] "v"(voa[1]), [voa2] "v"(voa[2]), [voa3] "v"(voa[3]),                           \
                 [vob0] "v"(vob[0]), [vob1] "v"(vob[1]), [vob2] "v"(vob[2]), [vob3] "v"(vob[3]),                           \
                 [am0] "v"(am[0]), [am1] "v"(am[1]), [am2] "v"(am[2]), [am3] "v"(am[3]), [voob] "v"(voob),                 \
                 [ra0] "v"(ra[0]), [ra1] "v"(ra[1]), [rb0] "v"(rb[0]), [rb1] "v"(rb[1]),                                   \
                 [srda] "s"(srdA), [srdb] "s"(srdB), [wb] "s"(wb), [tab] "s"(tab)                                          \
               : CONV_ASM_CLOBBERS)

template <int... Is, class F>
__device__ __forceinline__ void asm_static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void asm_static_for(F&& f) { asm_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// the asm text of a (type, staging form, tile) - wave group 0 runs the MT0-row text, group 1 the MT1-row text (tools/gen_conv_asm.py TILES)
#define CONV_ASM_RUN(TN_, MN_, M0_, M1_, AP_)                                                                              \
  if (wm == 0) { CONV_ASM_STMT(CONV_ASM_LOOP_##TN_##_##MN_##_M##M0_##_P##AP_##_G0); }                                      \
  else { CONV_ASM_STMT(CONV_ASM_LOOP_##TN_##_##MN_##_M##M1_##_P##AP_##_G1); }
#define CONV_ASM_TILES(TN_, MN_)                                                                                           \
  if constexpr (MT0 == 8 && MT1 == 8) { CONV_ASM_RUN(TN_, MN_, 8, 8, 4) }                                                  \
  else if constexpr (MT0 == 7 && MT1 == 7) { CONV_ASM_RUN(TN_, MN_, 7, 7, 4) }                                             \
  else if constexpr (MT0 == 7 && MT1 == 6) { CONV_ASM_RUN(TN_, MN_, 7, 6, 4) }                                             \
  else if constexpr (MT0 == 6 && MT1 == 6) { CONV_ASM_RUN(TN_, MN_, 6, 6, 3) }                                             \
  else { static_assert(MT0 == 4 && MT1 == 4, "tile not generated"); CONV_ASM_RUN(TN_, MN_, 4, 4, 2) }

// The epilogue of conv_common.h (conv_epilogue_impl: wave-private 16-row fp32 strips in LDS, + bias, activation, + residual, one rounding,
// coalesced 16-byte stores) for a 128 x 64 wave tile whose accumulators live in a[0:127]: strip i reads its four tiles right before use.
template <typename TH, int ACT, bool OUT_F32, int MT>
__device__ __forceinline__ void conv_epilogue_agpr(const ConvParams& p, unsigned char* smem, int m0, int n0, int row0, int wn, int wave, int lane, const float (&bias_v)[4]) {
  constexpr int WN = 64, NT = 4;
  const int lrow = lane & 15, lgrp = lane >> 4;
  constexpr int SLD = WN + 4;
  float* stage = reinterpret_cast<float*>(smem) + wave * (16 * SLD);
  constexpr int SG = CONV_ASM_STRIPS;                      // strips per group (each with its own 16-row stage per wave: SG x 34 KiB of the dead staging buffers)
  constexpr int VPRB = WN / 8, VPL = (16 * VPRB + 63) / 64, RDEPTH = SG;
  gran_t rpre[OUT_F32 ? 1 : RDEPTH][OUT_F32 ? 1 : VPL];
  const bool res_pre = !OUT_F32 && p.res != nullptr && !p.res_f32;   // uniform
  auto res_fetch = [&](int strip) {
#pragma unroll
    for (int v_ = 0; v_ < VPL; ++v_) {
      const int it_ = lane + v_ * 64;
      const int row_ = it_ / VPRB, col_ = (it_ - row_ * VPRB) * 8;
      const int m_ = m0 + row0 + strip * 16 + row_, n_ = n0 + wn * WN + col_;
      gran_t t_ = {0u, 0u, 0u, 0u};
      if (m_ < p.M && n_ < p.N) t_ = *reinterpret_cast<const gran_t*>(p.res + ((long)m_ * p.ldr + p.roff + n_) * 2);
      rpre[strip % RDEPTH][v_] = t_;
    }
  };
  if constexpr (!OUT_F32) {
    if (res_pre) {
#pragma unroll
      for (int s_ = 0; s_ < SG; ++s_)
        if (s_ < MT) res_fetch(s_);
    }
  }
  // strips go through LDS in PAIRS (two 16-row stages per wave): with eight waves per workgroup a wave walks eight strips, and one strip at a
  // time leaves the LDS round trip and the store latency of every strip exposed (round 4: 11.8 us against 8.5 for the 16-wave kernel)
  float* stage2[SG];
#pragma unroll
  for (int s_ = 0; s_ < SG; ++s_) stage2[s_] = stage + s_ * 8 * (16 * SLD);
  asm_static_for<(MT + SG - 1) / SG>([&](auto pc) {
    constexpr int i0 = SG * decltype(pc)::value;
    constexpr int NS = (i0 + SG <= MT) ? SG : MT - i0;     // (the last group may be short)
    asm_static_for<NS>([&](auto sc) {
      constexpr int s_ = decltype(sc)::value;
      asm_static_for<NT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const f32x4_t t = agpr_tile<(i0 + s_) * NT + j>();
#pragma unroll
        for (int e = 0; e < 4; ++e) stage2[s_][(lgrp * 4 + e) * SLD + j * 16 + lrow] = apply_act<ACT>(t[e] + bias_v[j]);
      });
    });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nbase = n0 + wn * WN;
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      const int i = i0 + s_;
      const float* st = stage2[s_];
      const int mbase = m0 + row0 + i * 16;
      if constexpr (OUT_F32) {
        constexpr int VPR = WN / 4;
#pragma unroll
        for (int vi = 0; vi < 4; ++vi) {
          const int it = lane + vi * 64;
          const int row = it / VPR, col = (it - row * VPR) * 4;
          const int m = mbase + row, n = nbase + col;
          if (m < p.M && n < p.N) {
            const f32x4_t sv = *reinterpret_cast<const f32x4_t*>(st + row * SLD + col);
            float v[4] = {sv[0], sv[1], sv[2], sv[3]};
            if (p.res != nullptr) {
              const long ro = (long)m * p.ldr + p.roff + n;
              if (p.res_f32) {
                const float4 rr = *reinterpret_cast<const float4*>(p.res + ro * 4);
                v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
              } else {
                const uint2 rr = *reinterpret_cast<const uint2*>(p.res + ro * 2);
                float r0, r1, r2, r3;
                Elem<TH>::unpack2(rr.x, r0, r1);
                Elem<TH>::unpack2(rr.y, r2, r3);
                v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3;
              }
            }
            *reinterpret_cast<f32x4_t*>(p.y + ((long)m * p.ldy + p.yoff + n) * 4) = f32x4_t{v[0], v[1], v[2], v[3]};
          }
        }
      } else {
        constexpr int VPR = WN / 8;
#pragma unroll
        for (int vi = 0; vi < VPL; ++vi) {
          const int it = lane + vi * 64;
          const int row = it / VPR, col = (it - row * VPR) * 8;
          const int m = mbase + row, n = nbase + col;
          if (m < p.M && n < p.N) {
            const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(st + row * SLD + col);
            const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(st + row * SLD + col + 4);
            float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
            if (res_pre) {
              float rf[8];
              Elem<TH>::unpack(rpre[(i0 + s_) % RDEPTH][vi], rf);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += rf[e];
            } else if (p.res != nullptr) {   // fp32 residual stream of the CFT block
              const long ro = (long)m * p.ldr + p.roff + n;
              const float4 r0v = *reinterpret_cast<const float4*>(p.res + ro * 4);
              const float4 r1v = *reinterpret_cast<const float4*>(p.res + ro * 4 + 16);
              v[0] += r0v.x; v[1] += r0v.y; v[2] += r0v.z; v[3] += r0v.w;
              v[4] += r1v.x; v[5] += r1v.y; v[6] += r1v.z; v[7] += r1v.w;
            }
            *reinterpret_cast<gran_t*>(p.y + ((long)m * p.ldy + p.yoff + n) * 2) = Elem<TH>::pack(v);
          }
        }
      }
    }
    if constexpr (!OUT_F32) {
      if (res_pre) {
#pragma unroll
        for (int s_ = 0; s_ < SG; ++s_)
          if (i0 + SG + s_ < MT) res_fetch(i0 + SG + s_);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  });
}

// Tile: 16 (MT0 + MT1) rows x 256 columns; wave group 0 (waves 0-3) owns the first MT0 16-row m-tiles, group 1 (waves 4-7) the next MT1 - a SIMD holds one
// wave of each, so MT0 + MT1 MFMA rows per SIMD whatever the split.  The height is chosen per launch (conv_asm_choose) so that the tile count fills whole
// rounds of the 256 CUs: 25 600 rows x 512 columns are 200 tiles of 256 rows (78 % of a round) or 230 of 224; 8 192 x 1 024: 128 tiles of 256 or 256 of 128.
template <typename T, bool MASKED, int MT0, int MT1>
__global__ void __launch_bounds__(512) conv_gemm_asm_kernel(const ConvAsmParams ap) {
  static_assert(sizeof(T) == 2, "16-bit operand types only");
  constexpr int BM = 16 * (MT0 + MT1), BN = 256, GE = 8, ES = 2, AP = (BM + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ConvParams& p = ap.p;

  const int nb = gridDim.x, bid = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, slot = bid >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;      // consecutive tiles on one XCD (as conv_gemm_kernel)
  const int tm = logical / p.tilesN, tn = logical - tm * p.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int rs = tid >> 3;                                             // staging row of this thread inside a 64-row pass
  const int g = (tid & 7) ^ (rs & 7);                                  // k-granule it fetches (source-side swizzle: slot ^ (row & 7))

  // Staging by buffer_load_dwordx4 ... lds: address = SRD base + per-thread voffset (constant) + SGPR offset from the K-walk table.  A masked
  // granule (tap outside the image, row beyond M or N) is fetched at the out-of-range voffset 2^31: the load returns - the DMA writes - zeros.
  // Only voffset is range-checked, so the input SRD starts `abias` bytes BELOW the tensor (the most negative tap of a border pixel) and every
  // voffset carries + abias.
  constexpr uint32_t OOB = 0x80000000u;
  const long abias = ((long)p.pad * p.W + p.pad) * p.ldx * ES;      // the most negative tap of a border pixel: pad rows up, pad pixels left
  uint32_t voa[4], vob[4], am[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + rs + i * 64;
    int a_off = 0;
    uint32_t mk = 0;
    if (i < AP && rs + i * 64 < BM && m < p.M) {
      const int t = fast_div(m, p.wo_mul, p.wo_sh);
      const int wo = m - t * p.Wo;
      const int b = fast_div(t, p.ho_mul, p.ho_sh);
      const int ho = t - b * p.Ho;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      a_off = ((b * p.H + hi0) * p.W + wi0) * p.ldx + p.xoff;
      uint32_t wbits = 0;
      for (int kw = 0; kw < p.KS; ++kw) wbits |= ((unsigned)(wi0 + kw) < (unsigned)p.W ? 1u : 0u) << kw;
      for (int kh = 0; kh < p.KS; ++kh)
        if ((unsigned)(hi0 + kh) < (unsigned)p.H) mk |= wbits << (kh * p.KS);
    }
    am[i] = mk;
    voa[i] = (uint32_t)(((long)a_off + g * GE) * ES + abias);
    if (!MASKED && mk == 0) voa[i] = OOB;                              // unmasked form (pointwise layers): only rows beyond M are masked, for good
    const int n = n0 + rs + i * 64;
    vob[i] = (n < p.N) ? (uint32_t)(((long)n * p.Kpad + g * GE) * ES) : OOB;
  }
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.x) - abias, 0, (int)(p.x_bytes + abias), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  // fragment reads: lane l -> row l & 15 of a 16-row MFMA tile, k-granule (half * 4 + (l >> 4)) of the K step, in slot granule ^ (row & 7)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const int lrow = lane & 15, lgrp = lane >> 4;
  // (LDS map: A images of even / odd steps at 0 / 32 KiB, B images at 64 / 96 KiB: the step parity goes into the instruction's offset field)
  uint32_t ra[2], rb[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t rd = (uint32_t)(lrow * 128 + (((h * 4 + lgrp) ^ (lrow & 7)) << 4));
    ra[h] = lds0 + (uint32_t)(wm * MT0 * 16 * 128) + rd;
    rb[h] = lds0 + 65536u + wn * 8192u + rd;
  }
  float bias_v[4];
  conv_load_bias<64>(p, n0, wn, lane, bias_v);

  const uint32_t wb = lds0 + (uint32_t)wave * 1024u;
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long tab = (unsigned long long)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ConvAsmParams, table);
#else
  const unsigned long long tab = 0;
#endif
  uint32_t toff = 0, cnt = (uint32_t)ap.npairs, m0s;
  const uint32_t voob = OOB;
  if constexpr (__is_same(T, f16_t)) { if constexpr (MASKED) { CONV_ASM_TILES(F16, MASK) } else { CONV_ASM_TILES(F16, NOMASK) } }
  else { if constexpr (MASKED) { CONV_ASM_TILES(BF16, MASK) } else { CONV_ASM_TILES(BF16, NOMASK) } }
  // (the asm block ends with vmcnt(0) lgkmcnt(0): this wave's requests have landed, its reads returned; the strips alias the staging buffers)
  __builtin_amdgcn_s_barrier();
  // The epilogue's launch parameters are re-read from the kernel-argument segment HERE (the pointer is laundered through an empty asm so
  // that the loads cannot be hoisted): held in SGPRs across the loop they would not fit beside the loop's own scalars.
#if defined(__HIP_DEVICE_COMPILE__)
  const __attribute__((address_space(4))) ConvParams* kp = (const __attribute__((address_space(4))) ConvParams*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  const ConvParams pe = *kp;
#else
  const ConvParams& pe = p;
#endif

#define CONV_ASM_EPI(MT_, ROW0_)                                                                                            \
  if (pe.out_f32) {                                                                                                        \
    if (pe.act == CFT_ACT_SILU) conv_epilogue_agpr<T, CFT_ACT_SILU, true, MT_>(pe, smem, m0, n0, ROW0_, wn, wave, lane, bias_v);       \
    else if (pe.act == CFT_ACT_GELU) conv_epilogue_agpr<T, CFT_ACT_GELU, true, MT_>(pe, smem, m0, n0, ROW0_, wn, wave, lane, bias_v);  \
    else conv_epilogue_agpr<T, CFT_ACT_NONE, true, MT_>(pe, smem, m0, n0, ROW0_, wn, wave, lane, bias_v);                  \
  } else {                                                                                                                 \
    if (pe.act == CFT_ACT_SILU) conv_epilogue_agpr<T, CFT_ACT_SILU, false, MT_>(pe, smem, m0, n0, ROW0_, wn, wave, lane, bias_v);      \
    else if (pe.act == CFT_ACT_GELU) conv_epilogue_agpr<T, CFT_ACT_GELU, false, MT_>(pe, smem, m0, n0, ROW0_, wn, wave, lane, bias_v); \
    else conv_epilogue_agpr<T, CFT_ACT_NONE, false, MT_>(pe, smem, m0, n0, ROW0_, wn, wave, lane, bias_v);                 \
  }
  if constexpr (MT0 == MT1) { CONV_ASM_EPI(MT0, wm * MT0 * 16) }
  else if (wm == 0) { CONV_ASM_EPI(MT0, 0) }
  else { CONV_ASM_EPI(MT1, MT0 * 16) }
#undef CONV_ASM_EPI
}

// ------------------------------------------------------------------------------------ the chained pair on the asm K loop
// conv_gemm_kernel<..., CHAIN, CRES> (conv_gemm.hip) with the hand-scheduled first K loop: a 256-channel first layer (a Bottleneck's 3x3, reference
// models/common.py:99-109, or the stride-2 Conv in front of a C3) whose 256 x 256 tile - after bias + SiLU (+ the shortcut, added in fp32 before the one
// rounding, :108) - is written to LDS AS the A operand of a pointwise second GEMM (the next Bottleneck's cv1 / the C3's cv1|cv2, N2 <= 256): four
// 64-channel images of 256 rows x 128 B in the dead staging buffers, the second layer's weights streaming through one extra 32-KiB buffer and the images
// already consumed.  With a shortcut the finished images are also stored to y1 (the next shortcut).  Same values, same roundings, same k order as the two
// launches and as the 16-wave chained kernel: bit-identical (tests/test_gpu_ops.py: the chain tests).  Wave (wm, wn) owns rows 128 wm .., columns 64 wn .. of
// the first layer's tile: its accumulators ARE image wn.
#define CONV_ASM_CHAIN2_STMT(text_)                                                                                        \
  asm volatile(text_ : : [ca0] "v"(ca[0]), [ca1] "v"(ca[1]), [cb0] "v"(cb[0]), [cb1] "v"(cb[1]) : CONV_ASM_CLOBBERS)

template <typename T>
__global__ void __launch_bounds__(512) conv_gemm_asm_chain_kernel(const ConvAsmParams ap) {
  static_assert(sizeof(T) == 2, "16-bit operand types only");
  constexpr int BM = 256, GE = 8, ES = 2, MT0 = 8, MT1 = 8;
  constexpr bool MASKED = true;
  constexpr int IMG = 32768;                                     // one 64-channel image: 256 rows x 128 B
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ConvParams& p = ap.p;

  const int nb = gridDim.x, bid = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, slot = bid >> 3;
  const int tm = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;          // one N tile: the logical tile index is the M tile
  const int m0 = tm * BM, n0 = 0;

  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int rs0 = tid0 >> 3;
  const int g0 = (tid0 & 7) ^ (rs0 & 7);

  constexpr uint32_t OOB = 0x80000000u;
  const long abias = ((long)p.pad * p.W + p.pad) * p.ldx * ES;      // the most negative tap of a border pixel: pad rows up, pad pixels left
  uint32_t voa[4], vob[4], am[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + rs0 + i * 64;
    int a_off = 0;
    uint32_t mk = 0;
    if (m < p.M) {
      const int t = fast_div(m, p.wo_mul, p.wo_sh);
      const int wo = m - t * p.Wo;
      const int b = fast_div(t, p.ho_mul, p.ho_sh);
      const int ho = t - b * p.Ho;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      a_off = ((b * p.H + hi0) * p.W + wi0) * p.ldx + p.xoff;
      uint32_t wbits = 0;
      for (int kw = 0; kw < p.KS; ++kw) wbits |= ((unsigned)(wi0 + kw) < (unsigned)p.W ? 1u : 0u) << kw;
      for (int kh = 0; kh < p.KS; ++kh)
        if ((unsigned)(hi0 + kh) < (unsigned)p.H) mk |= wbits << (kh * p.KS);
    }
    am[i] = mk;
    voa[i] = (uint32_t)(((long)a_off + g0 * GE) * ES + abias);
    const int n = rs0 + i * 64;
    vob[i] = (n < p.N) ? (uint32_t)(((long)n * p.Kpad + g0 * GE) * ES) : OOB;
  }
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.x) - abias, 0, (int)(p.x_bytes + abias), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  const uint32_t ldsk = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const int lrow0 = lane0 & 15, lgrp0 = lane0 >> 4;
  uint32_t ra[2], rb[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t rd = (uint32_t)(lrow0 * 128 + (((h * 4 + lgrp0) ^ (lrow0 & 7)) << 4));
    ra[h] = ldsk + (uint32_t)(wm * MT0 * 16 * 128) + rd;
    rb[h] = ldsk + 65536u + wn * 8192u + rd;
  }
  const uint32_t wb = ldsk + (uint32_t)wave * 1024u;
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long tab = (unsigned long long)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ConvAsmParams, table);
#else
  const unsigned long long tab = 0;
#endif
  uint32_t toff = 0, cnt = (uint32_t)ap.npairs, m0s;
  const uint32_t voob = OOB;
  if constexpr (__is_same(T, f16_t)) { CONV_ASM_TILES(F16, MASK) } else { CONV_ASM_TILES(BF16, MASK) }
  __builtin_amdgcn_s_barrier();                          // every wave's requests have landed and its reads returned: all four staging buffers are free

#if defined(__HIP_DEVICE_COMPILE__)
  const __attribute__((address_space(4))) ConvParams* kp = (const __attribute__((address_space(4))) ConvParams*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  const ConvParams pe = *kp;
#else
  const ConvParams& pe = p;
#endif
  const bool with_res = pe.y1 != nullptr;                // uniform
  // (every per-lane value of the phases below is re-derived from the thread id HERE: computed before the K loop they would not fit beside its operands and spill)
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  const int lane = tid2 & 63, rs = tid2 >> 3, g = (tid2 & 7) ^ (rs & 7), lrow = lane & 15, lgrp = lane >> 4, tid = tid2;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  float bias_v[4];
  conv_load_bias<64>(pe, 0, wn, lane, bias_v);             // (the first layer's bias: an L2 hit; loaded here for the same reason)
  const unsigned char* zero_page = reinterpret_cast<const unsigned char*>(cft_zero_page);
  unsigned char* sX = smem + 4 * IMG;                    // the extra weight buffer behind the four images
  auto load_w2 = [&](int k2, unsigned char* dst) {      // second layer's weights [N2][256], K step k2 -> dst, in the staging pattern of the K loop
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = rs + i * 64;
      const unsigned char* src = (n < pe.N2) ? pe.w2 + ((long)n * pe.N + k2 * 64 + g * GE) * ES : zero_page;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dst + i * 8192 + wave * 1024), 16, 0, 0);
    }
  };
  if (with_res) {                                        // the shortcut tile -> the image area, in the image layout; rows beyond M read the zero page
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + rs + i * 64;
        const unsigned char* src = (m < pe.M) ? pe.res + ((long)m * pe.ldr + pe.roff + k2 * 64 + g * GE) * ES : zero_page;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(smem + k2 * IMG + i * 8192 + wave * 1024), 16, 0, 0);
      }
  }
  load_w2(0, sX);
  float bias2_v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = wn * 64 + j * 16 + lrow;
    bias2_v[j] = (pe.bias2 != nullptr && n < pe.N2) ? pe.bias2[n] : 0.0f;
  }
  // first layer's bias + SiLU (+ shortcut) + rounding on the accumulators -> image wn; lanes l / l ^ 1 hold neighbouring channels of the same four
  // pixels and swap half of their values by DPP so that each writes two packed channel PAIRS (even lane: pixels 0, 1; odd: 2, 3)
  const bool odd = lane & 1;
  unsigned char* img = smem + wn * IMG;
  if (with_res) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // the shortcut tile (and W2 step 0) landed, visible to every wave
  }
  asm_static_for<8>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    asm_static_for<4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const f32x4_t t = agpr_tile<i * 4 + j>();
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = apply_act<CFT_ACT_SILU>(t[e] + bias_v[j]);
      const int row = wm * 128 + i * 16 + lgrp * 4 + (odd ? 2 : 0);
      const int c = j * 16 + (lrow & 14);                // channel inside this wave's 64-channel image
      uint32_t* pa = reinterpret_cast<uint32_t*>(img + (c & 7) * 2 + row * 128 + (((c >> 3) ^ (row & 7)) << 4));
      uint32_t* pb = reinterpret_cast<uint32_t*>(img + (c & 7) * 2 + (row + 1) * 128 + (((c >> 3) ^ ((row + 1) & 7)) << 4));
      if (with_res) {
        const float ga = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(odd ? v[0] : v[2]), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        const float gb = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(odd ? v[1] : v[3]), 0xB1, 0xF, 0xF, true));
        float lo0 = odd ? ga : v[0], hi0 = odd ? v[2] : ga;      // pixel A: channels c, c + 1
        float lo1 = odd ? gb : v[1], hi1 = odd ? v[3] : gb;      // pixel B = A + 1
        float r0l, r0h, r1l, r1h;
        Elem<T>::unpack2(*pa, r0l, r0h);
        Elem<T>::unpack2(*pb, r1l, r1h);
        lo0 += r0l; hi0 += r0h; lo1 += r1l; hi1 += r1h;
        *pa = Elem<T>::pack2(lo0, hi0);
        *pb = Elem<T>::pack2(lo1, hi1);
      } else {
        const uint32_t r01 = Elem<T>::pack2(v[0], v[1]), r23 = Elem<T>::pack2(v[2], v[3]);
        const uint32_t got = (uint32_t)__builtin_amdgcn_mov_dpp((int)(odd ? r01 : r23), 0xB1, 0xF, 0xF, true);
        *pa = odd ? ((got & 0xffffu) | (r23 << 16)) : ((r01 & 0xffffu) | (got << 16));
        *pb = odd ? ((got >> 16) | (r23 & 0xffff0000u)) : ((r01 >> 16) | (got & 0xffff0000u));
      }
    });
  });
  // image k2 IS a quarter of the first layer's output tile: with a shortcut it is stored to y1 as 16-byte granules (thread (rs, slot) holds granule
  // slot ^ (row & 7) of its rows: eight consecutive threads write one full 128-byte line), issued right before the MFMA step that consumes the image; the
  // barrier behind that step drains the stores, so an image is never overwritten (by a later W2 step) before it is on its way to memory
  auto store_image = [&](int k2) {
    if (with_res) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rs + i * 64, m = m0 + row;
        const gran_t t_ = *reinterpret_cast<const gran_t*>(smem + k2 * IMG + row * 128 + ((tid & 7) << 4));
        if (m < pe.M) *reinterpret_cast<gran_t*>(pe.y1 + ((long)m * pe.ldy1 + pe.yoff1 + k2 * 64 + g * GE) * ES) = t_;
      }
    }
  };
  // second GEMM: four K steps = the four images; fragment bases of this wave: A rows 128 wm .. of image k2, B rows 64 wn .. of a weight buffer
  const uint32_t rd0 = (uint32_t)(lrow * 128 + ((lgrp ^ (lrow & 7)) << 4));
  const uint32_t abase = lds0 + (uint32_t)(wm * 16384) + rd0, bX = lds0 + 4u * IMG + wn * 8192u + rd0, bI = lds0 + wn * 8192u + rd0;
  uint32_t ca[2], cb[2];
#define CONV_ASM_CHAIN2_RUN(aimg_, bbase_)                                                                                 \
  ca[0] = abase + (aimg_) * IMG; ca[1] = ca[0] ^ 64u; cb[0] = (bbase_); cb[1] = cb[0] ^ 64u;                               \
  if constexpr (__is_same(T, f16_t)) { CONV_ASM_CHAIN2_STMT(CONV_ASM_CHAIN2_F16); } else { CONV_ASM_CHAIN2_STMT(CONV_ASM_CHAIN2_BF16); }
  asm volatile(CONV_ASM_ZERO_ACC ::: CONV_ASM_CLOBBERS);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // images written, W2 step 0 in X
  store_image(0);
  CONV_ASM_CHAIN2_RUN(0, bX)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // X and image 0 are free (image 0 is in memory)
  load_w2(1, sX);
  load_w2(2, smem);
  store_image(1);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // both landed
  CONV_ASM_CHAIN2_RUN(1, bX)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // image 1 is free
  load_w2(3, smem + IMG);                                // lands under step 2
  store_image(2);
  CONV_ASM_CHAIN2_RUN(2, bI)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  store_image(3);
  CONV_ASM_CHAIN2_RUN(3, bI + IMG)
#undef CONV_ASM_CHAIN2_RUN
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // every wave is past its last image read: the strips may overwrite the images
  ConvParams p2 = pe;
  p2.N = pe.N2;
  p2.res = nullptr;
  const int row0 = wm * 128;
  int tid3 = threadIdx.x;                                // (the lane index once more from the thread id: kept across the four blocks above it spills)
  asm volatile("" : "+v"(tid3));
  const int lane3 = tid3 & 63;
  if (p2.act == CFT_ACT_SILU) conv_epilogue_agpr<T, CFT_ACT_SILU, false, 8>(p2, smem, m0, 0, row0, wn, wave, lane3, bias2_v);
  else if (p2.act == CFT_ACT_GELU) conv_epilogue_agpr<T, CFT_ACT_GELU, false, 8>(p2, smem, m0, 0, row0, wn, wave, lane3, bias2_v);
  else conv_epilogue_agpr<T, CFT_ACT_NONE, false, 8>(p2, smem, m0, 0, row0, wn, wave, lane3, bias2_v);
}

// ------------------------------------------------------------------------------------ host
// Eligibility: 16-bit operands, the uniform K walk (Cin a multiple of the 64-wide K step, no K padding), the walk fits the table, the
// buffer-addressing extents (masked granules are fetched at voffset 2^31, so both buffers must end below it), no split-K.
bool conv_asm_ok(const ConvParams& p, int dtype) {
  if (dtype != CFT_BF16 && dtype != CFT_F16) return false;
  if (p.Cin % 64 != 0 || p.Kpad != p.K || p.KS > 5 || p.ksplit > 1) return false;
  const int nk = p.Kpad / 64;
  if (nk < 2 || ((nk + 1) & ~1) + 3 > ASM_MAXE) return false;
  return p.x_bytes + 2L * ((long)p.pad * p.W + p.pad) * p.ldx * 2 < (1L << 31) && p.w_bytes < (1L << 31);
}

template <typename T, bool MASKED, int MT0, int MT1>
static int launch_asm_t(const ConvAsmParams& ap, int grid, hipStream_t stream) {
  constexpr int smem_bytes = 2 * (256 + 256) * 128;
  cft_allow_lds<&conv_gemm_asm_kernel<T, MASKED, MT0, MT1>>(smem_bytes);
  hipLaunchKernelGGL((conv_gemm_asm_kernel<T, MASKED, MT0, MT1>), dim3(grid), dim3(512), smem_bytes, stream, ap);
  return cft_check_launch("conv_gemm_asm_kernel");
}

template <typename T, bool MASKED>
static int launch_asm_tile(const ConvAsmParams& ap, int tile, int grid, hipStream_t stream) {
  switch (tile) {
    case 0: return launch_asm_t<T, MASKED, 8, 8>(ap, grid, stream);
    case 1: return launch_asm_t<T, MASKED, 7, 7>(ap, grid, stream);
    case 2: return launch_asm_t<T, MASKED, 7, 6>(ap, grid, stream);
    case 3: return launch_asm_t<T, MASKED, 6, 6>(ap, grid, stream);
    default: return launch_asm_t<T, MASKED, 4, 4>(ap, grid, stream);
  }
}

// the K-walk table of a launch (and its step-pair count)
static void fill_asm_table(ConvAsmParams& ap, const ConvParams& p) {
  const int nk = p.Kpad / 64, nkp = (nk + 1) & ~1;
  ap.npairs = nkp / 2;
  // K order: conv_gemm_kernel's - tap-major (k = (kh, kw, ci): the channel chunks of a tap, then the next tap), or CHUNK-major for 3x3 layers
  // with Cin >= 256 (all nine taps of a 64-channel chunk, then the next chunk: a tap re-reads the pixels its neighbour just read)
  const bool chunk_major = p.KS == 3 && p.Cin >= 256 && p.K == 9 * p.Cin;
  const int chunks = p.Cin / 64, taps = p.KS * p.KS;
  for (int t = 0; t < nkp + 3; ++t) {
    uint32_t* e = ap.table[t];
    if (t < nk) {
      const int tap = chunk_major ? t % taps : t / chunks, chunk = chunk_major ? t / taps : t % chunks;
      const int kh = tap / p.KS, kw = tap - kh * p.KS;
      e[0] = (uint32_t)((((long)kh * p.W + kw) * p.ldx + chunk * 64) * 2);
      e[1] = (uint32_t)(((long)tap * p.Cin + chunk * 64) * 2);
      e[2] = 1u << tap;
    } else {            // beyond K: the odd-count padding step stages zeros (tap bit 0 -> every A granule out of range); later entries are requested, never used
      e[0] = 0; e[1] = 0; e[2] = 0;
    }
    e[3] = 0;
  }
}

static const int kAsmTileRows[CONV_ASM_NTILES] = {256, 224, 208, 192, 128};
int conv_asm_tile_rows(int tile) { return tile >= 0 && tile < CONV_ASM_NTILES ? kAsmTileRows[tile] : 0; }

// Tile height of a launch: the one whose tile count wastes least of the 256 CUs' rounds.  Model (us), from tools/gemm_bench.py and the K-loop micro-benchmark:
// a K step costs max(1.27 BM / 256 [matrix pipe at the clock this load holds], 0.3 + 0.3 BM / 256 [the L2 -> LDS request stream: BM + 256 rows of 128 B]),
// a tile 5 + 6 BM / 256 on top (prologue, epilogue); rounds = ceil(tiles / 256).  Ties go to the taller tile (fewer weight re-fetches).  -1: too few tiles
// for any height (the caller falls back to the 16-wave tiles).
int conv_asm_choose(const ConvParams& p) {
  const int nk = p.Kpad / 64;
  const long tilesN = (p.N + 255) / 256;
  int best = -1;
  double best_t = 0.0;
  for (int c = 0; c < CONV_ASM_NTILES; ++c) {
    const int bm = kAsmTileRows[c];
    const long tiles = ((long)p.M + bm - 1) / bm * tilesN;
    if (tiles < 160) continue;
    const long rounds = (tiles + 255) / 256;
    const double f = bm / 256.0;
    const double step = 1.27 * f > 0.3 + 0.3 * f ? 1.27 * f : 0.3 + 0.3 * f;
    const double t = rounds * (nk * step + 5.0 + 6.0 * f);
    if (best < 0 || t < best_t * 0.90) { best = c; best_t = t; }
    // Shorter tiles only where the 256-row tiling does not even fill ONE round of the chip (measured, profiles/r06_asm_kloop.md: with two forwards in
    // flight the idle CUs of a partial LAST round of a multi-round launch are taken by the other forward's kernels, and shorter tiles then only add
    // weight re-fetches and prologues: -1.5 % pairs/s when every launch picks its best isolated height; a launch that leaves CUs idle for its whole
    // duration - 200 or 128 tiles - is different)
    if (c == 0 && tiles > 256) break;
  }
  return best;
}

int conv_asm_launch(const ConvParams& p, int dtype, int tile, hipStream_t stream) {
  if (!conv_asm_ok(p, dtype) || tile < 0 || tile >= CONV_ASM_NTILES) { cft_set_error("conv_gemm_asm_kernel: layer not eligible"); return CFT_EINVAL; }
  ConvAsmParams ap;
  ap.p = p;
  const int bm = kAsmTileRows[tile];
  const int tilesM = (p.M + bm - 1) / bm;
  ap.p.tilesN = (p.N + 255) / 256;
  ap.p.ksplit = 1;
  fill_asm_table(ap, p);
  const int nk = p.Kpad / 64;
  const bool masked = p.KS > 1 || (nk & 1);
  ap.masked = masked;
  const int grid = tilesM * ap.p.tilesN;
  if (dtype == CFT_BF16) return masked ? launch_asm_tile<uint16_t, true>(ap, tile, grid, stream) : launch_asm_tile<uint16_t, false>(ap, tile, grid, stream);
  return masked ? launch_asm_tile<f16_t, true>(ap, tile, grid, stream) : launch_asm_tile<f16_t, false>(ap, tile, grid, stream);
}

// The chained pair (cft_conv2d_chain / cft_conv2d_chain_res with a 256-channel first layer) on the asm K loop: eligibility as conv_asm_ok + one N tile of 256
// channels + at least 12 K steps; p carries w2 / bias2 / N2 (/ res, y1) as dispatch_chain passes them.
bool conv_asm_chain_ok(const ConvParams& p, int dtype) {
  return conv_asm_ok(p, dtype) && p.N == 256 && p.N2 > 0 && p.N2 <= 256 && p.Kpad >= 24 * 64;   // (18 steps - 3x3 s2 128 -> 256 - measured slower: 352 vs 336 us)
}

template <typename T>
static int launch_asm_chain_t(const ConvAsmParams& ap, int grid, hipStream_t stream) {
  constexpr int smem_bytes = 5 * 32768;
  cft_allow_lds<&conv_gemm_asm_chain_kernel<T>>(smem_bytes);
  hipLaunchKernelGGL((conv_gemm_asm_chain_kernel<T>), dim3(grid), dim3(512), smem_bytes, stream, ap);
  return cft_check_launch("conv_gemm_asm_chain_kernel");
}

int conv_asm_chain_launch(const ConvParams& p, int dtype, hipStream_t stream) {
  if (!conv_asm_chain_ok(p, dtype)) { cft_set_error("conv_gemm_asm_chain_kernel: layer pair not eligible"); return CFT_EINVAL; }
  ConvAsmParams ap;
  ap.p = p;
  ap.p.tilesN = 1;
  ap.p.ksplit = 1;
  fill_asm_table(ap, p);
  ap.masked = 1;
  const int grid = (p.M + 255) / 256;
  if (dtype == CFT_BF16) return launch_asm_chain_t<uint16_t>(ap, grid, stream);
  return launch_asm_chain_t<f16_t>(ap, grid, stream);
}
