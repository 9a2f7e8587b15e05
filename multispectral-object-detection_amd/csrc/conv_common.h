// Shared pieces of the implicit-GEMM kernels (conv_gemm.hip, conv_ring.hip): launch parameters, the zero page / zero region that
// masked granules fetch from (one copy per translation unit: device code is linked per file), the fused epilogue.
#pragma once
#include "cft_common.h"
#include <stdlib.h>

struct ConvParams {
  const unsigned char* x;
  const unsigned char* w;
  const float* bias;
  const unsigned char* res;
  unsigned char* y;
  int H, W, Cin, ldx, xoff;
  int Ho, Wo, N, Kpad, K;
  int ldy, yoff, ldr, roff;
  int KS, stride, pad;
  int act, out_f32, res_f32;
  int M, tilesN;
  const unsigned char* w2;   // chained pointwise layer (conv_gemm_kernel CHAIN): packed weights [N2][N], bias, width; y/ldy/yoff/act are ITS output
  const float* bias2;
  int N2;
  unsigned char* y1;         // CHAIN with a residual (res / ldr / roff): the FIRST layer's output (after + res) is also stored here (ldy1 / yoff1)
  int ldy1, yoff1;
  int ksplit, ksteps;        // split-K (pointwise layers on the uniform K walk): workgroup (tile, s) covers K steps [s * ksteps, (s + 1) * ksteps) and
                             // writes its fp32 partial sums to y + s * M * ldy (bias in split 0 only); ksplit <= 1: the whole K loop
  long x_bytes, w_bytes;   // extent of the input tensor (B*H*W*ldx elements) and of the packed weights (N*Kpad), in bytes
  uint32_t wo_mul, wo_sh, ho_mul, ho_sh;   // exact n / Wo and n / Ho for n < 2^31 as umulhi(n, mul) >> sh (mul == 0: divisor 1)
};

// floor(n / d) for 0 <= n < 2^31 with a host-computed (mul, sh): Granlund-Montgomery round-up method.
__device__ __forceinline__ int fast_div(int n, uint32_t mul, uint32_t sh) {
  return mul ? (int)(__umulhi((uint32_t)n, mul) >> sh) : n;
}

static __device__ __attribute__((aligned(16))) uint32_t cft_zero_page[4] = {0u, 0u, 0u, 0u};
// UNIK path: weight rows beyond N point at this zero REGION and still advance along K (no per-step select): 64 KiB >= 2 * Kpad + 128
constexpr int CFT_ZERO_REGION_BYTES = 65536;
static __device__ __attribute__((aligned(128))) uint32_t cft_zero_region[CFT_ZERO_REGION_BYTES / 4];   // zero-initialised

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// Epilogue (wave-private, no workgroup barriers - the LDS operations of one wave are ordered):
// acc (+bias, activation) -> 16-row fp32 LDS strip -> rows re-read as 16-B vectors -> (+residual)
// -> one rounding -> coalesced 16-B stores.  The caller guarantees (barrier) that no wave still
// reads the staging buffers that the strips alias.
template <typename TH, int WM, int WN, int ACT, bool OUT_F32>   // TH: the 16-bit storage type (bf16 bits or half)
__device__ __forceinline__ void conv_epilogue_impl(const ConvParams& p, f32x4_t (&acc)[WM / 16][WN / 16], unsigned char* smem,
                                                   int m0, int n0, int wm, int wn, int wave, int lane, const float (&bias_v)[WN / 16]) {
  constexpr int MT = WM / 16, NT = WN / 16;
  const int lrow = lane & 15, lgrp = lane >> 4;
  constexpr int SLD = WN + 4;  // fp32 strip leading dimension (+4: the four 4-row lane groups hit different banks)
  float* stage = reinterpret_cast<float*>(smem) + wave * (16 * SLD);
  // bf16 residual (Bottleneck shortcut): every strip's residual vectors are requested up front, so their HBM
  // latency runs under the activation / LDS work instead of once per strip.  (The lane that reads an element is
  // the lane that later stores it, so an in-place residual stays correct.)
  constexpr int VPRB = WN / 8;                       // 16-B bf16 vectors per strip row
  constexpr int VPL = (16 * VPRB + 63) / 64;         // vectors per lane per strip
  constexpr int RDEPTH = (MT * VPL <= 6) ? MT : 2;   // strips of residual in flight (register budget: 16-wave tiles keep 2)
  gran_t rpre[OUT_F32 ? 1 : RDEPTH][OUT_F32 ? 1 : VPL];
  const bool res_pre = !OUT_F32 && p.res != nullptr && !p.res_f32;   // uniform
#define CFT_RES_FETCH(strip_)                                                                          \
  _Pragma("unroll") for (int v_ = 0; v_ < VPL; ++v_) {                                                 \
    const int it_ = lane + v_ * 64;                                                                    \
    const int row_ = it_ / VPRB, col_ = (it_ - row_ * VPRB) * 8;                                       \
    const int m_ = m0 + wm * WM + (strip_) * 16 + row_, n_ = n0 + wn * WN + col_;                      \
    gran_t t_ = {0u, 0u, 0u, 0u};                                                                      \
    if (it_ < 16 * VPRB && m_ < p.M && n_ < p.N)                                                       \
      t_ = *reinterpret_cast<const gran_t*>(p.res + ((long)m_ * p.ldr + p.roff + n_) * 2);             \
    rpre[(strip_) % RDEPTH][v_] = t_;                                                                  \
  }
  if constexpr (!OUT_F32) {
    if (res_pre) {
#pragma unroll
      for (int i = 0; i < RDEPTH; ++i) CFT_RES_FETCH(i)
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        stage[(lgrp * 4 + e) * SLD + j * 16 + lrow] = apply_act<ACT>(acc[i][j][e] + bias_v[j]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int mbase = m0 + wm * WM + i * 16;
    const int nbase = n0 + wn * WN;
    if constexpr (OUT_F32) {
      constexpr int VPR = WN / 4;  // 16-B vectors per strip row
      for (int it = lane; it < 16 * VPR; it += 64) {
        const int row = it / VPR, col = (it - row * VPR) * 4;
        const int m = mbase + row, n = nbase + col;
        if (m < p.M && n < p.N) {
          const f32x4_t sv = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col);
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
        if (it < 16 * VPR && m < p.M && n < p.N) {
          const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col);
          const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col + 4);
          float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
          if (res_pre) {
            float rf[8];
            Elem<TH>::unpack(rpre[i % RDEPTH][vi], rf);
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
      if (res_pre && i + RDEPTH < MT) CFT_RES_FETCH(i + RDEPTH)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

#undef CFT_RES_FETCH

// This lane's bias values (output column j*16 + lrow of the wave tile).  Loaded BEFORE the K loop: at the epilogue the value
// is a register, not an exposed L2 round trip per workgroup (and an ordinary load result consumed next to LDS-DMA traffic
// makes hipcc drain vmcnt to 0 at that point).
template <int WN>
__device__ __forceinline__ void conv_load_bias(const ConvParams& p, int n0, int wn, int lane, float (&bias_v)[WN / 16]) {
#pragma unroll
  for (int j = 0; j < WN / 16; ++j) {
    const int n = n0 + wn * WN + j * 16 + (lane & 15);
    bias_v[j] = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.0f;
  }
}

// Uniform dispatch to the specialised epilogues (one activation / output type per launch).
template <typename TH, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x4_t (&acc)[WM / 16][WN / 16], unsigned char* smem,
                                              int m0, int n0, int wm, int wn, int wave, int lane, const float (&bias_v)[WN / 16]) {
  if (p.out_f32) {
    if (p.act == CFT_ACT_SILU) conv_epilogue_impl<TH, WM, WN, CFT_ACT_SILU, true>(p, acc, smem, m0, n0, wm, wn, wave, lane, bias_v);
    else if (p.act == CFT_ACT_GELU) conv_epilogue_impl<TH, WM, WN, CFT_ACT_GELU, true>(p, acc, smem, m0, n0, wm, wn, wave, lane, bias_v);
    else conv_epilogue_impl<TH, WM, WN, CFT_ACT_NONE, true>(p, acc, smem, m0, n0, wm, wn, wave, lane, bias_v);
  } else {
    if (p.act == CFT_ACT_SILU) conv_epilogue_impl<TH, WM, WN, CFT_ACT_SILU, false>(p, acc, smem, m0, n0, wm, wn, wave, lane, bias_v);
    else if (p.act == CFT_ACT_GELU) conv_epilogue_impl<TH, WM, WN, CFT_ACT_GELU, false>(p, acc, smem, m0, n0, wm, wn, wave, lane, bias_v);
    else conv_epilogue_impl<TH, WM, WN, CFT_ACT_NONE, false>(p, acc, smem, m0, n0, wm, wn, wave, lane, bias_v);
  }
}


// conv_gemm_asm.hip: the 8-wave 256 x 256 kernel with the hand-scheduled K loop (16-bit operands, uniform K walk): eligibility and launch
constexpr int CONV_ASM_NTILES = 5;                  // tile heights 256, 224, 208, 192, 128 rows (x 256 columns)
bool conv_asm_ok(const ConvParams& p, int dtype);
int conv_asm_choose(const ConvParams& p);            // tile index whose tile count fills the CUs' rounds best, -1: too few tiles for any
int conv_asm_tile_rows(int tile);
int conv_asm_launch(const ConvParams& p, int dtype, int tile, hipStream_t stream);
bool conv_asm_chain_ok(const ConvParams& p, int dtype);   // the chained pair with a 256-channel first layer (p.w2 / bias2 / N2 [/ res / y1] set)
int conv_asm_chain_launch(const ConvParams& p, int dtype, hipStream_t stream);

// conv_ring.hip: launch the ring kernel (dtype CFT_BF16 / CFT_F16; ablate: timing probes of -DCFT_PROBES builds, 0 otherwise)
int conv_ring_launch(const ConvParams& p, int dtype, int ablate, hipStream_t stream);
