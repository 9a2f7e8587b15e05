// Implicit-GEMM convolution / linear kernel for gfx950 (MI355X), fused bias + activation +
// residual + channel-slice epilogue.  This one kernel family carries 99.6 % of the FLOPs of the
// two-stream YOLOv5 + CFT forward.
//
// GEMM view:  C[M,N] = A[M,K] * Wt[N,K]^T
//   M = B*Ho*Wo output pixels (NHWC, so a row of C is one pixel's channel vector),
//   N = output channels, K = k*k*Cin with (kh,kw,ci) flattened, ci fastest.
//   A is never materialised: row m / k-granule (tap, ci) is gathered from the NHWC input
//   (a 16-byte granule never straddles a tap because Cin % granule == 0); out-of-image taps
//   and the K tail read as zero.
//
// Tiling (per 256-thread workgroup = 4 wave64 as 2x2):
//   BM x BN output tile, K step = 128 bytes per row (64 bf16 / 32 f32) = 8 granules.
//   global -> registers (16-B loads, 8 consecutive lanes cover one 128-B row segment)
//          -> LDS (row-major 128-B rows, granule index XOR (row & 7): conflict-free
//             ds_read_b128 fragment reads)  -> MFMA.
//   bf16: v_mfma_f32_16x16x32_bf16 (lane holds 8 consecutive k of one row = one granule);
//   f32 : 4 x v_mfma_f32_16x16x4_f32 per granule (exact fp32 products, fp32 accumulate).
//   Both operands use the same (lane-group, element) -> k assignment, so the reduction is a
//   permutation of k and needs no knowledge of the instruction's internal k order.
//   Double-buffered LDS, next tile's global loads are issued before the MFMAs of the current
//   one; one barrier per K step.
// Epilogue: acc (+bias, activation) -> per-wave LDS strip (fp32) -> rows re-read as 16-B
//   vectors -> (+residual) -> one rounding -> coalesced 16-B global stores.
// Workgroup order is remapped so that consecutive logical tiles (which share the A rows or
// neighbouring image rows) run on the same XCD and hit the same 4 MiB L2.
#include "cft_common.h"

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
};

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == CFT_ACT_SILU) return v / (1.0f + __expf(-v));
  if (act == CFT_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
  return v;
}

template <typename T, int BM, int BN>
__global__ void __launch_bounds__(256) conv_gemm_kernel(const ConvParams p) {
  constexpr int GE = Elem<T>::GE;
  constexpr int BK = 8 * GE;
  constexpr int A_PER = BM / 32, B_PER = BN / 32;
  constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 16, NT = WN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int ES = (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * A_BYTES;

  // ---- XCD-aware tile assignment (bijective for any grid size) ----
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, slot = bid >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  const int tm = logical / p.tilesN, tn = logical - tm * p.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = tid & 7, r0 = tid >> 3;

  // ---- per-thread gather state: A_PER pixel rows, one k-granule column g ----
  int a_off[A_PER];
  uint32_t a_mask[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int m = m0 + r0 + i * 32;
    a_off[i] = 0;
    a_mask[i] = 0;
    if (m < p.M) {
      const int wo = m % p.Wo;
      const int t = m / p.Wo;
      const int ho = t % p.Ho;
      const int b = t / p.Ho;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      a_off[i] = ((b * p.H + hi0) * p.W + wi0) * p.ldx + p.xoff;
      uint32_t mk = 0;
      for (int kh = 0; kh < p.KS; ++kh)
        for (int kw = 0; kw < p.KS; ++kw)
          if ((unsigned)(hi0 + kh) < (unsigned)p.H && (unsigned)(wi0 + kw) < (unsigned)p.W) mk |= 1u << (kh * p.KS + kw);
      a_mask[i] = mk;
    }
  }
  int ci = g * GE, kh = 0, kw = 0, tap = 0;
  while (ci >= p.Cin) { ci -= p.Cin; ++tap; if (++kw == p.KS) { kw = 0; ++kh; } }

  gran_t ra[A_PER], rb[B_PER];
  const int swz = (g ^ (r0 & 7)) << 4;

// Gather the next K step of this thread's A/B granules into registers, then advance (tap, ci).
#define CFT_LOAD_TILE(kt_)                                                                             \
  {                                                                                                    \
    const int kglob = (kt_) * BK + g * GE;                                                             \
    const bool kin = kglob < p.K;                                                                      \
    const long tapoff = ((long)kh * p.W + kw) * p.ldx + ci;                                            \
    _Pragma("unroll") for (int i = 0; i < A_PER; ++i) {                                                \
      const bool v = kin && ((a_mask[i] >> tap) & 1u);                                                 \
      gran_t t_ = {0u, 0u, 0u, 0u};                                                                  \
      if (v) t_ = *reinterpret_cast<const gran_t*>(p.x + ((long)a_off[i] + tapoff) * ES);              \
      ra[i] = t_;                                                                                      \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < B_PER; ++i) {                                                \
      const int n = n0 + r0 + i * 32;                                                                  \
      gran_t t_ = {0u, 0u, 0u, 0u};                                                                  \
      if (n < p.N) t_ = *reinterpret_cast<const gran_t*>(p.w + ((long)n * p.Kpad + kglob) * ES);      \
      rb[i] = t_;                                                                                      \
    }                                                                                                  \
    ci += BK;                                                                                          \
    while (ci >= p.Cin) { ci -= p.Cin; ++tap; if (++kw == p.KS) { kw = 0; ++kh; } }                    \
  }
#define CFT_STORE_TILE(buf_)                                                                           \
  {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < A_PER; ++i)                                                  \
      *reinterpret_cast<gran_t*>(sA + (buf_) * A_BYTES + (r0 + i * 32) * 128 + swz) = ra[i];           \
    _Pragma("unroll") for (int i = 0; i < B_PER; ++i)                                                  \
      *reinterpret_cast<gran_t*>(sB + (buf_) * B_BYTES + (r0 + i * 32) * 128 + swz) = rb[i];           \
  }

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = p.Kpad / BK;
  CFT_LOAD_TILE(0)
  CFT_STORE_TILE(0)
  __syncthreads();
  const int lrow = lane & 15, lgrp = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) CFT_LOAD_TILE(kt + 1)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kg = ks * 4 + lgrp;
      gran_t af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * WM + i * 16 + lrow;
        af[i] = *reinterpret_cast<const gran_t*>(sA + buf * A_BYTES + row * 128 + ((kg ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * WN + j * 16 + lrow;
        bf[j] = *reinterpret_cast<const gran_t*>(sB + buf * B_BYTES + row * 128 + ((kg ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mma_granule<T>(af[i], bf[j], acc[i][j]);
    }
    if (kt + 1 < nk) CFT_STORE_TILE(buf ^ 1)
    __syncthreads();
  }

  // ---- epilogue ----
  constexpr int SLD = WN + 4;  // fp32 strip leading dimension (+4: the four 4-row lane groups hit different banks)
  float* stage = reinterpret_cast<float*>(smem) + wave * (16 * SLD);
  float bias_v[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * WN + j * 16 + lrow;
    bias_v[j] = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.0f;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        stage[(lgrp * 4 + e) * SLD + j * 16 + lrow] = apply_act(acc[i][j][e] + bias_v[j], p.act);
    __syncthreads();
    const int mbase = m0 + wm * WM + i * 16;
    const int nbase = n0 + wn * WN;
    if (p.out_f32) {
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
              v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
              v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
            }
          }
          *reinterpret_cast<f32x4_t*>(p.y + ((long)m * p.ldy + p.yoff + n) * 4) = f32x4_t{v[0], v[1], v[2], v[3]};
        }
      }
    } else {
      constexpr int VPR = WN / 8;
      for (int it = lane; it < 16 * VPR; it += 64) {
        const int row = it / VPR, col = (it - row * VPR) * 8;
        const int m = mbase + row, n = nbase + col;
        if (m < p.M && n < p.N) {
          const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col);
          const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col + 4);
          float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
          if (p.res != nullptr) {
            const long ro = (long)m * p.ldr + p.roff + n;
            if (p.res_f32) {
              const float4 r0v = *reinterpret_cast<const float4*>(p.res + ro * 4);
              const float4 r1v = *reinterpret_cast<const float4*>(p.res + ro * 4 + 16);
              v[0] += r0v.x; v[1] += r0v.y; v[2] += r0v.z; v[3] += r0v.w;
              v[4] += r1v.x; v[5] += r1v.y; v[6] += r1v.z; v[7] += r1v.w;
            } else {
              const gran_t rr = *reinterpret_cast<const gran_t*>(p.res + ro * 2);
              float rf[8];
              Elem<uint16_t>::unpack(rr, rf);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += rf[e];
            }
          }
          *reinterpret_cast<gran_t*>(p.y + ((long)m * p.ldy + p.yoff + n) * 2) = Elem<uint16_t>::pack(v);
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ host
template <typename T, int BM, int BN>
static int launch_conv(const ConvParams& p, hipStream_t stream) {
  constexpr int smem_bytes = 2 * (BM + BN) * 128;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<T, BM, BN>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    attr_done = true;
  }
  ConvParams q = p;
  const int tilesM = (p.M + BM - 1) / BM;
  q.tilesN = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN>), dim3(tilesM * q.tilesN), dim3(256), smem_bytes, stream, q);
  return cft_check_launch("conv_gemm_kernel");
}

template <typename T>
static int dispatch_conv(const ConvParams& p, hipStream_t stream) {
  // Tile choice: BN = 64 for narrow outputs; BM = 64 when 128-row tiles would leave most of the
  // 256 CUs idle (deep layers at small batch).
  const bool narrow = p.N <= 64;
  const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + (narrow ? 63 : 127)) / (narrow ? 64 : 128));
  const bool small = tiles128 < 384;
  if (narrow) return small ? launch_conv<T, 64, 64>(p, stream) : launch_conv<T, 128, 64>(p, stream);
  return small ? launch_conv<T, 64, 128>(p, stream) : launch_conv<T, 128, 128>(p, stream);
}

extern "C" int cft_conv2d(const void* x, const void* w, const float* bias, const void* res, void* y,
                          int B, int H, int W, int cin, int ldx, int xoff,
                          int n, int kpad, int ksize, int stride,
                          int ldy, int yoff, int ldr, int roff,
                          int act, int dtype, int out_dtype, int res_dtype, void* stream) {
  CFT_REQUIRE(x && w && y, "cft_conv2d: null pointer");
  CFT_REQUIRE(dtype == CFT_BF16 || dtype == CFT_F32, "cft_conv2d: dtype must be CFT_BF16 or CFT_F32");
  CFT_REQUIRE(out_dtype == CFT_BF16 || out_dtype == CFT_F32, "cft_conv2d: bad out_dtype");
  CFT_REQUIRE(!(dtype == CFT_F32 && out_dtype == CFT_BF16), "cft_conv2d: f32 compute writes f32");
  CFT_REQUIRE(B > 0 && H > 0 && W > 0 && cin > 0 && n > 0, "cft_conv2d: non-positive size");
  CFT_REQUIRE(ksize >= 1 && ksize <= 5 && (ksize & 1) && stride >= 1, "cft_conv2d: ksize must be 1, 3 or 5");
  const int ge = dtype == CFT_BF16 ? 8 : 4, bk = 8 * ge;
  CFT_REQUIRE(cin % ge == 0 && ldx % ge == 0 && xoff % ge == 0, "cft_conv2d: input channels/ld/offset not granule aligned");
  CFT_REQUIRE(kpad % bk == 0 && kpad >= ksize * ksize * cin, "cft_conv2d: kpad must cover k*k*cin and be a multiple of the K step");
  CFT_REQUIRE(n % 8 == 0 && ldy % 8 == 0 && yoff % 8 == 0, "cft_conv2d: n/ldy/yoff must be multiples of 8");
  CFT_REQUIRE(res == nullptr || (ldr % 8 == 0 && roff % 8 == 0), "cft_conv2d: residual ld/offset must be multiples of 8");
  const int pad = ksize / 2;
  const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
  const long M = (long)B * Ho * Wo;
  CFT_REQUIRE(M < (1L << 31) && (long)B * H * W * ldx < (1L << 31) && M * ldy < (1L << 31),
              "cft_conv2d: tensor exceeds 2^31 elements (split the batch)");
  ConvParams p;
  p.x = (const unsigned char*)x; p.w = (const unsigned char*)w; p.bias = bias;
  p.res = (const unsigned char*)res; p.y = (unsigned char*)y;
  p.H = H; p.W = W; p.Cin = cin; p.ldx = ldx; p.xoff = xoff;
  p.Ho = Ho; p.Wo = Wo; p.N = n; p.Kpad = kpad; p.K = ksize * ksize * cin;
  p.ldy = ldy; p.yoff = yoff; p.ldr = ldr; p.roff = roff;
  p.KS = ksize; p.stride = stride; p.pad = pad;
  p.act = act; p.out_f32 = out_dtype == CFT_F32; p.res_f32 = res_dtype == CFT_F32;
  p.M = (int)M; p.tilesN = 0;
  return dtype == CFT_BF16 ? dispatch_conv<uint16_t>(p, as_stream(stream)) : dispatch_conv<float>(p, as_stream(stream));
}
