// Focus as ONE kernel for gfx950: space-to-depth of the image (models/common.py:176-179) + its 3x3 Conv
// (+ folded BN + SiLU, models/common.py:45-50) straight from the NCHW image, no intermediate tensor.
//
// The generic path (cft_focus_s2d + cft_conv2d) writes and re-reads the [B,H/2,W/2,16] tensor and gathers
// 32-byte taps from it (K = 9 taps x 16 channels); at 640x640 it ran at ~2 TB/s.  Here a workgroup owns a band
// of 4 output rows and walks along it in tiles of 64 pixels:
//   (a) the 6 x 66 pixel halo patch of the space-to-depth tensor is built in LDS directly from the image
//       (6 x 8-byte loads per patch pixel for fp32 images, 6 x 2-byte loads for uint8 ones, /255 fused),
//       16 bf16 channels per pixel (12 real + 4 zero), zero outside the image (= the conv padding);
//       the image loads of tile t+1 are issued before the MFMAs of tile t and land in registers under them;
//   (b) two waves per output row, 32 pixels x N channels each, K = 160 (9 taps x 16 + one zero tap) as
//       5 x v_mfma_f32_16x16x32_bf16 per 16x16 tile; the A fragment of (pixel, tap, half) is one ds_read_b128
//       of the patch (pixel stride 32 B: conflict-free), the weights sit in LDS in the GEMM's swizzled B layout;
//   (c) bias + activation -> fp32 LDS strip -> 16-byte row vectors -> bf16 NHWC stores (as in conv_gemm.hip).
// The products, the 32-wide k chunks and their order are those of the generic path, so the result is
// bit-identical to it (tests/test_gpu_ops.py).
#include "cft_common.h"
#include "focus_common.h"

struct FocusConvParams {
  const unsigned char* in;
  long sb, sc, sh;          // element strides of the image: batch, channel, row (row elements contiguous)
  float scale;              // 1 for float images, 1/255 for uint8 ones
  const unsigned char* w;   // bf16 [N][kpad], k = (kh*3 + kw)*16 + ci (the layout ops.pack_conv(cin_pad=16) produces)
  const float* bias;
  unsigned char* y;
  int kpad, ldy, yoff;
  int Ho, Wo, tiles_x, bands;
};

template <typename T, typename IN, int NT, int ACT>   // T: uint16_t (bf16) or f16_t compute/output; IN: image element
__global__ void __launch_bounds__(512) focus_conv_kernel(const FocusConvParams p) {
  constexpr int N = NT * 16, TW = 64, TH = 4, PW = TW + 2, PH = TH + 2;
  constexpr int W_BYTES = 3 * N * 128;        // three 64-wide K steps of the [N][192] weight, 128-B rows
  constexpr int ZERO_OFF = PH * PW * 32;      // a zero granule behind the patch (the 10th, all-zero tap)
  constexpr int SLD = N + 4;
  constexpr int VPR = N / 8;                  // 16-B bf16 vectors per output pixel
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW = smem;
  unsigned char* sZ = smem + W_BYTES;         // patch; re-used for the epilogue strips

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int b = blockIdx.x / p.bands, band = blockIdx.x - b * p.bands;
  const int y0 = band * TH;

  // weights -> LDS, slot = k-granule ^ (row & 7) (conflict-free ds_read_b128 fragments, as in conv_gemm.hip)
  for (int idx = tid; idx < 3 * N * 8; idx += 512) {
    const int kt = idx / (N * 8);
    const int r = idx - kt * (N * 8);
    const int n = r >> 3, s = r & 7, g = s ^ (n & 7);
    *reinterpret_cast<gran_t*>(sW + kt * (N * 128) + n * 128 + (s << 4)) =
        *reinterpret_cast<const gran_t*>(p.w + ((long)n * p.kpad + kt * 64 + g * 8) * 2);
  }
  float bias_v[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bias_v[j] = p.bias != nullptr ? p.bias[j * 16 + lrow] : 0.0f;
  float* stage = reinterpret_cast<float*>(sZ) + wave * (16 * SLD);
  const IN* img = reinterpret_cast<const IN*>(p.in) + (long)b * p.sb;
  const int wrow = wave >> 1, wx = (wave & 1) * 32;   // this wave's output row of the band and its 32-pixel half
  const int y = y0 + wrow;

  // patch pixel owned by this thread (one of 6 x 66; threads >= 396 idle in the build phase)
  const bool owner = tid < PH * PW;
  const int py = tid / PW, px = tid - py * PW;
  const int zy = y0 - 1 + py;
  float raw[12];
// request the 12 image samples of patch pixel (py, px) of tile tx_ (zero outside the image = conv padding)
#define FOCUS_FETCH(tx_)                                                                            \
  {                                                                                                 \
    const int zx = (tx_) * TW - 1 + px;                                                             \
    _Pragma("unroll") for (int e = 0; e < 12; ++e) raw[e] = 0.0f;                                   \
    if (owner && (unsigned)zy < (unsigned)p.Ho && (unsigned)zx < (unsigned)p.Wo) {                  \
      _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                               \
        const IN* base = img + (long)c * p.sc + (long)(2 * zy) * p.sh + 2 * zx;                     \
        load_pair<IN>(base, p.scale, raw[0 + c], raw[6 + c]);        /* (dy=0,dx=0), (dy=0,dx=1) */ \
        load_pair<IN>(base + p.sh, p.scale, raw[3 + c], raw[9 + c]); /* (dy=1,dx=0), (dy=1,dx=1) */ \
      }                                                                                             \
    }                                                                                               \
  }
  FOCUS_FETCH(0)

  for (int tx = 0; tx < p.tiles_x; ++tx) {
    const int x0 = tx * TW;
    // ---- (a) halo patch of the space-to-depth tensor: registers -> bf16 -> LDS
    if (owner) {
      float v[16];
#pragma unroll
      for (int e = 0; e < 12; ++e) v[e] = raw[e];
      v[12] = v[13] = v[14] = v[15] = 0.0f;
      gran_t* o = reinterpret_cast<gran_t*>(sZ + tid * 32);
      o[0] = Elem<T>::pack(v);
      o[1] = Elem<T>::pack(v + 8);
    }
    if (tid == PH * PW) *reinterpret_cast<gran_t*>(sZ + ZERO_OFF) = gran_t{0u, 0u, 0u, 0u};
    lds_barrier();
    if (tx + 1 < p.tiles_x) FOCUS_FETCH(tx + 1)   // lands under the MFMAs and the epilogue of this tile

    // ---- (b) 32 pixels x N channels per wave
    f32x4_t acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const int tap = ks * 2 + (lgrp >> 1), half = lgrp & 1;
      const int kh = tap / 3, kw = tap - kh * 3;
      const int abase = tap < 9 ? ((wrow + kh) * PW + wx + lrow + kw) * 32 + half * 16 : ZERO_OFF;
      const int astep = tap < 9 ? 16 * 32 : 0;
      gran_t af[2], bf[NT];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const gran_t*>(sZ + abase + i * astep);
      const int kg = (ks & 1) * 4 + lgrp;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = j * 16 + lrow;
        bf[j] = *reinterpret_cast<const gran_t*>(sW + (ks >> 1) * (N * 128) + n * 128 + ((kg ^ (n & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mma_granule<T>(af[i], bf[j], acc[i][j]);
    }
    lds_barrier();   // every wave is done with the patch before the strips overwrite it

    // ---- (c) bias + activation -> strip -> bf16 rows
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          stage[(lgrp * 4 + e) * SLD + j * 16 + lrow] = apply_act<ACT>(acc[i][j][e] + bias_v[j]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int it = lane; it < 16 * VPR; it += 64) {
        const int row = it / VPR, col = (it - row * VPR) * 8;
        const int x = x0 + wx + i * 16 + row;
        if (x < p.Wo && y < p.Ho) {
          const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col);
          const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col + 4);
          const float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
          const long m = ((long)b * p.Ho + y) * p.Wo + x;
          *reinterpret_cast<gran_t*>(p.y + (m * p.ldy + p.yoff + col) * 2) = Elem<T>::pack(v);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    lds_barrier();   // strips are dead before the next patch is written
  }
#undef FOCUS_FETCH
}

template <typename T, typename IN, int NT, int ACT>
static int launch_focus_conv(const FocusConvParams& p, int B, hipStream_t stream) {
  constexpr int N = NT * 16;
  constexpr int patch = 6 * 66 * 32 + 16, strips = 8 * 16 * (N + 4) * 4;
  constexpr int smem_bytes = 3 * N * 128 + (patch > strips ? patch : strips);
  cft_allow_lds<&focus_conv_kernel<T, IN, NT, ACT>>(smem_bytes);
  hipLaunchKernelGGL((focus_conv_kernel<T, IN, NT, ACT>), dim3(B * p.bands), dim3(512), smem_bytes, stream, p);
  return cft_check_launch("focus_conv_kernel");
}

template <typename T, typename IN, int ACT>
static int dispatch_focus_conv(const FocusConvParams& p, int B, int n, hipStream_t stream) {
  switch (n) {
    case 32: return launch_focus_conv<T, IN, 2, ACT>(p, B, stream);
    case 48: return launch_focus_conv<T, IN, 3, ACT>(p, B, stream);
    case 64: return launch_focus_conv<T, IN, 4, ACT>(p, B, stream);
    default: return launch_focus_conv<T, IN, 5, ACT>(p, B, stream);
  }
}
template <typename T, typename IN>
static int dispatch_focus_act(const FocusConvParams& p, int B, int n, int act, hipStream_t stream) {
  return act == CFT_ACT_SILU ? dispatch_focus_conv<T, IN, CFT_ACT_SILU>(p, B, n, stream) : dispatch_focus_conv<T, IN, CFT_ACT_NONE>(p, B, n, stream);
}
template <typename T>
static int dispatch_focus_in(const FocusConvParams& p, int B, int n, int act, int in_kind, hipStream_t stream) {
  if (in_kind == 1) return dispatch_focus_act<T, unsigned char>(p, B, n, act, stream);
  if constexpr (sizeof(T) == 2 && !__is_same(T, uint16_t)) {   // half images go with half compute (`model.half()` + `img.half()`)
    if (in_kind == 2) return dispatch_focus_act<T, f16_t>(p, B, n, act, stream);
  }
  return dispatch_focus_act<T, float>(p, B, n, act, stream);
}

extern "C" int cft_focus_conv(const void* in, int in_kind, long stride_b, long stride_c, long stride_h, float scale,
                              const void* w, int kpad, const float* bias, void* y, int ldy, int yoff,
                              int B, int H, int W, int n, int act, int dtype, void* stream) {
  CFT_REQUIRE(in && w && y, "cft_focus_conv: null pointer");
  CFT_REQUIRE(dtype == CFT_BF16 || dtype == CFT_F16, "cft_focus_conv: dtype must be CFT_BF16 or CFT_F16");
  CFT_REQUIRE(in_kind >= 0 && in_kind <= 2, "cft_focus_conv: in_kind must be 0 (float), 1 (uint8) or 2 (half)");
  CFT_REQUIRE(in_kind != 2 || dtype == CFT_F16, "cft_focus_conv: half images require dtype CFT_F16");
  CFT_REQUIRE(B > 0 && H > 0 && W > 0 && (H % 2 == 0) && (W % 2 == 0), "cft_focus_conv: H and W must be even");
  CFT_REQUIRE(n == 32 || n == 48 || n == 64 || n == 80, "cft_focus_conv: n must be 32, 48, 64 or 80 (use cft_focus_s2d + cft_conv2d otherwise)");
  CFT_REQUIRE(kpad == 192, "cft_focus_conv: weights must be packed as [n][192] (3x3 taps x 16 channels, zero padded)");
  CFT_REQUIRE(act == CFT_ACT_SILU || act == CFT_ACT_NONE, "cft_focus_conv: activation must be SiLU or none");
  CFT_REQUIRE(ldy % 8 == 0 && yoff % 8 == 0 && ldy >= yoff + n, "cft_focus_conv: bad output ld/offset");
  CFT_REQUIRE(stride_h >= W && stride_c > 0 && stride_b > 0, "cft_focus_conv: bad strides");
  const long es = in_kind == 1 ? 1 : (in_kind == 2 ? 2 : 4);
  CFT_REQUIRE(((long)(size_t)in * 1 % (2 * es) == 0) && stride_h % 2 == 0 && stride_c % 2 == 0 && stride_b % 2 == 0,
              "cft_focus_conv: image rows must start on pixel-pair boundaries (pointer and strides even)");
  FocusConvParams p;
  p.in = (const unsigned char*)in; p.sb = stride_b; p.sc = stride_c; p.sh = stride_h; p.scale = scale;
  p.w = (const unsigned char*)w; p.bias = bias; p.y = (unsigned char*)y;
  p.kpad = kpad; p.ldy = ldy; p.yoff = yoff;
  p.Ho = H / 2; p.Wo = W / 2; p.tiles_x = (p.Wo + 63) / 64; p.bands = (p.Ho + 3) / 4;
  CFT_REQUIRE((long)B * p.bands < (1L << 31) && (long)B * p.Ho * p.Wo * ldy < (1L << 40), "cft_focus_conv: tensor too large");
  hipStream_t s = as_stream(stream);
  return dtype == CFT_BF16 ? dispatch_focus_in<uint16_t>(p, B, n, act, in_kind, s) : dispatch_focus_in<f16_t>(p, B, n, act, in_kind, s);
}
