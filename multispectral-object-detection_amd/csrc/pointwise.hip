// Bandwidth-bound kernels of the two-stream YOLOv5 + CFT forward (gfx950): Focus space-to-depth,
// SPP max pools, nearest-upsample/concat copy, Add/Add2, CFT tokeniser (adaptive avg-pool +
// pos_emb), LayerNorm, CFT de-tokeniser (bilinear upsample + residual add) and Detect decode.
// All tensors are NHWC and are moved in 16-byte granules (8 bf16 / 4 f32) per lane so that a
// wave's accesses coalesce into full 128-B lines.
#include "cft_common.h"

static inline int grid_for(long work, int block) {
  long g = (work + block - 1) / block;
  const long cap = 256L * 16;  // 256 CUs x 16 resident 256-thread blocks; grid-stride beyond that
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------- Focus
// One thread per output pixel: 3 channels x (2 rows x float2) in, 16 channels out.
template <typename T>
__global__ void __launch_bounds__(256) focus_s2d_kernel(const float* __restrict__ in, unsigned char* __restrict__ out,
                                                        int B, int H, int W) {
  const int Ho = H >> 1, Wo = W >> 1;
  const long total = (long)B * Ho * Wo;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % Wo);
    const long t = idx / Wo;
    const int y = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float v[16];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* base = in + (((long)b * 3 + c) * H + 2 * y) * W + 2 * x;
      const float2 top = *reinterpret_cast<const float2*>(base);
      const float2 bot = *reinterpret_cast<const float2*>(base + W);
      v[0 + c] = top.x;   // (dy=0, dx=0)
      v[3 + c] = bot.x;   // (dy=1, dx=0)
      v[6 + c] = top.y;   // (dy=0, dx=1)
      v[9 + c] = bot.y;   // (dy=1, dx=1)
    }
    v[12] = v[13] = v[14] = v[15] = 0.f;
    constexpr int GE = Elem<T>::GE;
    gran_t* o = reinterpret_cast<gran_t*>(out + idx * 16 * sizeof(T));
#pragma unroll
    for (int k = 0; k < 16 / GE; ++k) o[k] = Elem<T>::pack(v + k * GE);
  }
}

extern "C" int cft_focus_s2d(const float* in, void* out, int B, int H, int W, int dtype, void* stream) {
  CFT_REQUIRE(in && out, "cft_focus_s2d: null pointer");
  CFT_REQUIRE(B > 0 && H > 0 && W > 0 && (H % 2 == 0) && (W % 2 == 0), "cft_focus_s2d: H and W must be even");
  CFT_REQUIRE(cft_is_dtype(dtype), "cft_focus_s2d: bad dtype");
  const long total = (long)B * (H / 2) * (W / 2);
  const int grid = grid_for(total, 256);
  CFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(focus_s2d_kernel<T>, dim3(grid), dim3(256), 0, as_stream(stream), in, (unsigned char*)out, B, H, W));
  return cft_check_launch("focus_s2d_kernel");
}

// uint8 variant: the reference's callers hold the pair as ONE uint8 [B,6,H,W] tensor and do
// `img.float() / 255` then `img[:, :3]` / `img[:, 3:]` on the host side of the model (test.py:106-113,
// detect_twostream.py:69-79).  Here the normalisation, the channel split (via strides) and the cast are folded
// into the space-to-depth gather: 4x less input traffic, no fp32 image copy at all.
template <typename T>
__global__ void __launch_bounds__(256) focus_s2d_u8_kernel(const unsigned char* __restrict__ in, long sb, long sc, long sh,
                                                           unsigned char* __restrict__ out, int B, int H, int W, float scale) {
  const int Ho = H >> 1, Wo = W >> 1;
  const long total = (long)B * Ho * Wo;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % Wo);
    const long t = idx / Wo;
    const int y = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float v[16];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const unsigned char* base = in + (long)b * sb + (long)c * sc + (long)(2 * y) * sh + 2 * x;
      v[0 + c] = (float)base[0] * scale;        // (dy=0, dx=0)
      v[6 + c] = (float)base[1] * scale;        // (dy=0, dx=1)
      v[3 + c] = (float)base[sh] * scale;       // (dy=1, dx=0)
      v[9 + c] = (float)base[sh + 1] * scale;   // (dy=1, dx=1)
    }
    v[12] = v[13] = v[14] = v[15] = 0.f;
    constexpr int GE = Elem<T>::GE;
    gran_t* o = reinterpret_cast<gran_t*>(out + idx * 16 * sizeof(T));
#pragma unroll
    for (int k = 0; k < 16 / GE; ++k) o[k] = Elem<T>::pack(v + k * GE);
  }
}

extern "C" int cft_focus_s2d_u8(const unsigned char* in, long stride_b, long stride_c, long stride_h, void* out,
                                int B, int H, int W, float scale, int dtype, void* stream) {
  CFT_REQUIRE(in && out, "cft_focus_s2d_u8: null pointer");
  CFT_REQUIRE(B > 0 && H > 0 && W > 0 && (H % 2 == 0) && (W % 2 == 0), "cft_focus_s2d_u8: H and W must be even");
  CFT_REQUIRE(cft_is_dtype(dtype), "cft_focus_s2d_u8: bad dtype");
  CFT_REQUIRE(stride_h >= W && stride_c > 0 && stride_b > 0, "cft_focus_s2d_u8: bad strides");
  const long total = (long)B * (H / 2) * (W / 2);
  const int grid = grid_for(total, 256);
  CFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(focus_s2d_u8_kernel<T>, dim3(grid), dim3(256), 0, as_stream(stream), in, stride_b, stride_c, stride_h, (unsigned char*)out, B, H, W, scale));
  return cft_check_launch("focus_s2d_u8_kernel");
}

// ------------------------------------------------------------------------------- SPP
// One workgroup per (image, 16-byte channel granule).  The HxW plane of that granule is staged
// in LDS, then two separable passes (row maxima for the three radii, then column maxima).
template <typename T>
__device__ __forceinline__ void gmax(float* a, const gran_t& g) {
  float f[Elem<T>::GE];
  Elem<T>::unpack(g, f);
#pragma unroll
  for (int i = 0; i < Elem<T>::GE; ++i) a[i] = fmaxf(a[i], f[i]);
}

template <typename T>
__global__ void __launch_bounds__(256) spp_maxpool_kernel(unsigned char* buf, int H, int W, int C, int ld,
                                                          int r1, int r2, int r3) {
  constexpr int GE = Elem<T>::GE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  gran_t* plane = reinterpret_cast<gran_t*>(smem);   // [H*W]
  gran_t* rowm = plane + H * W;                      // [3][H*W]
  const int gpc = C / GE;
  const int b = blockIdx.x / gpc, cg = blockIdx.x % gpc;
  const int HW = H * W;
  unsigned char* base = buf + ((long)b * HW * ld + (long)cg * GE) * sizeof(T);
  const long pix_stride = (long)ld * sizeof(T);
  for (int i = threadIdx.x; i < HW; i += blockDim.x) plane[i] = *reinterpret_cast<const gran_t*>(base + i * pix_stride);
  __syncthreads();
  const int rad[3] = {r1, r2, r3};
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    const int y = i / W, x = i - y * W;
    float m[GE];
#pragma unroll
    for (int e = 0; e < GE; ++e) m[e] = -INFINITY;
    int done = -1;  // radius already covered
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      for (int d = done + 1; d <= rad[k]; ++d) {
        if (d == 0) { gmax<T>(m, plane[i]); continue; }
        if (x - d >= 0) gmax<T>(m, plane[i - d]);
        if (x + d < W) gmax<T>(m, plane[i + d]);
      }
      done = rad[k];
      rowm[k * HW + i] = Elem<T>::pack(m);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    const int y = i / W;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float m[GE];
#pragma unroll
      for (int e = 0; e < GE; ++e) m[e] = -INFINITY;
      const gran_t* src = rowm + k * HW;
      for (int d = -rad[k]; d <= rad[k]; ++d)
        if ((unsigned)(y + d) < (unsigned)H) gmax<T>(m, src[i + d * W]);
      *reinterpret_cast<gran_t*>(base + i * pix_stride + (long)(k + 1) * C * sizeof(T)) = Elem<T>::pack(m);
    }
  }
}

// Fast path for the SPP the networks use (k, 2k-1, 3k-2, e.g. 5/9/13): max-pooling composes exactly,
// mp(2r) = mp(r) o mp(r) and mp(3r) = mp(r) o mp(2r) (windows clipped to the map, -inf padding), so the three
// outputs are a chain of three identical separable radius-r passes.  One workgroup owns G consecutive 16-byte
// channel granules of one image (64-byte runs per pixel for G = 4 - the generic kernel above touches 16 B per
// 128-B line), two LDS planes ping-pong: A --rows--> B --columns--> A (+ store), three times.
template <typename T>
__global__ void __launch_bounds__(512) spp_chain_kernel(unsigned char* buf, int H, int W, int C, int ld, int r, int G) {
  constexpr int GE = Elem<T>::GE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int HW = H * W, n = HW * G;
  gran_t* A = reinterpret_cast<gran_t*>(smem);
  gran_t* Bp = A + n;
  const int groups = (C / GE) / G;
  const int b = blockIdx.x / groups, g0 = (blockIdx.x - b * groups) * G;
  unsigned char* base = buf + ((long)b * HW * ld + (long)g0 * GE) * sizeof(T);
  const long pix_stride = (long)ld * sizeof(T);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int px = i / G, g = i - px * G;
    A[i] = *reinterpret_cast<const gran_t*>(base + px * pix_stride + g * 16);
  }
  __syncthreads();
  for (int k = 0; k < 3; ++k) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {      // rows: A -> Bp
      const int px = i / G;
      const int x = px % W;
      float m[GE];
      Elem<T>::unpack(A[i], m);
      for (int d = 1; d <= r; ++d) {
        if (x - d >= 0) gmax<T>(m, A[i - d * G]);
        if (x + d < W) gmax<T>(m, A[i + d * G]);
      }
      Bp[i] = Elem<T>::pack(m);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {      // columns: Bp -> A, and out
      const int px = i / G, g = i - px * G;
      const int y = px / W;
      float m[GE];
      Elem<T>::unpack(Bp[i], m);
      for (int d = 1; d <= r; ++d) {
        if (y - d >= 0) gmax<T>(m, Bp[i - d * W * G]);
        if (y + d < H) gmax<T>(m, Bp[i + d * W * G]);
      }
      const gran_t o = Elem<T>::pack(m);
      A[i] = o;
      *reinterpret_cast<gran_t*>(base + px * pix_stride + ((long)(k + 1) * C * sizeof(T)) + g * 16) = o;
    }
    __syncthreads();
  }
}

extern "C" int cft_spp_maxpool(void* buf, int B, int H, int W, int C, int ld, int k1, int k2, int k3,
                               int dtype, void* stream) {
  CFT_REQUIRE(buf != nullptr, "cft_spp_maxpool: null pointer");
  CFT_REQUIRE(cft_is_dtype(dtype), "cft_spp_maxpool: bad dtype");
  const int ge = cft_granule(dtype);
  CFT_REQUIRE(C % ge == 0 && ld % ge == 0 && ld >= 4 * C, "cft_spp_maxpool: C/ld not granule aligned or ld < 4C");
  CFT_REQUIRE((k1 & 1) && (k2 & 1) && (k3 & 1) && k1 <= k2 && k2 <= k3 && k3 <= 13 && k1 >= 1, "cft_spp_maxpool: kernel sizes must be odd, ascending, <= 13");
  const int r1 = k1 / 2, r2 = k2 / 2, r3 = k3 / 2, gpc = C / ge;
  if (r1 >= 1 && r2 == 2 * r1 && r3 == 3 * r1 && (size_t)2 * H * W * 16 <= 160 * 1024) {   // chained passes (5/9/13)
    int G = 1;
    for (int g = 4; g > 1; g >>= 1)
      if (gpc % g == 0 && (size_t)2 * H * W * g * 16 <= 64 * 1024) { G = g; break; }
    const size_t smem = (size_t)2 * H * W * G * 16;
    const int grid = B * (gpc / G);
    CFT_DISPATCH_DTYPE(dtype, T, {
      cft_allow_lds<&spp_chain_kernel<T>>(160 * 1024);
      hipLaunchKernelGGL(spp_chain_kernel<T>, dim3(grid), dim3(512), smem, as_stream(stream), (unsigned char*)buf, H, W, C, ld, r1, G);
    });
    return cft_check_launch("spp_chain_kernel");
  }
  const size_t smem = (size_t)4 * H * W * 16;
  CFT_REQUIRE(smem <= 160 * 1024, "cft_spp_maxpool: feature map too large for the LDS plane (H*W <= 2560)");
  const int grid = B * gpc;
  CFT_DISPATCH_DTYPE(dtype, T, {
    cft_allow_lds<&spp_maxpool_kernel<T>>(160 * 1024);
    hipLaunchKernelGGL(spp_maxpool_kernel<T>, dim3(grid), dim3(256), smem, as_stream(stream), (unsigned char*)buf, H, W, C, ld, r1, r2, r3);
  });
  return cft_check_launch("spp_maxpool_kernel");
}

// ------------------------------------------------------------------------------- copy / upsample
__global__ void __launch_bounds__(256) copy_channels_kernel(const unsigned char* __restrict__ in, long ldi_b, long ioff_b,
                                                            unsigned char* __restrict__ out, long ldo_b, long ooff_b,
                                                            int B, int Ho, int Wo, int gpp, int up) {
  const long total = (long)B * Ho * Wo * gpp;
  const int Hi = Ho >> up, Wi = Wo >> up;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int gidx = (int)(idx % gpp);
    long t = idx / gpp;
    const int x = (int)(t % Wo); t /= Wo;
    const int y = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const long ipix = ((long)b * Hi + (y >> up)) * Wi + (x >> up);
    const long opix = ((long)b * Ho + y) * Wo + x;
    *reinterpret_cast<gran_t*>(out + opix * ldo_b + ooff_b + gidx * 16L) =
        *reinterpret_cast<const gran_t*>(in + ipix * ldi_b + ioff_b + gidx * 16L);
  }
}

extern "C" int cft_copy_channels(const void* in, int ldi, int ioff, void* out, int ldo, int ooff,
                                 int B, int Ho, int Wo, int C, int up, int dtype, void* stream) {
  CFT_REQUIRE(in && out, "cft_copy_channels: null pointer");
  CFT_REQUIRE(cft_is_dtype(dtype), "cft_copy_channels: bad dtype");
  const int ge = cft_granule(dtype), es = cft_elem_size(dtype);
  CFT_REQUIRE(C % ge == 0 && ldi % ge == 0 && ioff % ge == 0 && ldo % ge == 0 && ooff % ge == 0, "cft_copy_channels: not granule aligned");
  CFT_REQUIRE(up >= 0 && up <= 3 && (Ho % (1 << up) == 0) && (Wo % (1 << up) == 0), "cft_copy_channels: bad upsample shift");
  const int gpp = C / ge;
  const long total = (long)B * Ho * Wo * gpp;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream),
                     (const unsigned char*)in, (long)ldi * es, (long)ioff * es, (unsigned char*)out, (long)ldo * es, (long)ooff * es,
                     B, Ho, Wo, gpp, up);
  return cft_check_launch("copy_channels_kernel");
}

// ------------------------------------------------------------------------------- layout / dtype conversion
// Any strided [B,C,H,W] tensor (fp32 / bf16 / half, e.g. a plain NCHW torch tensor handed to one of the modules)
// -> NHWC channel slice in the compute dtype; channels [C, Cpad) are zero filled.  Boundary helper, not on the
// hot path: inside the network every producer already writes NHWC.
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) to_nhwc_kernel(const unsigned char* __restrict__ in, long sb, long sc, long sh, long sw,
                                                      unsigned char* __restrict__ out, long ldo_b, long ooff_b,
                                                      int B, int C, int H, int W, int gpp) {
  constexpr int GE = Elem<TO>::GE;
  const long total = (long)B * H * W * gpp;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int gidx = (int)(idx % gpp);
    long t = idx / gpp;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    float v[GE];
#pragma unroll
    for (int e = 0; e < GE; ++e) {
      const int c = gidx * GE + e;
      float f = 0.0f;
      if (c < C) {
        const long o = (long)b * sb + (long)c * sc + (long)y * sh + (long)x * sw;
        if constexpr (sizeof(TI) == 4) f = reinterpret_cast<const float*>(in)[o];
        else {
          float lo, hi;
          Elem<TI>::unpack2((uint32_t)reinterpret_cast<const uint16_t*>(in)[o], lo, hi);
          f = lo;
        }
      }
      v[e] = f;
    }
    *reinterpret_cast<gran_t*>(out + (((long)b * H + y) * W + x) * ldo_b + ooff_b + gidx * 16L) = Elem<TO>::pack(v);
  }
}

extern "C" int cft_to_nhwc(const void* in, int in_dtype, long stride_b, long stride_c, long stride_h, long stride_w,
                           void* out, int ldo, int ooff, int B, int C, int cpad, int H, int W, int dtype, void* stream) {
  CFT_REQUIRE(in && out, "cft_to_nhwc: null pointer");
  CFT_REQUIRE(cft_is_dtype(in_dtype) && cft_is_dtype(dtype), "cft_to_nhwc: bad dtype");
  CFT_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "cft_to_nhwc: non-positive size");
  const int ge = cft_granule(dtype), es = cft_elem_size(dtype);
  CFT_REQUIRE(cpad >= C && cpad % ge == 0, "cft_to_nhwc: cpad must be >= C and a granule multiple");
  CFT_REQUIRE(ldo % ge == 0 && ooff % ge == 0 && ldo >= ooff + cpad, "cft_to_nhwc: output ld/offset not granule aligned or too small");
  const int gpp = cpad / ge;
  const long total = (long)B * H * W * gpp;
  const int grid = grid_for(total, 256);
  CFT_DISPATCH_DTYPE(in_dtype, TI, CFT_DISPATCH_DTYPE(dtype, TO,
      hipLaunchKernelGGL((to_nhwc_kernel<TI, TO>), dim3(grid), dim3(256), 0, as_stream(stream), (const unsigned char*)in, stride_b, stride_c, stride_h, stride_w,
                         (unsigned char*)out, (long)ldo * es, (long)ooff * es, B, C, H, W, gpp)));
  return cft_check_launch("to_nhwc_kernel");
}

// ------------------------------------------------------------------------------- add
template <typename T>
__global__ void __launch_bounds__(256) add_kernel(const unsigned char* a, long lda_b, long aoff_b,
                                                  const unsigned char* b, long ldb_b, long boff_b,
                                                  unsigned char* out, long ldo_b, long ooff_b, long M, int gpp) {
  constexpr int GE = Elem<T>::GE;
  const long total = M * gpp;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / gpp;
    const long go = (idx - m * gpp) * 16L;
    float fa[GE], fb[GE];
    Elem<T>::unpack(*reinterpret_cast<const gran_t*>(a + m * lda_b + aoff_b + go), fa);
    Elem<T>::unpack(*reinterpret_cast<const gran_t*>(b + m * ldb_b + boff_b + go), fb);
#pragma unroll
    for (int e = 0; e < GE; ++e) fa[e] += fb[e];
    *reinterpret_cast<gran_t*>(out + m * ldo_b + ooff_b + go) = Elem<T>::pack(fa);
  }
}

extern "C" int cft_add(const void* a, int lda, int aoff, const void* b, int ldb, int boff,
                       void* out, int ldo, int ooff, long M, int C, int dtype, void* stream) {
  CFT_REQUIRE(a && b && out, "cft_add: null pointer");
  CFT_REQUIRE(cft_is_dtype(dtype), "cft_add: bad dtype");
  const int ge = cft_granule(dtype), es = cft_elem_size(dtype);
  CFT_REQUIRE(C % ge == 0 && lda % ge == 0 && aoff % ge == 0 && ldb % ge == 0 && boff % ge == 0 && ldo % ge == 0 && ooff % ge == 0,
              "cft_add: not granule aligned");
  const int gpp = C / ge;
  const int grid = grid_for(M * gpp, 256);
  CFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(add_kernel<T>, dim3(grid), dim3(256), 0, as_stream(stream), (const unsigned char*)a, (long)lda * es, (long)aoff * es,
                                                   (const unsigned char*)b, (long)ldb * es, (long)boff * es, (unsigned char*)out, (long)ldo * es, (long)ooff * es, M, gpp));
  return cft_check_launch("add_kernel");
}

// ------------------------------------------------------------------------------- CFT tokeniser
// grid = B*128 token cells; each thread owns channel granules and walks the pooling window.
template <typename T>
__global__ void __launch_bounds__(256) gpt_tokenize_kernel(const unsigned char* rgb, long ld_rgb_b, long off_rgb_b,
                                                           const unsigned char* ir, long ld_ir_b, long off_ir_b,
                                                           const float* __restrict__ pos_emb, float* __restrict__ tokens,
                                                           int H, int W, int C) {
  constexpr int GE = Elem<T>::GE;
  const int cell = blockIdx.x & 127, b = blockIdx.x >> 7;
  const int s = cell >> 6, i = (cell >> 3) & 7, j = cell & 7;
  const int h0 = (i * H) / 8, h1 = ((i + 1) * H + 7) / 8;
  const int w0 = (j * W) / 8, w1 = ((j + 1) * W + 7) / 8;
  const unsigned char* src = s ? ir : rgb;
  const long ldb = s ? ld_ir_b : ld_rgb_b, offb = s ? off_ir_b : off_rgb_b;
  const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
  for (int cg = threadIdx.x; cg < C / GE; cg += blockDim.x) {
    float acc[GE];
#pragma unroll
    for (int e = 0; e < GE; ++e) acc[e] = 0.f;
    for (int y = h0; y < h1; ++y)
      for (int x = w0; x < w1; ++x) {
        float f[GE];
        Elem<T>::unpack(*reinterpret_cast<const gran_t*>(src + (((long)b * H + y) * W + x) * ldb + offb + cg * 16L), f);
#pragma unroll
        for (int e = 0; e < GE; ++e) acc[e] += f[e];
      }
    float* o = tokens + ((long)b * 128 + cell) * C + cg * GE;
    const float* pe = pos_emb + (long)cell * C + cg * GE;
#pragma unroll
    for (int e = 0; e < GE; ++e) o[e] = acc[e] * inv + pe[e];
  }
}

extern "C" int cft_gpt_tokenize(const void* rgb, int ld_rgb, int off_rgb, const void* ir, int ld_ir, int off_ir,
                                const float* pos_emb, float* tokens, int B, int H, int W, int C,
                                int dtype, void* stream) {
  CFT_REQUIRE(rgb && ir && pos_emb && tokens, "cft_gpt_tokenize: null pointer");
  CFT_REQUIRE(cft_is_dtype(dtype), "cft_gpt_tokenize: bad dtype");
  const int ge = cft_granule(dtype), es = cft_elem_size(dtype);
  CFT_REQUIRE(C % ge == 0 && ld_rgb % ge == 0 && off_rgb % ge == 0 && ld_ir % ge == 0 && off_ir % ge == 0, "cft_gpt_tokenize: not granule aligned");
  CFT_REQUIRE(B > 0 && H >= 1 && W >= 1, "cft_gpt_tokenize: bad shape");
  int threads = C / ge;
  threads = threads < 64 ? 64 : (threads > 256 ? 256 : ((threads + 63) / 64) * 64);
  CFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(gpt_tokenize_kernel<T>, dim3(B * 128), dim3(threads), 0, as_stream(stream), (const unsigned char*)rgb, (long)ld_rgb * es, (long)off_rgb * es,
                                                   (const unsigned char*)ir, (long)ld_ir * es, (long)off_ir * es, pos_emb, tokens, H, W, C));
  return cft_check_launch("gpt_tokenize_kernel");
}

// ------------------------------------------------------------------------------- LayerNorm
// One wave64 per row (C <= 64*32 floats kept in registers as float4 chunks), 4 rows per workgroup.
template <int MAXV>
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, unsigned char* __restrict__ y,
                                                        long rows, int C, float eps, int out_dtype) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = C >> 2;  // float4 per row
  const float4* xr = reinterpret_cast<const float4*>(x + row * C);
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int idx = lane + k * 64;
    v[k] = idx < nv ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += v[k].x + v[k].y + v[k].z + v[k].w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int idx = lane + k * 64;
    if (idx < nv) {
      const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
      q += a * a + b * b + c * c + d * d;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int idx = lane + k * 64;
    if (idx < nv) {
      const float4 gm = reinterpret_cast<const float4*>(gamma)[idx];
      const float4 bt = reinterpret_cast<const float4*>(beta)[idx];
      float o[4] = {(v[k].x - mean) * rstd * gm.x + bt.x, (v[k].y - mean) * rstd * gm.y + bt.y,
                    (v[k].z - mean) * rstd * gm.z + bt.z, (v[k].w - mean) * rstd * gm.w + bt.w};
      if (out_dtype == CFT_F32) {
        *reinterpret_cast<float4*>(y + (row * C + idx * 4L) * 4) = *reinterpret_cast<float4*>(o);
      } else {
        uint2 pk;
        if (out_dtype == CFT_BF16) { pk.x = pack_bf16x2(o[0], o[1]); pk.y = pack_bf16x2(o[2], o[3]); }
        else { pk.x = pack_f16x2(o[0], o[1]); pk.y = pack_f16x2(o[2], o[3]); }
        *reinterpret_cast<uint2*>(y + (row * C + idx * 4L) * 2) = pk;
      }
    }
  }
}

// LayerNorm that first folds split-K partial sums into the residual stream: x[row] += parts[0][row] + parts[1][row] + ... (fixed order;
// x is written back), then y = LayerNorm(x).  One wave per row, the row lives in registers between the two steps.
template <int MAXV>
__global__ void __launch_bounds__(256) layernorm_reduce_kernel(float* __restrict__ x, const float* __restrict__ parts, int nparts, long part_stride,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               unsigned char* __restrict__ y, long rows, int C, float eps, int out_dtype) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = C >> 2;
  float4* xr = reinterpret_cast<float4*>(x + row * C);
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int idx = lane + k * 64;
    v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < nv) {
      float4 a = xr[idx];
      for (int sp = 0; sp < nparts; ++sp) {
        const float4 q = reinterpret_cast<const float4*>(parts + sp * part_stride + row * C)[idx];
        a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
      }
      xr[idx] = a;
      v[k] = a;
    }
    s += v[k].x + v[k].y + v[k].z + v[k].w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int idx = lane + k * 64;
    if (idx < nv) {
      const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
      q += a * a + b * b + c * c + d * d;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int idx = lane + k * 64;
    if (idx < nv) {
      const float4 gm = reinterpret_cast<const float4*>(gamma)[idx];
      const float4 bt = reinterpret_cast<const float4*>(beta)[idx];
      float o[4] = {(v[k].x - mean) * rstd * gm.x + bt.x, (v[k].y - mean) * rstd * gm.y + bt.y,
                    (v[k].z - mean) * rstd * gm.z + bt.z, (v[k].w - mean) * rstd * gm.w + bt.w};
      if (out_dtype == CFT_F32) {
        *reinterpret_cast<float4*>(y + (row * C + idx * 4L) * 4) = *reinterpret_cast<float4*>(o);
      } else {
        uint2 pk;
        if (out_dtype == CFT_BF16) { pk.x = pack_bf16x2(o[0], o[1]); pk.y = pack_bf16x2(o[2], o[3]); }
        else { pk.x = pack_f16x2(o[0], o[1]); pk.y = pack_f16x2(o[2], o[3]); }
        *reinterpret_cast<uint2*>(y + (row * C + idx * 4L) * 2) = pk;
      }
    }
  }
}

// x (fp32 [rows][C], updated in place) += parts[0 .. nparts) (fp32 [nparts][rows][C], e.g. from cft_linear_splitk); y = LayerNorm(x).
extern "C" int cft_layernorm_reduce(float* x, const float* parts, int nparts, const float* gamma, const float* beta, void* y,
                                    long rows, int C, float eps, int out_dtype, void* stream) {
  CFT_REQUIRE(x && parts && gamma && beta && y, "cft_layernorm_reduce: null pointer");
  CFT_REQUIRE(nparts >= 1 && nparts <= 8, "cft_layernorm_reduce: 1 <= nparts <= 8");
  CFT_REQUIRE(C % 4 == 0 && C >= 4 && C <= 4096, "cft_layernorm_reduce: C must be a multiple of 4 and <= 4096");
  CFT_REQUIRE(cft_is_dtype(out_dtype), "cft_layernorm_reduce: bad out dtype");
  CFT_REQUIRE(rows > 0, "cft_layernorm_reduce: rows must be positive");
  const int nv = C >> 2;
  const long grid = (rows + 3) / 4;
  const long ps = rows * (long)C;
  if (nv <= 64 * 2)
    hipLaunchKernelGGL(layernorm_reduce_kernel<2>, dim3(grid), dim3(256), 0, as_stream(stream), x, parts, nparts, ps, gamma, beta, (unsigned char*)y, rows, C, eps, out_dtype);
  else if (nv <= 64 * 5)
    hipLaunchKernelGGL(layernorm_reduce_kernel<5>, dim3(grid), dim3(256), 0, as_stream(stream), x, parts, nparts, ps, gamma, beta, (unsigned char*)y, rows, C, eps, out_dtype);
  else
    hipLaunchKernelGGL(layernorm_reduce_kernel<16>, dim3(grid), dim3(256), 0, as_stream(stream), x, parts, nparts, ps, gamma, beta, (unsigned char*)y, rows, C, eps, out_dtype);
  return cft_check_launch("layernorm_reduce_kernel");
}

extern "C" int cft_layernorm(const float* x, const float* gamma, const float* beta, void* y,
                             long rows, int C, float eps, int out_dtype, void* stream) {
  CFT_REQUIRE(x && gamma && beta && y, "cft_layernorm: null pointer");
  CFT_REQUIRE(C % 4 == 0 && C >= 4 && C <= 4096, "cft_layernorm: C must be a multiple of 4 and <= 4096");
  CFT_REQUIRE(cft_is_dtype(out_dtype), "cft_layernorm: bad out dtype");
  CFT_REQUIRE(rows > 0, "cft_layernorm: rows must be positive");
  const int grid = (int)((rows + 3) / 4);
  const int nv = C / 4;
  const int of32 = out_dtype;
  if (nv <= 64 * 2)
    hipLaunchKernelGGL(layernorm_kernel<2>, dim3(grid), dim3(256), 0, as_stream(stream), x, gamma, beta, (unsigned char*)y, rows, C, eps, of32);
  else if (nv <= 64 * 5)
    hipLaunchKernelGGL(layernorm_kernel<5>, dim3(grid), dim3(256), 0, as_stream(stream), x, gamma, beta, (unsigned char*)y, rows, C, eps, of32);
  else
    hipLaunchKernelGGL(layernorm_kernel<16>, dim3(grid), dim3(256), 0, as_stream(stream), x, gamma, beta, (unsigned char*)y, rows, C, eps, of32);
  return cft_check_launch("layernorm_kernel");
}

// ------------------------------------------------------------------------------- CFT de-tokeniser
// out = base + bilinear(tokens 8x8 -> HxW), PyTorch align_corners=False source index:
// src = (dst + 0.5) * (8 / size) - 0.5, clamped at 0; neighbour clamped at 7.
// One workgroup per output image row (b, y): the two token rows that the bilinear filter touches are blended
// in y ONCE into LDS (8 cells x C floats), then every output granule needs two LDS reads instead of four
// 32-byte token fetches from L2 (the token traffic was 8x the output traffic).
template <typename T>
__global__ void __launch_bounds__(256) gpt_upsample_add_kernel(const float* __restrict__ tokens, int s,
                                                               const unsigned char* base, long ldb_b, long boff_b,
                                                               unsigned char* out, long ldo_b, long ooff_b,
                                                               int B, int H, int W, int C) {
  constexpr int GE = Elem<T>::GE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* R = reinterpret_cast<float*>(smem);          // [8][C]
  const int b = blockIdx.x / H, y = blockIdx.x - b * H;
  float fy = ((float)y + 0.5f) * (8.0f / (float)H) - 0.5f; fy = fy < 0.f ? 0.f : fy;
  const int y0 = (int)fy, y1 = y0 + (y0 < 7 ? 1 : 0);
  const float ly = fy - (float)y0, hy = 1.f - ly;
  const float* t0 = tokens + ((long)b * 128 + s * 64 + y0 * 8) * C;
  const float* t1 = tokens + ((long)b * 128 + s * 64 + y1 * 8) * C;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {          // 8*C/4 float4
    const float4 a = reinterpret_cast<const float4*>(t0)[i], c = reinterpret_cast<const float4*>(t1)[i];
    reinterpret_cast<float4*>(R)[i] = make_float4(hy * a.x + ly * c.x, hy * a.y + ly * c.y, hy * a.z + ly * c.z, hy * a.w + ly * c.w);
  }
  __syncthreads();
  const int gpp = C / GE;
  const float sx = 8.0f / (float)W;
  const long rowpix = ((long)b * H + y) * W;
  for (int idx = threadIdx.x; idx < W * gpp; idx += blockDim.x) {
    const int x = idx / gpp, cg = idx - x * gpp;
    float fx = ((float)x + 0.5f) * sx - 0.5f; fx = fx < 0.f ? 0.f : fx;
    const int x0 = (int)fx, x1 = x0 + (x0 < 7 ? 1 : 0);
    const float lx = fx - (float)x0, hx = 1.f - lx;
    const float* r0 = R + x0 * C + cg * GE;
    const float* r1 = R + x1 * C + cg * GE;
    float v[GE];
    const long pix = rowpix + x;
    if (base != nullptr) {
      Elem<T>::unpack(*reinterpret_cast<const gran_t*>(base + pix * ldb_b + boff_b + cg * 16L), v);
    } else {
#pragma unroll
      for (int e = 0; e < GE; ++e) v[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < GE; ++e) v[e] += hx * r0[e] + lx * r1[e];
    *reinterpret_cast<gran_t*>(out + pix * ldo_b + ooff_b + cg * 16L) = Elem<T>::pack(v);
  }
}

extern "C" int cft_gpt_upsample_add(const float* tokens, int s, const void* base, int ldb, int boff,
                                    void* out, int ldo, int ooff, int B, int H, int W, int C,
                                    int dtype, void* stream) {
  CFT_REQUIRE(tokens && out, "cft_gpt_upsample_add: null pointer");
  CFT_REQUIRE(cft_is_dtype(dtype), "cft_gpt_upsample_add: bad dtype");
  CFT_REQUIRE(s == 0 || s == 1, "cft_gpt_upsample_add: stream index must be 0 or 1");
  const int ge = cft_granule(dtype), es = cft_elem_size(dtype);
  CFT_REQUIRE(C % ge == 0 && ldo % ge == 0 && ooff % ge == 0 && (base == nullptr || (ldb % ge == 0 && boff % ge == 0)), "cft_gpt_upsample_add: not granule aligned");
  CFT_REQUIRE(C % 4 == 0 && (long)B * H < (1L << 31), "cft_gpt_upsample_add: C must be a multiple of 4");
  const int grid = B * H;
  const size_t smem = (size_t)8 * C * sizeof(float);
  CFT_REQUIRE(smem <= 64 * 1024, "cft_gpt_upsample_add: C too large for the LDS row (C <= 2048)");
  CFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(gpt_upsample_add_kernel<T>, dim3(grid), dim3(256), smem, as_stream(stream), tokens, s, (const unsigned char*)base, (long)ldb * es, (long)boff * es,
                                                   (unsigned char*)out, (long)ldo * es, (long)ooff * es, B, H, W, C));
  return cft_check_launch("gpt_upsample_add_kernel");
}

// Both streams of a CFT block in ONE launch, with the Add that follows them in the graph (yaml rows 11-12 + 29 of the x3 configs):
//   out0 = base0 + up(tokens[:, :64]),  out1 = base1 + up(tokens[:, 64:]),  sum = out0 + out1 (optional)
// (models/common.py:626-637 twice + Add2 :238-243 twice + Add :228-229).  One launch instead of three, the two base maps are read once
// instead of once + once more by Add, and the sum is formed from the UNROUNDED fp32 sums: one rounding per output tensor (the
// reference adds in fp32 throughout; profiles/r04_bf16_sites.md lists Add / Add2 among the activation-side rounding sites).
template <typename T>
__global__ void __launch_bounds__(256) gpt_upsample_add2_kernel(const float* __restrict__ tokens,
                                                                const unsigned char* base0, long ldb0_b, long boff0_b,
                                                                const unsigned char* base1, long ldb1_b, long boff1_b,
                                                                unsigned char* out0, long ldo0_b, long ooff0_b,
                                                                unsigned char* out1, long ldo1_b, long ooff1_b,
                                                                unsigned char* sum, long lds_b, long soff_b,
                                                                int B, int H, int W, int C) {
  constexpr int GE = Elem<T>::GE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* R = reinterpret_cast<float*>(smem);          // [2][8][C]: the row's vertical interpolation of both streams' tokens
  const int b = blockIdx.x / H, y = blockIdx.x - b * H;
  float fy = ((float)y + 0.5f) * (8.0f / (float)H) - 0.5f; fy = fy < 0.f ? 0.f : fy;
  const int y0 = (int)fy, y1 = y0 + (y0 < 7 ? 1 : 0);
  const float ly = fy - (float)y0, hy = 1.f - ly;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float* t0 = tokens + ((long)b * 128 + s * 64 + y0 * 8) * C;
    const float* t1 = tokens + ((long)b * 128 + s * 64 + y1 * 8) * C;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
      const float4 a = reinterpret_cast<const float4*>(t0)[i], c = reinterpret_cast<const float4*>(t1)[i];
      reinterpret_cast<float4*>(R + s * 8 * C)[i] = make_float4(hy * a.x + ly * c.x, hy * a.y + ly * c.y, hy * a.z + ly * c.z, hy * a.w + ly * c.w);
    }
  }
  __syncthreads();
  const int gpp = C / GE;
  const float sx = 8.0f / (float)W;
  const long rowpix = ((long)b * H + y) * W;
  for (int idx = threadIdx.x; idx < W * gpp; idx += blockDim.x) {
    const int x = idx / gpp, cg = idx - x * gpp;
    float fx = ((float)x + 0.5f) * sx - 0.5f; fx = fx < 0.f ? 0.f : fx;
    const int x0 = (int)fx, x1 = x0 + (x0 < 7 ? 1 : 0);
    const float lx = fx - (float)x0, hx = 1.f - lx;
    const long pix = rowpix + x;
    float v0[GE], v1[GE];
    Elem<T>::unpack(*reinterpret_cast<const gran_t*>(base0 + pix * ldb0_b + boff0_b + cg * 16L), v0);
    Elem<T>::unpack(*reinterpret_cast<const gran_t*>(base1 + pix * ldb1_b + boff1_b + cg * 16L), v1);
    const float* r0 = R + x0 * C + cg * GE;
    const float* r1 = R + x1 * C + cg * GE;
#pragma unroll
    for (int e = 0; e < GE; ++e) {            // the same expression as gpt_upsample_add_kernel: out0 / out1 are bit-identical to it
      v0[e] += hx * r0[e] + lx * r1[e];
      v1[e] += hx * r0[8 * C + e] + lx * r1[8 * C + e];
    }
    *reinterpret_cast<gran_t*>(out0 + pix * ldo0_b + ooff0_b + cg * 16L) = Elem<T>::pack(v0);
    *reinterpret_cast<gran_t*>(out1 + pix * ldo1_b + ooff1_b + cg * 16L) = Elem<T>::pack(v1);
    if (sum != nullptr) {
#pragma unroll
      for (int e = 0; e < GE; ++e) v0[e] += v1[e];
      *reinterpret_cast<gran_t*>(sum + pix * lds_b + soff_b + cg * 16L) = Elem<T>::pack(v0);
    }
  }
}

extern "C" int cft_gpt_upsample_add2(const float* tokens, const void* base0, int ldb0, int boff0, const void* base1, int ldb1, int boff1,
                                     void* out0, int ldo0, int ooff0, void* out1, int ldo1, int ooff1, void* sum, int lds, int soff,
                                     int B, int H, int W, int C, int dtype, void* stream) {
  CFT_REQUIRE(tokens && base0 && base1 && out0 && out1, "cft_gpt_upsample_add2: null pointer");
  CFT_REQUIRE(cft_is_dtype(dtype), "cft_gpt_upsample_add2: bad dtype");
  const int ge = cft_granule(dtype), es = cft_elem_size(dtype);
  CFT_REQUIRE(C % ge == 0 && ldb0 % ge == 0 && boff0 % ge == 0 && ldb1 % ge == 0 && boff1 % ge == 0 && ldo0 % ge == 0 && ooff0 % ge == 0 &&
              ldo1 % ge == 0 && ooff1 % ge == 0 && (sum == nullptr || (lds % ge == 0 && soff % ge == 0)), "cft_gpt_upsample_add2: not granule aligned");
  CFT_REQUIRE(C % 4 == 0 && (long)B * H < (1L << 31), "cft_gpt_upsample_add2: C must be a multiple of 4");
  const int smem = 2 * 8 * C * (int)sizeof(float);
  CFT_REQUIRE(smem <= 128 * 1024, "cft_gpt_upsample_add2: C too large for the LDS rows (C <= 2048)");
  CFT_DISPATCH_DTYPE(dtype, T, {
    cft_allow_lds<&gpt_upsample_add2_kernel<T>>(128 * 1024);
    hipLaunchKernelGGL(gpt_upsample_add2_kernel<T>, dim3(B * H), dim3(256), smem, as_stream(stream), tokens,
                       (const unsigned char*)base0, (long)ldb0 * es, (long)boff0 * es, (const unsigned char*)base1, (long)ldb1 * es, (long)boff1 * es,
                       (unsigned char*)out0, (long)ldo0 * es, (long)ooff0 * es, (unsigned char*)out1, (long)ldo1 * es, (long)ooff1 * es,
                       (unsigned char*)sum, (long)lds * es, (long)soff * es, B, H, W, C);
  });
  return cft_check_launch("gpt_upsample_add2_kernel");
}

// ------------------------------------------------------------------------------- Detect decode
__global__ void __launch_bounds__(256) detect_decode_kernel(const float* __restrict__ logits, int ldl, float* __restrict__ raw,
                                                            float* __restrict__ pred, const float* __restrict__ anchors,
                                                            int B, int ny, int nx, int na, int no, float stride,
                                                            long row0, long total_rows) {
  const long total = (long)B * na * ny * nx * no;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int o = (int)(idx % no);
    long t = idx / no;
    const int x = (int)(t % nx); t /= nx;
    const int y = (int)(t % ny); t /= ny;
    const int a = (int)(t % na);
    const int b = (int)(t / na);
    const float v = logits[(((long)b * ny + y) * nx + x) * ldl + a * no + o];
    raw[idx] = v;
    const float sg = 1.0f / (1.0f + __expf(-v));
    float r;
    if (o == 0) r = (sg * 2.0f - 0.5f + (float)x) * stride;
    else if (o == 1) r = (sg * 2.0f - 0.5f + (float)y) * stride;
    else if (o < 4) { const float w = sg * 2.0f; r = w * w * anchors[a * 2 + (o - 2)]; }
    else r = sg;
    pred[((long)b * total_rows + row0 + ((long)a * ny + y) * nx + x) * no + o] = r;
  }
}

extern "C" int cft_detect_decode(const float* logits, int ldl, float* raw, float* pred, const float* anchors,
                                 int B, int ny, int nx, int na, int no, float stride,
                                 long row0, long total_rows, void* stream) {
  CFT_REQUIRE(logits && raw && pred && anchors, "cft_detect_decode: null pointer");
  CFT_REQUIRE(B > 0 && ny > 0 && nx > 0 && na > 0 && no >= 5 && ldl >= na * no, "cft_detect_decode: bad shape");
  CFT_REQUIRE(row0 >= 0 && row0 + (long)na * ny * nx <= total_rows, "cft_detect_decode: rows out of range");
  const long total = (long)B * na * ny * nx * no;
  hipLaunchKernelGGL(detect_decode_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), logits, ldl, raw, pred, anchors,
                     B, ny, nx, na, no, stride, row0, total_rows);
  return cft_check_launch("detect_decode_kernel");
}

// ------------------------------------------------------------------------------- letterbox
// Caller-side pre-processing one step before the hot path (reference utils/datasets.py:1698-1728):
// resize an 8-bit HWC image to `new_unpad` with cv2.INTER_LINEAR and pad it to the letterboxed shape with a constant
// colour, in ONE pass on the device.  The resize restates OpenCV's published 8-bit bilinear path (resize.cpp,
// INTER_RESIZE_COEF_BITS = 11): source coordinate fx = (float)((dx + 0.5) * scale - 0.5) with scale = 1 / (dst / src) in
// double, coefficients rounded to 1/2048, horizontal pass in int, vertical pass
// ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.  cv2 is a third-party dependency that is absent from this
// image (and unpinned by the reference): that part is "parity unpinned"; the geometry is the reference's own code.
// The destination is addressed through element strides (y, x, c) and an optional channel flip, so the same kernel writes
// cv2's HWC BGR image or directly the CHW RGB plane of a [B,6,H,W] batch (the `img[:, :, ::-1].transpose(2, 0, 1)` of
// utils/datasets.py:1276-1281).
__device__ __forceinline__ int cft_cv_round(float v) { return __float2int_rn(v); }   // cvRound: nearest, ties to even

__global__ void __launch_bounds__(256) letterbox_u8_kernel(const unsigned char* __restrict__ src, int sh, int sw, long src_row_stride,
                                                           unsigned char* __restrict__ dst, int dh, int dw, long dsy, long dsx, long dsc, int flip,
                                                           int rh, int rw, int top, int left, int c0, int c1, int c2) {
  const long total = (long)dh * dw;
  const double scale_x = 1.0 / ((double)rw / (double)sw), scale_y = 1.0 / ((double)rh / (double)sh);
  const bool resize = rh != sh || rw != sw;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int y = (int)(idx / dw), x = (int)(idx - (long)y * dw);
    const int ry = y - top, rx = x - left;
    int v[3] = {c0, c1, c2};
    if (ry >= 0 && ry < rh && rx >= 0 && rx < rw) {
      if (!resize) {
        const unsigned char* s = src + (long)ry * src_row_stride + rx * 3L;
        v[0] = s[0]; v[1] = s[1]; v[2] = s[2];
      } else {
        float fx = (float)(((double)rx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
        float fy = (float)(((double)ry + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        // cv2 clamps rows by index (sy0 = clip(sy), sy1 = clip(sy + 1)) and keeps the fractional weights
        const int sy0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        const int sy1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        const int a0 = cft_cv_round((1.f - fx) * 2048.f), a1 = cft_cv_round(fx * 2048.f);
        const int b0 = cft_cv_round((1.f - fy) * 2048.f), b1 = cft_cv_round(fy * 2048.f);
        const int sx1 = sx + 1 > sw - 1 ? sw - 1 : sx + 1;
        const unsigned char* r0 = src + (long)sy0 * src_row_stride;
        const unsigned char* r1 = src + (long)sy1 * src_row_stride;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int h0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
          const int h1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
          v[c] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int o = v[c];
      o = o < 0 ? 0 : (o > 255 ? 255 : o);
      dst[(long)y * dsy + (long)x * dsx + (long)(flip ? 2 - c : c) * dsc] = (unsigned char)o;
    }
  }
}

extern "C" int cft_letterbox_u8(const unsigned char* src, int src_h, int src_w, long src_row_stride,
                                unsigned char* dst, int dst_h, int dst_w, long dst_stride_y, long dst_stride_x, long dst_stride_c, int flip_channels,
                                int resized_h, int resized_w, int top, int left, int color0, int color1, int color2, void* stream) {
  CFT_REQUIRE(src && dst, "cft_letterbox_u8: null pointer");
  CFT_REQUIRE(src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0 && resized_h > 0 && resized_w > 0, "cft_letterbox_u8: non-positive size");
  CFT_REQUIRE(top >= 0 && left >= 0 && top + resized_h <= dst_h && left + resized_w <= dst_w, "cft_letterbox_u8: the resized image does not fit the destination");
  CFT_REQUIRE(src_row_stride >= 3L * src_w, "cft_letterbox_u8: bad source row stride");
  const long total = (long)dst_h * dst_w;
  hipLaunchKernelGGL(letterbox_u8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), src, src_h, src_w, src_row_stride,
                     dst, dst_h, dst_w, dst_stride_y, dst_stride_x, dst_stride_c, flip_channels, resized_h, resized_w, top, left, color0, color1, color2);
  return cft_check_launch("letterbox_u8_kernel");
}
