// Bottleneck as ONE kernel for the 64-channel stage (gfx950):  y = x + SiLU(conv3x3(SiLU(conv1x1(x))))
// (reference models/common.py:99-109 with e = 1.0 as used inside C3, Conv = conv + folded BN + SiLU :45-50).
//
// As two cft_conv2d launches this stage is the least efficient part of the forward (160x160 maps, 64 channels:
// K = 64 / 576 is too short to amortise a GEMM workgroup's prologue and epilogue, and the hidden tensor `t` makes a
// round trip through HBM).  Here a workgroup of 8 waves owns a band of 8 output rows and walks along it in tiles of
// 32 pixels; the 3x3 weights (9 x 64 x 64 bf16 = 72 KiB) stay in LDS for the whole band:
//   phase 1  t = SiLU(W1 x + b1) on the 10 x 34 pixel halo patch (zero outside the image = the 3x3 conv's padding):
//            A fragments straight from global memory (a pixel's 64 channels are one 128-B run), W1 fragments live in
//            registers, result rounded to bf16 into LDS (128-B rows, granule slot ^ (row & 7));
//   phase 2  one output row per wave: 32 pixels x 64 channels, 9 taps x 2 MFMAs per 16x16 tile, A fragments are
//            shifted ds_read_b128 of the t patch, B fragments ds_read_b128 of the resident weights;
//   epilogue bias + SiLU -> fp32 strip -> 16-B row vectors -> + shortcut -> bf16 -> store (as in conv_gemm.hip).
// The image loads of tile i+1 (phase-1 operands) and the shortcut vectors are requested before phase 2 of tile i.
// Products, 32-wide k chunks, their order and every rounding are those of the two-launch path: bit-identical.
#include "cft_common.h"
#include <stdlib.h>

extern int g_conv_variant;   // conv_gemm.hip (cft_set_conv_variant)

__device__ __attribute__((aligned(16))) uint32_t cft_zero_page_b[4] = {0u, 0u, 0u, 0u};
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

struct BneckParams {
  const unsigned char* x;    // bf16 NHWC, ldx channels per pixel, slice offset xoff
  const unsigned char* w1;   // bf16 [C][kpad1]
  const unsigned char* w2;   // bf16 [C][kpad2], k = (kh*3 + kw)*C + ci
  const float* b1;
  const float* b2;
  unsigned char* y;          // bf16 NHWC, ldy / yoff; must not overlap x (halo reads)
  int ldx, xoff, ldy, yoff, kpad1, kpad2;
  int H, W, tiles_x, bands, shortcut;
};

// ABL (timing probes only, results wrong): 1 = no phase-1 MFMAs/SiLU (zeros), 2 = no phase-2 MFMAs, 4 = no epilogue
template <typename T, int NT, int ABL = 0>   // T: uint16_t (bf16) or f16_t
__global__ void __launch_bounds__(512) bottleneck_kernel(const BneckParams p) {
  static_assert(NT == 4, "the LDS layout below is for 64 channels (128-byte pixel rows)");
  constexpr int C = NT * 16, TW = 32, TH = 8, PW = TW + 2, PH = TH + 2;
  constexpr int NPIX = PH * PW;                 // 340 patch pixels
  constexpr int NRT = (NPIX + 15) / 16;         // 22 MFMA row tiles of the patch
  constexpr int RTW = (NRT + 7) / 8;            // row tiles per wave (3)
  constexpr int W2_BYTES = 9 * C * 128;
  constexpr int SLD = C + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW2 = smem;
  unsigned char* sT = smem + W2_BYTES;          // t patch [NRT*16][128 B]; re-used for the epilogue strips

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int b = blockIdx.x / p.bands, band = blockIdx.x - b * p.bands;
  const int y0 = band * TH;
  const long img_pix = (long)b * p.H * p.W;

  // 3x3 weights -> LDS, per tap a [C][128 B] tile, slot = k-granule ^ (row & 7)
#pragma unroll
  for (int k = 0; k < 9 * C * 8 / 512; ++k) {   // 9 independent 16-B loads per thread, then the LDS writes
    const int idx = tid + k * 512;
    const int tap = idx / (C * 8);
    const int r = idx - tap * (C * 8);
    const int n = r >> 3, s = r & 7, g = s ^ (n & 7);
    *reinterpret_cast<gran_t*>(sW2 + tap * (C * 128) + n * 128 + (s << 4)) =
        *reinterpret_cast<const gran_t*>(p.w2 + ((long)n * p.kpad2 + tap * C + g * 8) * 2);
  }
  // 1x1 weights: phase 1 runs transposed (t^T = W1 x^T), so W1 is the ROW operand: lane (row n = j*16 + lrow,
  // k = ks*32 + lgrp*8 ..) and the accumulator of lane (pixel lrow, group lgrp) holds 4 CONSECUTIVE channels
  // j*16 + lgrp*4 + e of one pixel - one 8-byte LDS write instead of four scattered 2-byte ones.
  gran_t w1f[NT][2];
  float b1v[NT][4], b2v[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = j * 16 + lrow;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      w1f[j][ks] = *reinterpret_cast<const gran_t*>(p.w1 + ((long)n * p.kpad1 + ks * 32 + lgrp * 8) * 2);
#pragma unroll
    for (int e = 0; e < 4; ++e) b1v[j][e] = p.b1 != nullptr ? p.b1[j * 16 + lgrp * 4 + e] : 0.0f;
    b2v[j] = p.b2 != nullptr ? p.b2[n] : 0.0f;
  }
  float* stage = reinterpret_cast<float*>(sT) + wave * (16 * SLD);
  const int y = y0 + wave;   // this wave's output row

  gran_t a1[RTW][2];
// phase-1 operands of tile tx_: patch pixel q = rt*16 + lrow of row tile rt = wave + 8*it (zero outside the image)
#define BNECK_FETCH(tx_)                                                                                \
  _Pragma("unroll") for (int it = 0; it < RTW; ++it) {                                                  \
    const int q = (wave + it * 8) * 16 + lrow;                                                          \
    const int py = q / PW, px = q - py * PW;                                                            \
    const int zy = y0 - 1 + py, zx = (tx_) * TW - 1 + px;                                               \
    gran_t t0 = {0u, 0u, 0u, 0u}, t1 = {0u, 0u, 0u, 0u};                                                \
    if (q < NPIX && (unsigned)zy < (unsigned)p.H && (unsigned)zx < (unsigned)p.W) {                     \
      const unsigned char* src = p.x + ((img_pix + (long)zy * p.W + zx) * p.ldx + p.xoff + lgrp * 8) * 2; \
      t0 = *reinterpret_cast<const gran_t*>(src);                                                       \
      t1 = *reinterpret_cast<const gran_t*>(src + 64);                                                  \
    }                                                                                                   \
    a1[it][0] = t0;                                                                                     \
    a1[it][1] = t1;                                                                                     \
  }
  BNECK_FETCH(0)

  for (int tx = 0; tx < p.tiles_x; ++tx) {
    const int x0 = tx * TW;
    // ---- phase 1: t patch -> LDS
#pragma unroll
    for (int it = 0; it < RTW; ++it) {
      const int rt = wave + it * 8;
      if (rt < NRT) {   // wave-uniform
        f32x4_t acc1[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc1[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            if constexpr (!(ABL & 1)) acc1[j] = mma_granule<T>(w1f[j][ks], a1[it][ks], acc1[j]);
        }
        // this lane's pixel: patch row q (the pixel whose operands it fetched); outside the image t = +0
        const int q = rt * 16 + lrow;
        const int py = q / PW, px = q - py * PW;
        const int zy = y0 - 1 + py, zx = x0 - 1 + px;
        const bool inside = q < NPIX && (unsigned)zy < (unsigned)p.H && (unsigned)zx < (unsigned)p.W;
        const uint32_t keep = inside ? 0xffffffffu : 0u;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (ABL & 1) ? 0.0f : apply_act<CFT_ACT_SILU>(acc1[j][e] + b1v[j][e]);
          const int n0 = j * 16 + lgrp * 4;   // first of this lane's 4 channels
          uint2 w;
          w.x = Elem<T>::pack2(v[0], v[1]) & keep;
          w.y = Elem<T>::pack2(v[2], v[3]) & keep;
          *reinterpret_cast<uint2*>(sT + q * 128 + ((((n0 >> 3) ^ (q & 7)) << 4) | ((n0 & 7) << 1))) = w;
        }
      }
    }
    lds_barrier();   // t patch (and, the first time, the 3x3 weights) complete

    // requests that land under phase 2: next tile's phase-1 operands, this tile's shortcut vectors
    if (tx + 1 < p.tiles_x) BNECK_FETCH(tx + 1)
    gran_t rs[2][2];
    if (p.shortcut) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const int it = lane + v * 64;
          const int row = it >> 3, col = (it & 7) * 8;
          const int x = x0 + i * 16 + row;
          gran_t t = {0u, 0u, 0u, 0u};
          if (x < p.W && y < p.H)
            t = *reinterpret_cast<const gran_t*>(p.x + ((img_pix + (long)y * p.W + x) * p.ldx + p.xoff + col) * 2);
          rs[i][v] = t;
        }
    }

    // ---- phase 2: 3x3 conv of the t patch, one output row (32 pixels) per wave
    f32x4_t acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {   // not unrolled: full unrolling hoists all 72 weight fragments (spills); unroll 3 measured no faster
      const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int kg = ks * 4 + lgrp;
        gran_t af[2], bf[NT];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int q = (wave + kh) * PW + i * 16 + lrow + kw;
          af[i] = *reinterpret_cast<const gran_t*>(sT + q * 128 + ((kg ^ (q & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = j * 16 + lrow;
          bf[j] = *reinterpret_cast<const gran_t*>(sW2 + tap * (C * 128) + n * 128 + ((kg ^ (n & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if constexpr (ABL & 2) { asm volatile("" ::"v"(af[i]), "v"(bf[j])); }
            else acc[i][j] = mma_granule<T>(af[i], bf[j], acc[i][j]);
          }
      }
    }
    lds_barrier();   // every wave is done with the t patch before the strips overwrite it

    // ---- epilogue: bias + SiLU -> strip -> (+ shortcut) -> bf16 rows
    if constexpr (ABL & 4) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          stage[(lgrp * 4 + e) * SLD + j * 16 + lrow] = apply_act<CFT_ACT_SILU>(acc[i][j][e] + b2v[j]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int it = lane + v * 64;
        const int row = it >> 3, col = (it & 7) * 8;
        const int x = x0 + i * 16 + row;
        if (x < p.W && y < p.H) {
          const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col);
          const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col + 4);
          float o[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
          if (p.shortcut) {
            float rf[8];
            Elem<T>::unpack(rs[i][v], rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += rf[e];
          }
          *reinterpret_cast<gran_t*>(p.y + ((img_pix + (long)y * p.W + x) * p.ldy + p.yoff + col) * 2) = Elem<T>::pack(o);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    lds_barrier();   // strips are dead before the next t patch is written
  }
#undef BNECK_FETCH
}

// ------------------------------------------------------------------------------------ 128 channels
// The same Bottleneck for the 128-channel stage (yolov5l: 80 x 80 maps, 18 Bottlenecks per forward, as two launches
// the largest line item of the forward: the 3x3 conv re-stages every input tap through the LDS-DMA path, nine times the
// activation bytes, and the hidden tensor makes an HBM round trip).  Here the 3x3 weights (288 KiB) no longer fit in
// LDS, so the roles flip: the ACTIVATIONS stay resident and only weights stream.
//   * One workgroup (8 waves) per 16 x 16 pixel tile.  The hidden tensor t = SiLU(W1 x + b1) of the 18 x 18 halo patch
//     is computed once into LDS (two 64-channel planes of 128-byte pixel rows, granule slot ^ (pixel & 7)); the nine taps
//     of the 3x3 conv are shifted ds_read_b128 of it.  Staging traffic per tile drops from 878 KiB (9 taps of A + W2)
//     to 320 KiB (W1 + W2), all of it L2 hits.
//   * W1 (2 K tiles of 64) and W2 (18 K tiles: tap x channel half) stream through a 3-deep ring of 16-KiB LDS buffers
//     (global_load_lds, source-side swizzle), two tiles ahead of the compute tile, one request per thread per phase;
//     s_waitcnt vmcnt(2) leaves one tile in flight across the barriers.
//   * K loop = the staggered two-group schedule of conv_gemm8n_kernel (conv_gemm.hip): two phases per K tile, waves
//     0-3 / 4-7 one barrier apart so that each SIMD always has one wave in its 16-MFMA segment.  Wave (wmr, wnc) owns
//     4 tile rows x 64 channels; the eight W2 fragments stay in registers for both phases of a K tile.
//   * t patch phase: x fragments straight from global memory (requested first thing, 12 x 16 B per lane), W1 is the row
//     operand so a lane ends up with 4 consecutive channels of one pixel (8-byte LDS writes), zero outside the image.
//   * epilogue: bias + SiLU -> fp32 strip (aliases the dead t patch) -> 16-byte row vectors -> + shortcut -> 16-bit stores.
// Every product, the 32-wide k chunks, their order and every rounding are those of the two-launch path: bit-identical.
static unsigned long long* g_bneck_dbg = nullptr;
extern "C" int cft_set_debug_buffer(void* p) { g_bneck_dbg = (unsigned long long*)p; return CFT_OK; }   // timing probes only

struct Bneck128Params {
  const unsigned char* x;
  const unsigned char* w1;   // [128][kpad1]
  const unsigned char* w2;   // [128][kpad2], k = (kh*3 + kw)*128 + ci
  const unsigned char* w2s;  // the same weights as 36 stage images of 8 KiB (cft_bottleneck_pack_w2), or null
  const float* b1;
  const float* b2;
  unsigned char* y;
  int ldx, xoff, ldy, yoff, kpad1, kpad2;
  int H, W, tiles_x, tiles_y, ntiles, shortcut;
  unsigned long long* dbg;   // timing probe (variant 932): [workgroup][wave 0 / 4][tile][8] s_memtime stamps
};

template <typename T, int ABL = 0>
__global__ void __launch_bounds__(512) bottleneck128_kernel(const Bneck128Params p) {
  constexpr int C = 128, TS = 16, PW = TS + 2, NPIX = PW * PW;     // 324 patch pixels
  constexpr int NRT = (NPIX + 15) / 16;                             // 21 row tiles of the patch
  constexpr int PLANE = NRT * 16 * 128;                             // one 64-channel plane of the t patch (43008 B)
  constexpr int RING = 16384;                                       // one K tile of weights: 128 rows x 128 B
  constexpr int NBUF = 4;                                           // ring depth: three K tiles ahead of the compute tile
  constexpr int NKT = 4 + 18;                                       // per pixel tile: W1 (2 K halves, streamed twice) + W2 (9 taps x 2 channel halves)
  constexpr int SLD = 64 + 4;
  constexpr int XTRA = 12;                                          // prefetch requests issued at the head of the 3x3 loop
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sT = smem;                    // [2][NRT*16][128 B]; re-used for the epilogue strips
  unsigned char* sR = smem + 2 * PLANE;        // [NBUF][128][128 B]
  float* sB1 = reinterpret_cast<float*>(smem + 2 * PLANE + NBUF * RING);   // [128] bias of the 1x1 conv

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int grp = wave >> 2;                   // stagger group (waves w, w + 4 share a SIMD)
  const int wmr = wave >> 1, wnc = wave & 1;   // 3x3 loop: tile rows 4 wmr .. +4, channels 64 wnc .. +64
  const int tiles = p.tiles_x * p.tiles_y;
  if (tid < C) sB1[tid] = p.b1 != nullptr ? p.b1[tid] : 0.0f;     // (a global load inside the tile loop would drain the DMA ring)
  else if (tid < 2 * C) sB1[tid] = p.b2 != nullptr ? p.b2[tid - C] : 0.0f;    // sB1[128..255] = bias of the 3x3 conv
  const int my_tiles = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // this workgroup walks tiles bid, bid + grid, ...
  const unsigned char* zero_page = reinterpret_cast<const unsigned char*>(cft_zero_page_b);

  // ---- weight ring: K tile = rows n (128) x 64 k; this thread stages rows r0 and r0 + 64, k-granule g.  The stream of
  // K tiles is W1.k0 W1.k1 W1.k0 W1.k1 W2.(tap,half) x 18, repeated for every pixel tile of this workgroup (zero page afterwards).
  const int r0 = tid >> 3, slot_s = tid & 7, g = slot_s ^ (r0 & 7);
  int st_kt = 0, st_left = my_tiles, sb = 0;
#define BN128_STAGE(part_)                                                                               \
  {                                                                                                      \
    const int n_ = r0 + 64 * (part_);                                                                    \
    const unsigned char* src_;                                                                           \
    if (st_left <= 0) src_ = zero_page;                                                                  \
    else if (st_kt < 4) src_ = p.w1 + ((long)n_ * p.kpad1 + (st_kt & 1) * 64 + g * 8) * 2;               \
    else src_ = p.w2 + ((long)n_ * p.kpad2 + (st_kt - 4) * 64 + g * 8) * 2;                              \
    if constexpr (!(ABL & 8))                                                                            \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src_, (lds_void_t*)(sR + sb * RING + (part_) * 8192 + wave * 1024), 16, 0, 0); \
    if ((part_) == 1) {                                                                                  \
      sb = sb == NBUF - 1 ? 0 : sb + 1;                                                                  \
      if (++st_kt == NKT) { st_kt = 0; --st_left; }                                                      \
    }                                                                                                    \
  }
#define BN128_BARRIER() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

  // ---- x fragments of a tile's halo patch: row tile rt = wave + 8 it, lane = (pixel lrow, k-group lgrp), 4 k chunks of 32
  // The fragments of the NEXT tile are requested into a1n while the current tile's 3x3 loop runs and handed over to a1
  // after it, so that no global-load destination is live across the DMA ring's counted waits (hipcc drains vmcnt to 0
  // at the first use of a pending load result: that first use is the single hand-over point below).
  gran_t a1[3][4], a1n[3][4];
  static_assert(PW == 18, "the mul-shift below divides by 18");
#define BN128_FETCH_X(tile_)                                                                             \
  {                                                                                                      \
    const int b_ = (tile_) / tiles, tt_ = (tile_) - b_ * tiles;                                          \
    const int ty_ = tt_ / p.tiles_x, tx_ = tt_ - ty_ * p.tiles_x;                                        \
    const long ip_ = (long)b_ * p.H * p.W;                                                               \
    _Pragma("unroll") for (int it = 0; it < 3; ++it) {                                                   \
      const int q = (wave + it * 8) * 16 + lrow;                                                         \
      const int py = (q * 3641) >> 16, px = q - py * PW;      /* q / 18, exact for q < 32768 */          \
      const int zy = ty_ * TS - 1 + py, zx = tx_ * TS - 1 + px;                                          \
      const bool in_ = (tile_) >= 0 && q < NPIX && (unsigned)zy < (unsigned)p.H && (unsigned)zx < (unsigned)p.W; \
      /* always exactly 12 loads per wave (the counted waits of the 3x3 loop rely on it): outside pixels read the zero page */ \
      const unsigned char* px_ = in_ ? p.x + ((ip_ + (long)zy * p.W + zx) * p.ldx + p.xoff + lgrp * 8) * 2 : zero_page; \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                   \
        a1n[it][ks] = *reinterpret_cast<const gran_t*>(px_ + (in_ ? ks * 64 : 0));                       \
    }                                                                                                    \
  }
#define BN128_HANDOVER()                                                                                 \
  {                                                                                                      \
    _Pragma("unroll") for (int it = 0; it < 3; ++it)                                                     \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                 \
        asm volatile("" : "+v"(a1n[it][ks]));   /* a real use: the load has landed from here on */       \
        a1[it][ks] = a1n[it][ks];                                                                        \
      }                                                                                                  \
  }
  BN128_FETCH_X((int)blockIdx.x)
  BN128_STAGE(0) BN128_STAGE(1)
  BN128_STAGE(0) BN128_STAGE(1)
  BN128_STAGE(0) BN128_STAGE(1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // K tiles 0-2, the x fragments, the biases: everything landed
  BN128_HANDOVER()
  BN128_BARRIER()

  const int fb = lrow * 128 + ((lgrp ^ (lrow & 7)) << 4);          // fragment read base inside a ring buffer (row lrow)
  const int fbn = (wnc * 64 + lrow) * 128 + ((lgrp ^ (lrow & 7)) << 4);
  int cb = 0;
  float* stage = reinterpret_cast<float*>(sT) + wave * (16 * SLD);
  const uint32_t t_lds = (uint32_t)(uintptr_t)(lds_void_t*)sT;    // LDS byte address of the t patch

#pragma unroll 1
  for (int ti = 0; ti < my_tiles; ++ti) {
    const int tile = (int)blockIdx.x + ti * (int)gridDim.x;
    const int b = tile / tiles, tt = tile - b * tiles;
    const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
    const int y0 = ty * TS, x0 = tx * TS;
    const long img_pix = (long)b * p.H * p.W;
#define BN128_STAMP(k_)                                                                                  \
    if constexpr (ABL & 32) {                                                                            \
      if (p.dbg != nullptr && lane == 0 && (wave & 3) == 0 && ti < 8)                                    \
        p.dbg[(((long)blockIdx.x * 2 + grp) * 8 + ti) * 8 + (k_)] = __builtin_readcyclecounter();        \
    }
    BN128_STAMP(0)
    if (grp == 1) BN128_BARRIER()              // group 1 runs one barrier behind group 0

    // ---- t^T = W1 x^T on the patch, output channels 0-63 then 64-127 (two passes over the two K halves of W1: half
    // the accumulators and fragments of one pass over all 128 - the register budget of the kernel is set here);
    // one load segment + one MFMA segment (24 MFMAs) per K tile; each pass ends with bias + SiLU -> 16-bit -> t patch
    // (zero outside the image: the 3x3 conv's padding) while the other group is still in its MFMA segment
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
      f32x4_t acc1[3][4];
#pragma unroll
      for (int it = 0; it < 3; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[it][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        gran_t wf[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          wf[j][0] = *reinterpret_cast<const gran_t*>(sR + cb * RING + (jh * 4 + j) * 2048 + fb);
          wf[j][1] = *reinterpret_cast<const gran_t*>(sR + cb * RING + (jh * 4 + j) * 2048 + (fb ^ 64));
        }
        BN128_STAGE(0) BN128_STAGE(1)            // (K tiles 0-2 of this pixel tile landed before the loop / at the hand-over)
        if (jh == 1) { asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); }   // covers K tiles 3 (the second W1.k1) / 4 (the first W2 tile)
        else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   // fragment reads retired BEFORE the barrier: the other group may re-stage this buffer right after it
        BN128_BARRIER()
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int it = 0; it < 3; ++it)
            if (wave + it * 8 < NRT) {   // wave-uniform
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if constexpr (!(ABL & 1)) acc1[it][j] = mma_granule<T>(wf[j][ks], a1[it][kt * 2 + ks], acc1[it][j]);
            }
        __builtin_amdgcn_s_setprio(0);
        BN128_BARRIER()
        cb = cb == NBUF - 1 ? 0 : cb + 1;
      }
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int rt = wave + it * 8;
        if (rt < NRT) {
          const int q = rt * 16 + lrow;
          const int py = (q * 3641) >> 16, px = q - py * PW;
          const uint32_t keep = (q < NPIX && (unsigned)(y0 - 1 + py) < (unsigned)p.H && (unsigned)(x0 - 1 + px) < (unsigned)p.W) ? 0xffffffffu : 0u;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int cc = j * 16 + lgrp * 4;        // first of this lane's 4 channels inside the 64-channel plane jh
            float v[4];
            const f32x4_t b1q = *reinterpret_cast<const f32x4_t*>(sB1 + jh * 64 + cc);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (ABL & 1) ? 0.0f : apply_act<CFT_ACT_SILU>(acc1[it][j][e] + b1q[e]);
            uint2 w;
            w.x = Elem<T>::pack2(v[0], v[1]) & keep;
            w.y = Elem<T>::pack2(v[2], v[3]) & keep;
            // (inline asm: for a compiler-visible LDS store hipcc first drains vmcnt to 0 - the LDS-DMA ring could alias it)
            const uint32_t ta = t_lds + jh * PLANE + q * 128 + ((((cc >> 3) ^ (q & 7)) << 4) | ((cc & 7) << 1));
            const unsigned long long wq = ((unsigned long long)w.y << 32) | w.x;
            asm volatile("ds_write_b64 %0, %1" ::"v"(ta), "v"(wq) : "memory");
          }
        }
      }
    }
    BN128_STAMP(1)
    if (grp == 0) BN128_BARRIER()                // both groups aligned again
    // requests that land under the 3x3 loop: the NEXT tile's x fragments (ALWAYS issued, 12 per wave - masked ones
    // read the zero page - because the counted vmcnt of the first two K tiles below allows for exactly that many)
    {
      const int nt_ = tile + (int)gridDim.x;
      BN128_FETCH_X(nt_ < p.ntiles ? nt_ : -1)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BN128_STAMP(2)
    BN128_BARRIER()                                // the whole t patch is visible
    if (grp == 1) BN128_BARRIER()                  // stagger again

    // ---- 3x3 conv of the t patch: 18 K tiles (tap, channel half), two phases (tile rows 4 wmr + {0,1} / {2,3})
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kk = 0; kk < 18; ++kk) {
      const int tap = kk >> 1, half = kk & 1;
      const int kh = tap / 3, kw = tap - kh * 3;
      const int qb = (wmr * 4 + kh) * PW + kw + lrow;      // patch pixel of tile row 4 wmr, this lane's column, this tap
      gran_t af[4][2], bf[4][2];
      // one load segment (8 A + 8 B fragment reads, the two staging requests of K tile kk + 3, the counted wait for
      // K tile kk + 1) and one MFMA segment (32 MFMAs) per K tile: an inter-barrier interval is as long as its LONGER
      // segment, and a 12-read load segment takes ~450 cycles against 272 for 16 MFMAs (tools/bneck_probe.py)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = qb + i * PW;
        const unsigned char* rowp = sT + half * PLANE + q * 128;
        af[i][0] = *reinterpret_cast<const gran_t*>(rowp + ((lgrp ^ (q & 7)) << 4));
        af[i][1] = *reinterpret_cast<const gran_t*>(rowp + (((4 + lgrp) ^ (q & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf[j][0] = *reinterpret_cast<const gran_t*>(sR + cb * RING + j * 2048 + fbn);
        bf[j][1] = *reinterpret_cast<const gran_t*>(sR + cb * RING + j * 2048 + (fbn ^ 64));
      }
      BN128_STAGE(0) BN128_STAGE(1)
      // the wait covers the next K tile.  During the first two K tiles the prefetch requests issued just before the loop
      // (x fragments of the next pixel tile) may still be in flight behind it.
      if (kk < 2) { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 + XTRA) : "memory"); }
      else { asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); }
      BN128_BARRIER()
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if constexpr (ABL & 2) { asm volatile("" ::"v"(af[i][ks]), "v"(bf[j][ks])); }
            else acc[i][j] = mma_granule<T>(af[i][ks], bf[j][ks], acc[i][j]);
          }
      __builtin_amdgcn_s_setprio(0);
      BN128_BARRIER()
      cb = cb == NBUF - 1 ? 0 : cb + 1;
      if (kk == 2) BN128_STAMP(3)
    }
    BN128_STAMP(4)
    // hand-over point: this tile's shortcut vectors are requested (L2 hits: the patch was read a few microseconds ago;
    // holding them in registers through the 3x3 loop would spill) and the next tile's x fragments are consumed, hipcc
    // waits vmcnt(0) for both - which also lands the first three K tiles of the next pixel tile
    gran_t rs[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int it = lane + v * 64;
        const int row = it >> 3, col = (it & 7) * 8;
        const int x = x0 + row, y = y0 + wmr * 4 + i;
        const unsigned char* rp_ = (p.shortcut && x < p.W && y < p.H)
            ? p.x + ((img_pix + (long)y * p.W + x) * p.ldx + p.xoff + wnc * 64 + col) * 2 : zero_page;
        rs[i][v] = *reinterpret_cast<const gran_t*>(rp_);
      }
    BN128_HANDOVER()
#pragma unroll
    for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(rs[i][0])); asm volatile("" : "+v"(rs[i][1])); }
    BN128_STAMP(5)
    if (grp == 0) BN128_BARRIER()
    BN128_BARRIER()                                     // nobody reads the t patch any more: the strips may overwrite it

    // ---- epilogue: bias + SiLU -> strip -> (+ shortcut) -> 16-bit rows; strip i = tile row 4 wmr + i, 16 pixels x 64 channels
    if constexpr (ABL & 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(rs[i][0]), "v"(rs[i][1]));
    } else {
      float b2v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b2v[j] = sB1[C + wnc * 64 + j * 16 + lrow];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            stage[(lgrp * 4 + e) * SLD + j * 16 + lrow] = apply_act<CFT_ACT_SILU>(acc[i][j][e] + b2v[j]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const int it = lane + v * 64;
          const int row = it >> 3, col = (it & 7) * 8;
          const int x = x0 + row, y = y0 + wmr * 4 + i;
          if (x < p.W && y < p.H) {
            const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col);
            const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col + 4);
            float o[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
            if (p.shortcut) {
              float rf[8];
              Elem<T>::unpack(rs[i][v], rf);
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] += rf[e];
            }
            *reinterpret_cast<gran_t*>(p.y + ((img_pix + (long)y * p.W + x) * p.ldy + p.yoff + wnc * 64 + col) * 2) = Elem<T>::pack(o);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BN128_STAMP(6)
    BN128_BARRIER()                                     // strips are dead before the next tile's t patch is written
    BN128_STAMP(7)
  }
#undef BN128_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero-page K tiles staged past the last one
#undef BN128_STAGE
#undef BN128_BARRIER
#undef BN128_FETCH_X
#undef BN128_HANDOVER
}

// ------------------------------------------------------------------------------------ 128 channels, two workgroups per CU
// The same fused Bottleneck with the OTHER answer to its latency exposure (profiles/r02_bottleneck128.md): instead of one
// persistent workgroup per CU that prefetches across tiles, TWO co-resident workgroups per CU (80 KiB LDS each, 8 x 16-pixel
// tiles: 48 KiB t patch + two 16-KiB ring buffers), plain lock-step K loop (stage K tile c + 1, compute K tile c, barrier):
// one workgroup's x loads / t write / epilogue / DMA waits run under the other's MFMAs - the mechanism that makes the
// 192 x 128 tile the best 128-wide GEMM tile.  Wave (wmr, wnc) owns tile rows 2 wmr, 2 wmr + 1 x 64 channels.
// Biases live in the 12 spare rows of plane 1 of the t patch (patch pixels 180..191 do not exist).
template <typename T, int NW, int ABL = 0>   // NW = waves per workgroup: 8 (wave tile 2 rows x 64 ch) or 4 (4 rows x 64 ch: 0.5 instead of
                                             // 0.75 fragment reads per MFMA - with 16 waves per CU the 8-wave form is LDS-read bound)
__global__ void __launch_bounds__(64 * NW, NW / 2) bottleneck128b_kernel(const Bneck128Params p) {
  constexpr int C = 128, TH = 8, TW = 16, PW = TW + 2, PH = TH + 2, NPIX = PW * PH;   // 180 patch pixels
  constexpr int NRT = (NPIX + 15) / 16;                             // 12 row tiles of the patch
  constexpr int PLANE = NRT * 16 * 128;                             // 24576 B
  constexpr int RING = 16384;
  constexpr int NKT = 4 + 18;
  constexpr int SLD = 64 + 4;
  constexpr int NTHR = 64 * NW, RW = 16 / NW, RT1 = (NRT + NW - 1) / NW, SPT = 1024 / NTHR, SROWS = NTHR / 8;
  static_assert(NW == 8 || NW == 4, "8 or 4 waves");
  static_assert(PW == 18, "the mul-shift below divides by 18");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sT = smem;
  unsigned char* sR = smem + 2 * PLANE;
  float* sB = reinterpret_cast<float*>(smem + PLANE + NPIX * 128);   // [256]: b1 then b2, in the spare rows of plane 1

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int wmr = wave >> 1, wnc = wave & 1;      // 3x3 loop: tile rows RW wmr .. + RW, channels 64 wnc .. + 64
  const int tiles = p.tiles_x * p.tiles_y;
  // XCD-aware tile assignment (bijective for any grid size): workgroup ids go round-robin over the 8 XCDs, so consecutive
  // LOGICAL tiles - spatial neighbours that share halo pixels - are given to one XCD and meet in its L2
  const int nb_ = gridDim.x, bid_ = blockIdx.x;
  const int xq_ = nb_ >> 3, xr_ = nb_ & 7, xcd_ = bid_ & 7, xslot_ = bid_ >> 3;
  const int ltile = (xcd_ < xr_ ? xcd_ * (xq_ + 1) : xr_ * (xq_ + 1) + (xcd_ - xr_) * xq_) + xslot_;
  const int b = ltile / tiles, tt = ltile - b * tiles;
  const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const long img_pix = (long)b * p.H * p.W;
  const uint32_t t_lds = (uint32_t)(uintptr_t)(lds_void_t*)sT;

  // x fragments: row tile rt = wave + NW it
  gran_t a1[RT1][4];
#pragma unroll
  for (int it = 0; it < RT1; ++it) {
    const int q = (wave + it * NW) * 16 + lrow;
    const int py = (q * 3641) >> 16, px = q - py * PW;
    const int zy = y0 - 1 + py, zx = x0 - 1 + px;
    const bool in_ = q < NPIX && (unsigned)zy < (unsigned)p.H && (unsigned)zx < (unsigned)p.W;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      gran_t t = {0u, 0u, 0u, 0u};
      if (in_) t = *reinterpret_cast<const gran_t*>(p.x + ((img_pix + (long)zy * p.W + zx) * p.ldx + p.xoff + ks * 32 + lgrp * 8) * 2);
      a1[it][ks] = t;
    }
  }
  if (tid < C) sB[tid] = p.b1 != nullptr ? p.b1[tid] : 0.0f;
  else if (tid < 2 * C) sB[tid] = p.b2 != nullptr ? p.b2[tid - C] : 0.0f;

  const int r0 = tid >> 3, slot_s = tid & 7, g = slot_s ^ (r0 & 7);
#define BNB_STAGE(kt_)                                                                                   \
  _Pragma("unroll") for (int part_ = 0; part_ < SPT; ++part_) {                                          \
    const int n_ = r0 + SROWS * part_;                                                                   \
    const unsigned char* src_ = (kt_) < 4 ? p.w1 + ((long)n_ * p.kpad1 + ((kt_) & 1) * 64 + g * 8) * 2   \
                                          : p.w2 + ((long)n_ * p.kpad2 + ((kt_) - 4) * 64 + g * 8) * 2;  \
    if constexpr (!(ABL & 8))                                                                            \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src_, (lds_void_t*)(sR + ((kt_) & 1) * RING + part_ * (SROWS * 128) + wave * 1024), 16, 0, 0); \
  }
  BNB_STAGE(0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int fb = lrow * 128 + ((lgrp ^ (lrow & 7)) << 4);
  const int fbn = (wnc * 64 + lrow) * 128 + ((lgrp ^ (lrow & 7)) << 4);

  // ---- t^T = W1 x^T: output channels 0-63 / 64-127 (K tiles 0,1 / 2,3), then bias + SiLU -> t patch
#pragma unroll
  for (int jh = 0; jh < 2; ++jh) {
    f32x4_t acc1[RT1][4];
#pragma unroll
    for (int it = 0; it < RT1; ++it)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc1[it][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int c = jh * 2 + kt;
      BNB_STAGE(c + 1)
      gran_t wf[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wf[j][0] = *reinterpret_cast<const gran_t*>(sR + (c & 1) * RING + (jh * 4 + j) * 2048 + fb);
        wf[j][1] = *reinterpret_cast<const gran_t*>(sR + (c & 1) * RING + (jh * 4 + j) * 2048 + (fb ^ 64));
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int it = 0; it < RT1; ++it)
          if (wave + it * NW < NRT) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if constexpr (!(ABL & 1)) acc1[it][j] = mma_granule<T>(wf[j][ks], a1[it][kt * 2 + ks], acc1[it][j]);
          }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // K tile c + 1 landed; this tile's fragment reads retired
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int it = 0; it < RT1; ++it) {
      const int q = (wave + it * NW) * 16 + lrow;
      if (wave + it * NW < NRT && q < NPIX) {
        const int py = (q * 3641) >> 16, px = q - py * PW;
        const uint32_t keep = ((unsigned)(y0 - 1 + py) < (unsigned)p.H && (unsigned)(x0 - 1 + px) < (unsigned)p.W) ? 0xffffffffu : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cc = j * 16 + lgrp * 4;
          const f32x4_t b1q = *reinterpret_cast<const f32x4_t*>(sB + jh * 64 + cc);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (ABL & 1) ? 0.0f : apply_act<CFT_ACT_SILU>(acc1[it][j][e] + b1q[e]);
          uint2 w;
          w.x = Elem<T>::pack2(v[0], v[1]) & keep;
          w.y = Elem<T>::pack2(v[2], v[3]) & keep;
          const uint32_t ta = t_lds + jh * PLANE + q * 128 + ((((cc >> 3) ^ (q & 7)) << 4) | ((cc & 7) << 1));
          const unsigned long long wq = ((unsigned long long)w.y << 32) | w.x;
          asm volatile("ds_write_b64 %0, %1" ::"v"(ta), "v"(wq) : "memory");
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                       // the whole t patch is visible

  // ---- 3x3 conv of the t patch
  f32x4_t acc[RW][4];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int kk = 0; kk < 18; ++kk) {
    const int c = kk + 4;
    if (kk + 1 < 18) BNB_STAGE(c + 1)
    const int tap = kk >> 1, half = kk & 1;
    const int kh = tap / 3, kw = tap - kh * 3;
    const int qb = (wmr * RW + kh) * PW + kw + lrow;
    gran_t af[RW][2], bf[4][2];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int q = qb + i * PW;
      const unsigned char* rowp = sT + half * PLANE + q * 128;
      af[i][0] = *reinterpret_cast<const gran_t*>(rowp + ((lgrp ^ (q & 7)) << 4));
      af[i][1] = *reinterpret_cast<const gran_t*>(rowp + (((4 + lgrp) ^ (q & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf[j][0] = *reinterpret_cast<const gran_t*>(sR + (c & 1) * RING + j * 2048 + fbn);
      bf[j][1] = *reinterpret_cast<const gran_t*>(sR + (c & 1) * RING + j * 2048 + (fbn ^ 64));
    }
    if constexpr (ABL & 16) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (ABL & 2) { asm volatile("" ::"v"(af[i][ks]), "v"(bf[j][ks])); }
          else acc[i][j] = mma_granule<T>(af[i][ks], bf[j][ks], acc[i][j]);
        }
    if constexpr (ABL & 16) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#undef BNB_STAGE

  // ---- epilogue: strip i = tile row RW wmr + i, 16 pixels x 64 channels
  if constexpr (ABL & 4) {
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  float b2v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b2v[j] = sB[C + wnc * 64 + j * 16 + lrow];
  gran_t rs[RW][2];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int it = lane + v * 64;
      const int row = it >> 3, col = (it & 7) * 8;
      const int x = x0 + row, y = y0 + wmr * RW + i;
      gran_t t = {0u, 0u, 0u, 0u};
      if (p.shortcut && x < p.W && y < p.H)
        t = *reinterpret_cast<const gran_t*>(p.x + ((img_pix + (long)y * p.W + x) * p.ldx + p.xoff + wnc * 64 + col) * 2);
      rs[i][v] = t;
    }
  __builtin_amdgcn_s_barrier();                       // (after b2v was read) the strips may overwrite the t patch; the bias rows lie beyond them
  float* stage = reinterpret_cast<float*>(sT) + wave * (16 * SLD);
#pragma unroll
  for (int i = 0; i < RW; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        stage[(lgrp * 4 + e) * SLD + j * 16 + lrow] = apply_act<CFT_ACT_SILU>(acc[i][j][e] + b2v[j]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int it = lane + v * 64;
      const int row = it >> 3, col = (it & 7) * 8;
      const int x = x0 + row, y = y0 + wmr * RW + i;
      if (x < p.W && y < p.H) {
        const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col);
        const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col + 4);
        float o[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        if (p.shortcut) {
          float rf[8];
          Elem<T>::unpack(rs[i][v], rf);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += rf[e];
        }
        *reinterpret_cast<gran_t*>(p.y + ((img_pix + (long)y * p.W + x) * p.ldy + p.yoff + wnc * 64 + col) * 2) = Elem<T>::pack(o);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ------------------------------------------------------------------------------------ 128 channels, two per CU, 4-slot ring
// bottleneck128b_kernel is bound by the latency of its weight stream: one 16-KiB K tile of lookahead (issued at the head of
// a K tile, needed at its tail) against 0.21 us of MFMA work per K tile.  Same tile, same LDS budget (48 KiB t patch +
// 32 KiB ring), same products in the same order, but
//   * the ring is FOUR slots of 8 KiB: a stage is 32 k wide (W2: 128 rows x 64 B, slot = k-granule ^ h(row / 4)) and is
//     issued three stages ahead with counted s_waitcnt vmcnt(2); W1 streams once (four stages of 64 rows x 128 B, one per
//     output-channel half and k half);
//   * the 3x3 loop is software-pipelined by hand: step s READS the fragments of stage s and runs the MFMAs of step s - 1
//     under those reads;
//   * the shortcut pixels are requested right after the LAST stage: vmcnt is in-order, so an earlier request would be
//     waited for together with the next stage, three steps later;
//   * the patch's 12 row tiles are split evenly: wave w owns row tile w in both output-channel passes of the W1 stage and
//     row tile 8 + w / 2 in ONE of them (the two waves of a SIMD in different ones): 8 x requests per lane for every wave.
// (A persistent form - 512 workgroups walking the tiles, the next tile's x fragments and W1 stages requested during the
// epilogue - was built and measured: 156 vs 158 us.  The x requests cost their bandwidth, not their latency; the form was
// dropped.  What it taught about hipcc is in profiles/r02_bottleneck128.md section 3.)
template <typename T, int ABL = 0>
__global__ void __launch_bounds__(512, 4) bottleneck128c_kernel(const Bneck128Params p) {
  constexpr int C = 128, TH = 8, TW = 16, PW = TW + 2, PH = TH + 2, NPIX = PW * PH;   // 180 patch pixels
  constexpr int NRT = (NPIX + 15) / 16;                             // 12 row tiles of the patch
  constexpr int PLANE = NRT * 16 * 128;                             // 24576 B
  constexpr int SLOT = 8192;
  constexpr int SLD = 64 + 4;
  constexpr int RW = 2;
  constexpr int XS = (ABL & 4) ? 0 : 2 * RW;                        // shortcut requests per lane
  static_assert(PW == 18 && NRT == 12, "the mul-shift below divides by 18; 8 + 4 row tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sT = smem;
  unsigned char* sR = smem + 2 * PLANE;
  float* sB = reinterpret_cast<float*>(smem + PLANE + NPIX * 128);   // [256]: b1 then b2, in the spare rows of plane 1

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wmr = wave >> 1, wnc = wave & 1;      // 3x3 loop: tile rows 2 wmr, 2 wmr + 1, channels 64 wnc .. + 64
  const int tiles = p.tiles_x * p.tiles_y;
  const int rt1 = 8 + (wave >> 1), hp = (wave & 1) ^ (wave >> 2);   // second W1-stage unit: row tile rt1 in pass hp (the two waves of a SIMD take different passes)

  // stage s: 0..3 = W1 rows 64 (s >> 1) .. + 64, k 64 (s & 1) .. + 64 (64 rows x 128 B); 4 + u = stage image u of W2 (128 rows x 64 B)
  const int tid = threadIdx.x, lane = tid & 63;
  const int lrow = lane & 15, lgrp = lane >> 4;
  const uint32_t tl = (uint32_t)(uintptr_t)(lds_void_t*)sT;
  const int r1 = tid >> 3, g1 = (tid & 7) ^ (r1 & 7);
  const unsigned char* src1 = p.w1 + ((long)r1 * p.kpad1 + g1 * 8) * 2;
  const unsigned char* src2 = p.w2s + tid * 16;           // stage images: a linear copy, 1 KiB contiguous per wave
  const int fb = lrow * 128 + ((lgrp ^ (lrow & 7)) << 4);
  const int fb2 = (wnc * 64 + lrow) * 64 + ((lgrp ^ ((0x1320 >> (4 * ((lrow >> 2) & 3))) & 3)) << 4);
#define BNC_STAGE(s_)                                                                                    \
  {                                                                                                      \
    const int ss_ = (s_);                                                                                \
    const unsigned char* src_ = ss_ < 4 ? src1 + ((long)(ss_ >> 1) * 64 * p.kpad1 + (ss_ & 1) * 64) * 2  \
                                        : src2 + (long)(ss_ - 4) * SLOT;                                 \
    if constexpr (!(ABL & 8))                                                                            \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src_, (lds_void_t*)(sR + (ss_ & 3) * SLOT + wave * 1024), 16, 0, 0); \
  }
#define BNC_SYNC(n_)                                                                                     \
  {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    __builtin_amdgcn_s_waitcnt((n_) | 0x70);              /* vmcnt(n) lgkmcnt(0): a builtin, so the compiler's counter model sees it */ \
    __builtin_amdgcn_s_barrier();                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  }
  // XCD-aware tile assignment (bijective for any grid size): workgroup ids go round-robin over the 8 XCDs, so consecutive
  // LOGICAL tiles - spatial neighbours that share halo pixels - are given to one XCD and meet in its L2
  const int nb_ = gridDim.x, bid_ = blockIdx.x;
  const int xq_ = nb_ >> 3, xr_ = nb_ & 7, xcd_ = bid_ & 7, xslot_ = bid_ >> 3;
  const int ltile = (xcd_ < xr_ ? xcd_ * (xq_ + 1) : xr_ * (xq_ + 1) + (xcd_ - xr_) * xq_) + xslot_;
  const int tile = ltile;
  const int b = tile / tiles, tt = tile - b * tiles;
  const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const long img_pix = (long)b * p.H * p.W;
  // x fragments: unconditional requests at clamped coordinates, masked below (row tile `wave`, row tile rt1)
  gran_t a1n[2][4];
  {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int q = (it == 0 ? wave : rt1) * 16 + lrow;
      const int py = (q * 3641) >> 16, px = q - py * PW;
      const int zy = min(max(y0 - 1 + py, 0), p.H - 1), zx = min(max(x0 - 1 + px, 0), p.W - 1);
      const unsigned char* xp = p.x + ((img_pix + (long)zy * p.W + zx) * p.ldx + p.xoff + lgrp * 8) * 2;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if constexpr (ABL & 256) a1n[it][ks] = gran_t{0x3c003c00u + (unsigned)lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};   // probe: no x requests
        else a1n[it][ks] = *reinterpret_cast<const gran_t*>(xp + ks * 64);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    BNC_STAGE(0)
    BNC_STAGE(1)
    BNC_STAGE(2)
    __builtin_amdgcn_sched_barrier(0);
    // biases last: their LDS store makes the compiler drain vmcnt (an LDS-DMA request is pending), which is what the first
    // step needs anyway - x, the biases and stages 0-2 land together instead of one after the other
    float bq = 0.0f;
    if (tid < C) { if (p.b1 != nullptr) bq = p.b1[tid]; }
    else if (tid < 2 * C) { if (p.b2 != nullptr) bq = p.b2[tid - C]; }
    if (tid < 2 * C) sB[tid] = bq;
  }
  const int hp_t = hp;
  {
    BNC_SYNC(0)
    // ---- hand-over: this tile's x fragments, zero outside the image / the patch
    uint32_t keep[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int q = (it == 0 ? wave : rt1) * 16 + lrow;
      const int py = (q * 3641) >> 16, px = q - py * PW;
      keep[it] = (q < NPIX && (unsigned)(y0 - 1 + py) < (unsigned)p.H && (unsigned)(x0 - 1 + px) < (unsigned)p.W) ? 0xffffffffu : 0u;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        a1n[it][ks].x &= keep[it]; a1n[it][ks].y &= keep[it]; a1n[it][ks].z &= keep[it]; a1n[it][ks].w &= keep[it];
      }
    }

    // ---- t^T = W1 x^T: output channels 0-63 (stages 0, 1), 64-127 (stages 2, 3), then bias + SiLU -> t patch
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
      f32x4_t acc1[2][4];
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[it][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int s = jh * 2 + kt;
        BNC_STAGE(s + 3)
        gran_t wf[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          wf[j][0] = *reinterpret_cast<const gran_t*>(sR + (s & 3) * SLOT + j * 2048 + fb);
          wf[j][1] = *reinterpret_cast<const gran_t*>(sR + (s & 3) * SLOT + j * 2048 + (fb ^ 64));
        }
        if constexpr (!(ABL & 1)) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[0][j] = mma_granule<T>(wf[j][ks], a1n[0][kt * 2 + ks], acc1[0][j]);
          if (hp_t == jh) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc1[1][j] = mma_granule<T>(wf[j][ks], a1n[1][kt * 2 + ks], acc1[1][j]);
          }
        }
        BNC_SYNC(2)                                      // stage s + 1 landed (s + 2, s + 3 in flight); this slot's reads retired
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int q = (it == 0 ? wave : rt1) * 16 + lrow;
        if ((it == 0 || hp_t == jh) && q < NPIX) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int cc = j * 16 + lgrp * 4;
            const f32x4_t b1q = *reinterpret_cast<const f32x4_t*>(sB + jh * 64 + cc);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (ABL & 1) ? 0.0f : apply_act<(ABL & 512) ? CFT_ACT_NONE : CFT_ACT_SILU>(acc1[it][j][e] + b1q[e]);
            uint2 w;
            w.x = Elem<T>::pack2(v[0], v[1]) & keep[it];
            w.y = Elem<T>::pack2(v[2], v[3]) & keep[it];
            const uint32_t ta = tl + jh * PLANE + q * 128 + ((((cc >> 3) ^ (q & 7)) << 4) | ((cc & 7) << 1));
            const unsigned long long wq = ((unsigned long long)w.y << 32) | w.x;
            asm volatile("ds_write_b64 %0, %1" ::"v"(ta), "v"(wq) : "memory");
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // the whole t patch is visible

    // ---- 3x3 conv of the t patch: step (tap, kq) = stage 4 + 4 tap + kq in slot kq; kq = plane * 2 + k half of the plane
    f32x4_t acc[RW][4];
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // Software pipeline: step s issues stage s + 3, READS the fragments of stage s into register set kq & 1 and runs the MFMAs
    // of step s - 1 from the other set under those reads; the counted wait + barrier at its end publishes stage s + 1.
    gran_t af[2][RW], bf[2][4];
#pragma unroll
    for (int i = 0; i < RW; ++i) af[1][i] = gran_t{0u, 0u, 0u, 0u};    // "step -1": zero products, the accumulators stay 0
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[1][j] = gran_t{0u, 0u, 0u, 0u};
#define BNC_READ(kq_)                                                                                    \
    {                                                                                                    \
      _Pragma("unroll") for (int i = 0; i < RW; ++i) {                                                   \
        const int q = qb + i * PW;                                                                       \
        af[(kq_) & 1][i] = *reinterpret_cast<const gran_t*>(sT + ((kq_) >> 1) * PLANE + q * 128 + (((((kq_) & 1) * 4 + lgrp) ^ (q & 7)) << 4)); \
      }                                                                                                  \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                      \
        bf[(kq_) & 1][j] = *reinterpret_cast<const gran_t*>(sR + (kq_) * SLOT + j * 1024 + fb2);         \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
#define BNC_MMA(set_)                                                                                    \
    {                                                                                                    \
      _Pragma("unroll") for (int i = 0; i < RW; ++i)                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                  \
          if constexpr (ABL & 2) { asm volatile("" ::"v"(af[set_][i]), "v"(bf[set_][j])); }              \
          else acc[i][j] = mma_granule<T>(af[set_][i], bf[set_][j], acc[i][j]);                          \
        }                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
    gran_t rs[RW][2];
#define BNC_FETCH_RS()                                                                                   \
    _Pragma("unroll") for (int i = 0; i < RW; ++i)                                                       \
      _Pragma("unroll") for (int v = 0; v < 2; ++v) {                                                    \
        const int it = lane + v * 64;                                                                    \
        const int row = it >> 3, col = (it & 7) * 8;                                                     \
        const int x = min(x0 + row, p.W - 1), y = min(y0 + wmr * RW + i, p.H - 1);                       \
        if constexpr (XS != 0)                                                                           \
          rs[i][v] = *reinterpret_cast<const gran_t*>(p.x + ((img_pix + (long)y * p.W + x) * p.ldx + p.xoff + wnc * 64 + col) * 2); \
        else                                                                                             \
          rs[i][v] = gran_t{0u, 0u, 0u, 0u};                                                             \
      }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int tap = 0; tap < 8; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      const int qb = (wmr * RW + kh) * PW + kw + lrow;
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        BNC_STAGE(4 + tap * 4 + kq + 3)
        BNC_READ(kq)
        BNC_MMA((kq + 1) & 1)
        BNC_SYNC(2)
      }
    }
    {
      const int qb = (wmr * RW + 2) * PW + 2 + lrow;     // tap 8: kh = kw = 2
      BNC_STAGE(39)
      // shortcut pixels: unconditional (clamped address) so that every wave has exactly XS more requests in flight
      BNC_FETCH_RS()
      __builtin_amdgcn_sched_barrier(0);
      BNC_READ(0)
      BNC_MMA(1)
      BNC_SYNC(2 + XS)                                   // stage 37 landed; 38, 39 and the four shortcut requests may be in flight
      BNC_READ(1)
      BNC_MMA(0)
      BNC_SYNC(1 + XS)
      BNC_READ(2)
      BNC_MMA(1)
      BNC_SYNC(XS)
      BNC_READ(3)
      BNC_MMA(0)
      BNC_MMA(1)
    }
#undef BNC_READ
#undef BNC_MMA
#undef BNC_FETCH_RS

    // ---- epilogue: strip i = tile row 2 wmr + i, 16 pixels x 64 channels
    if constexpr (ABL & 4) {
#pragma unroll
      for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else {
      float b2v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b2v[j] = sB[C + wnc * 64 + j * 16 + lrow];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                     // every wave is past its last t read: the strips may overwrite the patch
      float* stage = reinterpret_cast<float*>(sT) + wave * (16 * SLD);
#pragma unroll
      for (int i = 0; i < RW; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            stage[(lgrp * 4 + e) * SLD + j * 16 + lrow] = apply_act<(ABL & 512) ? CFT_ACT_NONE : CFT_ACT_SILU>(acc[i][j][e] + b2v[j]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const int it = lane + v * 64;
          const int row = it >> 3, col = (it & 7) * 8;
          const int x = x0 + row, y = y0 + wmr * RW + i;
          if (x < p.W && y < p.H) {
            const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col);
            const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col + 4);
            float o[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
            if (p.shortcut) {
              float rf[8];
              Elem<T>::unpack(rs[i][v], rf);
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] += rf[e];
            }
            *reinterpret_cast<gran_t*>(p.y + ((img_pix + (long)y * p.W + x) * p.ldy + p.yoff + wnc * 64 + col) * 2) = Elem<T>::pack(o);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
#undef BNC_STAGE
#undef BNC_SYNC
}

// ------------------------------------------------------------------------------------ 64 channels, two per CU, 4-slot ring
// The recipe of bottleneck128c_kernel for the 64-channel stage (160 x 160 maps): the activation patch of a 16 x 16-pixel tile
// stays in LDS (18 x 18 x 128 B = 42 KiB, one 64-channel plane) and the weights stream - W1 as one stage, W2 as one stage per
// tap (64 rows x 128 B = 8 KiB, straight from the cft_conv2d layout: a row's 64 k are one cache line) - through a 4-slot
// ring three stages ahead.  42 + 32 KiB: two workgroups per CU, where bottleneck_kernel (3x3 weights resident, 117 KiB) has
// one whose 16 waves run their SiLU passes and their MFMAs in lock step.  Wave w owns tile rows 2 w, 2 w + 1 and all 64
// channels; per tap it reads 4 + 8 fragments for 16 MFMAs, the second k half under the MFMAs of the first.
template <typename T, int ABL = 0>
__global__ void __launch_bounds__(512, 4) bottleneck64r_kernel(const Bneck128Params p) {
  constexpr int C = 64, TS = 16, PW = TS + 2, NPIX = PW * PW;       // 324 patch pixels
  constexpr int NRT = (NPIX + 15) / 16;                             // 21 row tiles of the patch
  constexpr int PATCH = NRT * 16 * 128;                             // 43008 B
  constexpr int SLOT = 8192;
  constexpr int SLD = 64 + 4;
  constexpr int RW = 2, RT1 = 3;
  static_assert(PW == 18, "the mul-shift below divides by 18");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sT = smem;
  unsigned char* sR = smem + PATCH;
  float* sB = reinterpret_cast<float*>(smem + NPIX * 128);           // [128]: b1 then b2, in the spare rows of the patch

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int tiles = p.tiles_x * p.tiles_y;
  // XCD-aware tile assignment (bijective for any grid size): workgroup ids go round-robin over the 8 XCDs, so consecutive
  // LOGICAL tiles - spatial neighbours that share halo pixels - are given to one XCD and meet in its L2
  const int nb_ = gridDim.x, bid_ = blockIdx.x;
  const int xq_ = nb_ >> 3, xr_ = nb_ & 7, xcd_ = bid_ & 7, xslot_ = bid_ >> 3;
  const int ltile = (xcd_ < xr_ ? xcd_ * (xq_ + 1) : xr_ * (xq_ + 1) + (xcd_ - xr_) * xq_) + xslot_;
  const int b = ltile / tiles, tt = ltile - b * tiles;
  const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
  const int y0 = ty * TS, x0 = tx * TS;
  const long img_pix = (long)b * p.H * p.W;
  const uint32_t tl = (uint32_t)(uintptr_t)(lds_void_t*)sT;

  // stage 0 = W1, stage 1 + tap = W2[:, 64 tap .. 64 tap + 63]: 64 rows x 128 B, k-granule g of row r in slot g ^ (r & 7)
  const int r1 = tid >> 3, g1 = (tid & 7) ^ (r1 & 7);
  const unsigned char* src1 = p.w1 + ((long)r1 * p.kpad1 + g1 * 8) * 2;
  const unsigned char* src2 = p.w2 + ((long)r1 * p.kpad2 + g1 * 8) * 2;
#define BNR_STAGE(s_)                                                                                    \
  {                                                                                                      \
    const int ss_ = (s_);                                                                                \
    const unsigned char* src_ = ss_ == 0 ? src1 : src2 + (long)(ss_ - 1) * 128;                          \
    if constexpr (!(ABL & 8))                                                                            \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src_, (lds_void_t*)(sR + (ss_ & 3) * SLOT + wave * 1024), 16, 0, 0); \
  }
#define BNR_SYNC(n_)                                                                                     \
  {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    __builtin_amdgcn_s_waitcnt((n_) | 0x70);              /* vmcnt(n) lgkmcnt(0) */                      \
    __builtin_amdgcn_s_barrier();                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  }
  const int fb = lrow * 128 + ((lgrp ^ (lrow & 7)) << 4);

  // x fragments: row tiles wave, wave + 8, wave + 16 (the last one for waves 0-4), clamped coordinates, masked below
  gran_t a1[RT1][2];
  uint32_t keep[RT1];
#pragma unroll
  for (int it = 0; it < RT1; ++it) {
    const int q = (wave + 8 * it) * 16 + lrow;
    const int py = (q * 3641) >> 16, px = q - py * PW;
    const int zy = min(max(y0 - 1 + py, 0), p.H - 1), zx = min(max(x0 - 1 + px, 0), p.W - 1);
    keep[it] = (q < NPIX && (unsigned)(y0 - 1 + py) < (unsigned)p.H && (unsigned)(x0 - 1 + px) < (unsigned)p.W) ? 0xffffffffu : 0u;
    const unsigned char* xp = p.x + ((img_pix + (long)zy * p.W + zx) * p.ldx + p.xoff + lgrp * 8) * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      gran_t t = {0u, 0u, 0u, 0u};
      if (wave + 8 * it < NRT) t = *reinterpret_cast<const gran_t*>(xp + ks * 64);
      a1[it][ks] = t;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  BNR_STAGE(0)
  BNR_STAGE(1)
  BNR_STAGE(2)
  __builtin_amdgcn_sched_barrier(0);
  {   // biases last: their LDS store makes the compiler drain vmcnt, which is what the first step needs anyway
    float bq = 0.0f;
    if (tid < C) { if (p.b1 != nullptr) bq = p.b1[tid]; }
    else if (tid < 2 * C) { if (p.b2 != nullptr) bq = p.b2[tid - C]; }
    if (tid < 2 * C) sB[tid] = bq;
  }
  BNR_SYNC(0)
#pragma unroll
  for (int it = 0; it < RT1; ++it)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a1[it][ks].x &= keep[it]; a1[it][ks].y &= keep[it]; a1[it][ks].z &= keep[it]; a1[it][ks].w &= keep[it];
    }

  // ---- t^T = W1 x^T (stage 0), bias + SiLU -> t patch
  {
    f32x4_t acc1[RT1][4];
#pragma unroll
    for (int it = 0; it < RT1; ++it)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc1[it][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    BNR_STAGE(3)
    gran_t wf[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wf[j][0] = *reinterpret_cast<const gran_t*>(sR + j * 2048 + fb);
      wf[j][1] = *reinterpret_cast<const gran_t*>(sR + j * 2048 + (fb ^ 64));
    }
    if constexpr (!(ABL & 1)) {
#pragma unroll
      for (int it = 0; it < RT1; ++it)
        if (wave + 8 * it < NRT) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[it][j] = mma_granule<T>(wf[j][ks], a1[it][ks], acc1[it][j]);
        }
    }
    BNR_SYNC(2)                                          // stage 1 landed (2, 3 in flight); slot 0's reads retired
#pragma unroll
    for (int it = 0; it < RT1; ++it) {
      const int q = (wave + 8 * it) * 16 + lrow;
      if (wave + 8 * it < NRT && q < NPIX) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cc = j * 16 + lgrp * 4;
          const f32x4_t b1q = *reinterpret_cast<const f32x4_t*>(sB + cc);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (ABL & 1) ? 0.0f : apply_act<CFT_ACT_SILU>(acc1[it][j][e] + b1q[e]);
          uint2 w;
          w.x = Elem<T>::pack2(v[0], v[1]) & keep[it];
          w.y = Elem<T>::pack2(v[2], v[3]) & keep[it];
          const uint32_t ta = tl + q * 128 + ((((cc >> 3) ^ (q & 7)) << 4) | ((cc & 7) << 1));
          const unsigned long long wq = ((unsigned long long)w.y << 32) | w.x;
          asm volatile("ds_write_b64 %0, %1" ::"v"(ta), "v"(wq) : "memory");
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                         // the whole t patch is visible

  // ---- 3x3 conv of the t patch: step tap = stage 1 + tap in slot (1 + tap) & 3, two k halves per step
  f32x4_t acc[RW][4];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  gran_t af[2][RW], bf[2][4];
#pragma unroll
  for (int i = 0; i < RW; ++i) af[1][i] = gran_t{0u, 0u, 0u, 0u};      // "step -1": zero products
#pragma unroll
  for (int j = 0; j < 4; ++j) bf[1][j] = gran_t{0u, 0u, 0u, 0u};
#define BNR_READ(slot_, ks_)                                                                             \
  {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < RW; ++i) {                                                     \
      const int q = qb + i * PW;                                                                         \
      af[ks_][i] = *reinterpret_cast<const gran_t*>(sT + q * 128 + ((((ks_) * 4 + lgrp) ^ (q & 7)) << 4)); \
    }                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
      bf[ks_][j] = *reinterpret_cast<const gran_t*>(sR + (slot_) * SLOT + j * 2048 + ((ks_) ? (fb ^ 64) : fb)); \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  }
#define BNR_MMA(set_)                                                                                    \
  {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < RW; ++i)                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                    \
        if constexpr (ABL & 2) { asm volatile("" ::"v"(af[set_][i]), "v"(bf[set_][j])); }                \
        else acc[i][j] = mma_granule<T>(af[set_][i], bf[set_][j], acc[i][j]);                            \
      }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  }
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int kh = tap / 3, kw = tap - kh * 3;
    const int qb = (wave * RW + kh) * PW + kw + lrow;
    if (tap + 4 <= 9) BNR_STAGE(tap + 4)
    BNR_READ((1 + tap) & 3, 0)
    BNR_MMA(1)
    BNR_READ((1 + tap) & 3, 1)
    BNR_MMA(0)
    // stage 2 + tap must have landed; younger: stages 3 + tap, 4 + tap (while they exist)
    if (tap <= 5) { BNR_SYNC(2) }
    else if (tap == 6) { BNR_SYNC(1) }
    else if (tap == 7) { BNR_SYNC(0) }
  }
  BNR_MMA(1)
#undef BNR_READ
#undef BNR_MMA
#undef BNR_STAGE
#undef BNR_SYNC

  // ---- epilogue: strip i = tile row 2 wave + i, 16 pixels x 64 channels
  if constexpr (ABL & 4) {
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  float b2v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b2v[j] = sB[C + j * 16 + lrow];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                         // every wave is past its last t / ring read: strips and x centre may overwrite them
  // The shortcut is x at the tile's own pixels - which this workgroup already holds: the interior of the patch whose x
  // fragments fed the W1 stage.  They are parked in the (now dead) 32 KiB of the ring, pixel-major, instead of being read
  // from memory a second time (210 MB per launch at 160 x 160, a third of the kernel's HBM traffic).
  if (p.shortcut) {
    int lane_e;                                          // recomputed: carried from the prologue these indices would be spilled
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int lrow_e = lane_e & 15, lgrp_e = lane_e >> 4;
#pragma unroll
    for (int it = 0; it < RT1; ++it) {
      const int q = (wave + 8 * it) * 16 + lrow_e;
      const int py = (q * 3641) >> 16, px = q - py * PW;
      if (wave + 8 * it < NRT && py >= 1 && py <= TS && px >= 1 && px <= TS) {
        const int c = (py - 1) * TS + (px - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          *reinterpret_cast<gran_t*>(sR + c * 128 + (((ks * 4 + lgrp_e) ^ (c & 7)) << 4)) = a1[it][ks];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  float* stage = reinterpret_cast<float*>(sT) + wave * (16 * SLD);
#pragma unroll
  for (int i = 0; i < RW; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        stage[(lgrp * 4 + e) * SLD + j * 16 + lrow] = apply_act<CFT_ACT_SILU>(acc[i][j][e] + b2v[j]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int it = lane + v * 64;
      const int row = it >> 3, col = (it & 7) * 8;
      const int x = x0 + row, y = y0 + wave * RW + i;
      if (x < p.W && y < p.H) {
        const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col);
        const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col + 4);
        float o[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        if (p.shortcut) {
          const int c = (wave * RW + i) * TS + row;
          const gran_t rsv = *reinterpret_cast<const gran_t*>(sR + c * 128 + (((it & 7) ^ (c & 7)) << 4));
          float rf[8];
          Elem<T>::unpack(rsv, rf);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += rf[e];
        }
        *reinterpret_cast<gran_t*>(p.y + ((img_pix + (long)y * p.W + x) * p.ldy + p.yoff + col) * 2) = Elem<T>::pack(o);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// Stage-major image of the 3x3 weights for bottleneck128c_kernel: stage u (k = 32 u .. 32 u + 31 of every row) as the 8 KiB
// the kernel wants in an LDS slot - row n at n * 64 B, k-granule kg of the stage in 16-byte slot kg ^ h((n / 4) & 3).
__global__ void __launch_bounds__(256) bneck_pack_w2_kernel(const gran_t* __restrict__ w2, int kpad2, gran_t* __restrict__ out, int total) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int slot = i & 3, n = (i >> 2) & 127, u = i >> 9;
  const int kg = slot ^ ((0x1320 >> (4 * ((n >> 2) & 3))) & 3);
  out[i] = w2[(long)n * (kpad2 / 8) + u * 4 + kg];
}

extern "C" int cft_bottleneck_pack_w2(const void* w2, int kpad2, int c, void* w2_stages, int dtype, void* stream) {
  CFT_REQUIRE(w2 && w2_stages, "cft_bottleneck_pack_w2: null pointer");
  CFT_REQUIRE(dtype == CFT_BF16 || dtype == CFT_F16, "cft_bottleneck_pack_w2: dtype must be CFT_BF16 or CFT_F16");
  CFT_REQUIRE(c == 128 && kpad2 == 9 * 128, "cft_bottleneck_pack_w2: 128 channels, kpad2 = 1152");
  const int total = 36 * 128 * 4;
  hipLaunchKernelGGL(bneck_pack_w2_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream),
                     (const gran_t*)w2, kpad2, (gran_t*)w2_stages, total);
  return cft_check_launch("bneck_pack_w2_kernel");
}

// CFT_BNECK128=persistent selects the one-workgroup-per-CU kernel; the default is the two-per-CU kernel.
static bool bneck128_two_per_cu() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("CFT_BNECK128"); v = (e && e[0] == 'p') ? 0 : 1; }
  return v == 1;
}
// CFT_BNECK64=resident selects bottleneck_kernel (3x3 weights LDS-resident, one workgroup per CU) for the 64-channel stage.
static bool bneck64_ring() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("CFT_BNECK64"); v = (e && e[0] == 'r') ? 0 : 1; }
  return v == 1;
}
// CFT_BNECK128=b selects the 16-KiB-K-tile kernel (bottleneck128b_kernel) for A/B runs; the default is the 4-slot-ring kernel.
static bool bneck128_ring() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("CFT_BNECK128"); v = (e && e[0] == 'b') ? 0 : 1; }
  return v == 1;
}

extern "C" int cft_bottleneck(const void* x, int ldx, int xoff, const void* w1, int kpad1, const float* b1,
                              const void* w2, int kpad2, const void* w2_stages, const float* b2, void* y, int ldy, int yoff,
                              int B, int H, int W, int c, int shortcut, int dtype, void* stream) {
  CFT_REQUIRE(x && w1 && w2 && y, "cft_bottleneck: null pointer");
  CFT_REQUIRE(dtype == CFT_BF16 || dtype == CFT_F16, "cft_bottleneck: dtype must be CFT_BF16 or CFT_F16");
  CFT_REQUIRE(c == 64 || c == 128, "cft_bottleneck: the fused kernels cover 64 and 128 channels (use two cft_conv2d calls otherwise)");
  CFT_REQUIRE(B > 0 && H > 0 && W > 0, "cft_bottleneck: non-positive size");
  CFT_REQUIRE(kpad1 >= c && kpad1 % 64 == 0 && kpad2 >= 9 * c && kpad2 % 64 == 0, "cft_bottleneck: weights must be packed as for cft_conv2d");
  CFT_REQUIRE(ldx % 8 == 0 && xoff % 8 == 0 && ldy % 8 == 0 && yoff % 8 == 0 && ldx >= xoff + c && ldy >= yoff + c,
              "cft_bottleneck: ld/offset must be multiples of 8 and cover the channel slice");
  CFT_REQUIRE((long)B * H * W * ldx < (1L << 31) && (long)B * H * W * ldy < (1L << 31), "cft_bottleneck: tensor exceeds 2^31 elements");
  {   // the kernel reads a halo of x: the output may share a buffer with x only as a disjoint channel slice
    const char* xa = (const char*)x + (long)xoff * 2;
    const char* ya = (const char*)y + (long)yoff * 2;
    const long xbytes = (long)B * H * W * ldx * 2, ybytes = (long)B * H * W * ldy * 2;
    const long d = ya > xa ? ya - xa : xa - ya;
    const bool disjoint_mem = ya + ybytes <= xa || xa + xbytes <= ya;
    const bool disjoint_slice = ldx == ldy && d >= (long)c * 2 && d + (long)c * 2 <= (long)ldx * 2;   // same pixel grid, other channels
    CFT_REQUIRE(disjoint_mem || disjoint_slice, "cft_bottleneck: output overlaps the input (halo reads forbid in-place)");
  }
  if (c == 128 && bneck128_two_per_cu() && g_conv_variant != 9128 && g_conv_variant != 932) {   // variant 9128 / 932: the persistent kernel
    Bneck128Params q;
    q.x = (const unsigned char*)x; q.w1 = (const unsigned char*)w1; q.w2 = (const unsigned char*)w2;
    q.w2s = (const unsigned char*)w2_stages;
    q.b1 = b1; q.b2 = b2; q.y = (unsigned char*)y;
    q.ldx = ldx; q.xoff = xoff; q.ldy = ldy; q.yoff = yoff; q.kpad1 = kpad1; q.kpad2 = kpad2;
    q.H = H; q.W = W; q.tiles_x = (W + 15) / 16; q.tiles_y = (H + 7) / 8; q.shortcut = shortcut ? 1 : 0;
    q.ntiles = B * q.tiles_x * q.tiles_y; q.dbg = nullptr;
    CFT_REQUIRE((long)B * q.tiles_x * q.tiles_y < (1L << 31), "cft_bottleneck: too many tiles");
    constexpr int smemb = 2 * 12 * 16 * 128 + 2 * 16384;
    const dim3 gridb(q.ntiles);
    hipStream_t sb_ = as_stream(stream);
#define BNB_LAUNCH(T_, NW_, ABL_)                                                                       \
    {                                                                                                   \
      cft_allow_lds<&bottleneck128b_kernel<T_, NW_, ABL_>>(smemb);                                      \
      hipLaunchKernelGGL((bottleneck128b_kernel<T_, NW_, ABL_>), gridb, dim3(64 * NW_), smemb, sb_, q); \
    }
#define BNC_LAUNCH(T_, ABL_)                                                                            \
    {                                                                                                   \
      cft_allow_lds<&bottleneck128c_kernel<T_, ABL_>>(smemb);                                           \
      hipLaunchKernelGGL((bottleneck128c_kernel<T_, ABL_>), gridb, dim3(512), smemb, sb_, q);           \
    }
    // Default: the 4-slot-ring kernel when the caller supplies the stage-major weights, else the 16-KiB-K-tile kernel.
    // Variants: 9100 (or CFT_BNECK128=b) = the 16-KiB-K-tile kernel, 9004 = its 4-wave form, 92xx / 94xx / 97xx probes.
    const int var = g_conv_variant;
    const bool ring = q.w2s != nullptr && bneck128_ring() && var != 9100 && var != 9004 && !(var >= 901 && var <= 916) && !(var >= 9014 && var <= 9084);
    if (dtype == CFT_F16) {
      if (!ring) { if (var == 9004) BNB_LAUNCH(f16_t, 4, 0) else BNB_LAUNCH(f16_t, 8, 0) }
      else BNC_LAUNCH(f16_t, 0)
    } else if (ring) {
      switch (var) {
        case 9201: BNC_LAUNCH(uint16_t, 1) break;           // probes: no W1-stage MFMAs / no 3x3 MFMAs / no epilogue / no weight DMA
        case 9202: BNC_LAUNCH(uint16_t, 2) break;
        case 9204: BNC_LAUNCH(uint16_t, 4) break;
        case 9208: BNC_LAUNCH(uint16_t, 8) break;
        case 9456: BNC_LAUNCH(uint16_t, 256) break;         // no x requests / no SiLU
        case 9712: BNC_LAUNCH(uint16_t, 512) break;
        default: BNC_LAUNCH(uint16_t, 0) break;
      }
    } else {
      switch (var) {
        case 901: BNB_LAUNCH(uint16_t, 8, 1) break;
        case 902: BNB_LAUNCH(uint16_t, 8, 2) break;
        case 904: BNB_LAUNCH(uint16_t, 8, 4) break;
        case 908: BNB_LAUNCH(uint16_t, 8, 8) break;
        case 916: BNB_LAUNCH(uint16_t, 8, 16) break;
        case 9004: BNB_LAUNCH(uint16_t, 4, 0) break;        // four waves per workgroup, wave tile 4 rows x 64 channels
        case 9014: BNB_LAUNCH(uint16_t, 4, 1) break;
        case 9024: BNB_LAUNCH(uint16_t, 4, 2) break;
        case 9044: BNB_LAUNCH(uint16_t, 4, 4) break;
        case 9084: BNB_LAUNCH(uint16_t, 4, 8) break;
        default: BNB_LAUNCH(uint16_t, 8, 0) break;
      }
    }
#undef BNB_LAUNCH
#undef BNC_LAUNCH
    return cft_check_launch("bottleneck128b_kernel");
  }
  if (c == 128) {
    Bneck128Params q;
    q.x = (const unsigned char*)x; q.w1 = (const unsigned char*)w1; q.w2 = (const unsigned char*)w2;
    q.b1 = b1; q.b2 = b2; q.y = (unsigned char*)y;
    q.ldx = ldx; q.xoff = xoff; q.ldy = ldy; q.yoff = yoff; q.kpad1 = kpad1; q.kpad2 = kpad2;
    q.H = H; q.W = W; q.tiles_x = (W + 15) / 16; q.tiles_y = (H + 15) / 16; q.shortcut = shortcut ? 1 : 0;
    CFT_REQUIRE((long)B * q.tiles_x * q.tiles_y < (1L << 31), "cft_bottleneck: too many tiles");
    q.ntiles = B * q.tiles_x * q.tiles_y;
    q.dbg = g_bneck_dbg;
    constexpr int smem128 = 2 * 21 * 16 * 128 + 4 * 16384 + 1024;
    int cus = 256;
    {   // persistent: one workgroup per CU (148 KiB LDS each) walks tiles bid, bid + grid, ...
      static int cached = 0;
      if (!cached) { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cached = prop.multiProcessorCount; else cached = 256; }
      cus = cached;
    }
    const dim3 grid128(q.ntiles < cus ? q.ntiles : cus), block128(512);
    hipStream_t s128 = as_stream(stream);
#define BN128_LAUNCH(T_, ABL_)                                                                    \
    {                                                                                             \
      cft_allow_lds<&bottleneck128_kernel<T_, ABL_>>(smem128);                                    \
      hipLaunchKernelGGL((bottleneck128_kernel<T_, ABL_>), grid128, block128, smem128, s128, q);  \
    }
    if (dtype == CFT_F16) {
      BN128_LAUNCH(f16_t, 0)
    } else {
      switch (g_conv_variant) {   // timing probes: 1 = no t-patch MFMAs, 2 = no 3x3 MFMAs, 4 = no epilogue, 8 = no weight DMA
        case 901: BN128_LAUNCH(uint16_t, 1) break;
        case 902: BN128_LAUNCH(uint16_t, 2) break;
        case 904: BN128_LAUNCH(uint16_t, 4) break;
        case 908: BN128_LAUNCH(uint16_t, 8) break;
        case 932: BN128_LAUNCH(uint16_t, 32) break;
        default: BN128_LAUNCH(uint16_t, 0) break;
      }
    }
#undef BN128_LAUNCH
    return cft_check_launch("bottleneck128_kernel");
  }
  if (bneck64_ring() && g_conv_variant != 9640 && !(g_conv_variant >= 901 && g_conv_variant <= 907)) {   // 9640 / CFT_BNECK64=resident: the weights-resident kernel
    Bneck128Params q;
    q.x = (const unsigned char*)x; q.w1 = (const unsigned char*)w1; q.w2 = (const unsigned char*)w2; q.w2s = nullptr;
    q.b1 = b1; q.b2 = b2; q.y = (unsigned char*)y;
    q.ldx = ldx; q.xoff = xoff; q.ldy = ldy; q.yoff = yoff; q.kpad1 = kpad1; q.kpad2 = kpad2;
    q.H = H; q.W = W; q.tiles_x = (W + 15) / 16; q.tiles_y = (H + 15) / 16; q.shortcut = shortcut ? 1 : 0;
    CFT_REQUIRE((long)B * q.tiles_x * q.tiles_y < (1L << 31), "cft_bottleneck: too many tiles");
    q.ntiles = B * q.tiles_x * q.tiles_y; q.dbg = nullptr;
    constexpr int smemr = 21 * 16 * 128 + 4 * 8192;
    const dim3 gridr(q.ntiles);
    hipStream_t sr_ = as_stream(stream);
#define BNR_LAUNCH(T_, ABL_)                                                                            \
    {                                                                                                   \
      cft_allow_lds<&bottleneck64r_kernel<T_, ABL_>>(smemr);                                            \
      hipLaunchKernelGGL((bottleneck64r_kernel<T_, ABL_>), gridr, dim3(512), smemr, sr_, q);            \
    }
    if (dtype == CFT_F16) {
      BNR_LAUNCH(f16_t, 0)
    } else {
      switch (g_conv_variant) {   // probes: no W1-stage MFMAs / no 3x3 MFMAs / no epilogue / no weight DMA
        case 9601: BNR_LAUNCH(uint16_t, 1) break;
        case 9602: BNR_LAUNCH(uint16_t, 2) break;
        case 9604: BNR_LAUNCH(uint16_t, 4) break;
        case 9608: BNR_LAUNCH(uint16_t, 8) break;
        default: BNR_LAUNCH(uint16_t, 0) break;
      }
    }
#undef BNR_LAUNCH
    return cft_check_launch("bottleneck64r_kernel");
  }
  BneckParams p;
  p.x = (const unsigned char*)x; p.w1 = (const unsigned char*)w1; p.w2 = (const unsigned char*)w2;
  p.b1 = b1; p.b2 = b2; p.y = (unsigned char*)y;
  p.ldx = ldx; p.xoff = xoff; p.ldy = ldy; p.yoff = yoff; p.kpad1 = kpad1; p.kpad2 = kpad2;
  p.H = H; p.W = W; p.tiles_x = (W + 31) / 32; p.bands = (H + 7) / 8; p.shortcut = shortcut ? 1 : 0;
  constexpr int smem_bytes = 9 * 64 * 128 + 22 * 16 * 128;
  const dim3 grid(B * p.bands), block(512);
  hipStream_t s = as_stream(stream);
#define BNECK_LAUNCH(T_, ABL_)                                                             \
  {                                                                                        \
    cft_allow_lds<&bottleneck_kernel<T_, 4, ABL_>>(smem_bytes);                            \
    hipLaunchKernelGGL((bottleneck_kernel<T_, 4, ABL_>), grid, block, smem_bytes, s, p);   \
  }
  if (dtype == CFT_F16) {
    BNECK_LAUNCH(f16_t, 0)
  } else {
    switch (g_conv_variant) {   // timing probes (tools/bneck_bench.py)
      case 901: BNECK_LAUNCH(uint16_t, 1) break;
      case 902: BNECK_LAUNCH(uint16_t, 2) break;
      case 904: BNECK_LAUNCH(uint16_t, 4) break;
      case 907: BNECK_LAUNCH(uint16_t, 7) break;
      default: BNECK_LAUNCH(uint16_t, 0) break;
    }
  }
#undef BNECK_LAUNCH
  return cft_check_launch("bottleneck_kernel");
}
