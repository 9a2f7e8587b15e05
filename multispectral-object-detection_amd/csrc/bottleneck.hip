// Bottleneck as ONE kernel for the 64- and 128-channel stages (gfx950):  y = x (+) SiLU(conv3x3(SiLU(conv1x1(x))))
// (reference models/common.py:99-109 with e = 1.0 as used inside C3, Conv = conv + folded BN + SiLU :45-50).
//
// As two cft_conv2d launches these stages are the least efficient part of the forward: the 3x3 conv re-stages every
// activation nine times through the LDS-DMA path (once per tap) and the hidden tensor t makes a round trip through HBM.
// Both kernels here keep the ACTIVATIONS resident instead and stream only the weights:
//   * one 8-wave workgroup per spatial tile computes t = SiLU(W1 x + b1) on the tile's halo patch (zero outside the image
//     = the 3x3 conv's padding) into LDS (128-byte pixel rows, granule slot ^ (pixel & 7)); x fragments come straight
//     from global memory, W1 is the ROW operand so a lane ends up with 4 consecutive channels of one pixel;
//   * the nine taps of the 3x3 conv are shifted ds_read_b128 of that patch; W1 / W2 stream through a 4-slot LDS ring
//     (global_load_lds, three stages ahead, counted s_waitcnt vmcnt(2));
//   * two workgroups per CU (<= 80 KiB LDS each): one's x loads / SiLU / epilogue run under the other's MFMAs;
//   * epilogue: bias + SiLU -> fp32 strip (aliases the dead patch) -> 16-byte row vectors -> + shortcut -> 16-bit stores.
// Products, 32-wide k chunks, their order and every rounding are those of the two-launch path: bit-identical to it.
// (Three earlier implementations - 3x3 weights LDS-resident for 64 channels, a persistent one-workgroup-per-CU kernel and a
// 16-KiB-K-tile kernel for 128 - were measured slower, profiles/r02_bottleneck128.md, and left the library in round 3.)
// ABL template arguments are timing probes (results wrong): compiled only with -DCFT_PROBES (tools/build_probes.sh).
#include "cft_common.h"
#include <stdlib.h>

#ifdef CFT_PROBES
extern thread_local int g_conv_variant;   // conv_gemm.hip (cft_set_conv_variant)
#endif

__device__ __attribute__((aligned(16))) uint32_t cft_zero_page_b[4] = {0u, 0u, 0u, 0u};
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

#include "bneck_common.h"

// ------------------------------------------------------------------------------------ 128 channels
// 80 x 80 maps, 21 Bottlenecks per yolov5l forward.  8 x 16-pixel tiles: t patch 10 x 18 pixels in two 64-channel planes
// (48 KiB) + the ring (32 KiB).
//   * the ring is FOUR slots of 8 KiB: a W2 stage is 32 k wide (128 rows x 64 B, slot = k-granule ^ h(row / 4)), fetched from
//     the stage-major, pre-swizzled copy of the weights (cft_bottleneck_pack_w2: every LDS-DMA request is 1 KiB contiguous),
//     issued three stages ahead with counted s_waitcnt vmcnt(2); W1 streams once (four stages of 64 rows x 128 B, one per
//     output-channel half and k half);
//   * the 3x3 loop is software-pipelined by hand: step s READS the fragments of stage s and runs the MFMAs of step s - 1
//     under those reads;
//   * the shortcut pixels are requested right after the LAST stage: vmcnt is in-order, so an earlier request would be
//     waited for together with the next stage, three steps later;
//   * the patch's 12 row tiles are split evenly: wave w owns row tile w in both output-channel passes of the W1 stage and
//     row tile 8 + w / 2 in ONE of them (the two waves of a SIMD in different ones): 8 x requests per lane for every wave;
//   * wave (wmr, wnc) owns tile rows 2 wmr, 2 wmr + 1 x 64 channels; biases live in the 12 spare rows of plane 1 of the patch.
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
// ring three stages ahead.  42 + 32 KiB: two workgroups per CU (a weights-resident form, 117 KiB, had
// one whose 16 waves ran their SiLU passes and their MFMAs in lock step).  Wave w owns tile rows 2 w, 2 w + 1 and all 64
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

extern "C" int cft_bottleneck(const void* x, int ldx, int xoff, const void* w1, int kpad1, const float* b1,
                              const void* w2, int kpad2, const void* w2_stages, const float* b2, void* y, int ldy, int yoff,
                              int B, int H, int W, int c, int shortcut, int dtype, void* stream) {
  CFT_REQUIRE(x && w1 && w2 && y, "cft_bottleneck: null pointer");
  CFT_REQUIRE(dtype == CFT_BF16 || dtype == CFT_F16, "cft_bottleneck: dtype must be CFT_BF16 or CFT_F16");
  CFT_REQUIRE(c == 64 || c == 128, "cft_bottleneck: the fused kernels cover 64 and 128 channels (use two cft_conv2d calls otherwise)");
  CFT_REQUIRE(c != 128 || w2_stages != nullptr, "cft_bottleneck: 128 channels need the stage-major weights (cft_bottleneck_pack_w2)");
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
  Bneck128Params q;
  q.x = (const unsigned char*)x; q.w1 = (const unsigned char*)w1; q.w2 = (const unsigned char*)w2;
  q.w2s = (const unsigned char*)w2_stages;
  q.b1 = b1; q.b2 = b2; q.y = (unsigned char*)y;
  q.ldx = ldx; q.xoff = xoff; q.ldy = ldy; q.yoff = yoff; q.kpad1 = kpad1; q.kpad2 = kpad2;
  q.H = H; q.W = W; q.shortcut = shortcut ? 1 : 0;
  q.tiles_x = (W + 15) / 16; q.tiles_y = c == 128 ? (H + 7) / 8 : (H + 15) / 16;     // 8 x 16 / 16 x 16 pixel tiles
  CFT_REQUIRE((long)B * q.tiles_x * q.tiles_y < (1L << 31), "cft_bottleneck: too many tiles");
  q.ntiles = B * q.tiles_x * q.tiles_y;
  hipStream_t s_ = as_stream(stream);
  const dim3 grid(q.ntiles);
#define BNC_LAUNCH(T_, ABL_)                                                                          \
  {                                                                                                   \
    constexpr int smem_ = 2 * 12 * 16 * 128 + 2 * 16384;                                              \
    cft_allow_lds<&bottleneck128c_kernel<T_, ABL_>>(smem_);                                           \
    hipLaunchKernelGGL((bottleneck128c_kernel<T_, ABL_>), grid, dim3(512), smem_, s_, q);             \
  }
#define BNR_LAUNCH(T_, ABL_)                                                                          \
  {                                                                                                   \
    constexpr int smem_ = 21 * 16 * 128 + 4 * 8192;                                                   \
    cft_allow_lds<&bottleneck64r_kernel<T_, ABL_>>(smem_);                                            \
    hipLaunchKernelGGL((bottleneck64r_kernel<T_, ABL_>), grid, dim3(512), smem_, s_, q);              \
  }
#ifdef CFT_PROBES   // timing probes (results wrong): no W1-stage MFMAs / no 3x3 MFMAs / no epilogue / no weight DMA / no x requests / no SiLU
  if (dtype == CFT_BF16) {
    switch (g_conv_variant) {
      case 9201: if (c == 128) { BNC_LAUNCH(uint16_t, 1) return cft_check_launch("bottleneck128c_kernel(probe)"); } break;
      case 9202: if (c == 128) { BNC_LAUNCH(uint16_t, 2) return cft_check_launch("bottleneck128c_kernel(probe)"); } break;
      case 9204: if (c == 128) { BNC_LAUNCH(uint16_t, 4) return cft_check_launch("bottleneck128c_kernel(probe)"); } break;
      case 9208: if (c == 128) { BNC_LAUNCH(uint16_t, 8) return cft_check_launch("bottleneck128c_kernel(probe)"); } break;
      case 9456: if (c == 128) { BNC_LAUNCH(uint16_t, 256) return cft_check_launch("bottleneck128c_kernel(probe)"); } break;
      case 9712: if (c == 128) { BNC_LAUNCH(uint16_t, 512) return cft_check_launch("bottleneck128c_kernel(probe)"); } break;
      case 9601: if (c == 64) { BNR_LAUNCH(uint16_t, 1) return cft_check_launch("bottleneck64r_kernel(probe)"); } break;
      case 9602: if (c == 64) { BNR_LAUNCH(uint16_t, 2) return cft_check_launch("bottleneck64r_kernel(probe)"); } break;
      case 9604: if (c == 64) { BNR_LAUNCH(uint16_t, 4) return cft_check_launch("bottleneck64r_kernel(probe)"); } break;
      case 9608: if (c == 64) { BNR_LAUNCH(uint16_t, 8) return cft_check_launch("bottleneck64r_kernel(probe)"); } break;
      default: break;
    }
  }
#endif
  if (c == 128) {
#ifdef CFT_PROBES
    if (g_conv_variant == 98 || (g_conv_variant >= 9300 && g_conv_variant < 9400)) {   // the 16 x 16-tile kernel with the hand-scheduled 3x3 loop (probes/bottleneck_asm.hip: slower, A/B only)
      q.tiles_y = (H + 15) / 16;
      q.ntiles = B * q.tiles_x * q.tiles_y;
      return bneck128_asm_launch(q, dtype, s_);
    }
#endif
    if (dtype == CFT_F16) BNC_LAUNCH(f16_t, 0) else BNC_LAUNCH(uint16_t, 0)
    return cft_check_launch("bottleneck128c_kernel");
  }
  if (dtype == CFT_F16) BNR_LAUNCH(f16_t, 0) else BNR_LAUNCH(uint16_t, 0)
#undef BNC_LAUNCH
#undef BNR_LAUNCH
  return cft_check_launch("bottleneck64r_kernel");
}
