// PROBE BUILD ONLY since round 6 (tools/build_probes.sh): the one-kernel stem is bit-identical and removes 3.35 GB of HBM traffic per forward, but is
// slower than the two kernels it replaces (profiles/r05_stem.md: 943 us against 313 + 463) - VERDICT r5 item 8: it left the product library, the
// executor switch (Model.fuse_stem) and the C ABI (cft_stem / cft_stem_ok, ABI v10) with it.
#ifdef CFT_PROBES
// The stem of a backbone as ONE kernel for gfx950 (yolov5l widths):  image -> Focus -> Conv(64 -> 128, 3x3, stride 2) -> C3.cv1 | C3.cv2
// (reference models/common.py:168-179 Focus, :45-50 Conv, :141-143 the two 1x1 convs of the C3 behind it, packed as one [N2][128] weight;
// yaml rows 0-2 / 5-7 of the fusion configs).
//
// As separate launches (cft_focus_conv, then cft_conv2d_chain) the Focus output - [B, 320, 320, 64] at 640 x 640: 839 MB per stream at 64
// pairs - is written once and read once: 3.4 GB of a forward's 48 GB of HBM traffic, for 1.3 % of its FLOPs.  Here it never exists:
//   * persistent 8-wave workgroups, TWO per CU (73 KiB of LDS each), walk 8 x 8-pixel tiles of the stride-2 conv's output in an XCD-aware
//     order (spatial neighbours share image rows in one L2);
//   (a) the 19 x 19 halo patch of the space-to-depth tensor is built in LDS straight from the image (two column-parity planes of 32-byte
//       pixels, so that stride-2 pixel walks are stride-1 in LDS); the NEXT tile's image samples are requested late in the tile and land
//       in registers under its tail;
//   (b) Focus: F = SiLU(Wf s2d + bf) on the 17 x 17 halo patch the stride-2 conv needs (289 pixels, 19 MFMA row tiles; the Focus weights
//       are the ROW operand: a lane ends up with 4 consecutive channels of a pixel), rounded once and written to an LDS patch of 128-byte
//       pixels (even / odd columns de-interleaved per row); pixels outside the image are ZERO (= the stride-2 conv's padding, not Focus
//       of padding);
//   (c) the stride-2 3x3 conv reads its nine taps as shifted ds_read_b128 of that patch; only its weights stream - one tap (128 rows x
//       128 B = 16 KiB) per stage through a two-slot LDS ring;
//   (d) bias + SiLU + rounding on the accumulators, which are written to LDS as the A operand of the pointwise GEMM (two swizzled
//       [64 pixels][64 channels] images over the dead patch; lanes l / l^1 swap half of their values so that every write is a packed
//       channel pair); its weights arrive as ring stages 9 and 10;
//   (e) epilogue: bias + activation -> fp32 strip -> 16-byte row vectors -> 16-bit NHWC stores (as conv_gemm.hip).
// Products, 32-wide k chunks, their order and every rounding are those of cft_focus_conv followed by cft_conv2d_chain: bit-identical to
// them (tests/test_gpu_ops.py).
// MEASURED (profiles/r05_stem.md): 905 us per launch against 376 + 541 us for the two kernels back to back, 943 against 313 + 463 inside the
// forward, -1 % pairs/s on the whole forward: the tile loop is a chain of ~14 barriers and exposed L2 / HBM round trips that two workgroups
// per CU do not hide (the skeleton alone - barriers, patch build, image writes, pointwise GEMM - costs 6.4 us per tile).  An 8 x 16-tile,
// one-workgroup-per-CU form measured 1020 - 1030 us and was removed.  The kernel is therefore OPT-IN (Model.fuse_stem, default off): it
// trades 7 % of the forward's HBM traffic for no time.
#include "../cft_common.h"
#include "../focus_common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
static __device__ __attribute__((aligned(16))) uint32_t cft_zero_page_st[4] = {0u, 0u, 0u, 0u};

struct StemParams {
  const unsigned char* in;
  long sb, sc, sh;          // element strides of the image: batch, channel, row
  float scale;              // 1 for float images, 1 / 255 for uint8 ones
  const unsigned char* wf;  // Focus conv [64][192], k = tap * 16 + ci (ops.pack_conv(cin_pad = 16))
  const float* bf;
  const unsigned char* w1;  // stride-2 conv [128][576], k = tap * 64 + ci
  const float* b1;
  const unsigned char* w2;  // pointwise layer [N2][128]
  const float* b2;
  unsigned char* y;
  int ldy, yoff, N2, act2;
  int Hf, Wf, Ho, Wo;       // Focus output (= space-to-depth) size, stride-2 conv output size
  int tiles_x, tiles_y, ntiles;
  int abl;                  // timing probes of stem8_kernel (results wrong; cft_set_conv_variant(8800 + bits)): 1 no weight ring, 2 no Focus phase, 4 no conv taps, 8 no epilogue, 16 no image samples
};

// 8 x 8 output pixels per tile: F patch 17 x 17 = 38 KiB, two-slot weight ring 32 KiB whose second slot first holds the s2d patch: 73 KiB, so
// that TWO workgroups share a CU and one's VALU / LDS / barrier phases run under the other's MFMAs - what the Bottleneck kernels do.  Mapping:
//   * F patch line = 17 slots (9 even columns, then 8 odd ones), granule slot ^ ((p + 3 py) & 7): the stride-2 reads of two output rows
//     (lanes 0-7 / 8-15 of an MFMA row tile) are conflict-free;
//   * Focus: wave (q, jh) computes row tiles q, q + 4, ... x output-channel tiles 2 jh, 2 jh + 1; its 10 weight fragments live in registers;
//   * stride-2 conv / pointwise GEMM: wave (wmr, wnq) owns MFMA row tiles 2 wmr, 2 wmr + 1 (output rows 4 wmr .. + 3) x channels 32 wnq .. + 32;
//   * ring: stage s in slot s & 1, requested one step ahead (the other workgroup's work covers the L2 latency).
template <typename T, typename IN>
__global__ void __launch_bounds__(512, 2) stem8_kernel(const StemParams p) {
  constexpr int TS = 8;
  constexpr int FPW = 17, NFP = 17 * FPW;                     // F halo patch: 17 x 17 = 289 pixels
  constexpr int NRT = 19;                                     // its MFMA row tiles (304 slots)
  constexpr int SPW = 19, NSP = 19 * SPW;                     // s2d halo patch: 19 x 19 = 361 pixels
  constexpr int SCOLS = 10;                                   // storage columns per column parity
  constexpr int F_BYTES = NRT * 16 * 128;                     // 38912
  constexpr int SLOT = 16384;
  constexpr int IMG = 64 * 128;                               // one [64 pixels][64 channels] image of the pointwise GEMM's A operand
  constexpr int SLD = 32 + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sF = smem;
  unsigned char* sR = smem + F_BYTES;
  unsigned char* sS = sR + SLOT;                              // the s2d patch (12 160 B) lives in ring slot 1 until the Focus phase is over
  unsigned char* sZ = sR + 2 * SLOT;
  float* sB = reinterpret_cast<float*>(sZ + 16);              // bf[64], b1[128], b2[128]

  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int wmr = wave >> 2, wnq = wave & 3;                  // stride-2 conv / pointwise GEMM: row tiles 2 wmr, 2 wmr + 1; channels 32 wnq .. + 32
  const int fq = wave >> 1, fjh = wave & 1;                   // Focus: row tiles fq, fq + 4, ...; channel tiles 2 fjh, 2 fjh + 1
  const unsigned char* zero_page = reinterpret_cast<const unsigned char*>(cft_zero_page_st);
  const uint32_t fl = (uint32_t)(uintptr_t)(lds_void_t*)sF;

  if (tid0 < 64) sB[tid0] = p.bf != nullptr ? p.bf[tid0] : 0.0f;
  else if (tid0 < 192) sB[tid0] = p.b1 != nullptr ? p.b1[tid0 - 64] : 0.0f;
  else if (tid0 < 320) sB[tid0] = (p.b2 != nullptr && tid0 - 192 < p.N2) ? p.b2[tid0 - 192] : 0.0f;
  if (tid0 == 320) *reinterpret_cast<gran_t*>(sZ) = gran_t{0u, 0u, 0u, 0u};

#define ST8_STAGE(s_)                                                                                    \
  {                                                                                                      \
    constexpr int ss_ = (s_);                                                                            \
    if (!(p.abl & 1)) _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                 \
      const int row_ = r1 + h_ * 64;                                                                     \
      const unsigned char* src_ = ss_ < 9 ? p.w1 + ((long)row_ * 576 + ss_ * 64 + g1 * 8) * 2            \
                                          : (row_ < p.N2 ? p.w2 + ((long)row_ * 128 + (ss_ - 9) * 64 + g1 * 8) * 2 : zero_page); \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src_, (lds_void_t*)(sR + (ss_ & 1) * SLOT + h_ * 8192 + wave * 1024), 16, 0, 0); \
    }                                                                                                    \
  }
#define ST8_SYNC(n_)                                                                                     \
  {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    __builtin_amdgcn_s_waitcnt((n_) | 0x70);              /* vmcnt(n) lgkmcnt(0) */                      \
    __builtin_amdgcn_s_barrier();                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  }
  const int nt = p.ntiles, xq = nt >> 3, xr = nt & 7;
  const int tiles = p.tiles_x * p.tiles_y;
  const IN* img_base = reinterpret_cast<const IN*>(p.in);
  float raw[12];
// the 12 image samples of this thread's s2d patch pixel for tile (b_, oy0_, ox0_): unconditional loads at clamped coordinates (every
// wave issues the same 6 requests - the counted waits rely on it), masked when the patch is built
#define ST8_FETCH(b_, oy0_, ox0_)                                                                        \
  {                                                                                                      \
    const IN* img_ = img_base + (long)((p.abl & 16) ? 0 : (b_)) * p.sb;                                  \
    const int si_ = (p.abl & 16) ? 0 : min(tid, NSP - 1);                                                                   \
    const int sy_ = (si_ * 3450) >> 16, sx_ = si_ - sy_ * SPW;                                           \
    const int zy_ = min(max(2 * (oy0_) - 2 + sy_, 0), p.Hf - 1), zx_ = min(max(2 * (ox0_) - 2 + sx_, 0), p.Wf - 1); \
    _Pragma("unroll") for (int c_ = 0; c_ < 3; ++c_) {                                                   \
      const IN* base_ = img_ + (long)c_ * p.sc + (long)(2 * zy_) * p.sh + 2 * zx_;                       \
      load_pair<IN>(base_, p.scale, raw[0 + c_], raw[6 + c_]);                                           \
      load_pair<IN>(base_ + p.sh, p.scale, raw[3 + c_], raw[9 + c_]);                                    \
    }                                                                                                    \
  }
#define ST8_DECODE(v_, lt_, b_, oy0_, ox0_)                                                              \
  const int xcd_##v_ = (v_) & 7, xslot_##v_ = (v_) >> 3;                                                 \
  const int lt_ = (xcd_##v_ < xr ? xcd_##v_ * (xq + 1) : xr * (xq + 1) + (xcd_##v_ - xr) * xq) + xslot_##v_; \
  const int b_ = lt_ / tiles, tt_##v_ = lt_ - b_ * tiles;                                                \
  const int ty_##v_ = tt_##v_ / p.tiles_x, tx_##v_ = tt_##v_ - ty_##v_ * p.tiles_x;                      \
  const int oy0_ = ty_##v_ * TS, ox0_ = tx_##v_ * TS;

  int vslot = blockIdx.x;
  if (vslot < nt) {
    const int tid = tid0;
    ST8_DECODE(vslot, lt0, b0, oy00, ox00)
    ST8_FETCH(b0, oy00, ox00)
  }
  __syncthreads();

  for (; vslot < nt; vslot += gridDim.x) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));                             // opaque copy: otherwise ~60 registers of loop-invariant addresses (ring stages, fragment offsets) are hoisted out of the persistent loop
    const int lane = tid & 63, lrow = lane & 15, lgrp = lane >> 4;
    const int r1 = tid >> 3, g1 = (tid & 7) ^ (r1 & 7);
    ST8_DECODE(vslot, lt, b, oy0, ox0)
    const int fy0 = 2 * oy0 - 1, fx0 = 2 * ox0 - 1;
    // ---- (a) s2d halo patch -> ring slot 1; ring stage 0 -> slot 0
    ST8_STAGE(0)
    if (tid < NSP) {
      const int sy = (tid * 3450) >> 16, sx = tid - sy * SPW;
      const bool in_img = (unsigned)(fy0 - 1 + sy) < (unsigned)p.Hf && (unsigned)(fx0 - 1 + sx) < (unsigned)p.Wf;
      float v[16];
#pragma unroll
      for (int e = 0; e < 12; ++e) v[e] = in_img ? raw[e] : 0.0f;
      v[12] = v[13] = v[14] = v[15] = 0.0f;
      gran_t* o = reinterpret_cast<gran_t*>(sS + ((sy * 2 + (sx & 1)) * SCOLS + (sx >> 1)) * 32);
      o[0] = Elem<T>::pack(v);
      o[1] = Elem<T>::pack(v + 8);
    }
    // this wave's Focus weight fragments (channel tiles 2 fjh, 2 fjh + 1): 10 granules, re-read per tile from L1 / L2
    gran_t wfrag[5][2];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        wfrag[ks][jj] = *reinterpret_cast<const gran_t*>(p.wf + ((long)((2 * fjh + jj) * 16 + lrow) * 192 + (ks * 4 + lgrp) * 8) * 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // the s2d patch is visible

    // ---- (b) Focus on the F halo patch
#pragma unroll 1
    for (int rt = (p.abl & 2) ? NRT : fq; rt < NRT; rt += 4) {
      const int pq = rt * 16 + lrow, pc = min(pq, NFP - 1);
      const int py = (pc * 3856) >> 16, rr = pc - py * FPW;
      const int px = rr < 9 ? 2 * rr : 2 * (rr - 9) + 1;
      const uint32_t keep = (pq < NFP && (unsigned)(fy0 + py) < (unsigned)p.Hf && (unsigned)(fx0 + px) < (unsigned)p.Wf) ? 0xffffffffu : 0u;
      const int qy = (pq * 3856) >> 16;                       // the patch line of slot pq (also for the 15 padding slots): the swizzle's row term
      f32x4_t acc1[2];
      acc1[0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      acc1[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      gran_t af[5];
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        const int tap = ks * 2 + (lgrp >> 1), half = lgrp & 1;
        const int kh = tap / 3, kw = tap - kh * 3;
        const int sy = py + kh, sx = px + kw;
        const unsigned char* ap = tap < 9 ? sS + ((sy * 2 + (sx & 1)) * SCOLS + (sx >> 1)) * 32 + half * 16 : sZ;
        af[ks] = *reinterpret_cast<const gran_t*>(ap);
      }
#pragma unroll
      for (int ks = 0; ks < 5; ++ks)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) acc1[jj] = mma_granule<T>(wfrag[ks][jj], af[ks], acc1[jj]);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int cc = (2 * fjh + jj) * 16 + lgrp * 4;
        const f32x4_t bq = *reinterpret_cast<const f32x4_t*>(sB + cc);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act<CFT_ACT_SILU>(acc1[jj][e] + bq[e]);
        uint2 w;
        w.x = Elem<T>::pack2(v[0], v[1]) & keep;
        w.y = Elem<T>::pack2(v[2], v[3]) & keep;
        const uint32_t ta = fl + pq * 128 + ((((cc >> 3) ^ ((pq + 3 * qy) & 7)) << 4) | ((cc & 7) << 1));
        const unsigned long long wq = ((unsigned long long)w.y << 32) | w.x;
        asm volatile("ds_write_b64 %0, %1" ::"v"(ta), "v"(wq) : "memory");
      }
    }
    ST8_SYNC(0)                                               // F patch visible, stage 0 landed, the s2d patch (slot 1) is dead

    // ---- (c) stride-2 3x3 conv of the F patch: tap t = stage t in slot t & 1
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int brow = (wnq * 32 + lrow) * 128;
    const int aoy = 4 * wmr + (lrow >> 3), aox = lrow & 7;   // output row (of row tile 2 wmr; + 2 for the second tile) and column of this lane's pixel
#define ST8_TAP(t_)                                                                                      \
    {                                                                                                    \
      constexpr int kh_ = (t_) / 3, kw_ = (t_) - 3 * ((t_) / 3);                                         \
      constexpr int kwoff_ = kw_ == 0 ? 0 : (kw_ == 1 ? 9 : 1);                                          \
      gran_t af_[2][2], bf_[2][2];                                                                       \
      _Pragma("unroll") for (int kq = 0; kq < 2; ++kq) {                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                  \
          const int py_ = 2 * (aoy + 2 * i) + kh_;                                                       \
          const int pp_ = py_ * FPW + kwoff_ + aox;                                                      \
          af_[kq][i] = *reinterpret_cast<const gran_t*>(sF + pp_ * 128 + (((kq * 4 + lgrp) ^ ((pp_ + 3 * py_) & 7)) << 4)); \
        }                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                    \
          bf_[kq][j] = *reinterpret_cast<const gran_t*>(sR + ((t_) & 1) * SLOT + j * 2048 + brow + (((kq * 4 + lgrp) ^ (lrow & 7)) << 4)); \
      }                                                                                                  \
      if (!(p.abl & 4)) _Pragma("unroll") for (int kq = 0; kq < 2; ++kq)                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                    \
          _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mma_granule<T>(af_[kq][i], bf_[kq][j], acc[i][j]); \
    }
    // step t: request stage t + 1 into the other slot (its readers passed the barrier that ended step t - 1), multiply tap t, wait for the stage
    ST8_STAGE(1) ST8_TAP(0) ST8_SYNC(0)
    ST8_STAGE(2) ST8_TAP(1) ST8_SYNC(0)
    ST8_STAGE(3) ST8_TAP(2) ST8_SYNC(0)
    ST8_STAGE(4) ST8_TAP(3) ST8_SYNC(0)
    ST8_STAGE(5) ST8_TAP(4) ST8_SYNC(0)
    ST8_STAGE(6) ST8_TAP(5) ST8_SYNC(0)
    ST8_STAGE(7) ST8_TAP(6) ST8_SYNC(0)
    ST8_STAGE(8) ST8_TAP(7) ST8_SYNC(0)
    ST8_STAGE(9) ST8_TAP(8) ST8_SYNC(0)                       // stage 9 (pointwise weights, k 0-63) landed; every wave is past its last F read
#undef ST8_TAP
    ST8_STAGE(10)                                             // -> slot 0 (tap 8's readers are past the barrier)
    __builtin_amdgcn_sched_barrier(0);                        // the counted waits below assume stage 10's two requests are OLDER than the image loads
    {   // the next tile's image samples: the youngest requests of the tile (the two waits below leave them in flight)
      const int vnext = vslot + gridDim.x;
      if (vnext < nt) {
        ST8_DECODE(vnext, ltn, bn, oy0n, ox0n)
        ST8_FETCH(bn, oy0n, ox0n)
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool more = vslot + gridDim.x < nt;                 // uniform: 6 more requests in flight per wave

    // ---- (d) bias + SiLU + rounding; the tile becomes the pointwise GEMM's A operand (two [64 pixels][64 channels] images over the dead patch)
    {
      const bool odd = lane & 1;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float b1v = sB[64 + wnq * 32 + j * 16 + lrow];
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = apply_act<CFT_ACT_SILU>(acc[i][j][e] + b1v);
          const uint32_t r01 = Elem<T>::pack2(v[0], v[1]), r23 = Elem<T>::pack2(v[2], v[3]);
          const uint32_t got = (uint32_t)__builtin_amdgcn_mov_dpp((int)(odd ? r01 : r23), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
          const uint32_t d0 = odd ? ((got & 0xffffu) | (r23 << 16)) : ((r01 & 0xffffu) | (got << 16));
          const uint32_t d1 = odd ? ((got >> 16) | (r23 & 0xffff0000u)) : ((r01 >> 16) | (got & 0xffff0000u));
          const int row = (2 * wmr + i) * 16 + lgrp * 4 + (odd ? 2 : 0);
          const int c = (wnq & 1) * 32 + j * 16 + (lrow & 14);
          unsigned char* im = sF + (wnq >> 1) * IMG + (c & 7) * 2;
          *reinterpret_cast<uint32_t*>(im + row * 128 + (((c >> 3) ^ (row & 7)) << 4)) = d0;
          *reinterpret_cast<uint32_t*>(im + (row + 1) * 128 + (((c >> 3) ^ ((row + 1) & 7)) << 4)) = d1;
          acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    }
    // images visible; stage 10 (2 requests) and the next tile's samples (6) may stay in flight
    if (more) { ST8_SYNC(8) } else { ST8_SYNC(2) }
#define ST8_PW(k2_)                                                                                      \
    {                                                                                                    \
      gran_t af_[2][2], bf_[2][2];                                                                       \
      _Pragma("unroll") for (int kq = 0; kq < 2; ++kq) {                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                  \
          const int row_ = (2 * wmr + i) * 16 + lrow;                                                    \
          af_[kq][i] = *reinterpret_cast<const gran_t*>(sF + (k2_) * IMG + row_ * 128 + (((kq * 4 + lgrp) ^ (row_ & 7)) << 4)); \
        }                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                    \
          bf_[kq][j] = *reinterpret_cast<const gran_t*>(sR + ((9 + (k2_)) & 1) * SLOT + j * 2048 + brow + (((kq * 4 + lgrp) ^ (lrow & 7)) << 4)); \
      }                                                                                                  \
      _Pragma("unroll") for (int kq = 0; kq < 2; ++kq)                                                   \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                    \
          _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mma_granule<T>(af_[kq][i], bf_[kq][j], acc[i][j]); \
    }
    ST8_PW(0)
    if (more) { ST8_SYNC(6) } else { ST8_SYNC(0) }            // stage 10 landed (the samples may stay in flight)
    ST8_PW(1)
#undef ST8_PW

    // ---- (e) epilogue: strip i = MFMA row tile 2 wmr + i (output rows 4 wmr + 2 i, + 1), 16 pixels x 32 channels
    if (p.abl & 8) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else {
      float b2v[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) b2v[j] = sB[192 + wnq * 32 + j * 16 + lrow];
      float* stage = reinterpret_cast<float*>(sF + 2 * IMG) + wave * (16 * SLD);
      const long img_pix = (long)b * p.Ho * p.Wo;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = acc[i][j][e] + b2v[j];
            stage[(lgrp * 4 + e) * SLD + j * 16 + lrow] = p.act2 == CFT_ACT_SILU ? apply_act<CFT_ACT_SILU>(a) : a;
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
          const int row = lane >> 2, col = (lane & 3) * 8;
          const int x = ox0 + (row & 7), y = oy0 + 4 * wmr + 2 * i + (row >> 3);
          if (x < p.Wo && y < p.Ho && wnq * 32 + col < p.N2) {
            const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col);
            const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(stage + row * SLD + col + 4);
            const float o[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
            *reinterpret_cast<gran_t*>(p.y + ((img_pix + (long)y * p.Wo + x) * p.ldy + p.yoff + wnq * 32 + col) * 2) = Elem<T>::pack(o);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // ring, images and strips are free for the next tile
  }
#undef ST8_STAGE
#undef ST8_SYNC
#undef ST8_FETCH
#undef ST8_DECODE
}

extern thread_local int g_conv_variant;   // conv_gemm.hip (cft_set_conv_variant): 8800 + bits = timing probes (StemParams.abl)

template <typename T, typename IN>
static int launch_stem(const StemParams& p0, hipStream_t stream) {   int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  cus = cus > 8 ? (cus / 8) * 8 : 8;                     // a multiple of the 8 XCDs: a workgroup's tiles stay on its XCD
  StemParams p = p0;
  p.abl = (g_conv_variant >= 8800 && g_conv_variant < 8864) ? g_conv_variant - 8800 : 0;
  constexpr int smem_bytes = 19 * 16 * 128 + 2 * 16384 + 16 + 320 * 4;                          // 72 976 B: two workgroups per CU
  cft_allow_lds<&stem8_kernel<T, IN>>(smem_bytes);
  const int grid = p.ntiles < 2 * cus ? p.ntiles : 2 * cus;
  hipLaunchKernelGGL((stem8_kernel<T, IN>), dim3(grid), dim3(512), smem_bytes, stream, p);
  return cft_check_launch("stem8_kernel");
}

template <typename T>
static int dispatch_stem(const StemParams& p, int in_kind, hipStream_t stream) {
  if (in_kind == 1) return launch_stem<T, unsigned char>(p, stream);
  if constexpr (!__is_same(T, uint16_t)) {                // half images go with half compute (`model.half()` + `img.half()`)
    if (in_kind == 2) return launch_stem<T, f16_t>(p, stream);
  }
  return launch_stem<T, float>(p, stream);
}

extern "C" int cft_stem_ok(int H, int W, int n_focus, int kpad_focus, int n1, int kpad1, int n2, int dtype) {
  return (dtype == CFT_BF16 || dtype == CFT_F16) && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && n_focus == 64 && kpad_focus == 192 &&
         n1 == 128 && kpad1 == 576 && n2 >= 8 && n2 <= 128 && n2 % 8 == 0 ? 1 : 0;
}

extern "C" int cft_stem(const void* in, int in_kind, long stride_b, long stride_c, long stride_h, float scale,
                        const void* wf, int kpad_f, const float* bf, const void* w1, int kpad1, const float* b1,
                        const void* w2, const float* b2, void* y, int ldy, int yoff,
                        int B, int H, int W, int n_focus, int n1, int n2, int act2, int dtype, void* stream) {
  CFT_REQUIRE(in && wf && w1 && w2 && y, "cft_stem: null pointer");
  CFT_REQUIRE(cft_stem_ok(H, W, n_focus, kpad_f, n1, kpad1, n2, dtype),
              "cft_stem: not eligible (16-bit compute, Focus 64 channels [64][192], conv 128 channels [128][576], n2 <= 128; ask cft_stem_ok)");
  CFT_REQUIRE(in_kind >= 0 && in_kind <= 2, "cft_stem: in_kind must be 0 (float), 1 (uint8) or 2 (half)");
  CFT_REQUIRE(in_kind != 2 || dtype == CFT_F16, "cft_stem: half images require dtype CFT_F16");
  CFT_REQUIRE(B > 0, "cft_stem: non-positive batch");
  CFT_REQUIRE(act2 == CFT_ACT_SILU || act2 == CFT_ACT_NONE, "cft_stem: activation must be SiLU or none");
  CFT_REQUIRE(ldy % 8 == 0 && yoff % 8 == 0 && ldy >= yoff + n2, "cft_stem: bad output ld/offset");
  CFT_REQUIRE(stride_h >= W && stride_c > 0 && stride_b > 0, "cft_stem: bad strides");
  const long es = in_kind == 1 ? 1 : (in_kind == 2 ? 2 : 4);
  CFT_REQUIRE(((long)(size_t)in % (2 * es) == 0) && stride_h % 2 == 0 && stride_c % 2 == 0 && stride_b % 2 == 0,
              "cft_stem: image rows must start on pixel-pair boundaries (pointer and strides even)");
  StemParams p;
  p.in = (const unsigned char*)in; p.sb = stride_b; p.sc = stride_c; p.sh = stride_h; p.scale = scale;
  p.wf = (const unsigned char*)wf; p.bf = bf; p.w1 = (const unsigned char*)w1; p.b1 = b1; p.w2 = (const unsigned char*)w2; p.b2 = b2;
  p.y = (unsigned char*)y; p.ldy = ldy; p.yoff = yoff; p.N2 = n2; p.act2 = act2;
  p.Hf = H / 2; p.Wf = W / 2;
  p.Ho = (p.Hf - 1) / 2 + 1; p.Wo = (p.Wf - 1) / 2 + 1;          // 3x3, stride 2, padding 1
  p.tiles_x = (p.Wo + 7) / 8; p.tiles_y = (p.Ho + 7) / 8; p.abl = 0;
  CFT_REQUIRE((long)B * p.tiles_x * p.tiles_y < (1L << 31) && (long)B * p.Ho * p.Wo * ldy < (1L << 31), "cft_stem: tensor too large (split the batch)");
  p.ntiles = B * p.tiles_x * p.tiles_y;
  hipStream_t s = as_stream(stream);
  return dtype == CFT_BF16 ? dispatch_stem<uint16_t>(p, in_kind, s) : dispatch_stem<f16_t>(p, in_kind, s);
}

#endif
