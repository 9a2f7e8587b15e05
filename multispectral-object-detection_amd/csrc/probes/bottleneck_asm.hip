#ifdef CFT_PROBES
// The 128-channel Bottleneck (reference models/common.py:99-109 with e = 1.0 as inside C3: y = x (+) SiLU(conv3x3(SiLU(conv1x1(x)))), Conv = conv +
// folded BN + SiLU :45-50) as ONE kernel on 16 x 16-pixel tiles with a HAND-SCHEDULED 3x3 loop (round 6).  What it changes against
// bottleneck128c_kernel (bottleneck.hip: 8 x 16 tiles, two workgroups per CU, 36 stages of 8 MFMAs per wave between barriers, hipcc-scheduled):
//   * 16 x 16 tiles: the halo patch is 18 x 18 = 1.27 x the tile (8 x 16: 1.41 x) - less of the 1x1 conv, its SiLU and its x reads is recomputed;
//   * the 3x3 conv is NINE K steps of 64 MFMAs per wave (one per tap, 128 channels), one s_barrier each, as one inline-asm block written by
//     tools/gen_bneck_asm.py (bottleneck_asm.inc): fragments double-buffered in registers, the A fragments shifted reads of the patch whose
//     swizzle class is an immediate per (tap, m-tile), the weights' LDS-DMA requests spread between the MFMAs of the step's last quarter;
//   * wave tiles of 64 pixels x 64 channels: 0.5 fragment reads per MFMA (8 x 16 tiles with 32-pixel wave tiles: 0.75).
// Products, 32-wide k chunks, their order and every rounding are those of bottleneck128c_kernel and of the two-launch path: bit-identical to both
// (checked on the 44 fused-Bottleneck cases of tests/test_gpu_ops.py while it was the default).
// MEASURED SLOWER than bottleneck128c_kernel (profiles/r06_bottleneck128_asm.md: 178-199 us against 154-161 per launch at the bench shape): its
// 3x3 loop runs at the matrix pipe's pace (77 us = 10.7 us per 256-pixel tile), but with 152 KiB of LDS ONE workgroup fits a CU, so the W1 stage
// (64 us) and the epilogue (48 us) run beside nothing, where the two 80-KiB workgroups of the 8 x 16-tile kernel hide each other's.  It is
// compiled into the PROBE build only (tools/build_probes.sh; variant 98 of cft_set_conv_variant selects it there).
#include "../bneck_common.h"
#include <utility>
#include "bottleneck_asm.inc"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

#define BNECK_ASM_CLOBBERS                                                                                                 \
  "memory", "scc",                                                                                                         \
  "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
  "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63", \
  "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
  "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
  "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107"

#define BNECK_ASM_STMT(text_)                                                                                              \
  asm volatile(text_                                                                                                       \
               : [sof] "+s"(sof), [m0s] "=&s"(m0s)                                                                         \
               : [qb0] "v"(qb[0]), [qb1] "v"(qb[1]), [qb2] "v"(qb[2]), [qb3] "v"(qb[3]),                                   \
                 [sw00] "v"(sw[0][0]), [sw10] "v"(sw[1][0]), [sw20] "v"(sw[2][0]), [sw30] "v"(sw[3][0]),                   \
                 [sw40] "v"(sw[4][0]), [sw50] "v"(sw[5][0]), [sw60] "v"(sw[6][0]), [sw70] "v"(sw[7][0]),                   \
                 [sw01] "v"(sw[0][1]), [sw11] "v"(sw[1][1]), [sw21] "v"(sw[2][1]), [sw31] "v"(sw[3][1]),                   \
                 [sw41] "v"(sw[4][1]), [sw51] "v"(sw[5][1]), [sw61] "v"(sw[6][1]), [sw71] "v"(sw[7][1]),                   \
                 [rb0] "v"(rb[0]), [rb1] "v"(rb[1]),                                                                       \
                 [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]),                                   \
                 [srd] "s"(srdW), [wb] "s"(wb)                                                                             \
               : BNECK_ASM_CLOBBERS)

template <int... Is, class F>
__device__ __forceinline__ void bn_static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void bn_static_for(F&& f) { bn_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// LDS map: [t patch plane 0: 336 pixel rows x 128 B][plane 1][weight buffer 0: 32 KiB][weight buffer 1][b1, b2: 256 floats]
// PERSISTENT: one workgroup per CU walks the tiles vb = blockIdx.x, + gridDim.x, ...; the shortcut pixels of tile n are requested before its W1
// stage, and the x fragments, W1 and tap 0 of tile n + 1 before tile n's epilogue (their HBM / L2 latency runs under it; the x registers are dead
// between the W1 stage and that point, so the prefetch costs none).
// ABL (timing probes, -DCFT_PROBES builds only; results wrong): 1 = no W1 stage (no 1x1 MFMAs, no SiLU: the patch is not written),
// 2 = no 3x3 loop, 4 = no epilogue, 8 = no x requests
template <typename T, int ABL = 0>
__global__ void __launch_bounds__(512) bottleneck128a_kernel(const Bneck128Params p) {
  constexpr int C = 128, TH = 16, TW = 16, PW = 18, NPIX = PW * (TH + 2);     // 324 patch pixels
  constexpr int NRT = 21, PLANE = NRT * 16 * 128, RING = 2 * PLANE, SB = RING + 65536;
  constexpr int SLD = 64 + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sT = smem;
  unsigned char* sR = smem + RING;
  float* sB = reinterpret_cast<float*>(smem + SB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane & 15, lgrp = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  const int tiles = p.tiles_x * p.tiles_y;
  const bool third = wave < 5;                                  // row tiles wave, wave + 8 and (waves 0-4) wave + 16 of the 21

  // virtual block -> tile, XCD-aware (as bottleneck128c_kernel): consecutive logical tiles - spatial neighbours sharing halo pixels - meet in one L2
  auto decode = [&](int vb, int& b, int& y0, int& x0) {
    const int nb_ = p.ntiles;
    const int xq_ = nb_ >> 3, xr_ = nb_ & 7, xcd_ = vb & 7, xslot_ = vb >> 3;
    const int tile = (xcd_ < xr_ ? xcd_ * (xq_ + 1) : xr_ * (xq_ + 1) + (xcd_ - xr_) * xq_) + xslot_;
    b = tile / tiles;
    const int tt = tile - b * tiles;
    const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
    y0 = ty * TH; x0 = tx * TW;
  };
  // this wave's x fragments of a tile's halo patch: unconditional requests at clamped coordinates (masked when consumed)
  auto load_x = [&](gran_t (&xa)[3][4], int b, int y0, int x0) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int q = (wave + 8 * it) * 16 + lrow;
      const int py = (q * 3641) >> 16, px = q - py * PW;        // q / 18 for q < 336
      const int zy = min(max(y0 - 1 + py, 0), p.H - 1), zx = min(max(x0 - 1 + px, 0), p.W - 1);
      const unsigned char* xp = p.x + (((long)b * p.H * p.W + (long)zy * p.W + zx) * p.ldx + p.xoff + lgrp * 8) * 2;
      if ((it < 2 || third) && !(ABL & 8)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xa[it][ks] = *reinterpret_cast<const gran_t*>(xp + ks * 64);
      } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xa[it][ks] = gran_t{0u, 0u, 0u, 0u};
      }
    }
  };
  const int rs = tid >> 3, g = (tid & 7) ^ (rs & 7);
  uint32_t vo[4];
#pragma unroll
  for (int pz = 0; pz < 4; ++pz) vo[pz] = (uint32_t)((((long)(rs + 64 * (pz & 1))) * p.kpad2 + (pz >> 1) * 64 + g * 8) * 2);
  const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.w2), 0, (int)((long)C * p.kpad2 * 2), 0x00020000);
  const uint32_t wb = lds0 + RING + (uint32_t)wave * 1024u;
  // W1 -> weight buffer 1 (two 64-wide k halves of 128 rows x 128 B), tap 0 of W2 -> weight buffer 0
  auto stage_w1_tap0 = [&]() {
#pragma unroll
    for (int pz = 0; pz < 4; ++pz) {
      const unsigned char* s1 = p.w1 + (((long)(rs + 64 * (pz & 1))) * p.kpad1 + (pz >> 1) * 64 + g * 8) * 2;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)s1, (lds_void_t*)(sR + 32768 + (pz >> 1) * 16384 + (pz & 1) * 8192 + wave * 1024), 16, 0, 0);
      const unsigned char* s2 = p.w2 + vo[pz];
      __builtin_amdgcn_global_load_lds((gbl_void_t*)s2, (lds_void_t*)(sR + (pz >> 1) * 16384 + (pz & 1) * 8192 + wave * 1024), 16, 0, 0);
    }
  };
  uint32_t qb[4], sw[8][2], rb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) qb[i] = lds0 + (uint32_t)(((wm * 4 + i) * PW + lrow) * 128);
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int k = 0; k < 2; ++k) sw[c][k] = (uint32_t)(((lgrp + 4 * k) ^ ((lrow + c) & 7)) << 4);
#pragma unroll
  for (int k = 0; k < 2; ++k) rb[k] = lds0 + RING + (uint32_t)((wn * 64 + lrow) * 128 + (((lgrp + 4 * k) ^ (lrow & 7)) << 4));
  const int fbw = lrow * 128 + ((lgrp ^ (lrow & 7)) << 4);

  int vb = blockIdx.x;
  int b, y0, x0;
  decode(vb, b, y0, x0);
  gran_t xa[3][4];
  load_x(xa, b, y0, x0);
  stage_w1_tap0();
  {
    float bq = 0.0f;
    if (tid < C) { if (p.b1 != nullptr) bq = p.b1[tid]; }
    else if (tid < 2 * C) { if (p.b2 != nullptr) bq = p.b2[tid - C]; }
    if (tid < 2 * C) sB[tid] = bq;
  }
#pragma unroll 1
  for (;;) {
    // [A] this tile's x fragments, W1 and tap 0 have landed; the previous tile's strips are read and its stores drained
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const long img_pix = (long)b * p.H * p.W;
    const int vbn = vb + (int)gridDim.x;
    const bool more = vbn < p.ntiles;                     // uniform
    int bn = b, y0n = y0, x0n = x0;
    // [C] t^T = W1 x^T (W1 the ROW operand: a lane ends up with 4 consecutive channels of one pixel), bias + SiLU -> t patch; zero outside the
    //     image (= the 3x3 conv's padding) and beyond the patch.  A W1 fragment is read once and multiplies all of this wave's row tiles.
    if constexpr (!(ABL & 1)) {
      uint32_t keep[3];
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int q = (wave + 8 * it) * 16 + lrow;
        const int py = (q * 3641) >> 16, px = q - py * PW;
        keep[it] = (q < NPIX && (unsigned)(y0 - 1 + py) < (unsigned)p.H && (unsigned)(x0 - 1 + px) < (unsigned)p.W) ? 0xffffffffu : 0u;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { xa[it][ks].x &= keep[it]; xa[it][ks].y &= keep[it]; xa[it][ks].z &= keep[it]; xa[it][ks].w &= keep[it]; }
      }
      // rows tiles `wave` and `wave + 8` together (a W1 fragment is read once for both), then (waves 0-4) row tile `wave + 16`
      auto w1_rows = [&](auto nrc, int it0) {
        constexpr int NR = decltype(nrc)::value;
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {                  // output channels 64 jh .. + 63
          f32x4_t acc1[NR][4];
#pragma unroll
          for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[r][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const gran_t wf = *reinterpret_cast<const gran_t*>(sR + 32768 + (ks >> 1) * 16384 + (jh * 64 + j * 16) * 128 + ((ks & 1) ? (fbw ^ 64) : fbw));
#pragma unroll
              for (int r = 0; r < NR; ++r) acc1[r][j] = mma_granule<T>(wf, xa[it0 + r][ks], acc1[r][j]);
            }
          }
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const int q = (wave + 8 * (it0 + r)) * 16 + lrow;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int cc = j * 16 + lgrp * 4;
              const f32x4_t b1q = *reinterpret_cast<const f32x4_t*>(sB + jh * 64 + cc);
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = apply_act<CFT_ACT_SILU>(acc1[r][j][e] + b1q[e]);
              uint2 w;
              w.x = Elem<T>::pack2(v[0], v[1]) & keep[it0 + r];
              w.y = Elem<T>::pack2(v[2], v[3]) & keep[it0 + r];
              const uint32_t ta = lds0 + jh * PLANE + q * 128 + ((((cc >> 3) ^ (q & 7)) << 4) | ((cc & 7) << 1));
              const unsigned long long wq = ((unsigned long long)w.y << 32) | w.x;
              asm volatile("ds_write_b64 %0, %1" ::"v"(ta), "v"(wq) : "memory");
            }
          }
        }
      };
      w1_rows(std::integral_constant<int, 2>{}, 0);
      if (third) w1_rows(std::integral_constant<int, 1>{}, 2);
    }
    // [D] the whole t patch is visible; W1 (buffer 1) is dead.  (No vmcnt wait: [B]'s requests stay in flight; the loop's first barrier covers them.)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // [E] 3x3 conv of the patch: the asm block (bottleneck_asm.inc)
    uint32_t sof = 0, m0s;
    if constexpr (ABL & 2) {
    } else if ((wave >> 2) == 0) {
      if constexpr (__is_same(T, f16_t)) { BNECK_ASM_STMT(BNECK_ASM_LOOP_F16_G0); } else { BNECK_ASM_STMT(BNECK_ASM_LOOP_BF16_G0); }
    } else {
      if constexpr (__is_same(T, f16_t)) { BNECK_ASM_STMT(BNECK_ASM_LOOP_F16_G1); } else { BNECK_ASM_STMT(BNECK_ASM_LOOP_BF16_G1); }
    }
    // (every wave is past the loop's last barrier: all reads of the patch and of both weight buffers have returned)
    if constexpr (ABL & 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    // [F] this tile's shortcut pixels (strips of tile rows 4 wm + i, channels 64 wn ..: held across the loop they would not fit its register
    //     budget; their latency runs under the first strips' SiLU), then the next tile's x fragments, W1 and tap 0: they land under this epilogue
    gran_t rsv[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int itx = lane + v * 64;
        const int row = itx >> 3, col = (itx & 7) * 8;
        const int x = min(x0 + row, p.W - 1), y = min(y0 + wm * 4 + i, p.H - 1);
        rsv[i][v] = gran_t{0u, 0u, 0u, 0u};
        if (p.shortcut && !(ABL & 4)) rsv[i][v] = *reinterpret_cast<const gran_t*>(p.x + ((img_pix + (long)y * p.W + x) * p.ldx + p.xoff + wn * 64 + col) * 2);
      }
    if (more) { decode(vbn, bn, y0n, x0n); load_x(xa, bn, y0n, x0n); stage_w1_tap0(); }
    else {                                                   // (ends the old fragments' live range: nothing of them crosses the loop)
#pragma unroll
      for (int it = 0; it < 3; ++it)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xa[it][ks] = gran_t{0u, 0u, 0u, 0u};
    }
    // [H] epilogue: strip i = tile row 4 wm + i, 16 pixels x 64 channels: bias + SiLU -> fp32 strip (aliases the dead patch) -> 16-byte row vectors -> + shortcut -> stores
    if constexpr (!(ABL & 4)) {
      float b2v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b2v[j] = sB[C + wn * 64 + j * 16 + lrow];
      float* stage2[2] = {reinterpret_cast<float*>(sT) + wave * (16 * SLD), reinterpret_cast<float*>(sT) + (8 + wave) * (16 * SLD)};
      bn_static_for<2>([&](auto pc) {
        constexpr int i0 = 2 * decltype(pc)::value;
        bn_static_for<2>([&](auto sc) {
          constexpr int s_ = decltype(sc)::value;
          bn_static_for<4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const f32x4_t t = bneck_agpr_tile<(i0 + s_) * 4 + j>();
#pragma unroll
            for (int e = 0; e < 4; ++e) stage2[s_][(lgrp * 4 + e) * SLD + j * 16 + lrow] = apply_act<CFT_ACT_SILU>(t[e] + b2v[j]);
          });
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          const int i = i0 + s_;
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            const int itx = lane + v * 64;
            const int row = itx >> 3, col = (itx & 7) * 8;
            const int x = x0 + row, y = y0 + wm * 4 + i;
            if (x < p.W && y < p.H) {
              const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(stage2[s_] + row * SLD + col);
              const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(stage2[s_] + row * SLD + col + 4);
              float o[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
              if (p.shortcut) {
                float rf[8];
                Elem<T>::unpack(rsv[i][v], rf);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += rf[e];
              }
              *reinterpret_cast<gran_t*>(p.y + ((img_pix + (long)y * p.W + x) * p.ldy + p.yoff + wn * 64 + col) * 2) = Elem<T>::pack(o);
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      });
    }
    if (!more) break;
    vb = vbn; b = bn; y0 = y0n; x0 = x0n;
  }
}

template <typename T, int ABL = 0>
static int launch_b128a(const Bneck128Params& q, hipStream_t stream) {
  constexpr int smem_bytes = 2 * 21 * 16 * 128 + 65536 + 1024;
  cft_allow_lds<&bottleneck128a_kernel<T, ABL>>(smem_bytes);
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    ncu = n;
  }
  const int grid = q.ntiles < ncu ? q.ntiles : ncu;          // persistent: one workgroup per CU walks the tiles
  hipLaunchKernelGGL((bottleneck128a_kernel<T, ABL>), dim3(grid), dim3(512), smem_bytes, stream, q);
  return cft_check_launch("bottleneck128a_kernel");
}

// p: as cft_bottleneck fills it for 128 channels, with tiles_y / ntiles for 16 x 16-pixel tiles
int cft_set_conv_variant_peek();
int bneck128_asm_launch(const Bneck128Params& p, int dtype, hipStream_t stream) {
  switch (cft_set_conv_variant_peek()) {
    case 9301: return launch_b128a<uint16_t, 1>(p, stream);
    case 9302: return launch_b128a<uint16_t, 2>(p, stream);
    case 9304: return launch_b128a<uint16_t, 4>(p, stream);
    case 9308: return launch_b128a<uint16_t, 8>(p, stream);
    case 9303: return launch_b128a<uint16_t, 3>(p, stream);
    case 9306: return launch_b128a<uint16_t, 6>(p, stream);
    case 9305: return launch_b128a<uint16_t, 5>(p, stream);
    default: break;
  }
  if (dtype == CFT_F16) return launch_b128a<f16_t>(p, stream);
  return launch_b128a<uint16_t>(p, stream);
}

#endif
