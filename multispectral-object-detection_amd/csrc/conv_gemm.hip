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
// Tiling: BM x BN output tile per workgroup of WGM x WGN wave64s (each wave a (BM/WGM)x(BN/WGN)
//   sub-tile of 16x16 MFMA tiles), K step = 128 bytes per row (64 bf16 / 32 f32) = 8 granules.
//   LDS image: row-major 128-B rows, the 16-B granule slot index XOR (row & 7) -> conflict-free
//   ds_read_b128 fragment reads.  Two staging paths:
//     GLDS = true  global_load_lds_dwordx4: HBM/L2 -> LDS directly.  The LDS destination of that
//                  instruction is wave-uniform base + lane*16, i.e. linear, so the XOR swizzle is
//                  applied on the SOURCE side: lane (row r, slot s) fetches k-granule s ^ (r & 7).
//                  Masked granules (image border, K tail, N tail) fetch from a zero page.
//     GLDS = false global -> VGPR -> ds_write_b128 (register staging, loads issued one K step ahead)
//   Double-buffered, one barrier per K step.  (Deeper vmcnt-counted rings, 32x32 MFMA with swapped
//   operands + permlane epilogue, a register-double-buffered 8-wave variant and two hand-scheduled
//   "8-phase" kernels with staggered wave groups were built and measured slower on MI355X - see
//   profiles/r01_gemm_experiments.md, profiles/r02_gemm_experiments.md - and are not part of the library.)
//   bf16: v_mfma_f32_16x16x32_bf16 (lane holds 8 consecutive k of one row = one granule);
//   f32 : 4 x v_mfma_f32_16x16x4_f32 per granule (exact fp32 products, fp32 accumulate).
//   Both operands use the same (lane-group, element) -> k assignment, so the reduction is a
//   permutation of k and needs no knowledge of the instruction's internal k order.
// Epilogue (wave-private, no workgroup barriers): acc (+bias, activation) -> 16-row fp32 LDS strip
//   -> rows re-read as 16-B vectors -> (+residual) -> one rounding -> coalesced 16-B stores.
// Workgroup order is remapped so that consecutive logical tiles (which share A rows / neighbouring
// image rows) run on the same XCD and hit the same 4 MiB L2.
#include "conv_common.h"

// ABLATE (tuning only, bit mask): 1 = no global loads after the first tile (compute-only bound); 2 = no MFMA
// (staging-only bound); 16 = no epilogue (no bias/activation/residual/stores).
// UNIK (uniform K walk; needs GLDS, Cin % K-step == 0, Kpad == K): every thread of the workgroup is at the same (tap, channel
// chunk) in a K step, so the walk over taps / chunks is SCALAR (SGPR) arithmetic and a thread only keeps constant pointers:
// the staging block shrinks from ~80 to ~25 instructions per K step (16 waves issue it in lock step after every barrier,
// with the matrix pipe idle).  Same fetches, same LDS image, same products and k order: bit-identical to the generic path.
// CHAIN (needs GLDS, one N tile, N == 2 K steps): a second, pointwise GEMM runs on the tile before anything is stored - the
// stride-2 Conv in front of a C3 and that C3's packed cv1|cv2 (both consume exactly this tile's pixels).  After the K loop the
// accumulators get the first layer's bias + activation + rounding and are written to LDS AS the second GEMM's A tile (same
// swizzled [row][64 k] images the staging buffers held; the k-th image = channels 64k..64k+63), the second layer's weights are
// fetched into the B buffers, and the same fragment reads / MFMAs walk the two images.  Same values, same roundings, same k
// order as the two launches; the intermediate tensor (M x N elements written, then read) never exists.
// CHAIN + residual (p.res, four-image form only): the first layer is a Bottleneck's 3x3 whose output gets the shortcut added BEFORE the
// one rounding (reference models/common.py:108) and is itself an output (the next shortcut) - the residual tile arrives by LDS-DMA in the
// image layout, every lane adds "its" dwords in fp32 and writes the rounded sum back in place, and the finished images are stored to y1
// granule by granule (32 KiB right before each MFMA step of the second GEMM, drained at the barrier behind it).
// split-K (p.ksplit > 1, pointwise layers on the uniform K walk): see ConvParams.
template <typename T, int BM, int BN, int WGM, int WGN, bool GLDS, int ABLATE = 0, bool UNIK = false, bool CHAIN = false, bool CRES = false>
__global__ void __launch_bounds__(64 * WGM * WGN) conv_gemm_kernel(const ConvParams p) {
  constexpr int NTHR = 64 * WGM * WGN;
  constexpr int GE = Elem<T>::GE;
  constexpr int BK = 8 * GE;
  constexpr int RPP = NTHR / 8;  // tile rows staged per pass of the whole workgroup
  constexpr int A_PER = BM / RPP, B_PER = (BN + RPP - 1) / RPP;   // BN % RPP != 0: the last pass stages only the waves below BN
  constexpr int WM = BM / WGM, WN = BN / WGN, MT = WM / 16, NT = WN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int ES = (int)sizeof(T);
  static_assert(BM % RPP == 0 && BN % 8 == 0 && WM % 16 == 0 && WN % 16 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * A_BYTES;

  // ---- XCD-aware tile assignment (bijective for any grid size) ----
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, slot = bid >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  int ltile = logical, split = 0;
  if (p.ksplit > 1) {                    // uniform
    const int per = nb / p.ksplit;       // host: the grid is tiles * ksplit
    split = logical / per;
    ltile = logical - split * per;
  }
  const int tm = ltile / p.tilesN, tn = ltile - tm * p.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave - wm * WGN;
  const int r0 = tid >> 3;                               // tile row within a pass
  const int slot_s = tid & 7;                            // LDS slot this thread fills
  const int g = GLDS ? (slot_s ^ (r0 & 7)) : slot_s;     // k-granule this thread fetches

  // ---- per-thread gather state: A_PER pixel rows, one k-granule column g ----
  int a_off[A_PER];
  uint32_t a_mask[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int m = m0 + r0 + i * RPP;
    a_off[i] = 0;
    a_mask[i] = 0;
    if (m < p.M) {
      const int t = fast_div(m, p.wo_mul, p.wo_sh);
      const int wo = m - t * p.Wo;
      const int b = fast_div(t, p.ho_mul, p.ho_sh);
      const int ho = t - b * p.Ho;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      a_off[i] = ((b * p.H + hi0) * p.W + wi0) * p.ldx + p.xoff;
      uint32_t wbits = 0, mk = 0;   // tap validity = (row kh in image) x (column kw in image)
      for (int kw = 0; kw < p.KS; ++kw) wbits |= ((unsigned)(wi0 + kw) < (unsigned)p.W ? 1u : 0u) << kw;
      for (int kh = 0; kh < p.KS; ++kh)
        if ((unsigned)(hi0 + kh) < (unsigned)p.H) mk |= wbits << (kh * p.KS);
      a_mask[i] = mk;
    }
  }
  int ci = g * GE, kh = 0, kw = 0, tap = 0;
  while (ci >= p.Cin) { ci -= p.Cin; ++tap; if (++kw == p.KS) { kw = 0; ++kh; } }
  int koff = (kh * p.W + kw) * p.ldx + ci;          // element offset of this thread's k-granule inside the pixel neighbourhood
  const int tap_step = p.ldx - p.Cin;                // next tap in the same kernel row: +ldx, channel index restarts
  const int row_step = (p.W - p.KS) * p.ldx;         // next kernel row: additionally skip to the next image row
  const bool wide_cin = p.Cin >= BK;   // uniform
  // K order.  Default: tap-major - k = (kh, kw, ci), the K steps walk the channels of one tap, then the next tap.  For 3x3
  // convs with Cin >= 4 K steps the walk is CHUNK-major instead - all nine taps of one 64-channel chunk, then the next
  // chunk: a tap re-reads the pixels its neighbour tap just read, and with the chunk fixed that re-use comes one K step
  // later (working set per XCD: 32 CUs x 43 KB) instead of Cin / 64 steps later (32 x 172 KB for Cin = 256 - more than the
  // 4 MiB L2, PMC: 77 % hit rate, 1.8 x the algorithmic HBM reads, and every K step waits for a miss).  Same products,
  // another summation order: results differ from the tap-major walk in the last bits (all tile variants walk alike).
  const bool chunk_major = (ABLATE & 64) ? false : (p.KS == 3 && p.Cin >= 4 * BK && p.Cin % BK == 0 && p.K == 9 * p.Cin);
  int kglob = g * GE;                  // this thread's k position (column of the packed weights)

  gran_t ra[GLDS ? 1 : A_PER], rb[GLDS ? 1 : B_PER];
  const int swz = (slot_s ^ (r0 & 7)) << 4;              // register path: where this thread's granule lands
  const unsigned char* zero_page = reinterpret_cast<const unsigned char*>(cft_zero_page);

// Fetch K step kt_ of this thread's A/B granules (into LDS buffer buf_ when GLDS, else into
// registers), then advance (tap, ci).
#define CFT_LOAD_TILE(kt_, buf_)                                                                       \
  {                                                                                                    \
    const bool kin = kglob < p.K;                                                                      \
    const int tapoff = koff;                                                                           \
    _Pragma("unroll") for (int i = 0; i < A_PER; ++i) {                                                \
      const bool v = kin && ((a_mask[i] >> tap) & 1u);                                                 \
      if constexpr (GLDS) {                                                                            \
        const unsigned char* src = v ? p.x + (long)(a_off[i] + tapoff) * ES : zero_page;               \
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src,                                             \
            (lds_void_t*)(sA + (buf_) * A_BYTES + i * (RPP * 128) + wave * 1024), 16, 0, 0);           \
      } else {                                                                                         \
        gran_t t_ = {0u, 0u, 0u, 0u};                                                                  \
        if (v) t_ = *reinterpret_cast<const gran_t*>(p.x + (long)(a_off[i] + tapoff) * ES);            \
        ra[i] = t_;                                                                                    \
      }                                                                                                \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < B_PER; ++i) {                                                \
      const int n = n0 + r0 + i * RPP;                                                                 \
      if (BN % RPP != 0 && i * RPP + wave * 8 >= BN) continue; /* wave-uniform: no B rows in this pass */ \
      if constexpr (GLDS) {                                                                            \
        const unsigned char* src = (n < p.N) ? p.w + ((long)n * p.Kpad + kglob) * ES : zero_page;      \
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src,                                             \
            (lds_void_t*)(sB + (buf_) * B_BYTES + i * (RPP * 128) + wave * 1024), 16, 0, 0);           \
      } else {                                                                                         \
        gran_t t_ = {0u, 0u, 0u, 0u};                                                                  \
        if (n < p.N) t_ = *reinterpret_cast<const gran_t*>(p.w + ((long)n * p.Kpad + kglob) * ES);    \
        rb[i] = t_;                                                                                    \
      }                                                                                                \
    }                                                                                                  \
    if (chunk_major) { /* same channel chunk, next tap; after the ninth tap the next chunk */          \
      ++tap; ++kw; koff += p.ldx; kglob += p.Cin;                                                      \
      const bool roww = kw == 3;                                                                       \
      kw = roww ? 0 : kw;                                                                              \
      koff += roww ? row_step : 0;                                                                     \
      const bool nextc = tap == 9;                                                                     \
      tap = nextc ? 0 : tap;                                                                           \
      koff += nextc ? BK - 3 * p.W * p.ldx : 0;                                                        \
      kglob += nextc ? BK - 9 * p.Cin : 0;                                                             \
    } else {                                                                                           \
    kglob += BK;                                                                                       \
    ci += BK;                                                                                          \
    koff += BK;                                                                                        \
    if (wide_cin) { /* Cin >= K step: at most one tap boundary per step, branch-free */                \
      const bool wrap = ci >= p.Cin;                                                                   \
      ci -= wrap ? p.Cin : 0;                                                                          \
      koff += wrap ? tap_step : 0;                                                                     \
      tap += wrap ? 1 : 0;                                                                             \
      kw += wrap ? 1 : 0;                                                                              \
      const bool roww = kw == p.KS;                                                                    \
      kw = roww ? 0 : kw;                                                                              \
      koff += roww ? row_step : 0;                                                                     \
    } else {                                                                                           \
      while (ci >= p.Cin) { ci -= p.Cin; koff += tap_step; ++tap; if (++kw == p.KS) { kw = 0; koff += row_step; } } \
    }                                                                                                  \
    }                                                                                                  \
  }
#define CFT_STORE_TILE(buf_)                                                                           \
  if constexpr (!GLDS) {                                                                               \
    _Pragma("unroll") for (int i = 0; i < A_PER; ++i)                                                  \
      *reinterpret_cast<gran_t*>(sA + (buf_) * A_BYTES + (r0 + i * RPP) * 128 + swz) = ra[i];          \
    _Pragma("unroll") for (int i = 0; i < B_PER; ++i)                                                  \
      if (BN % RPP == 0 || r0 + i * RPP < BN)                                                          \
        *reinterpret_cast<gran_t*>(sB + (buf_) * B_BYTES + (r0 + i * RPP) * 128 + swz) = rb[i];        \
  }

  // ---- UNIK: constant per-thread pointers + scalar walk ----
  const unsigned char* ua[UNIK ? A_PER : 1];     // x + (pixel origin + this thread's k-granule) : only dereferenced for in-image taps
  const unsigned char* ub[UNIK ? B_PER : 1];     // w + row n, this thread's k-granule (rows >= N: the zero region)
  int u_tap = 0, u_kw = 0;                       // scalar: tap index and its column
  long u_offa = 0;                               // scalar: byte offset of (tap, chunk) inside the pixel neighbourhood
  int u_offb = 0;                                // scalar: byte offset of the K step inside a weight row
  if constexpr (UNIK) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) ua[i] = p.x + ((long)a_off[i] + g * GE) * ES;
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int n = n0 + r0 + i * RPP;
      ub[i] = (n < p.N) ? p.w + ((long)n * p.Kpad + g * GE) * ES
                        : reinterpret_cast<const unsigned char*>(cft_zero_region) + g * GE * ES;
    }
  }
// One K step of the UNIK path: 2 scalar adds per operand + per A row {mask test, 64-bit add, select}; then the scalar advance
// (tap-major: next 64-channel chunk, after Cin channels the next tap; chunk-major: next tap, after nine taps the next chunk).
#define CFT_LOAD_TILE_U(buf_)                                                                          \
  {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < A_PER; ++i) {                                                \
      const bool v = (a_mask[i] >> u_tap) & 1u;                                                        \
      const unsigned char* src = v ? ua[i] + u_offa : zero_page;                                       \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src,                                               \
          (lds_void_t*)(sA + (buf_) * A_BYTES + i * (RPP * 128) + wave * 1024), 16, 0, 0);             \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < B_PER; ++i) {                                                \
      if (BN % RPP != 0 && i * RPP + wave * 8 >= BN) continue;                                         \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(ub[i] + u_offb),                                  \
          (lds_void_t*)(sB + (buf_) * B_BYTES + i * (RPP * 128) + wave * 1024), 16, 0, 0);             \
    }                                                                                                  \
    if (chunk_major) {                                                                                 \
      ++u_tap; ++u_kw; u_offa += (long)p.ldx * ES; u_offb += p.Cin * ES;                               \
      if (u_kw == 3) { u_kw = 0; u_offa += (long)row_step * ES; }                                      \
      if (u_tap == 9) { u_tap = 0; u_offa += ((long)BK - 3L * p.W * p.ldx) * ES; u_offb += (BK - 9 * p.Cin) * ES; } \
    } else {                                                                                           \
      u_offb += BK * ES; u_offa += BK * ES; u_ci += BK;                                                \
      if (u_ci == p.Cin) {                                                                             \
        u_ci = 0; ++u_tap; ++u_kw; u_offa += (long)tap_step * ES;                                      \
        if (u_kw == p.KS) { u_kw = 0; u_offa += (long)row_step * ES; }                                 \
      }                                                                                                \
    }                                                                                                  \
  }
  int u_ci = 0;                                  // scalar: channel offset of the K step inside its tap (tap-major walk)
  if constexpr (UNIK) {
    if (p.ksplit > 1) {                          // pointwise layer (host): one tap, the walk starts at this split's first K step
      const int k0 = split * p.ksteps * BK;
      u_ci = k0; u_offa = (long)k0 * ES; u_offb = k0 * ES;
    }
  }

// The two 32-wide k sub-steps of one staged K step: fragment reads + MFMAs of LDS buffer buf_.
#define CFT_COMPUTE_STEP(buf_) CFT_COMPUTE_STEP_AT(sA + (buf_) * A_BYTES, sB + (buf_) * B_BYTES)
#define CFT_COMPUTE_STEP_AT(abase_, bbase_)                                                            \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                   \
    const int kg = ks * 4 + lgrp;                                                                      \
    gran_t af[MT], bf[NT];                                                                             \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                   \
      const int row = wm * WM + i * 16 + lrow;                                                         \
      af[i] = *reinterpret_cast<const gran_t*>((abase_) + row * 128 + ((kg ^ (row & 7)) << 4));         \
    }                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                   \
      const int row = wn * WN + j * 16 + lrow;                                                         \
      bf[j] = *reinterpret_cast<const gran_t*>((bbase_) + row * 128 + ((kg ^ (row & 7)) << 4));         \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                     \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                 \
        if constexpr (ABLATE & 2) { asm volatile("" ::"v"(af[i]), "v"(bf[j])); }                       \
        else acc[i][j] = mma_granule<T>(af[i], bf[j], acc[i][j]);                                      \
      }                                                                                                \
  }

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bias_v[NT];
  if constexpr (!(ABLATE & 32)) conv_load_bias<WN>(p, n0, wn, lane, bias_v);

  const int nk = (UNIK && p.ksplit > 1) ? p.ksteps : p.Kpad / BK;
  if constexpr (UNIK) { CFT_LOAD_TILE_U(0) } else { CFT_LOAD_TILE(0, 0) }
  CFT_STORE_TILE(0)
  __syncthreads();
  const int lrow = lane & 15, lgrp = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk && !(ABLATE & 1)) {
      if constexpr (UNIK) { CFT_LOAD_TILE_U(buf ^ 1) } else { CFT_LOAD_TILE(kt + 1, buf ^ 1) }
    }
    CFT_COMPUTE_STEP(buf)
    if (kt + 1 < nk) CFT_STORE_TILE(buf ^ 1)
    __syncthreads();
  }
#undef CFT_LOAD_TILE
#undef CFT_LOAD_TILE_U
#undef CFT_STORE_TILE

  if constexpr (CHAIN) {
    static_assert(GLDS && sizeof(T) == 2 && BN % RPP == 0, "chained GEMM: 16-bit operands, LDS-DMA staging");
    static_assert(BN % 64 == 0, "chained GEMM: the tile is cut into 64-channel images");
    typedef typename Half16<T>::type TH;
    // (every wave is past the barrier that ended the last K step: both staging buffers are free)
    // K steps of the second GEMM = 64-channel images of this tile (host: N == BN).  Two images: they and the second layer's weights
    // take the places of the staging buffers.  Four (256-wide tile): the images fill all of the staging area and the weights stream
    // through ONE extra buffer behind it and through the images already consumed (launch_conv sizes the allocation).
    constexpr int NK2 = BN / 64;
    constexpr bool W2_RESIDENT = NK2 <= 2;
    unsigned char* sB2 = W2_RESIDENT ? sB : smem + NK2 * A_BYTES;
#define CFT_LOAD_W2(k2_, dst_)   /* second layer's weights [N2][N], K step k2_ -> dst_: the staging pattern of the K loop */   \
  _Pragma("unroll") for (int i = 0; i < B_PER; ++i) {                                                   \
    const int n = r0 + i * RPP;                                                                        \
    const unsigned char* src = (n < p.N2) ? p.w2 + ((long)n * p.N + (k2_) * 64 + g * GE) * ES : zero_page; \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)((dst_) + i * (RPP * 128) + wave * 1024), 16, 0, 0); \
  }
    constexpr bool with_res = CRES;                             // (its own instantiation: the plain chained kernel keeps its registers)
    static_assert(!CRES || !W2_RESIDENT, "chained GEMM with a shortcut: four-image form only");
    if constexpr (!W2_RESIDENT) {
      if (with_res) {
        // the shortcut tile -> the image area, in the image layout (row r, slot s holds channel granule s ^ (r & 7) of image k): the staging
        // pattern of the K loop; rows beyond M read the zero page
#pragma unroll
        for (int k2 = 0; k2 < NK2; ++k2)
#pragma unroll
          for (int i = 0; i < A_PER; ++i) {
            const int m = m0 + r0 + i * RPP;
            const unsigned char* src = (m < p.M) ? p.res + ((long)m * p.ldr + p.roff + k2 * 64 + g * GE) * ES : zero_page;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(sA + k2 * A_BYTES + i * (RPP * 128) + wave * 1024), 16, 0, 0);
          }
      }
    }
    if constexpr (W2_RESIDENT) {
#pragma unroll
      for (int k2 = 0; k2 < NK2; ++k2) CFT_LOAD_W2(k2, sB2 + k2 * B_BYTES)
    } else {
      CFT_LOAD_W2(0, sB2)
    }
    float bias2_v[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = wn * WN + j * 16 + lrow;
      bias2_v[j] = (p.bias2 != nullptr && n < p.N2) ? p.bias2[n] : 0.0f;
    }
    // First layer's bias + SiLU + rounding on the accumulators; lanes l / l^1 hold neighbouring channels of the same four pixels:
    // they swap half of their values (one DPP move) so that each writes two packed channel PAIRS (even lane: pixels 0,1; odd: 2,3).
    const bool odd = lane & 1;
    if (with_res) {
      __syncthreads();                                     // the shortcut tile (and W2 step 0) landed, visible to every wave
      // lanes l / l^1 exchange fp32 values (two DPP moves) so that each owns the channel PAIR (c, c + 1) of two pixels; it reads the shortcut's
      // dword for each, adds in fp32 - act(acc + bias) + shortcut, the operation order of the plain epilogue - rounds once and writes the dword back
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = apply_act<CFT_ACT_SILU>(acc[i][j][e] + bias_v[j]);
          const float ga = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(odd ? v[0] : v[2]), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
          const float gb = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(odd ? v[1] : v[3]), 0xB1, 0xF, 0xF, true));
          float lo0 = odd ? ga : v[0], hi0 = odd ? v[2] : ga;      // pixel A: channels c, c + 1
          float lo1 = odd ? gb : v[1], hi1 = odd ? v[3] : gb;      // pixel B = A + 1
          const int row = wm * WM + i * 16 + lgrp * 4 + (odd ? 2 : 0);
          const int col = wn * WN + j * 16 + (lrow & 14);
          const int c = col & 63;
          unsigned char* img = sA + (col >> 6) * A_BYTES + (c & 7) * 2;
          uint32_t* pa = reinterpret_cast<uint32_t*>(img + row * 128 + (((c >> 3) ^ (row & 7)) << 4));
          uint32_t* pb = reinterpret_cast<uint32_t*>(img + (row + 1) * 128 + (((c >> 3) ^ ((row + 1) & 7)) << 4));
          float r0l, r0h, r1l, r1h;
          Elem<TH>::unpack2(*pa, r0l, r0h);
          Elem<TH>::unpack2(*pb, r1l, r1h);
          lo0 += r0l; hi0 += r0h; lo1 += r1l; hi1 += r1h;
          *pa = Elem<TH>::pack2(lo0, hi0);
          *pb = Elem<TH>::pack2(lo1, hi1);
          acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    } else {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act<CFT_ACT_SILU>(acc[i][j][e] + bias_v[j]);
        const uint32_t r01 = Elem<TH>::pack2(v[0], v[1]), r23 = Elem<TH>::pack2(v[2], v[3]);
        const uint32_t got = (uint32_t)__builtin_amdgcn_mov_dpp((int)(odd ? r01 : r23), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        const uint32_t d0 = odd ? ((got & 0xffffu) | (r23 << 16)) : ((r01 & 0xffffu) | (got << 16));
        const uint32_t d1 = odd ? ((got >> 16) | (r23 & 0xffff0000u)) : ((r01 >> 16) | (got & 0xffff0000u));
        const int row = wm * WM + i * 16 + lgrp * 4 + (odd ? 2 : 0);
        const int col = wn * WN + j * 16 + (lrow & 14);
        const int c = col & 63;
        unsigned char* img = sA + (col >> 6) * A_BYTES + (c & 7) * 2;
        *reinterpret_cast<uint32_t*>(img + row * 128 + (((c >> 3) ^ (row & 7)) << 4)) = d0;
        *reinterpret_cast<uint32_t*>(img + (row + 1) * 128 + (((c >> 3) ^ ((row + 1) & 7)) << 4)) = d1;
        acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    if constexpr (W2_RESIDENT) {
      __syncthreads();
#pragma unroll
      for (int k2 = 0; k2 < NK2; ++k2) { CFT_COMPUTE_STEP_AT(sA + k2 * A_BYTES, sB2 + k2 * B_BYTES) }
    } else {
      // Four images I0..I3, one extra buffer X.  An image is dead once its K step is done, so the later weight steps land in dead
      // images: only the second step's weights are waited for with nothing to do (X and I0 are both free only after step 0).
      static_assert(NK2 == 4 && B_BYTES <= A_BYTES, "chained GEMM: streamed second-layer weights are scheduled for four K steps");
// residual form: image k2_ IS a quarter of the first layer's output tile (final values) - store it to y1 as 16-byte granules (thread (r0, slot)
// holds granule slot ^ (row & 7) of its rows: eight consecutive threads write one full 128-byte line).  Issued right before an MFMA step; the
// barrier behind that step drains the stores, so an image is never overwritten (by a later W2 step) before it is in flight to memory.
#define CFT_STORE_IMAGE(k2_)                                                                           \
      if (with_res) {                                                                                  \
        _Pragma("unroll") for (int i = 0; i < A_PER; ++i) {                                            \
          const int row = r0 + i * RPP, m = m0 + row;                                                  \
          const gran_t t_ = *reinterpret_cast<const gran_t*>(sA + (k2_) * A_BYTES + row * 128 + (slot_s << 4)); \
          if (m < p.M) *reinterpret_cast<gran_t*>(p.y1 + ((long)m * p.ldy1 + p.yoff1 + (k2_) * 64 + g * GE) * ES) = t_; \
        }                                                                                              \
      }
      __syncthreads();                                     // images written, W2 step 0 in X
      CFT_STORE_IMAGE(0)
      CFT_COMPUTE_STEP_AT(sA, sB2)
      __syncthreads();                                     // X and I0 are free (I0 is on its way to y1)
      CFT_LOAD_W2(1, sB2)
      CFT_LOAD_W2(2, sA)
      CFT_STORE_IMAGE(1)
      __syncthreads();                                     // both landed
      CFT_COMPUTE_STEP_AT(sA + A_BYTES, sB2)
      __syncthreads();                                     // I1 is free
      CFT_LOAD_W2(3, sA + A_BYTES)                         // lands under step 2
      CFT_STORE_IMAGE(2)
      CFT_COMPUTE_STEP_AT(sA + 2 * A_BYTES, sA)
      __syncthreads();
      CFT_STORE_IMAGE(3)
      CFT_COMPUTE_STEP_AT(sA + 3 * A_BYTES, sA + A_BYTES)
#undef CFT_STORE_IMAGE
    }
#undef CFT_LOAD_W2
    __syncthreads();
    ConvParams p2 = p;
    p2.N = p.N2;
    p2.res = nullptr;
    conv_epilogue<TH, WM, WN>(p2, acc, smem, m0, 0, wm, wn, wave, lane, bias2_v);
    return;
  }
  if constexpr (ABLATE & 16) {   // timing probe: no epilogue (keep the accumulators alive)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  if constexpr (ABLATE & 32) conv_load_bias<WN>(p, n0, wn, lane, bias_v);   // A/B probe: the round-1 placement
  if (p.ksplit > 1) {                    // uniform: fp32 partial sums of this split (host: out_f32, no activation, no residual)
    ConvParams ps = p;
    ps.y = p.y + (long)split * p.M * p.ldy * 4;
    if (split != 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j) bias_v[j] = 0.0f;
    }
    conv_epilogue_impl<typename Half16<T>::type, WM, WN, CFT_ACT_NONE, true>(ps, acc, smem, m0, n0, wm, wn, wave, lane, bias_v);
    return;
  }
  conv_epilogue<typename Half16<T>::type, WM, WN>(p, acc, smem, m0, n0, wm, wn, wave, lane, bias_v);
}

// ------------------------------------------------------------------------------------ host
// (mul, sh) with floor(n / d) == umulhi(n, mul) >> sh for all 0 <= n < 2^31 (d >= 2); mul = 0 encodes d == 1.
static void set_magic(int d, uint32_t& mul, uint32_t& sh) {
  if (d <= 1) { mul = 0; sh = 0; return; }
  int l = 0;
  while ((1L << l) < d) ++l;                         // l = ceil(log2 d)
  const unsigned long long k = 31ULL + l;           // 2^k / d < 2^32
  mul = (uint32_t)(((1ULL << k) / (unsigned long long)d) + 1ULL);
  sh = (uint32_t)(k - 32);
}


// Tile variants.  0 = automatic choice; the others force one configuration (tuning / A-B tests).
// Per host thread (two forwards may be issued from two threads; a test's cft_set_conv_variant must not reach the other's launches).
thread_local int g_conv_variant = 0;   // also read by bottleneck.hip in probe builds (-DCFT_PROBES)
extern "C" int cft_set_conv_variant(int v) {
  const int old = g_conv_variant;
  g_conv_variant = v;
  return old;
}

int cft_set_conv_variant_peek() { return g_conv_variant; }   // (bottleneck.hip: variant 97 = the round-5 kernels)

template <typename T, int BM, int BN, int WGM, int WGN, bool GLDS, int ABLATE = 0, bool UNIK = false, bool CHAIN = false, bool CRES = false>
static int launch_conv(const ConvParams& p, hipStream_t stream) {
  // (chained kernel with four 64-channel images: they take all of the staging area, the second layer's weights one buffer behind it)
  constexpr int smem_bytes = (CHAIN && BN / 64 > 2) ? (BN / 64) * BM * 128 + BN * 128 : 2 * (BM + BN) * 128;
  static_assert(!CHAIN || (BN / 64) * BM * 128 >= 2 * (BM + BN) * 128 || BN / 64 <= 2, "chained GEMM: LDS map");
  cft_allow_lds<&conv_gemm_kernel<T, BM, BN, WGM, WGN, GLDS, ABLATE, UNIK, CHAIN, CRES>>(smem_bytes);
  ConvParams q = p;
  const int tilesM = (p.M + BM - 1) / BM;
  q.tilesN = (p.N + BN - 1) / BN;
  if (q.ksplit < 1) q.ksplit = 1;
  if (!UNIK && q.ksplit > 1) {     // split-K exists on the uniform K walk only: parts[1..] would stay uninitialised
    cft_set_error("conv_gemm_kernel: split-K needs the uniform-K-walk instantiation (this variant forces the generic address path)");
    return CFT_EINVAL;
  }
  hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, WGM, WGN, GLDS, ABLATE, UNIK, CHAIN, CRES>), dim3(tilesM * q.tilesN * q.ksplit), dim3(64 * WGM * WGN), smem_bytes, stream, q);
  return cft_check_launch("conv_gemm_kernel");
}

// Automatic choice: the uniform-K-walk form of a tile configuration whenever the layer allows it (every layer of the CFT
// networks except the Focus conv and the 80 / 160 / 320-channel layers of yolov5x); variant 900 forces the generic path (A/B).
template <typename T, int BM, int BN, int WGM, int WGN>
static int launch_auto(const ConvParams& p, hipStream_t stream) {
  constexpr int BK = 8 * Elem<T>::GE;
  const bool unik = g_conv_variant != 900 && p.Cin % BK == 0 && p.Kpad == p.K && 2L * p.Kpad * (long)sizeof(T) + 128 <= CFT_ZERO_REGION_BYTES;
  if (unik) return launch_conv<T, BM, BN, WGM, WGN, true, 0, true>(p, stream);
  return launch_conv<T, BM, BN, WGM, WGN, true>(p, stream);
}

#ifdef CFT_PROBES
// Eligibility of the ring kernel: 16-bit operands, Cin a multiple of the 64-wide K step (uniform walk), no K padding.
template <typename T>
static bool ring_ok(const ConvParams& p) {
  // (the last clause: masked granules are fetched at the out-of-range buffer offset 2^31, so both buffers must end below it)
  return sizeof(T) == 2 && p.Cin % 64 == 0 && p.Kpad == p.K && p.KS <= 3 &&
         p.x_bytes + 2L * ((long)p.W + 1) * p.ldx < (1L << 31) && p.w_bytes < (1L << 31);
}

template <typename T, int ABLATE = 0>
static int launch_ring(const ConvParams& p, hipStream_t stream) {
  return conv_ring_launch(p, sizeof(T) != 2 ? CFT_F32 : (__is_same(T, f16_t) ? CFT_F16 : CFT_BF16), ABLATE, stream);   // conv_ring.hip
}
#endif

template <typename T>
static constexpr int dtype_code() { return sizeof(T) != 2 ? CFT_F32 : (__is_same(T, f16_t) ? CFT_F16 : CFT_BF16); }

template <typename T>
static int dispatch_conv(const ConvParams& p, hipStream_t stream) {
  switch (g_conv_variant) {
    case 96: case 961: case 962: case 963: case 964:   // the hand-scheduled 8-wave kernel wherever eligible, tile heights 256 / 224 / 208 / 192 / 128 (tests)
      if (conv_asm_ok(p, dtype_code<T>())) return conv_asm_launch(p, dtype_code<T>(), g_conv_variant == 96 ? 0 : g_conv_variant - 960, stream);
      break;
    case 1: return launch_conv<T, 128, 128, 2, 2, false>(p, stream);   // register-staged baseline
    case 2: return launch_auto<T, 128, 128, 2, 2>(p, stream);
    case 4: return launch_auto<T, 128, 64, 2, 2>(p, stream);
    case 6: return launch_auto<T, 256, 64, 4, 2>(p, stream);
    case 7: return launch_auto<T, 64, 128, 2, 2>(p, stream);
    case 8: return launch_auto<T, 64, 64, 2, 2>(p, stream);
    case 23: return launch_auto<T, 128, 128, 2, 4>(p, stream);
    case 27: return launch_auto<T, 256, 256, 4, 4>(p, stream);
    case 30: return launch_auto<T, 512, 128, 8, 2>(p, stream);
    case 32: return launch_auto<T, 512, 64, 8, 2>(p, stream);
    case 33: return launch_auto<T, 256, 128, 4, 2>(p, stream);
    case 51: return launch_auto<T, 192, 128, 2, 4>(p, stream);
    case 60: return launch_auto<T, 128, 256, 4, 4>(p, stream);
    case 63: return launch_auto<T, 64, 128, 2, 4>(p, stream);
    case 70: return launch_auto<T, 256, 160, 8, 2>(p, stream);   // 80/160-wide tiles: yolov5x / yolov5m channel counts
    case 71: return launch_auto<T, 256, 160, 4, 2>(p, stream);
    case 72: return launch_auto<T, 128, 160, 4, 2>(p, stream);
    case 73: return launch_auto<T, 256, 80, 4, 1>(p, stream);
    case 74: return launch_auto<T, 256, 80, 8, 1>(p, stream);
    case 75: return launch_auto<T, 128, 80, 4, 1>(p, stream);
#ifdef CFT_PROBES   // timing probes / A-B variants (tools/build_probes.sh): results of 1xx/2xx/3xx/16xx are wrong by construction
    case 3223: return launch_conv<T, 128, 128, 2, 4, true, 32>(p, stream);   // 32xx: bias loaded at the epilogue (round-1 placement)
    case 3251: return launch_conv<T, 192, 128, 2, 4, true, 32>(p, stream);
    case 3227: return launch_conv<T, 256, 256, 4, 4, true, 32>(p, stream);
    case 6427: return launch_conv<T, 256, 256, 4, 4, true, 64>(p, stream);   // 64xx: tap-major K walk for every layer (round-2 A/B)
    case 1627: return launch_conv<T, 256, 256, 4, 4, true, 16>(p, stream);
    case 127: return launch_conv<T, 256, 256, 4, 4, true, 1>(p, stream);
    case 227: return launch_conv<T, 256, 256, 4, 4, true, 2>(p, stream);
    case 327: return launch_conv<T, 256, 256, 4, 4, true, 3>(p, stream);
    case 91: if (ring_ok<T>(p)) return launch_ring<T>(p, stream); return launch_auto<T, 256, 256, 4, 4>(p, stream);   // the 8-wave kernel wherever eligible
    case 190: if (ring_ok<T>(p)) return launch_ring<T, 1>(p, stream); break;    // ring kernel: no global loads after the prologue's
    case 290: if (ring_ok<T>(p)) return launch_ring<T, 2>(p, stream); break;    // no MFMAs
    case 1690: if (ring_ok<T>(p)) return launch_ring<T, 16>(p, stream); break;  // no epilogue
#endif
    default: break;
  }
  // Automatic choice (measured on MI355X with tools/gemm_bench.py, yolov5l+CFTx3 layer shapes, bf16):
  //  * the more waves per CU the better the latency hiding: 16-wave workgroups with 64x64 wave tiles
  //    (256x256 for wide layers, 512x128 for 128-channel layers) reach 0.9-1.1 PFLOP/s on K >= 1152;
  //  * HBM-bound layers (1x1, K <= 256) run best on 8-wave 128x128 / 256x64 tiles (4.3-4.8 TB/s);
  //  * a tile configuration is only used if it yields at least one workgroup per CU.
  auto tiles = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * (p.ksplit > 1 ? p.ksplit : 1); };   // workgroups
  const long kCUs = 192;   // accept a configuration once it yields >= 0.75 workgroups per CU (256 CUs)
  if (p.N <= 64) {
    if (tiles(256, 64) >= kCUs) return launch_auto<T, 256, 64, 4, 2>(p, stream);
    if (tiles(128, 64) >= kCUs) return launch_auto<T, 128, 64, 2, 2>(p, stream);
    return launch_auto<T, 64, 64, 2, 2>(p, stream);
  }
  if (p.N > 64) {
    // channel counts that are multiples of 80 / 160 but not of 128 (yolov5x: 80, 160, 320): the 128/256-wide tiles
    // would compute up to 37 % padding.  128x160 / 128x80 (8 / 4 waves, two+ workgroups per CU) measured -14..-27 %
    // on those layers; taken only when they cut the padded width by >= 10 %, so 64/128/256-multiples are unaffected.
    const long p128 = (long)((p.N + 127) / 128) * 128, p256 = (long)((p.N + 255) / 256) * 256;
    const long cur = (p.N <= 128) ? p128 : ((p256 * 100 <= p128 * 115) ? p256 : p128);
    const long p160 = (long)((p.N + 159) / 160) * 160, p80 = (long)((p.N + 79) / 80) * 80;
    if (p160 * 100 <= cur * 90 && p160 <= p80 && tiles(128, 160) >= kCUs / 2) return launch_auto<T, 128, 160, 4, 2>(p, stream);
    if (p80 * 100 <= cur * 90 && p80 < p160 && tiles(128, 80) >= kCUs / 2) return launch_auto<T, 128, 80, 4, 1>(p, stream);
  }
  if (p.N <= 128) {
    // 192x128 with 8 waves is the largest 128-wide tile of which TWO workgroups fit a CU (80 KiB LDS each): the
    // store/residual burst of one workgroup's epilogue overlaps the other's K loop (+5..8 % over 512x128x16w).
    if (tiles(192, 128) >= 2 * kCUs) return launch_auto<T, 192, 128, 2, 4>(p, stream);
    if (tiles(128, 128) >= 2 * kCUs) return launch_auto<T, 128, 128, 2, 4>(p, stream);
    return launch_auto<T, 64, 128, 2, 4>(p, stream);          // few tiles: 3 workgroups of 8 waves per CU
  }
  // wide layers: prefer 256-wide tiles unless the N tail would waste much more than 128-wide tiles do
  const long pad256 = (long)((p.N + 255) / 256) * 256, pad128 = (long)((p.N + 127) / 128) * 128;
  const bool wide_ok = pad256 * 100 <= pad128 * 115;
  // the hand-scheduled 8-wave kernel (conv_gemm_asm.hip: bit-identical, K loop at 1.27 instead of 1.5 us per 256-row step) from 12 K steps on (below that
  // the 16-wave kernel's shorter prologue + epilogue outweigh the loop, profiles/r06_asm_kloop.md), its tile height picked so that the tile count
  // fills whole rounds of the CUs; variant 97: the round-5 choice
  static const int asm_min_steps = [] { const char* e = getenv("CFT_ASM_MIN_STEPS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 12; }();   // (tuning knob)
  if (wide_ok && g_conv_variant != 97 && p.Kpad >= asm_min_steps * 64 && p.ksplit <= 1 && conv_asm_ok(p, dtype_code<T>())) {
    const int tile = conv_asm_choose(p);
    if (tile >= 0) return conv_asm_launch(p, dtype_code<T>(), tile, stream);
  }
  if (wide_ok && p.Kpad >= 256 && tiles(256, 256) >= kCUs) {
#ifdef CFT_PROBES
    if (g_conv_variant == 90 && ring_ok<T>(p)) return launch_ring<T>(p, stream);    // A/B: the 8-wave kernel in place of the 16-wave 256x256 tile
#endif
    return launch_auto<T, 256, 256, 4, 4>(p, stream);
  }
  if (wide_ok && tiles(128, 256) >= kCUs) return launch_auto<T, 128, 256, 4, 4>(p, stream);   // e.g. CFT fc2 at M = 8192
  if (tiles(192, 128) >= 2 * kCUs) return launch_auto<T, 192, 128, 2, 4>(p, stream);
  if (tiles(128, 128) >= 2 * kCUs) return launch_auto<T, 128, 128, 2, 4>(p, stream);
  return launch_auto<T, 64, 128, 2, 4>(p, stream);
}

// Validate one conv layer's geometry and fill the launch parameters (shared by cft_conv2d and cft_conv2d_chain).
static int fill_conv_params(ConvParams& p, const void* x, const void* w, const float* bias, const void* res, void* y,
                            int B, int H, int W, int cin, int ldx, int xoff, int n, int kpad, int ksize, int stride,
                            int ldy, int yoff, int ldr, int roff, int act, int dtype, int out_dtype, int res_dtype) {
  CFT_REQUIRE(x && w && y, "cft_conv2d: null pointer");
  CFT_REQUIRE(cft_is_dtype(dtype), "cft_conv2d: dtype must be CFT_BF16, CFT_F16 or CFT_F32");
  CFT_REQUIRE(out_dtype == dtype || out_dtype == CFT_F32, "cft_conv2d: out_dtype must be the compute dtype or CFT_F32");
  CFT_REQUIRE(res == nullptr || res_dtype == dtype || res_dtype == CFT_F32, "cft_conv2d: res_dtype must be the compute dtype or CFT_F32");
  CFT_REQUIRE(B > 0 && H > 0 && W > 0 && cin > 0 && n > 0, "cft_conv2d: non-positive size");
  CFT_REQUIRE(ksize >= 1 && ksize <= 5 && (ksize & 1) && stride >= 1, "cft_conv2d: ksize must be 1, 3 or 5");
  const int ge = cft_granule(dtype), bk = 8 * ge;
  CFT_REQUIRE(cin % ge == 0 && ldx % ge == 0 && xoff % ge == 0, "cft_conv2d: input channels/ld/offset not granule aligned");
  CFT_REQUIRE(kpad % bk == 0 && kpad >= ksize * ksize * cin, "cft_conv2d: kpad must cover k*k*cin and be a multiple of the K step");
  CFT_REQUIRE(n % 8 == 0 && ldy % 8 == 0 && yoff % 8 == 0, "cft_conv2d: n/ldy/yoff must be multiples of 8");
  CFT_REQUIRE(res == nullptr || (ldr % 8 == 0 && roff % 8 == 0), "cft_conv2d: residual ld/offset must be multiples of 8");
  const int pad = ksize / 2;
  const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
  const long M = (long)B * Ho * Wo;
  CFT_REQUIRE(M < (1L << 31) && (long)B * H * W * ldx < (1L << 31) && M * ldy < (1L << 31),
              "cft_conv2d: tensor exceeds 2^31 elements (split the batch)");
  p.x = (const unsigned char*)x; p.w = (const unsigned char*)w; p.bias = bias;
  p.res = (const unsigned char*)res; p.y = (unsigned char*)y;
  p.H = H; p.W = W; p.Cin = cin; p.ldx = ldx; p.xoff = xoff;
  p.Ho = Ho; p.Wo = Wo; p.N = n; p.Kpad = kpad; p.K = ksize * ksize * cin;
  p.ldy = ldy; p.yoff = yoff; p.ldr = ldr; p.roff = roff;
  p.KS = ksize; p.stride = stride; p.pad = pad;
  p.act = act; p.out_f32 = out_dtype == CFT_F32; p.res_f32 = res_dtype == CFT_F32;
  p.M = (int)M; p.tilesN = 0;
  p.w2 = nullptr; p.bias2 = nullptr; p.N2 = 0;
  p.y1 = nullptr; p.ldy1 = 0; p.yoff1 = 0; p.ksplit = 1; p.ksteps = 0;
  p.x_bytes = (long)B * H * W * ldx * cft_elem_size(dtype);
  p.w_bytes = (long)n * kpad * cft_elem_size(dtype);
  set_magic(Wo, p.wo_mul, p.wo_sh);
  set_magic(Ho, p.ho_mul, p.ho_sh);
  return CFT_OK;
}

extern "C" int cft_conv2d(const void* x, const void* w, const float* bias, const void* res, void* y,
                          int B, int H, int W, int cin, int ldx, int xoff,
                          int n, int kpad, int ksize, int stride,
                          int ldy, int yoff, int ldr, int roff,
                          int act, int dtype, int out_dtype, int res_dtype, void* stream) {
  ConvParams p;
  const int rc = fill_conv_params(p, x, w, bias, res, y, B, H, W, cin, ldx, xoff, n, kpad, ksize, stride,
                                  ldy, yoff, ldr, roff, act, dtype, out_dtype, res_dtype);
  if (rc != CFT_OK) return rc;
  CFT_DISPATCH_DTYPE(dtype, T, return dispatch_conv<T>(p, as_stream(stream)));
  return CFT_EINVAL;
}

// Which layer pairs the chained kernel takes (cft_conv2d_chain / cft_conv2d_chain_ok): 16-bit operands, a first layer on the
// uniform K walk whose whole width is ONE tile (128 or 256 channels = two or four 64-channel images), SiLU, and a pointwise
// second layer no wider than the first - in the CFT networks the stride-2 Convs 64 -> 128 and 128 -> 256 in front of the
// first two C3s of each stream (the 256 -> 512 conv does not fit: 512 channels x 256 pixels x 2 B = 256 KiB).
static bool chain_ok(const ConvParams& p, int n2, int dtype) {
  return (dtype == CFT_BF16 || dtype == CFT_F16) && (p.N == 128 || p.N == 256) && n2 >= 8 && n2 <= p.N && n2 % 8 == 0 && p.Cin % 64 == 0 && p.Kpad == p.K &&
         2L * p.Kpad * 2 + 128 <= CFT_ZERO_REGION_BYTES;
}

template <typename T>
static int dispatch_chain(const ConvParams& p, hipStream_t stream) {
  if constexpr (sizeof(T) == 2) {
    if (p.N == 128) return launch_conv<T, 192, 128, 2, 4, true, 0, true, true>(p, stream);   // two workgroups per CU (80 KiB)
    // 256-channel first layer: the 8-wave kernel with the hand-scheduled first K loop (conv_gemm_asm.hip; variant 97: the 16-wave chained kernel)
    if (g_conv_variant != 97 && conv_asm_chain_ok(p, dtype_code<T>())) return conv_asm_chain_launch(p, dtype_code<T>(), stream);
    if (p.y1 != nullptr) return launch_conv<T, 256, 256, 4, 4, true, 0, true, true, true>(p, stream);   // + shortcut (cft_conv2d_chain_res)
    return launch_conv<T, 256, 256, 4, 4, true, 0, true, true>(p, stream);                   // 16 waves, all 160 KiB
  } else return CFT_EINVAL;
}

extern "C" int cft_conv2d_chain(const void* x, const void* w1, const float* bias1, const void* w2, const float* bias2, void* y,
                                int B, int H, int W, int cin, int ldx, int xoff,
                                int n1, int kpad1, int ksize, int stride, int n2,
                                int ldy, int yoff, int act2, int dtype, void* stream) {
  CFT_REQUIRE(w2 != nullptr, "cft_conv2d_chain: null pointer");
  ConvParams p;
  const int rc = fill_conv_params(p, x, w1, bias1, nullptr, y, B, H, W, cin, ldx, xoff, n1, kpad1, ksize, stride,
                                  ldy, yoff, 0, 0, act2, dtype, dtype, dtype);
  if (rc != CFT_OK) return rc;
  CFT_REQUIRE(n2 % 8 == 0 && n2 > 0, "cft_conv2d_chain: n2 must be a positive multiple of 8");
  CFT_REQUIRE(chain_ok(p, n2, dtype), "cft_conv2d_chain: layer pair not eligible (ask cft_conv2d_chain_ok; run the two layers with cft_conv2d)");
  p.w2 = (const unsigned char*)w2; p.bias2 = bias2; p.N2 = n2;
  CFT_DISPATCH_DTYPE(dtype, T, return dispatch_chain<T>(p, as_stream(stream)));
  return CFT_EINVAL;
}

// The chained pair with a shortcut on the FIRST layer (a Bottleneck's 3x3, reference models/common.py:108: x + cv2(cv1(x))) whose sum is
// also the next shortcut: y1 = SiLU(conv(x) + b1) + res (one rounding), y2 = act2(conv1x1(y1) + b2).  256-channel first layers only (the
// four-image form of the chained kernel); y1 may alias res.
extern "C" int cft_conv2d_chain_res(const void* x, const void* w1, const float* bias1, const void* res, void* y1,
                                    const void* w2, const float* bias2, void* y2,
                                    int B, int H, int W, int cin, int ldx, int xoff,
                                    int n1, int kpad1, int ksize, int stride, int ldr, int roff, int ldy1, int yoff1,
                                    int n2, int ldy2, int yoff2, int act2, int dtype, void* stream) {
  CFT_REQUIRE(w2 != nullptr && res != nullptr && y1 != nullptr, "cft_conv2d_chain_res: null pointer");
  ConvParams p;
  const int rc = fill_conv_params(p, x, w1, bias1, res, y2, B, H, W, cin, ldx, xoff, n1, kpad1, ksize, stride,
                                  ldy2, yoff2, ldr, roff, act2, dtype, dtype, dtype);
  if (rc != CFT_OK) return rc;
  CFT_REQUIRE(n2 % 8 == 0 && n2 > 0 && ldy2 >= yoff2 + n2, "cft_conv2d_chain_res: n2 must be a positive multiple of 8 and fit ldy2");
  CFT_REQUIRE(chain_ok(p, n2, dtype) && n1 == 256, "cft_conv2d_chain_res: layer pair not eligible (256-channel first layer, see cft_conv2d_chain_ok)");
  CFT_REQUIRE(ldy1 % 8 == 0 && yoff1 % 8 == 0 && ldy1 >= yoff1 + n1 && ldr >= roff + n1, "cft_conv2d_chain_res: y1 / res ld and offset must be multiples of 8 and cover n1 channels");
  CFT_REQUIRE((long)p.M * ldy1 < (1L << 31) && (long)p.M * ldr < (1L << 31), "cft_conv2d_chain_res: tensor exceeds 2^31 elements (split the batch)");
  p.w2 = (const unsigned char*)w2; p.bias2 = bias2; p.N2 = n2;
  p.y1 = (unsigned char*)y1; p.ldy1 = ldy1; p.yoff1 = yoff1;
  CFT_DISPATCH_DTYPE(dtype, T, return dispatch_chain<T>(p, as_stream(stream)));
  return CFT_EINVAL;
}

// split-K form of a pointwise layer / nn.Linear: parts[s] (fp32, [rows][n], s < splits) = x[:, s*K/splits : (s+1)*K/splits] . w^T (+ bias
// in s = 0).  For GEMMs with few output tiles and a long K loop (the CFT block's out_proj / fc2 at 8192 rows: reference
// models/common.py:511, :532-538): splits x the workgroups, each with 1 / splits of the K steps.  The partial sums are added to the fp32
// residual stream - in a FIXED order, so results are reproducible - by the LayerNorm that reads it next (cft_layernorm_reduce).
extern "C" int cft_linear_splitk(const void* x, const void* w, const float* bias, float* parts,
                                 int rows, int cin, int ldx, int n, int kpad, int splits, int dtype, void* stream) {
  CFT_REQUIRE(parts != nullptr, "cft_linear_splitk: null pointer");
  CFT_REQUIRE(splits >= 2 && splits <= 8, "cft_linear_splitk: 2 <= splits <= 8");
  ConvParams p;
  const int rc = fill_conv_params(p, x, w, bias, nullptr, parts, 1, 1, rows, cin, ldx, 0, n, kpad, 1, 1,
                                  n, 0, 0, 0, CFT_ACT_NONE, dtype, CFT_F32, dtype);
  if (rc != CFT_OK) return rc;
  const int bk = 8 * cft_granule(dtype);
  CFT_REQUIRE(cin % bk == 0 && kpad == cin && 2L * kpad * cft_elem_size(dtype) + 128 <= CFT_ZERO_REGION_BYTES,
              "cft_linear_splitk: needs the uniform K walk (cin a multiple of the K step, no K padding)");
  CFT_REQUIRE((kpad / bk) % splits == 0, "cft_linear_splitk: the K steps must divide evenly among the splits");
  CFT_REQUIRE((long)splits * rows * n < (1L << 31), "cft_linear_splitk: partial-sum buffer exceeds 2^31 elements");
  p.ksplit = splits; p.ksteps = kpad / bk / splits;
  const int saved = g_conv_variant;
  if (saved == 900 || saved == 1) g_conv_variant = 0;   // (variants 900 / 1 force the generic address path: split-K lives on the uniform walk)
  int st = CFT_EINVAL;
  CFT_DISPATCH_DTYPE(dtype, T, st = dispatch_conv<T>(p, as_stream(stream)));
  g_conv_variant = saved;
  return st;
}

// 1 if cft_conv2d_chain takes this pair of layers (first layer: its cft_conv2d geometry, SiLU; second: pointwise, n2 outputs; ldx / ldy: channels
// per pixel of the input / output buffers), else 0.  Runs the launcher's OWN validation (fill_conv_params + chain_ok) on dummy pointers, so
// "ok" and "cft_conv2d_chain accepts" cannot drift apart (the 2^31-element extents were missing here).
extern "C" int cft_conv2d_chain_ok(int B, int H, int W, int cin, int ldx, int n1, int kpad1, int ksize, int stride, int n2, int ldy, int dtype) {
  if (!cft_is_dtype(dtype) || n2 <= 0 || n2 % 8 != 0) return 0;
  static const char dummy = 0;
  ConvParams p;
  const int rc = fill_conv_params(p, &dummy, &dummy, nullptr, nullptr, (void*)&dummy, B, H, W, cin, ldx, 0, n1, kpad1, ksize, stride,
                                  ldy, 0, 0, 0, CFT_ACT_SILU, dtype, dtype, dtype);
  return rc == CFT_OK && ldy >= n2 && ldx >= cin && chain_ok(p, n2, dtype) ? 1 : 0;
}
