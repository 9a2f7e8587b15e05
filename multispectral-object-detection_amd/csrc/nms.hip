// Batched non-maximum suppression on the pre-NMS detection tensor (reference
// utils/general.py:455-543 `non_max_suppression`, which loops over images in Python and calls
// torchvision.ops.nms).  One 1024-thread workgroup per image, everything stays on the GPU:
//   phase 1  candidate filter + compaction: obj > conf_thres, conf = obj * cls, best class (or every class
//            above the threshold when multi_label), optional class filter, xywh -> xyxy, append to the
//            image's scratch list (LDS atomic cursor);
//   phase 1b the reference's `max_nms` pre-truncation (:469, :515-516): if more than max_nms candidates remain,
//            only the max_nms highest confidences take part - an 8-bit-per-pass radix select of the max_nms-th
//            score in LDS, then everything below it is dropped (candidates that TIE with that score all stay;
//            the reference's unstable argsort leaves their order unspecified);
//   phase 2  greedy selection on LDS-resident CHUNKS of the <= 2048 highest remaining scores (see nms_kernel): a chunk
//            is sorted once by (score descending, key ascending) and walked in blocks of 64 - in-block suppression
//            bits by all threads, one wave resolves the block in registers, all threads clear the later candidates
//            the block's detections suppress (round 2 ran one workgroup-wide arg-max per detection: 2.9 us each).
//            Boxes of different classes are separated by the reference's class offset (cls * 4096) unless agnostic.
//            This is exactly the order torchvision's nms produces (descending score) truncated to max_det.
// Ties between equal scores are broken by the lower (row, class) key, which makes the result deterministic.
#include "cft_common.h"

// Scratch layout per image (structure of arrays, cap = rows * (multi_label ? nc : 1) entries, 32 bytes per entry):
//   score[cap] float | box[cap] float4 (x1,y1,x2,y2) | cls[cap] int | key[cap] int
struct NmsView {
  float* score;
  float4* box;
  int* cls;
  int* key;
};
__device__ __forceinline__ NmsView nms_view(unsigned char* base, long cap) {
  NmsView v;
  v.score = reinterpret_cast<float*>(base);
  v.box = reinterpret_cast<float4*>(base + cap * 4);
  v.cls = reinterpret_cast<int*>(base + cap * 20);
  v.key = reinterpret_cast<int*>(base + cap * 24);
  return v;
}

__device__ __forceinline__ bool better(float s, int k, float s2, int k2) { return s > s2 || (s == s2 && k < k2); }

__device__ __forceinline__ bool iou_exceeds(float x1, float y1, float x2, float y2, float bx1, float by1, float bx2, float by2, float thr) {
  const float iw = fminf(x2, bx2) - fmaxf(x1, bx1), ih = fminf(y2, by2) - fmaxf(y1, by1);
  const float inter = (iw > 0.f ? iw : 0.f) * (ih > 0.f ? ih : 0.f);
  const float iou = inter / ((x2 - x1) * (y2 - y1) + (bx2 - bx1) * (by2 - by1) - inter);
  return iou > thr;
}

constexpr int NMS_CHUNK = 2048;       // candidates resident in LDS per chunk
constexpr int NMS_MAXDET_LDS = 1024;  // kept boxes remembered in LDS (max_det above this uses the global-memory path)

// Workgroup-wide radix select over the positive scores with bit pattern < hi: returns (through s_prefix) the bit pattern
// of the want-th largest of them (want >= 1; there must be at least `want`).  Four 8-bit passes, LDS histogram.
__device__ __forceinline__ unsigned nms_radix_select(const float* __restrict__ score, int n, unsigned hi, unsigned want,
                                                     unsigned* s_hist, unsigned* s_prefix, unsigned* s_want, int tid) {
  if (tid == 0) { *s_prefix = 0u; *s_want = want; }
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) s_hist[tid] = 0u;
    __syncthreads();
    const unsigned prefix = *s_prefix;
    const unsigned hi_mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
    for (int i = tid; i < n; i += 1024) {
      const float sc = score[i];
      const unsigned u = __float_as_uint(sc);
      if (sc > 0.f && u < hi && (u & hi_mask) == prefix) atomicAdd(&s_hist[(u >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned w = *s_want, d = 255u;
      for (;; --d) {
        if (s_hist[d] >= w || d == 0u) break;
        w -= s_hist[d];
      }
      *s_want = w;
      *s_prefix = prefix | (d << shift);
    }
    __syncthreads();
  }
  return *s_prefix;
}

// One 1024-thread workgroup per image.
//   phase 1   candidate filter + compaction into the image's scratch (see the file header);
//   phase 1b  max_nms pre-truncation (radix select of the max_nms-th score);
//   phase 2   greedy selection in CHUNKS of the highest remaining scores: the next <= 2048 candidates by score are
//             copied into LDS (a radix select finds the chunk's score threshold; candidates that tie with it all
//             belong to the chunk), first thinned by the boxes kept so far, then sorted and walked in blocks of 64
//             (see below).  Greedy NMS visits candidates in descending score, so nothing outside the current chunk can
//             precede anything in it; the 25200-row arrays are read two or three times per chunk.
//             (A chunk whose score ties overflow the LDS arrays falls back to arg-max rounds over global memory.)
__global__ void __launch_bounds__(1024) nms_kernel(const float* __restrict__ pred, int rows, int no, float conf_thres,
                                                   float iou_thres, int agnostic, int multi_label, const unsigned char* __restrict__ class_allow,
                                                   int max_det, int max_nms, int cap, unsigned char* __restrict__ scratch, float* __restrict__ dets,
                                                   int* __restrict__ counts) {
  __shared__ int s_n, s_m, s_best;
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_prefix, s_want;
  __shared__ float s_score[16];
  __shared__ int s_key[16], s_idx[16];
  __shared__ int c_org[NMS_CHUNK];        // sorted position -> chunk slot
  __shared__ unsigned d_lo[64], d_hi[64]; // in-block suppression bits of the 64 candidates being resolved
  __shared__ float s_box[4];
  __shared__ float c_x1[NMS_CHUNK], c_y1[NMS_CHUNK], c_x2[NMS_CHUNK], c_y2[NMS_CHUNK], c_sc[NMS_CHUNK];
  __shared__ int c_idx[NMS_CHUNK], c_key[NMS_CHUNK];   // c_x1.. hold the class-shifted boxes the IoU runs on; c_idx -> the exact box
  __shared__ float k_x1[NMS_MAXDET_LDS], k_y1[NMS_MAXDET_LDS], k_x2[NMS_MAXDET_LDS], k_y2[NMS_MAXDET_LDS];
  __shared__ float k_sc[NMS_MAXDET_LDS];                 // winners of the LDS-chunk rounds: score and candidate index (deferred emission)
  __shared__ int k_src[NMS_MAXDET_LDS];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nc = no - 5;
  const float* P = pred + (long)b * rows * no;
  const long cap_al = ((long)cap + 3) & ~3L;     // keeps the float4 box array 16-byte aligned
  const NmsView C = nms_view(scratch + (long)b * cap_al * 32, cap_al);
  const float max_wh = 4096.f;
  if (tid == 0) s_n = 0;
  __syncthreads();
  // ---- phase 1 ----
  for (int r = tid; r < rows; r += 1024) {
    const float* p = P + (long)r * no;
    const float obj = p[4];
    if (!(obj > conf_thres)) continue;
    const float cx = p[0], cy = p[1], hw = p[2] * 0.5f, hh = p[3] * 0.5f;
    if (multi_label) {
      for (int c = 0; c < nc; ++c) {
        const float conf = p[5 + c] * obj;
        if (conf > conf_thres && (class_allow == nullptr || class_allow[c])) {
          const int i = atomicAdd(&s_n, 1);
          if (i < cap) { C.score[i] = conf; C.box[i] = make_float4(cx - hw, cy - hh, cx + hw, cy + hh); C.cls[i] = c; C.key[i] = r * nc + c; }
        }
      }
    } else {
      float best = -1.f;
      int bc = 0;
      for (int c = 0; c < nc; ++c) {
        const float conf = p[5 + c] * obj;
        if (conf > best) { best = conf; bc = c; }
      }
      if (best > conf_thres && (class_allow == nullptr || class_allow[bc])) {
        const int i = atomicAdd(&s_n, 1);
        if (i < cap) { C.score[i] = best; C.box[i] = make_float4(cx - hw, cy - hh, cx + hw, cy + hh); C.cls[i] = bc; C.key[i] = r; }
      }
    }
  }
  __syncthreads();
  const int n = s_n < cap ? s_n : cap;
  __threadfence_block();
  // ---- phase 1b: keep only the max_nms best scores (positive floats order like their bit patterns) ----
  if (max_nms > 0 && n > max_nms) {
    const unsigned kth = nms_radix_select(C.score, n, 0xffffffffu, (unsigned)max_nms, s_hist, &s_prefix, &s_want, tid);
    for (int i = tid; i < n; i += 1024)
      if (__float_as_uint(C.score[i]) < kth) C.score[i] = -1.f;
    __syncthreads();
    __threadfence_block();
  }
  // ---- phase 2 ----
  int kept = 0, kept_lds = 0;         // kept_lds: detections [0, kept_lds) were selected by LDS-chunk rounds (emitted at the end)
  unsigned hi = 0xffffffffu;          // candidates with score bits >= hi are done (selected, suppressed or earlier chunks)
  const bool kept_in_lds = max_det <= NMS_MAXDET_LDS;
  while (kept < max_det) {
    // remaining live candidates below hi
    if (tid == 0) s_m = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = tid; i < n; i += 1024) {
      const float sc = C.score[i];
      cnt += (sc > 0.f && __float_as_uint(sc) < hi) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0 && cnt) atomicAdd(&s_m, cnt);
    __syncthreads();
    const int rem = s_m;
    if (rem == 0) break;
    unsigned thr = 0u;                // chunk = live candidates with thr <= bits < hi
    int m = rem;
    if (rem > NMS_CHUNK) {
      thr = nms_radix_select(C.score, n, hi, (unsigned)(NMS_CHUNK / 2), s_hist, &s_prefix, &s_want, tid);
      if (tid == 0) s_m = 0;
      __syncthreads();
      cnt = 0;
      for (int i = tid; i < n; i += 1024) {
        const float sc = C.score[i];
        const unsigned u = __float_as_uint(sc);
        cnt += (sc > 0.f && u < hi && u >= thr) ? 1 : 0;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
      if (lane == 0 && cnt) atomicAdd(&s_m, cnt);
      __syncthreads();
      m = s_m;
    }
    const bool in_lds = m <= NMS_CHUNK && kept_in_lds;    // uniform
    if (in_lds) {
      // gather the chunk into LDS (in any order: it is sorted by (score, key) below)
      __syncthreads();
      if (tid == 0) s_m = 0;
      __syncthreads();
      for (int i = tid; i < n; i += 1024) {
        const float sc = C.score[i];
        const unsigned u = __float_as_uint(sc);
        if (sc > 0.f && u < hi && u >= thr) {
          const int j = atomicAdd(&s_m, 1);
          const float4 bx = C.box[i];
          const int cl = C.cls[i];
          const float off = agnostic ? 0.f : (float)cl * max_wh;
          c_x1[j] = bx.x + off; c_y1[j] = bx.y + off; c_x2[j] = bx.z + off; c_y2[j] = bx.w + off;
          c_sc[j] = sc; c_idx[j] = i; c_key[j] = C.key[i];
        }
      }
      __syncthreads();
      // thin by the boxes kept so far (earlier chunks)
      if (kept > 0) {
        for (int j = tid; j < m; j += 1024) {
          const float x1 = c_x1[j], y1 = c_y1[j], x2 = c_x2[j], y2 = c_y2[j];
          bool dead = false;
          for (int k = 0; k < kept && !dead; ++k) dead = iou_exceeds(x1, y1, x2, y2, k_x1[k], k_y1[k], k_x2[k], k_y2[k], iou_thres);
          if (dead) c_sc[j] = -1.f;
        }
        __syncthreads();
      }
      // Greedy selection on the LDS chunk WITHOUT one workgroup-wide arg-max per detection (round 2: 2.9 us per detection):
      //   1. the chunk is sorted once by (score descending, key ascending) - the order greedy NMS visits candidates in;
      //   2. it is walked in BLOCKS of 64 sorted candidates: all threads compute the block's 64 x 64 suppression bits, wave 0
      //      resolves the block sequentially in registers (next live candidate is kept, the candidates it suppresses
      //      are cleared), then all threads clear the LATER candidates that the block's new detections suppress.
      // Every IoU test is the same call as before (later candidate against the kept box); exact.
      int npow2 = 64;
      while (npow2 < m) npow2 <<= 1;                                     // bitonic size
      for (int j = tid; j < npow2; j += 1024) {
        c_org[j] = j;
        if (j >= m) { c_sc[j] = -1.f; c_key[j] = 0x7fffffff; }
      }
      __syncthreads();
      for (int k = 2; k <= npow2; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
          for (int i = tid; i < npow2; i += 1024) {
            const int l = i ^ jj;
            if (l > i) {
              const bool first_half = (i & k) == 0;             // "better first" in the first half of each 2k run, reversed in the second
              const float si = c_sc[i], sl = c_sc[l];
              const int ki = c_key[i], kl = c_key[l];
              if (better(sl, kl, si, ki) == first_half) {
                c_sc[i] = sl; c_sc[l] = si; c_key[i] = kl; c_key[l] = ki;
                const int t = c_org[i]; c_org[i] = c_org[l]; c_org[l] = t;
              }
            }
          }
          __syncthreads();
        }
      }
      // live candidates (score >= 0) now sit in front, best first
      if (tid == 0) s_m = 0;
      __syncthreads();
      {
        int cnt = 0;
        for (int j = tid; j < m; j += 1024) cnt += c_sc[j] >= 0.f ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (lane == 0 && cnt) atomicAdd(&s_m, cnt);
      }
      __syncthreads();
      const int L = s_m;
      for (int blk = 0; blk * 64 < L; ++blk) {
        const int p0 = blk * 64;
        // (a) in-block suppression bits: thread (row i = tid >> 4, columns 4 (tid & 15) .. + 3): bit j of row i = candidate j > i of the
        //     block is suppressed by candidate i
        {
          const int i = tid >> 4, jb = (tid & 15) * 4;
          unsigned lo = 0u, hiw = 0u;
          if (p0 + i < L) {
            const int oi = c_org[p0 + i];
            const float ax1 = c_x1[oi], ay1 = c_y1[oi], ax2 = c_x2[oi], ay2 = c_y2[oi];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int j = jb + q;
              if (j > i && p0 + j < L) {
                const int oj = c_org[p0 + j];
                if (iou_exceeds(c_x1[oj], c_y1[oj], c_x2[oj], c_y2[oj], ax1, ay1, ax2, ay2, iou_thres)) {
                  if (j < 32) lo |= 1u << j; else hiw |= 1u << (j - 32);
                }
              }
            }
          }
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) { lo |= __shfl_xor(lo, o); hiw |= __shfl_xor(hiw, o); }    // OR over the 16 lanes of the row
          if ((tid & 15) == 0) { d_lo[i] = lo; d_hi[i] = hiw; }
        }
        __syncthreads();
        // (b) wave 0 resolves the block
        if (wave == 0) {
          const bool live = p0 + lane < L && c_sc[p0 + lane] >= 0.f;
          unsigned long long alive = __ballot(live), keepbits = 0ull;
          const unsigned long long mydiag = ((unsigned long long)d_hi[lane] << 32) | d_lo[lane];
          int nk = kept;
          while (alive != 0ull && nk < max_det) {
            const int i = __builtin_amdgcn_readfirstlane(__builtin_ctzll(alive));
            keepbits |= 1ull << i;
            ++nk;
            alive &= ~(1ull << i);
            const unsigned dl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mydiag, i);
            const unsigned dh = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mydiag >> 32), i);
            alive &= ~(((unsigned long long)dh << 32) | dl);
          }
          if ((keepbits >> lane) & 1ull) {
            const int pos = kept + __builtin_popcountll(keepbits & ((1ull << lane) - 1ull));
            const int o = c_org[p0 + lane];
            k_src[pos] = c_idx[o]; k_sc[pos] = c_sc[p0 + lane];
            k_x1[pos] = c_x1[o]; k_y1[pos] = c_y1[o]; k_x2[pos] = c_x2[o]; k_y2[pos] = c_y2[o];
          }
          if (lane == 0) s_best = nk;
        }
        __syncthreads();
        const int kept_new = s_best;
        if (kept_new == max_det) { kept = kept_new; break; }
        // (c) the block's new detections suppress later candidates
        if (kept_new > kept) {
          for (int pp = p0 + 64 + tid; pp < L; pp += 1024) {
            if (c_sc[pp] < 0.f) continue;
            const int o = c_org[pp];
            const float x1 = c_x1[o], y1 = c_y1[o], x2 = c_x2[o], y2 = c_y2[o];
            bool dead = false;
            for (int k = kept; k < kept_new && !dead; ++k) dead = iou_exceeds(x1, y1, x2, y2, k_x1[k], k_y1[k], k_x2[k], k_y2[k], iou_thres);
            if (dead) c_sc[pp] = -1.f;
          }
        }
        kept = kept_new;
        __syncthreads();
      }
      kept_lds = kept;
      __syncthreads();
      if (thr == 0u) break;               // that was everything
      hi = thr;
    } else {
      // fallback (score ties overflow the LDS chunk, or max_det beyond the LDS kept list): rounds over global memory
      // on everything that is still live; the boxes kept so far suppress through the first sweep
      for (int i = tid; i < n; i += 1024) {
        const float sc = C.score[i];
        if (!(sc > 0.f) || !(__float_as_uint(sc) < hi)) { if (sc > 0.f) C.score[i] = -1.f; continue; }
        if (kept_in_lds && kept > 0) {
          const float4 bx = C.box[i];
          const float off = agnostic ? 0.f : (float)C.cls[i] * max_wh;
          bool dead = false;
          for (int k = 0; k < kept && !dead; ++k) dead = iou_exceeds(bx.x + off, bx.y + off, bx.z + off, bx.w + off, k_x1[k], k_y1[k], k_x2[k], k_y2[k], iou_thres);
          if (dead) C.score[i] = -1.f;
        }
      }
      __syncthreads();
      __threadfence_block();
      float bx1 = 0.f, by1 = 0.f, bx2 = 0.f, by2 = 0.f;
      bool have_best = false;
      for (;;) {
        float ls = -1.f;
        int lk = 0x7fffffff, li = -1;
        for (int i = tid; i < n; i += 1024) {
          const float sc = C.score[i];
          if (sc < 0.f) continue;
          const float4 bx = C.box[i];
          const float off = agnostic ? 0.f : (float)C.cls[i] * max_wh;
          if (have_best && iou_exceeds(bx.x + off, bx.y + off, bx.z + off, bx.w + off, bx1, by1, bx2, by2, iou_thres)) { C.score[i] = -1.f; continue; }
          const int ky = C.key[i];
          if (better(sc, ky, ls, lk)) { ls = sc; lk = ky; li = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float s2 = __shfl_xor(ls, o);
          const int k2 = __shfl_xor(lk, o), i2 = __shfl_xor(li, o);
          if (better(s2, k2, ls, lk)) { ls = s2; lk = k2; li = i2; }
        }
        if (lane == 0) { s_score[wave] = ls; s_key[wave] = lk; s_idx[wave] = li; }
        __syncthreads();
        if (tid == 0) {
          float bs = s_score[0]; int bk = s_key[0], bi = s_idx[0];
          for (int w = 1; w < 16; ++w)
            if (better(s_score[w], s_key[w], bs, bk)) { bs = s_score[w]; bk = s_key[w]; bi = s_idx[w]; }
          s_best = bi;
          if (bi >= 0) {
            const float4 bx = C.box[bi];
            const int cl = C.cls[bi];
            float* d = dets + ((long)b * max_det + kept) * 6;
            d[0] = bx.x; d[1] = bx.y; d[2] = bx.z; d[3] = bx.w; d[4] = bs; d[5] = (float)cl;
            const float off = agnostic ? 0.f : (float)cl * max_wh;
            s_box[0] = bx.x + off; s_box[1] = bx.y + off; s_box[2] = bx.z + off; s_box[3] = bx.w + off;
            C.score[bi] = -1.f;
          }
        }
        __syncthreads();
        if (s_best < 0) break;
        bx1 = s_box[0]; by1 = s_box[1]; bx2 = s_box[2]; by2 = s_box[3];
        have_best = true;
        ++kept;
        if (kept == max_det) break;
        __syncthreads();
        __threadfence_block();
      }
      break;                               // the fallback consumed every remaining candidate
    }
  }
  __syncthreads();
  for (int k = tid; k < kept_lds; k += 1024) {     // deferred emission of the LDS-chunk winners
    const int i = k_src[k];
    const float4 ex = C.box[i];                    // the unshifted box, bit-exact (the reference returns x[i], not boxes[i])
    float* d = dets + ((long)b * max_det + k) * 6;
    d[0] = ex.x; d[1] = ex.y; d[2] = ex.z; d[3] = ex.w; d[4] = k_sc[k]; d[5] = (float)C.cls[i];
  }
  if (tid == 0) counts[b] = kept;
}

extern "C" int cft_nms(const float* pred, int B, int rows, int no, float conf_thres, float iou_thres,
                       int agnostic, int multi_label, const unsigned char* class_allow, int max_det, int max_nms,
                       void* scratch, long scratch_bytes, float* dets, int* counts, void* stream) {
  CFT_REQUIRE(pred && scratch && dets && counts, "cft_nms: null pointer");
  CFT_REQUIRE(B > 0 && rows > 0 && no >= 6 && max_det > 0 && max_nms >= 0, "cft_nms: bad shape (needs at least one class)");
  const int nc = no - 5;
  const long cap = (long)rows * (multi_label ? nc : 1);
  const long cap_al = (cap + 3) & ~3L;
  CFT_REQUIRE(cap < (1L << 30) && scratch_bytes >= (long)B * cap_al * 32L, "cft_nms: scratch too small (need B * round_up(rows*(multi_label?nc:1), 4) * 32 bytes)");
  CFT_REQUIRE(((size_t)scratch & 15) == 0, "cft_nms: scratch must be 16-byte aligned");
  // rows of `dets` beyond counts[b] are zero (the kernel writes only what it keeps): cleared here, on the stream, so that callers need no
  // fill kernel of their own
  if (hipMemsetAsync(dets, 0, (size_t)B * max_det * 6 * sizeof(float), as_stream(stream)) != hipSuccess) {
    cft_set_error("cft_nms: hipMemsetAsync failed");
    return CFT_EINVAL;
  }
  hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(1024), 0, as_stream(stream), pred, rows, no, conf_thres, iou_thres, agnostic, multi_label,
                     class_allow, max_det, max_nms, (int)cap, (unsigned char*)scratch, dets, counts);
  return cft_check_launch("nms_kernel");
}
