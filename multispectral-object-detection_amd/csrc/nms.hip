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
//   phase 2  greedy selection: up to max_det rounds of {workgroup arg-max of the live scores, emit it,
//            kill every live box whose IoU with it exceeds iou_thres}.  Boxes of different classes are
//            separated by the reference's class offset (cls * 4096) unless agnostic.  This is exactly the
//            order torchvision's nms produces (descending score) truncated to max_det, without sorting
//            all candidates: the work is O(max_det * n / 1024) per image.
// Ties between equal scores are broken by the lower (row, class) key, which makes the result deterministic.
#include "cft_common.h"

struct NmsCand { float x1, y1, x2, y2, score; int cls; int key; int pad; };   // 32 bytes

__device__ __forceinline__ bool better(float s, int k, float s2, int k2) { return s > s2 || (s == s2 && k < k2); }

__global__ void __launch_bounds__(1024) nms_kernel(const float* __restrict__ pred, int rows, int no, float conf_thres,
                                                   float iou_thres, int agnostic, int multi_label, const unsigned char* __restrict__ class_allow,
                                                   int max_det, int max_nms, int cap, NmsCand* __restrict__ scratch, float* __restrict__ dets,
                                                   int* __restrict__ counts) {
  __shared__ int s_n;
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_prefix, s_want;
  __shared__ float s_score[16];
  __shared__ int s_key[16], s_idx[16];
  __shared__ float s_box[4];
  __shared__ int s_best;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nc = no - 5;
  const float* P = pred + (long)b * rows * no;
  NmsCand* C = scratch + (long)b * cap;
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
          if (i < cap) C[i] = NmsCand{cx - hw, cy - hh, cx + hw, cy + hh, conf, c, r * nc + c, 0};
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
        if (i < cap) C[i] = NmsCand{cx - hw, cy - hh, cx + hw, cy + hh, best, bc, r, 0};
      }
    }
  }
  __syncthreads();
  const int n = s_n < cap ? s_n : cap;
  __threadfence_block();
  // ---- phase 1b: keep only the max_nms best scores (scores are positive floats: their bit patterns order like uints)
  if (max_nms > 0 && n > max_nms) {
    if (tid == 0) { s_prefix = 0u; s_want = (unsigned)max_nms; }
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) s_hist[tid] = 0u;
      __syncthreads();
      const unsigned prefix = s_prefix;
      const unsigned hi_mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
      for (int i = tid; i < n; i += 1024) {
        const unsigned u = __float_as_uint(C[i].score);
        if ((u & hi_mask) == prefix) atomicAdd(&s_hist[(u >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {   // walk the digits from the top: the bucket in which the want-th largest score falls
        unsigned want = s_want, d = 255u;
        for (;; --d) {
          if (s_hist[d] >= want || d == 0u) break;
          want -= s_hist[d];
        }
        s_want = want;
        s_prefix = prefix | (d << shift);
      }
      __syncthreads();
    }
    const unsigned kth = s_prefix;     // bit pattern of the max_nms-th largest score
    for (int i = tid; i < n; i += 1024)
      if (__float_as_uint(C[i].score) < kth) C[i].score = -1.f;
    __syncthreads();
    __threadfence_block();
  }
  // ---- phase 2 ----
  const float max_wh = 4096.f;
  int kept = 0;
  float bx1 = 0.f, by1 = 0.f, bx2 = 0.f, by2 = 0.f;
  bool have_best = false;
  for (int round = 0; round <= max_det; ++round) {
    // one sweep: kill what the previous winner suppresses, find this thread's best live candidate
    float ls = -1.f;
    int lk = 0x7fffffff, li = -1;
    for (int i = tid; i < n; i += 1024) {
      NmsCand c = C[i];
      if (c.score < 0.f) continue;
      if (have_best) {
        const float off = agnostic ? 0.f : (float)c.cls * max_wh;
        const float x1 = c.x1 + off, y1 = c.y1 + off, x2 = c.x2 + off, y2 = c.y2 + off;
        const float iw = fminf(x2, bx2) - fmaxf(x1, bx1), ih = fminf(y2, by2) - fmaxf(y1, by1);
        const float inter = (iw > 0.f ? iw : 0.f) * (ih > 0.f ? ih : 0.f);
        const float iou = inter / ((x2 - x1) * (y2 - y1) + (bx2 - bx1) * (by2 - by1) - inter);
        if (iou > iou_thres) { C[i].score = -1.f; continue; }
      }
      if (better(c.score, c.key, ls, lk)) { ls = c.score; lk = c.key; li = i; }
    }
    if (round == max_det) break;
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
        const NmsCand c = C[bi];
        float* d = dets + ((long)b * max_det + kept) * 6;
        d[0] = c.x1; d[1] = c.y1; d[2] = c.x2; d[3] = c.y2; d[4] = c.score; d[5] = (float)c.cls;
        const float off = agnostic ? 0.f : (float)c.cls * max_wh;
        s_box[0] = c.x1 + off; s_box[1] = c.y1 + off; s_box[2] = c.x2 + off; s_box[3] = c.y2 + off;
        C[bi].score = -1.f;   // the winner suppresses itself
      }
    }
    __syncthreads();
    if (s_best < 0) break;
    bx1 = s_box[0]; by1 = s_box[1]; bx2 = s_box[2]; by2 = s_box[3];
    have_best = true;
    ++kept;
    __syncthreads();   // s_box / s_best are rewritten next round; winner's score store is visible to all
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
  CFT_REQUIRE(cap < (1L << 30) && scratch_bytes >= (long)B * cap * (long)sizeof(NmsCand), "cft_nms: scratch too small (need B*rows*(multi_label?nc:1)*32 bytes)");
  hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(1024), 0, as_stream(stream), pred, rows, no, conf_thres, iou_thres, agnostic, multi_label,
                     class_allow, max_det, max_nms, (int)cap, (NmsCand*)scratch, dets, counts);
  return cft_check_launch("nms_kernel");
}
