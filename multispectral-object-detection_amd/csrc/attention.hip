// Multi-head self-attention core of the CFT block for gfx950: O = softmax(Q K^T / sqrt(dk)) V over a
// fixed T = 128 tokens (64 RGB + 64 IR cells; reference models/common.py:491-510).  One 256-thread
// workgroup per (image, head); wave w owns query rows [32w, 32w+32).  The whole 128x128 score tile
// lives in registers (no KV loop, no online softmax needed).
//   S = Q K^T : Q fragments straight from global (read once), K staged in LDS, MFMA 16x16.
//   softmax   : row max / sum via in-lane reduction over the 8 column tiles + 4 xor-shuffles.
//   O = P V   : P (unnormalised exp, compute dtype) to a wave-private LDS strip that aliases the K
//               tile; V is staged TRANSPOSED in <= 64-column chunks so both MFMA operands are
//               k(token)-contiguous 16-byte granules.  O is scaled by 1/rowsum in fp32 at the end.
// Head width arrives padded to dkp (multiple of 4 granules) with zero columns, so no K-tail code.
#include "cft_common.h"

template <typename T>
__global__ void __launch_bounds__(256) attention_kernel(const unsigned char* __restrict__ qkv, unsigned char* __restrict__ out,
                                                        int heads, int dkp, float scale) {
  constexpr int GE = Elem<T>::GE;
  constexpr int ES = (int)sizeof(T);
  constexpr int T_TOK = 128;
  constexpr int PS_B = T_TOK * ES + 16;   // P / V^T row stride in bytes (odd number of 16-B slots)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int KS_B = dkp * ES + 16;          // K row stride in bytes
  const int region0 = T_TOK * (KS_B > PS_B ? KS_B : PS_B);
  unsigned char* sK = smem;
  unsigned char* sVT = smem + region0;     // [<=64][PS_B]

  const int b = blockIdx.x / heads, head = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int G = dkp / GE;                  // granules per head row
  const long ldq_b = (long)3 * heads * dkp * ES;
  const long ldo_b = (long)heads * dkp * ES;
  const unsigned char* qbase = qkv + (long)b * T_TOK * ldq_b + (long)head * dkp * ES;
  const unsigned char* kbase = qbase + (long)heads * dkp * ES;
  const unsigned char* vbase = kbase + (long)heads * dkp * ES;

  // ---- stage K ----
  for (int i = tid; i < T_TOK * G; i += 256) {
    const int t = i / G, kg = i - t * G;
    *reinterpret_cast<gran_t*>(sK + t * KS_B + kg * 16) = *reinterpret_cast<const gran_t*>(kbase + t * ldq_b + kg * 16);
  }
  __syncthreads();

  // ---- S = Q K^T ----
  f32x4_t s[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int ksteps = G >> 2;
  for (int ks = 0; ks < ksteps; ++ks) {
    const int kg = ks * 4 + lgrp;
    gran_t qf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
      qf[i] = *reinterpret_cast<const gran_t*>(qbase + (long)(wave * 32 + i * 16 + lrow) * ldq_b + kg * 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const gran_t kf = *reinterpret_cast<const gran_t*>(sK + (j * 16 + lrow) * KS_B + kg * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i) s[i][j] = mma_granule<T>(qf[i], kf, s[i][j]);
    }
  }

  // ---- softmax over the 128 columns of each row (rows: i*16 + lgrp*4 + e) ----
  float inv_sum[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[i][j][e] *= scale; mx = fmaxf(mx, s[i][j][e]); }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float pv = __expf(s[i][j][e] - mx); s[i][j][e] = pv; sum += pv; }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o);
      inv_sum[i][e] = 1.0f / sum;
    }

  __syncthreads();   // every wave is done with K before P overwrites it
  unsigned char* sP = smem + wave * 32 * PS_B;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned char* dst = sP + (i * 16 + lgrp * 4 + e) * PS_B + (j * 16 + lrow) * ES;
        if (ES == 2) *reinterpret_cast<uint16_t*>(dst) = f32_to_bf16(s[i][j][e]);
        else *reinterpret_cast<float*>(dst) = s[i][j][e];
      }

  // ---- O = P V, in chunks of <= 64 head columns ----
  for (int c0 = 0; c0 < dkp; c0 += 64) {
    const int cw = (dkp - c0) < 64 ? (dkp - c0) : 64;
    const int cg = cw / GE;   // granules per token in this chunk
    __syncthreads();          // previous chunk's V^T fully consumed (and P visible on first pass)
    for (int i = tid; i < T_TOK * cg; i += 256) {
      const int t = i / cg, kg = i - t * cg;
      const gran_t g = *reinterpret_cast<const gran_t*>(vbase + t * ldq_b + (long)(c0 + kg * GE) * ES);
      if (ES == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          *reinterpret_cast<uint16_t*>(sVT + (kg * GE + 2 * e) * PS_B + t * 2) = (uint16_t)(g[e] & 0xffffu);
          *reinterpret_cast<uint16_t*>(sVT + (kg * GE + 2 * e + 1) * PS_B + t * 2) = (uint16_t)(g[e] >> 16);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<uint32_t*>(sVT + (kg * GE + e) * PS_B + t * 4) = g[e];
      }
    }
    __syncthreads();
    const int ntc = cw >> 4;
    f32x4_t o[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) o[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int PSTEPS = (T_TOK / GE) / 4;
#pragma unroll
    for (int ks = 0; ks < PSTEPS; ++ks) {
      const int kg = ks * 4 + lgrp;
      gran_t pf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) pf[i] = *reinterpret_cast<const gran_t*>(sP + (i * 16 + lrow) * PS_B + kg * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < ntc) {
          const gran_t vf = *reinterpret_cast<const gran_t*>(sVT + (j * 16 + lrow) * PS_B + kg * 16);
#pragma unroll
          for (int i = 0; i < 2; ++i) o[i][j] = mma_granule<T>(pf[i], vf, o[i][j]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < ntc) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = wave * 32 + i * 16 + lgrp * 4 + e;
            const float v = o[i][j][e] * inv_sum[i][e];
            unsigned char* dst = out + ((long)b * T_TOK + row) * ldo_b + ((long)head * dkp + c0 + j * 16 + lrow) * ES;
            if (ES == 2) *reinterpret_cast<uint16_t*>(dst) = f32_to_bf16(v);
            else *reinterpret_cast<float*>(dst) = v;
          }
        }
      }
  }
}

extern "C" int cft_attention(const void* qkv, void* out, int B, int heads, int dk, int dkp,
                             int dtype, void* stream) {
  CFT_REQUIRE(qkv && out, "cft_attention: null pointer");
  CFT_REQUIRE(dtype == CFT_BF16 || dtype == CFT_F32, "cft_attention: bad dtype");
  const int es = dtype == CFT_BF16 ? 2 : 4;
  const int kstep = dtype == CFT_BF16 ? 32 : 16;
  CFT_REQUIRE(B > 0 && heads > 0 && dk > 0 && dkp >= dk && dkp % kstep == 0 && dkp <= 256, "cft_attention: dkp must be a multiple of 32 (bf16) / 16 (f32), >= dk, <= 256");
  const int ps = 128 * es + 16, ks = dkp * es + 16;
  const size_t smem = (size_t)128 * (ks > ps ? ks : ps) + (size_t)64 * ps;
  CFT_REQUIRE(smem <= 160 * 1024, "cft_attention: head too wide for LDS");
  const float scale = 1.0f / sqrtf((float)dk);
  if (dtype == CFT_BF16) {
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = true; }
    hipLaunchKernelGGL(attention_kernel<uint16_t>, dim3(B * heads), dim3(256), smem, as_stream(stream), (const unsigned char*)qkv, (unsigned char*)out, heads, dkp, scale);
  } else {
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = true; }
    hipLaunchKernelGGL(attention_kernel<float>, dim3(B * heads), dim3(256), smem, as_stream(stream), (const unsigned char*)qkv, (unsigned char*)out, heads, dkp, scale);
  }
  return cft_check_launch("attention_kernel");
}
