// Multi-head self-attention core of the CFT block for gfx950: O = softmax(Q K^T / sqrt(dk)) V over a
// fixed T = 128 tokens (64 RGB + 64 IR cells; reference models/common.py:491-510).  One 256-thread
// workgroup per (image, head); wave w owns queries [32w, 32w+32).  The whole 128x128 score tile lives in
// registers (no KV loop, no online softmax), and it is computed TRANSPOSED:
//   S^T = K Q^T : K fragments (row operand) from LDS, Q fragments (column operand) straight from global; a lane
//                 then holds, for ONE query (its column), 4 consecutive keys of each of the 8 key tiles;
//   softmax     : per query = in-lane over 32 values + two xor-shuffles across the four lane groups;
//   O^T = V^T P^T: the exponentials never leave the registers - the 8 values a lane holds for a 32-key chunk
//                 (two key tiles) ARE its column-operand fragment, because the MFMA reduction does not care
//                 which k index a (lane-group, element) slot stands for as long as both operands agree; the
//                 V^T fragment reads the matching keys (two 8-byte runs) from the transposed LDS copy of V;
//   store       : a lane ends up with 4 consecutive head columns of one query -> one 8-byte (bf16) store.
// Head width arrives padded to dkp (multiple of 4 granules) with zero columns, so no K-tail code.
#include "cft_common.h"

template <typename T>
__global__ void __launch_bounds__(256) attention_kernel(const unsigned char* __restrict__ qkv, unsigned char* __restrict__ out,
                                                        int heads, int dkp, float scale, uint32_t drop_thresh, float inv_keep,
                                                        unsigned long long seed) {
  constexpr int GE = Elem<T>::GE;
  constexpr int ES = (int)sizeof(T);
  constexpr int T_TOK = 128;
  constexpr int KT = GE / 4;               // 16-key score tiles per MFMA k chunk (bf16: 2 -> 32 keys, f32: 1 -> 16 keys)
  constexpr int NCH = 8 / KT;              // k chunks over the 128 keys
  constexpr int PS_B = T_TOK * ES + 16;    // V^T row stride in bytes (odd number of 16-B slots)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int KS_B = dkp * ES + 16;          // K row stride in bytes
  unsigned char* sK = smem;
  unsigned char* sVT = smem + T_TOK * KS_B;   // [<=64 head columns][PS_B]

  const int b = blockIdx.x / heads, head = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int G = dkp / GE;                  // granules per head row
  const long ldq_b = (long)3 * heads * dkp * ES;
  const long ldo_b = (long)heads * dkp * ES;
  const unsigned char* qbase = qkv + (long)b * T_TOK * ldq_b + (long)head * dkp * ES;
  const unsigned char* kbase = qbase + (long)heads * dkp * ES;
  const unsigned char* vbase = kbase + (long)heads * dkp * ES;

// V^T of head columns [c0_, c0_ + cw_) -> LDS (2-/4-byte scattered writes; V is read once, coalesced)
#define ATT_STAGE_VT(c0_, cw_)                                                                             \
  {                                                                                                        \
    const int cg_ = (cw_) / GE;                                                                            \
    for (int i = tid; i < T_TOK * cg_; i += 256) {                                                         \
      const int t = i / cg_, kg = i - t * cg_;                                                             \
      const gran_t g = *reinterpret_cast<const gran_t*>(vbase + t * ldq_b + (long)((c0_) + kg * GE) * ES); \
      if (ES == 2) {                                                                                       \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                    \
          *reinterpret_cast<uint16_t*>(sVT + (kg * GE + 2 * e) * PS_B + t * 2) = (uint16_t)(g[e] & 0xffffu); \
          *reinterpret_cast<uint16_t*>(sVT + (kg * GE + 2 * e + 1) * PS_B + t * 2) = (uint16_t)(g[e] >> 16); \
        }                                                                                                  \
      } else {                                                                                             \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                      \
          *reinterpret_cast<uint32_t*>(sVT + (kg * GE + e) * PS_B + t * 4) = g[e];                         \
      }                                                                                                    \
    }                                                                                                      \
  }

  // ---- stage K and the first V^T chunk ----
  for (int i = tid; i < T_TOK * G; i += 256) {
    const int t = i / G, kg = i - t * G;
    *reinterpret_cast<gran_t*>(sK + t * KS_B + kg * 16) = *reinterpret_cast<const gran_t*>(kbase + t * ldq_b + kg * 16);
  }
  {
    const int cw0 = dkp < 64 ? dkp : 64;
    ATT_STAGE_VT(0, cw0)
  }
  __syncthreads();

  // ---- S^T = K Q^T: s[i][j][e] = score(query wave*32 + i*16 + lrow, key j*16 + lgrp*4 + e) ----
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
      for (int i = 0; i < 2; ++i) s[i][j] = mma_granule<T>(kf, qf[i], s[i][j]);
    }
  }

  // ---- softmax over the 128 keys of this lane's query: 32 values in the lane, the rest in lanes +-16, +-32 ----
  float inv_sum[2];
  gran_t pf[2][NCH];   // unnormalised exponentials as MFMA column-operand fragments
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[i][j][e] *= scale; mx = fmaxf(mx, s[i][j][e]); }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float pv = __expf(s[i][j][e] - mx); s[i][j][e] = pv; sum += pv; }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    inv_sum[i] = 1.0f / sum;
    if (drop_thresh != 0u) {   // training: attn_drop on the normalised probabilities (reference models/common.py:507); the
                               // normaliser is linear, so masking + 1/(1-p) scaling the unnormalised exponentials is the same
      const unsigned long long qrow = ((unsigned long long)blockIdx.x * T_TOK + (wave * 32 + i * 16 + lrow)) * T_TOK;
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          s[i][j][e] = cft_hash32(seed, qrow + (unsigned)(j * 16 + lgrp * 4 + e)) >= drop_thresh ? s[i][j][e] * inv_keep : 0.0f;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if constexpr (ES == 2) {
        typedef Elem<typename Half16<T>::type> E16;
        pf[i][c] = gran_t{E16::pack2(s[i][2 * c][0], s[i][2 * c][1]), E16::pack2(s[i][2 * c][2], s[i][2 * c][3]),
                          E16::pack2(s[i][2 * c + 1][0], s[i][2 * c + 1][1]), E16::pack2(s[i][2 * c + 1][2], s[i][2 * c + 1][3])};
      } else {
        pf[i][c] = gran_t{__float_as_uint(s[i][c][0]), __float_as_uint(s[i][c][1]), __float_as_uint(s[i][c][2]), __float_as_uint(s[i][c][3])};
      }
    }
  }

  // ---- O^T = V^T P^T, in chunks of <= 64 head columns ----
  for (int c0 = 0; c0 < dkp; c0 += 64) {
    const int cw = (dkp - c0) < 64 ? (dkp - c0) : 64;
    if (c0 > 0) {
      __syncthreads();          // previous chunk's V^T fully consumed
      ATT_STAGE_VT(c0, cw)
      __syncthreads();
    }
    const int ntc = cw >> 4;
    f32x4_t o[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) o[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < ntc) {
          const unsigned char* vrow = sVT + (j * 16 + lrow) * PS_B;
          gran_t vf;
          if constexpr (ES == 2) {   // keys (2c)*16 + lgrp*4 .. +3 and (2c+1)*16 + lgrp*4 .. +3: the slots of pf[.][c]
            const uint2 lo = *reinterpret_cast<const uint2*>(vrow + ((2 * c) * 16 + lgrp * 4) * 2);
            const uint2 hi = *reinterpret_cast<const uint2*>(vrow + ((2 * c + 1) * 16 + lgrp * 4) * 2);
            vf = gran_t{lo.x, lo.y, hi.x, hi.y};
          } else {
            vf = *reinterpret_cast<const gran_t*>(vrow + (c * 16 + lgrp * 4) * 4);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) o[i][j] = mma_granule<T>(vf, pf[i][c], o[i][j]);
        }
      }
    }
    // o[i][j][e] = O(query wave*32 + i*16 + lrow, head column c0 + j*16 + lgrp*4 + e)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < ntc) {
          const int row = wave * 32 + i * 16 + lrow;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = o[i][j][e] * inv_sum[i];
          unsigned char* dst = out + ((long)b * T_TOK + row) * ldo_b + ((long)head * dkp + c0 + j * 16 + lgrp * 4) * ES;
          if constexpr (ES == 2) {
            typedef Elem<typename Half16<T>::type> E16;
            *reinterpret_cast<uint2*>(dst) = uint2{E16::pack2(v[0], v[1]), E16::pack2(v[2], v[3])};
          }
          else *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{v[0], v[1], v[2], v[3]};
        }
      }
  }
#undef ATT_STAGE_VT
}

extern "C" int cft_attention(const void* qkv, void* out, int B, int heads, int dk, int dkp,
                             int dtype, float attn_pdrop, unsigned long long seed, void* stream) {
  CFT_REQUIRE(qkv && out, "cft_attention: null pointer");
  CFT_REQUIRE(attn_pdrop >= 0.0f && attn_pdrop < 1.0f, "cft_attention: attn_pdrop must be in [0, 1) (0 = inference)");
  CFT_REQUIRE(cft_is_dtype(dtype), "cft_attention: bad dtype");
  const int es = cft_elem_size(dtype);
  const int kstep = es == 2 ? 32 : 16;
  CFT_REQUIRE(B > 0 && heads > 0 && dk > 0 && dkp >= dk && dkp % kstep == 0 && dkp <= 256, "cft_attention: dkp must be a multiple of 32 (bf16, f16) / 16 (f32), >= dk, <= 256");
  const int ps = 128 * es + 16, ks = dkp * es + 16;
  const size_t smem = (size_t)128 * ks + (size_t)64 * ps;
  CFT_REQUIRE(smem <= 160 * 1024, "cft_attention: head too wide for LDS");
  const float scale = 1.0f / sqrtf((float)dk);
  CFT_DISPATCH_DTYPE(dtype, T, {
    cft_allow_lds<&attention_kernel<T>>(160 * 1024);
    hipLaunchKernelGGL(attention_kernel<T>, dim3(B * heads), dim3(256), smem, as_stream(stream), (const unsigned char*)qkv, (unsigned char*)out, heads, dkp, scale,
                       (uint32_t)((double)attn_pdrop * 4294967296.0), 1.0f / (1.0f - attn_pdrop), seed);
  });
  return cft_check_launch("attention_kernel");
}
