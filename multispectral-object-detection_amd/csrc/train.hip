// Training-mode forward pieces: BatchNorm with batch statistics + running-stat update fused with
// the activation / residual / channel-slice store, and dropout (reference models/common.py:45-47 `act(bn(conv(x)))` with
// `bn.training`, :456-457,507,511,537 and :611 `nn.Dropout`).  Forward only - no autograd graph is built.
//
// BatchNorm2d in training mode (torch semantics): per channel c over the M = B*H*W rows of the conv output
//   mean = sum(x)/M,  var = sum((x-mean)^2)/M (biased),  y = (x-mean)/sqrt(var+eps)*gamma + beta
//   running_mean = (1-m)*running_mean + m*mean,  running_var = (1-m)*running_var + m*var*M/(M-1)
// Three launches, deterministic (no atomics): per-slab partial sums (fp32 inside a slab of <= 4096 rows, Welford-free
// because the slab means are combined in double), a finalize kernel (double), and the apply kernel.
#include "cft_common.h"

__global__ void __launch_bounds__(1024) bn_partial_kernel(const float* __restrict__ x, long ldx, long xoff, long M, int G, int rpp,
                                                          long rows_per_blk, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4* red = reinterpret_cast<float4*>(smem);           // [rpp][G] sums, then [rpp][G] sums of squares
  const int g = threadIdx.x % G, r = threadIdx.x / G;
  const long m0 = (long)blockIdx.x * rows_per_blk;
  const long m1 = (m0 + rows_per_blk < M) ? m0 + rows_per_blk : M;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
  // sums of (x - pivot): the pivot (row 0 of the tensor, the same for every slab) removes the cancellation of
  // E[x^2] - mean^2 for channels with |mean| >> std
  const float4 pv = *reinterpret_cast<const float4*>(x + xoff + g * 4L);
  for (long m = m0 + r; m < m1; m += rpp) {
    float4 v = *reinterpret_cast<const float4*>(x + m * ldx + xoff + g * 4L);
    v.x -= pv.x; v.y -= pv.y; v.z -= pv.z; v.w -= pv.w;
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
  }
  red[r * G + g] = s;
  red[(rpp + r) * G + g] = q;
  __syncthreads();
  if (r == 0) {
    for (int k = 1; k < rpp; ++k) {
      const float4 a = red[k * G + g], b = red[(rpp + k) * G + g];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
    }
    float4* o = reinterpret_cast<float4*>(part + (long)blockIdx.x * 2 * G * 4);
    o[g] = s;
    o[G + g] = q;
  }
}

__global__ void __launch_bounds__(256) bn_finalize_kernel(const float* __restrict__ x, long xoff, const float* __restrict__ part, int nblk, int C, int Cp, long M,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float momentum, float eps, float* __restrict__ scale_shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cp) return;
  if (c >= C) { scale_shift[c] = 0.f; scale_shift[Cp + c] = 0.f; return; }   // padding channels of the GEMM output
  double s = 0.0, q = 0.0;
  for (int k = 0; k < nblk; ++k) {
    s += (double)part[(long)k * 2 * Cp + c];
    q += (double)part[(long)k * 2 * Cp + Cp + c];
  }
  const double dm = s / (double)M;                  // mean of (x - pivot)
  double var = q / (double)M - dm * dm;
  var = var < 0.0 ? 0.0 : var;
  const double mean = (double)x[xoff + c] + dm;
  const float sc = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
  scale_shift[c] = sc;
  scale_shift[Cp + c] = beta[c] - (float)mean * sc;
  if (running_mean != nullptr) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

template <typename TO, int ACT>
__global__ void __launch_bounds__(256) bn_apply_kernel(const float* __restrict__ x, long ldx, long xoff, const float* __restrict__ scale_shift, int Cp,
                                                       const unsigned char* res, long ldr_b, long roff_b, int res_f32,
                                                       unsigned char* y, long ldy_b, long yoff_b, long M, int gpp) {
  constexpr int GE = Elem<TO>::GE;
  const long total = M * gpp;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / gpp;
    const int c0 = (int)(idx - m * gpp) * GE;
    float v[GE];
#pragma unroll
    for (int e = 0; e < GE; ++e) v[e] = apply_act<ACT>(x[m * ldx + xoff + c0 + e] * scale_shift[c0 + e] + scale_shift[Cp + c0 + e]);
    if (res != nullptr) {
      if (res_f32) {
#pragma unroll
        for (int e = 0; e < GE; ++e) v[e] += reinterpret_cast<const float*>(res + m * ldr_b + roff_b)[c0 + e];
      } else {
        float rf[GE];
        Elem<TO>::unpack(*reinterpret_cast<const gran_t*>(res + m * ldr_b + roff_b + (long)c0 * sizeof(TO)), rf);
#pragma unroll
        for (int e = 0; e < GE; ++e) v[e] += rf[e];
      }
    }
    *reinterpret_cast<gran_t*>(y + m * ldy_b + yoff_b + (long)c0 * sizeof(TO)) = Elem<TO>::pack(v);
  }
}

extern "C" long cft_batchnorm_train_workspace(long M, int C) {
  const int Cp = (C + 7) / 8 * 8;
  const long nblk = (M + 4095) / 4096;
  return (nblk * 2 * Cp + 2 * Cp) * (long)sizeof(float);
}

extern "C" int cft_batchnorm_train(const float* x, int ldx, int xoff, long M, int C,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   float momentum, float eps, const void* res, int ldr, int roff, int res_dtype,
                                   void* y, int ldy, int yoff, int act, int out_dtype,
                                   void* workspace, long workspace_bytes, void* stream) {
  CFT_REQUIRE(x && gamma && beta && y && workspace, "cft_batchnorm_train: null pointer");
  CFT_REQUIRE(M > 0 && C > 0, "cft_batchnorm_train: non-positive size");
  CFT_REQUIRE(cft_is_dtype(out_dtype), "cft_batchnorm_train: bad out dtype");
  CFT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "cft_batchnorm_train: running_mean and running_var go together");
  CFT_REQUIRE(act == CFT_ACT_NONE || act == CFT_ACT_SILU, "cft_batchnorm_train: activation must be SiLU or none");
  const int Cp = (C + 7) / 8 * 8;       // the GEMM output is padded to 8 channels; padding columns hold zeros
  const int ge = cft_granule(out_dtype), es = cft_elem_size(out_dtype);
  CFT_REQUIRE(ldx % 4 == 0 && xoff % 4 == 0 && ldx >= xoff + Cp, "cft_batchnorm_train: input ld/offset must be multiples of 4 and cover pad8(C)");
  CFT_REQUIRE(ldy % ge == 0 && yoff % ge == 0 && ldy >= yoff + Cp, "cft_batchnorm_train: output ld/offset not granule aligned or too small");
  CFT_REQUIRE(res == nullptr || res_dtype == out_dtype || res_dtype == CFT_F32, "cft_batchnorm_train: residual dtype must be the output dtype or CFT_F32");
  CFT_REQUIRE(res == nullptr || (ldr % ge == 0 && roff % ge == 0), "cft_batchnorm_train: residual ld/offset not granule aligned");
  CFT_REQUIRE(workspace_bytes >= cft_batchnorm_train_workspace(M, C), "cft_batchnorm_train: workspace too small (cft_batchnorm_train_workspace)");
  CFT_REQUIRE(Cp <= 4096, "cft_batchnorm_train: at most 4096 channels");
  hipStream_t s = as_stream(stream);
  float* part = reinterpret_cast<float*>(workspace);
  const long nblk = (M + 4095) / 4096;
  float* scale_shift = part + nblk * 2 * Cp;
  const int G = Cp / 4;
  int rpp = 1024 / G;
  rpp = rpp < 1 ? 1 : (rpp > 16 ? 16 : rpp);
  CFT_REQUIRE(G <= 1024, "cft_batchnorm_train: too many channels for one workgroup");
  hipLaunchKernelGGL(bn_partial_kernel, dim3((unsigned)nblk), dim3(G * rpp), (size_t)2 * rpp * G * 16, s, x, (long)ldx, (long)xoff, M, G, rpp, 4096L, part);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((Cp + 255) / 256), dim3(256), 0, s, x, (long)xoff, part, (int)nblk, C, Cp, M, gamma, beta, running_mean, running_var,
                     momentum, eps, scale_shift);
  const int gpp = Cp / ge;
  long g = (M * gpp + 255) / 256;
  g = g > 4096 ? 4096 : g;
  const int rf32 = res_dtype == CFT_F32;
#define CFT_BN_APPLY(TO_, ACT_)                                                                                       \
  hipLaunchKernelGGL((bn_apply_kernel<TO_, ACT_>), dim3((unsigned)g), dim3(256), 0, s, x, (long)ldx, (long)xoff, scale_shift, Cp, \
                     (const unsigned char*)res, (long)ldr * (rf32 ? 4 : es), (long)roff * (rf32 ? 4 : es), rf32,            \
                     (unsigned char*)y, (long)ldy * es, (long)yoff * es, M, gpp)
  CFT_DISPATCH_DTYPE(out_dtype, TO, {
    if (act == CFT_ACT_SILU) CFT_BN_APPLY(TO, CFT_ACT_SILU);
    else CFT_BN_APPLY(TO, CFT_ACT_NONE);
  });
#undef CFT_BN_APPLY
  return cft_check_launch("bn_train kernels");
}

// ------------------------------------------------------------------------------- dropout
// Counter-based generator: element i of a call keeps its value iff hash(seed, i) >= p * 2^32 and is scaled by 1/(1-p)
// (torch.nn.Dropout semantics; torch's own Philox stream cannot be reproduced, parity is statistical - tests/test_train.py).
template <typename T>
__global__ void __launch_bounds__(256) dropout_kernel(unsigned char* x, long n_gran, uint32_t thresh, float inv_keep, unsigned long long seed) {
  constexpr int GE = Elem<T>::GE;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n_gran; idx += (long)gridDim.x * blockDim.x) {
    gran_t* p = reinterpret_cast<gran_t*>(x + idx * 16);
    float v[GE];
    Elem<T>::unpack(*p, v);
#pragma unroll
    for (int e = 0; e < GE; ++e) v[e] = cft_hash32(seed, (unsigned long long)(idx * GE + e)) >= thresh ? v[e] * inv_keep : 0.0f;
    *p = Elem<T>::pack(v);
  }
}

extern "C" int cft_dropout(void* x, long n, float p, unsigned long long seed, int dtype, void* stream) {
  CFT_REQUIRE(x != nullptr, "cft_dropout: null pointer");
  CFT_REQUIRE(cft_is_dtype(dtype), "cft_dropout: bad dtype");
  CFT_REQUIRE(p >= 0.0f && p < 1.0f, "cft_dropout: p must be in [0, 1)");
  const int ge = cft_granule(dtype);
  CFT_REQUIRE(n > 0 && n % ge == 0 && ((size_t)x & 15) == 0, "cft_dropout: contiguous, 16-byte aligned tensors with a granule multiple of elements");
  if (p == 0.0f) return CFT_OK;
  const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
  const long n_gran = n / ge;
  long g = (n_gran + 255) / 256;
  g = g > 4096 ? 4096 : g;
  CFT_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(dropout_kernel<T>, dim3((unsigned)g), dim3(256), 0, as_stream(stream), (unsigned char*)x, n_gran, thresh,
                                                   1.0f / (1.0f - p), seed));
  return cft_check_launch("dropout_kernel");
}
