// Shared device/host helpers for the gfx950 kernels of libcft_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cft_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
// Element type tags of the kernel templates: uint16_t = bfloat16 bits, f16_t = IEEE half, float.
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

// One 16-byte granule: 8 bf16 or 4 f32.  All NHWC tensors are addressed in granules.
typedef __attribute__((ext_vector_type(4))) uint32_t gran_t;

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {   // round-to-nearest-even, NaN kept quiet
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// two floats -> packed bf16 pair with the hardware round-to-nearest-even convert (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

// two floats -> packed half pair, round-to-nearest-even: ONE v_cvt_pk_f16_f32 on gfx950 (checked in the ISA, profiles/r04_bf16_sites.md)
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, f16x2_t));
}

// Element traits: T = uint16_t (bf16 bits), f16_t (IEEE half) or float.
template <typename T> struct Elem;
template <> struct Elem<uint16_t> {
  static constexpr int GE = 8;   // elements per 16-byte granule
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
  static __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) { lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ void unpack(const gran_t& g, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack2(g[i], f[2 * i], f[2 * i + 1]);
  }
  static __device__ __forceinline__ gran_t pack(const float* f) {
    gran_t g;
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = pack2(f[2 * i], f[2 * i + 1]);
    return g;
  }
};
template <> struct Elem<f16_t> {
  static constexpr int GE = 8;
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
  static __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
    const f32x2_t v = __builtin_convertvector(__builtin_bit_cast(f16x2_t, w), f32x2_t);
    lo = v[0]; hi = v[1];
  }
  static __device__ __forceinline__ void unpack(const gran_t& g, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack2(g[i], f[2 * i], f[2 * i + 1]);
  }
  static __device__ __forceinline__ gran_t pack(const float* f) {
    gran_t g;
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = pack2(f[2 * i], f[2 * i + 1]);
    return g;
  }
};
// 16-bit storage type that goes with compute type T (T itself when it is 16 bits wide; bf16 for the
// never-launched 16-bit-output instantiations of the fp32 kernels).
template <typename T> struct Half16 { typedef T type; };
template <> struct Half16<float> { typedef uint16_t type; };
template <> struct Elem<float> {
  static constexpr int GE = 4;
  static __device__ __forceinline__ void unpack(const gran_t& g, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(g[i]);
  }
  static __device__ __forceinline__ gran_t pack(const float* f) {
    gran_t g;
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = __float_as_uint(f[i]);
    return g;
  }
};

// One MFMA K-step on a pair of 16-byte granules (lane l: row l&15, k-group l>>4).
// bf16: v_mfma_f32_16x16x32_bf16; f32: 4 x v_mfma_f32_16x16x4_f32 (exact fp32 products).
// A and B use the same (k-group, element) -> k assignment, so any internal k order cancels.
template <typename T>
__device__ __forceinline__ f32x4_t mma_granule(const gran_t& a, const gran_t& b, f32x4_t c);

template <>
__device__ __forceinline__ f32x4_t mma_granule<uint16_t>(const gran_t& a, const gran_t& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t mma_granule<f16_t>(const gran_t& a, const gran_t& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t mma_granule<float>(const gran_t& a, const gran_t& b, f32x4_t c) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
  return c;
}

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain the wave's outstanding
// global loads/stores (s_waitcnt vmcnt(0)), so prefetched operands stay in flight and stores retire under the next
// phase.  Use it where the data handed over between waves lives in LDS.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Fused epilogue activations (one instantiation per activation, selected uniformly per launch).
template <int ACT>
__device__ __forceinline__ float apply_act(float v) {
  if constexpr (ACT == CFT_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));   // x*sigmoid(x): v_exp + v_rcp
  if constexpr (ACT == CFT_ACT_GELU) {
    // exact-erf GELU (reference nn.GELU(), models/common.py:498) = max(v, 0) - 0.5 |v| erfc(|v| / sqrt 2), erfc by Abramowitz &
    // Stegun 7.1.26 (|error| <= 1.5e-7, no cancellation in the negative tail): 14 VALU instructions, two of them
    // transcendental, against ~35 for erff() - the fc1 epilogue of the GPT blocks is VALU-bound (profiles/r02_bottleneck128.md section 3)
    const float z = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float e = __builtin_amdgcn_exp2f(z * (z * -1.4426950408889634f));
    float q = fmaf(1.061405429f, t, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    return fmaf(-0.5f * fabsf(v), q * t * e, fmaxf(v, 0.0f));
  }
  return v;
}

// Counter-based uniform 32-bit hash of (seed, element index) for dropout masks (splitmix64 finaliser).
__device__ __forceinline__ uint32_t cft_hash32(unsigned long long seed, unsigned long long i) {
  unsigned long long z = seed + (i + 1ull) * 0x9E3779B97F4A7C15ull;     // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

// ---- host side -----------------------------------------------------------------------------
void cft_set_error(const char* msg);
int cft_check_launch(const char* what);
static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

// Dynamic-LDS opt-in of one kernel, once per (kernel, device): the attribute is per device, so a process that
// drives several GPUs must set it on each.
template <auto Kernel>
static inline void cft_allow_lds(int bytes) {
  static bool done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return;
  }
  if (!done[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done[dev] = true;
  }
}

static inline bool cft_is_dtype(int d) { return d == CFT_BF16 || d == CFT_F16 || d == CFT_F32; }
static inline int cft_elem_size(int d) { return d == CFT_F32 ? 4 : 2; }
static inline int cft_granule(int d) { return d == CFT_F32 ? 4 : 8; }

// Run `...` with `T` bound to the element tag of dtype code `dt_`.
#define CFT_DISPATCH_DTYPE(dt_, T, ...)                              \
  do {                                                               \
    if ((dt_) == CFT_BF16) { using T = uint16_t; __VA_ARGS__; }      \
    else if ((dt_) == CFT_F16) { using T = f16_t; __VA_ARGS__; }     \
    else { using T = float; __VA_ARGS__; }                           \
  } while (0)

#define CFT_REQUIRE(cond, msg)            \
  do {                                    \
    if (!(cond)) {                        \
      cft_set_error(msg);                 \
      return CFT_EINVAL;                  \
    }                                     \
  } while (0)
