// Image-side helpers shared by the Focus kernels (focus_conv.hip, stem.hip): two horizontally adjacent image samples -> two floats,
// for fp32 / fp16 / uint8 images (uint8: the /255 of the reference's callers, test.py:107-108, is the `scale` factor).
#pragma once
#include "cft_common.h"

template <typename IN>
__device__ __forceinline__ void load_pair(const IN* p, float scale, float& a, float& b);
template <>
__device__ __forceinline__ void load_pair<float>(const float* p, float scale, float& a, float& b) {
  const float2 t = *reinterpret_cast<const float2*>(p);
  a = t.x * scale;
  b = t.y * scale;
}
template <>
__device__ __forceinline__ void load_pair<f16_t>(const f16_t* p, float scale, float& a, float& b) {   // half images (`img.half()`, test.py:107)
  const f32x2_t t = __builtin_convertvector(*reinterpret_cast<const f16x2_t*>(p), f32x2_t);
  a = t[0] * scale;
  b = t[1] * scale;
}
template <>
__device__ __forceinline__ void load_pair<unsigned char>(const unsigned char* p, float scale, float& a, float& b) {
  const unsigned short t = *reinterpret_cast<const unsigned short*>(p);
  a = (float)(t & 0xffu) * scale;
  b = (float)(t >> 8) * scale;
}

