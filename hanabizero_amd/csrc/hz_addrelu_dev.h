// hz_addrelu_dev.h -- y = relu(y + res) on packed 16-bit pairs / fp32 words, the arithmetic of hz_add_relu (include/hz_netglue.h:
// the residual-add + ReLU between the nets' Linear layers, config/hanabi_control/model.py:54-56, 81-82, 123-124).  Shared by the
// stand-alone kernel (hz_netglue.hip) and by the fused MLP's input staging (hz_mlp_dev.h), which applies it to the rows it
// gathers when a residual source is given -- the same expression per element, hence the same bits.
#pragma once
#include "hz_common.h"
#include "hz_tree.h"

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {  // round to nearest even; inputs here are finite sums
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

template <int DTYPE>
__device__ __forceinline__ uint32_t hz_add_relu_word(uint32_t a, uint32_t b) {
  if (DTYPE == HZ_F32) {
    const float v = __uint_as_float(a) + __uint_as_float(b);
    return __float_as_uint(v > 0.0f ? v : (v != v ? v : 0.0f));
  } else if (DTYPE == HZ_BF16) {
    const float lo = bf16_to_f32((uint16_t)(a & 0xffffu)) + bf16_to_f32((uint16_t)(b & 0xffffu));
    const float hi = bf16_to_f32((uint16_t)(a >> 16)) + bf16_to_f32((uint16_t)(b >> 16));
    const uint16_t l = f32_to_bf16(lo > 0.0f ? lo : (lo != lo ? lo : 0.0f));
    const uint16_t h = f32_to_bf16(hi > 0.0f ? hi : (hi != hi ? hi : 0.0f));
    return (uint32_t)l | ((uint32_t)h << 16);
  } else {
    const uint16_t al = (uint16_t)(a & 0xffffu), ah = (uint16_t)(a >> 16);
    const uint16_t bl = (uint16_t)(b & 0xffffu), bh = (uint16_t)(b >> 16);
    _Float16 lo = *reinterpret_cast<const _Float16*>(&al) + *reinterpret_cast<const _Float16*>(&bl);
    _Float16 hi = *reinterpret_cast<const _Float16*>(&ah) + *reinterpret_cast<const _Float16*>(&bh);
    if (!(lo > (_Float16)0) && lo == lo) lo = (_Float16)0;
    if (!(hi > (_Float16)0) && hi == hi) hi = (_Float16)0;
    return (uint32_t)(*reinterpret_cast<uint16_t*>(&lo)) | ((uint32_t)(*reinterpret_cast<uint16_t*>(&hi)) << 16);
  }
}
