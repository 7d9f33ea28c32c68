// hz_netglue.hip -- residual-add + ReLU between the dynamics/prediction GEMMs (gfx950); HBM-bound elementwise.
#include "hz_common.h"
#include "hz_netglue.h"
#include "hz_tree.h"

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {  // round to nearest even; inputs here are finite sums
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

template <int DTYPE>
__global__ __launch_bounds__(256) void k_add_relu_vec(uint4* __restrict__ y, long long ys16, const uint4* __restrict__ r,
                                                      long long rs16, int rows, int cols16) {
  // one uint4 (16 B = 8 x 16-bit or 4 x f32) per thread-iteration, grid-stride over rows*cols16
  const long long total = (long long)rows * cols16;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / cols16), c = (int)(i % cols16);
    uint4 a = y[row * ys16 + c];
    const uint4 b = r[row * rs16 + c];
    uint32_t* pa = reinterpret_cast<uint32_t*>(&a);
    const uint32_t* pb = reinterpret_cast<const uint32_t*>(&b);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (DTYPE == HZ_F32) {
        const float v = __uint_as_float(pa[k]) + __uint_as_float(pb[k]);
        pa[k] = __float_as_uint(v > 0.0f ? v : (v != v ? v : 0.0f));
      } else if (DTYPE == HZ_BF16) {
        const float lo = bf16_to_f32((uint16_t)(pa[k] & 0xffffu)) + bf16_to_f32((uint16_t)(pb[k] & 0xffffu));
        const float hi = bf16_to_f32((uint16_t)(pa[k] >> 16)) + bf16_to_f32((uint16_t)(pb[k] >> 16));
        const uint16_t l = f32_to_bf16(lo > 0.0f ? lo : (lo != lo ? lo : 0.0f));
        const uint16_t h = f32_to_bf16(hi > 0.0f ? hi : (hi != hi ? hi : 0.0f));
        pa[k] = (uint32_t)l | ((uint32_t)h << 16);
      } else {
        const uint16_t al = (uint16_t)(pa[k] & 0xffffu), ah = (uint16_t)(pa[k] >> 16);
        const uint16_t bl = (uint16_t)(pb[k] & 0xffffu), bh = (uint16_t)(pb[k] >> 16);
        _Float16 lo = *reinterpret_cast<const _Float16*>(&al) + *reinterpret_cast<const _Float16*>(&bl);
        _Float16 hi = *reinterpret_cast<const _Float16*>(&ah) + *reinterpret_cast<const _Float16*>(&bh);
        if (!(lo > (_Float16)0) && lo == lo) lo = (_Float16)0;
        if (!(hi > (_Float16)0) && hi == hi) hi = (_Float16)0;
        pa[k] = (uint32_t)(*reinterpret_cast<uint16_t*>(&lo)) | ((uint32_t)(*reinterpret_cast<uint16_t*>(&hi)) << 16);
      }
    }
    y[row * ys16 + c] = a;
  }
}

extern "C" int hz_add_relu(void* y, int64_t y_stride, const void* res, int64_t res_stride, int rows, int cols, int dtype,
                           void* stream) {
  HZ_REQUIRE(y && res && rows > 0 && cols > 0, "hz_add_relu: bad argument");
  HZ_REQUIRE(dtype == HZ_F32 || dtype == HZ_BF16 || dtype == HZ_F16, "hz_add_relu: bad dtype %d", dtype);
  const int es = dtype == HZ_F32 ? 4 : 2, per = 16 / es;
  HZ_REQUIRE(cols % per == 0 && y_stride % per == 0 && res_stride % per == 0 && ((uintptr_t)y % 16) == 0 &&
                 ((uintptr_t)res % 16) == 0,
             "hz_add_relu: cols, strides (elements) must be multiples of %d and pointers 16-B aligned", per);
  const int cols16 = cols / per;
  const long long total = (long long)rows * cols16;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  const long long ys = y_stride / per, rs = res_stride / per;
  if (dtype == HZ_F32)
    hipLaunchKernelGGL(k_add_relu_vec<HZ_F32>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint4*)y, ys, (const uint4*)res, rs, rows, cols16);
  else if (dtype == HZ_BF16)
    hipLaunchKernelGGL(k_add_relu_vec<HZ_BF16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint4*)y, ys, (const uint4*)res, rs, rows, cols16);
  else
    hipLaunchKernelGGL(k_add_relu_vec<HZ_F16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint4*)y, ys, (const uint4*)res, rs, rows, cols16);
  HZ_HIP(hipGetLastError());
  return 0;
}
