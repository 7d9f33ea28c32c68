// hz_netglue.hip -- residual-add + ReLU between the dynamics/prediction GEMMs (gfx950); HBM-bound elementwise.
#include "hz_addrelu_dev.h"
#include "hz_netglue.h"

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
    for (int k = 0; k < 4; ++k) pa[k] = hz_add_relu_word<DTYPE>(pa[k], pb[k]);
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
