// tools/mlp_loop.hip -- the product's fused inference (hz_mlp_dev.h::mlp_body, 16 waves x 2 tiles, exactly as the persistent
// search kernel instantiates it: rows handed over in registers, no final stage) ITERS times back to back inside one
// launch, nothing else: no tree phases, no launch gaps.  Stamps with s_memtime (shader clock) and s_memrealtime (100 MHz):
// bytes per shader cycle of the weight stream, comparable with tools/l2_stream_bench.hip's synthetic ceiling, and the shader
// clock the chip actually runs at under this load.  Built and driven by tools/mlp_loop_bench.py.
#include "hz_mlp_dev.h"

template <class EL, int RT>
__global__ __launch_bounds__(1024, 1) __attribute__((amdgpu_num_vgpr(HZ_ASMK_VGPRS))) void k_mlp_loop(
    hz_mlp_header_t H, const hz_mlp_job_t* __restrict__ jobs, const uint16_t* __restrict__ wstream, const float* __restrict__ bias,
    const float* __restrict__ act_tab, const int32_t* __restrict__ actions, uint16_t* __restrict__ hidden_out, int n_rows, int iters,
    unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  const int row0 = (int)blockIdx.x * 16 * RT;
  RowFrag rows[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const unsigned int x = 0x3c003c00u ^ ((threadIdx.x * 2654435761u) & 0x00ff00ffu);
    rows[rt].v[0] = make_uint4(x, x, x, x);
    rows[rt].v[1] = rows[rt].v[0];
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    mlp_body<EL, RT, 16, 2, STAGE_REGS, false>(H, jobs, wstream, bias, act_tab, nullptr, 0, nullptr, 0, actions, hidden_out, nullptr,
                                               nullptr, nullptr, n_rows, lds, row0, rows);
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && (blockIdx.x & 63) == 36) {
    stamps[2 * (blockIdx.x >> 6)] = t1 - t0;
    stamps[2 * (blockIdx.x >> 6) + 1] = r1 - r0;
  }
}

extern "C" int hz_mlp_loop(const hz_mlp_header_t* H, const hz_mlp_job_t* jobs, const void* wstream, const float* biases,
                           const float* action_table, const int32_t* actions, void* hidden_out, int num_rows, int rows_per_wg,
                           int iters, unsigned long long* stamps, void* stream) {
  const size_t lds_bytes = (size_t)rows_per_wg * H->row_stride * sizeof(uint16_t);
  const int grid = num_rows / rows_per_wg;
  if (rows_per_wg == 16) {
    if (hipFuncSetAttribute((const void*)k_mlp_loop<ElBf16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return 1;
    hipLaunchKernelGGL((k_mlp_loop<ElBf16, 1>), dim3(grid), dim3(1024), lds_bytes, (hipStream_t)stream, *H, jobs, (const uint16_t*)wstream,
                       biases, action_table, actions, (uint16_t*)hidden_out, num_rows, iters, stamps);
  } else {
    if (hipFuncSetAttribute((const void*)k_mlp_loop<ElBf16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return 1;
    hipLaunchKernelGGL((k_mlp_loop<ElBf16, 2>), dim3(grid), dim3(1024), lds_bytes, (hipStream_t)stream, *H, jobs, (const uint16_t*)wstream,
                       biases, action_table, actions, (uint16_t*)hidden_out, num_rows, iters, stamps);
  }
  return (int)hipGetLastError();
}
