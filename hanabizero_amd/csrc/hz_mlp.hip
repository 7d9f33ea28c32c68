// hz_mlp.hip -- the whole recurrent inference of one simulation (hidden-state gather + dynamics + reward/value/policy
// heads + scalar transforms) as ONE MFMA kernel for gfx950.  See include/hz_mlp.h for what it replaces and why.
//
// Work split (body in hz_mlp_dev.h, shared with the persistent search kernel hz_search.hip): a workgroup of NW waves
// (4, 8 or 16) owns MT rows (16 or 32) of the batch.  The rows' activations live in an LDS image (bf16 or fp16, one image row per
// batch row) for the whole layer chain.  The chain is a table of jobs; in job j wave w produces 16 * NT output columns
// (NT MFMA tiles: 4, or 2 with 16 waves) of one layer:
//     D[n][row] += W[n][k] * X[k][row]      v_mfma_f32_16x16x32_{bf16,f16}, A = weights, B = activations (ds_read_b128)
// so a lane ends up with 4 consecutive output columns of one batch row -> one ds_write_b64 per tile into the image.
// Each wave's weights are ONE stream in execution order (1 KiB-contiguous dwordx4 fragment loads; the waves' streams are
// interleaved k-step by k-step in memory), prefetched HZ_RING - 1 = 3 k-steps ahead in a register ring across job, layer
// and barrier boundaries (barriers wait for LDS traffic only).  The kernel is bound by the CU's L2 port: every weight byte
// is fetched once per workgroup and that stream, not the matrix cores, sets the time (DESIGN.md section 4 has the
// measurements: ring depth, burst size and stream layout do not matter; the layer boundaries do).
#include "hz_mlp_dev.h"
#include "hz_tree.h"

template <class EL, int RT, int NW, int NT>
__global__ __launch_bounds__(64 * NW, 1) void k_mlp_recurrent(
    hz_mlp_header_t H, const hz_mlp_job_t* __restrict__ jobs, const uint16_t* __restrict__ wstream,
    const float* __restrict__ bias, const float* __restrict__ act_tab, const uint16_t* __restrict__ state_src,
    long long state_row_stride, const int32_t* __restrict__ plane_index, long long plane_stride,
    const int32_t* __restrict__ actions, uint16_t* __restrict__ hidden_out, float* __restrict__ out_reward,
    float* __restrict__ out_value, float* __restrict__ out_policy, int n_rows, const uint16_t* __restrict__ state_res,
    long long state_res_stride) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  mlp_body<EL, RT, NW, NT>(H, jobs, wstream, bias, act_tab, state_src, state_row_stride, plane_index, plane_stride, actions,
                       hidden_out, out_reward, out_value, out_policy, n_rows, lds, (int)blockIdx.x * 16 * RT, nullptr, nullptr,
                       state_res, state_res_stride);
}

// The 16 waves x 2 tiles shape runs the hand-scheduled k-loop of hz_mlp_dev.h, whose weight ring lives in registers the
// compiler is kept away from (HZ_ASMK_VGPRS).
template <class EL, int RT>
__global__ __launch_bounds__(1024, 1) __attribute__((amdgpu_num_vgpr(HZ_ASMK_VGPRS))) void k_mlp_recurrent16(
    hz_mlp_header_t H, const hz_mlp_job_t* __restrict__ jobs, const uint16_t* __restrict__ wstream,
    const float* __restrict__ bias, const float* __restrict__ act_tab, const uint16_t* __restrict__ state_src,
    long long state_row_stride, const int32_t* __restrict__ plane_index, long long plane_stride,
    const int32_t* __restrict__ actions, uint16_t* __restrict__ hidden_out, float* __restrict__ out_reward,
    float* __restrict__ out_value, float* __restrict__ out_policy, int n_rows, const uint16_t* __restrict__ state_res,
    long long state_res_stride) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  mlp_body<EL, RT, 16, 2>(H, jobs, wstream, bias, act_tab, state_src, state_row_stride, plane_index, plane_stride, actions,
                          hidden_out, out_reward, out_value, out_policy, n_rows, lds, (int)blockIdx.x * 16 * RT, nullptr, nullptr,
                          state_res, state_res_stride);
}

extern "C" int hz_search_poll_giveups(unsigned int* count);
extern "C" int hz_mlp_poll_giveups(unsigned int* count) {
  HZ_REQUIRE(count != nullptr, "hz_mlp_poll_giveups: NULL argument");
  unsigned int mine = 0, theirs = 0;
  HZ_HIP(hipMemcpyFromSymbol(&mine, HIP_SYMBOL(hz_poll_giveups_dev), sizeof(unsigned int)));
  const int rc = hz_search_poll_giveups(&theirs);
  if (rc != 0) return rc;
  *count = mine + theirs;
  return 0;
}

extern "C" int hz_search_poll_giveups_async(unsigned int* host_pinned, void* stream);
extern "C" int hz_mlp_poll_giveups_async(unsigned int* host_pinned2, void* stream) {
  HZ_REQUIRE(host_pinned2 != nullptr, "hz_mlp_poll_giveups_async: NULL argument");
  HZ_HIP(hipMemcpyFromSymbolAsync(host_pinned2, HIP_SYMBOL(hz_poll_giveups_dev), sizeof(unsigned int), 0, hipMemcpyDeviceToHost,
                                  (hipStream_t)stream));
  return hz_search_poll_giveups_async(host_pinned2 + 1, stream);
}

extern "C" int hz_mlp_recurrent(const hz_mlp_header_t* H, const hz_mlp_job_t* jobs, const void* wstream,
                                const float* biases, const float* action_table, const void* state_src,
                                int64_t row_stride, const int32_t* plane_index, int64_t plane_stride,
                                const int32_t* actions, void* hidden_out, float* out_reward, float* out_value,
                                float* out_policy, int num_rows, int rows_per_wg, void* stream) {
  return hz_mlp_recurrent_res(H, jobs, wstream, biases, action_table, state_src, row_stride, plane_index, plane_stride, actions,
                              hidden_out, out_reward, out_value, out_policy, num_rows, rows_per_wg, nullptr, 0, stream);
}

extern "C" int hz_mlp_recurrent_res(const hz_mlp_header_t* H, const hz_mlp_job_t* jobs, const void* wstream,
                                    const float* biases, const float* action_table, const void* state_src,
                                    int64_t row_stride, const int32_t* plane_index, int64_t plane_stride,
                                    const int32_t* actions, void* hidden_out, float* out_reward, float* out_value,
                                    float* out_policy, int num_rows, int rows_per_wg, const void* state_res,
                                    int64_t res_stride, void* stream) {
  HZ_REQUIRE(H && jobs && wstream && biases && action_table && state_src && actions && hidden_out && out_reward &&
                 out_value && out_policy,
             "hz_mlp_recurrent: NULL argument");
  HZ_REQUIRE(num_rows > 0, "hz_mlp_recurrent: num_rows must be > 0");
  HZ_REQUIRE(state_res == nullptr || (plane_index == nullptr && res_stride % 8 == 0 && res_stride >= H->in_width && ((uintptr_t)state_res % 16) == 0),
             "hz_mlp_recurrent_res: the residual rows go with ungathered input rows, 16-B aligned, stride a multiple of 8 elements");
  HZ_REQUIRE(rows_per_wg == 16 || rows_per_wg == 32, "hz_mlp_recurrent: rows_per_wg must be 16 or 32");
  HZ_REQUIRE(H->n_jobs > 0 && H->n_jobs <= 32, "hz_mlp_recurrent: bad job count %d", H->n_jobs);
  HZ_REQUIRE(H->dtype == HZ_BF16 || H->dtype == HZ_F16 || H->dtype == HZ_F16X2,
             "hz_mlp_recurrent: header dtype must be HZ_BF16, HZ_F16 or HZ_F16X2 (got %d)", H->dtype);
  HZ_REQUIRE(H->dtype != HZ_F16X2 || (H->num_waves != 16 && state_res == nullptr && H->lo_plane > 0 && H->lo_plane % 8 == 0 &&
                                      2 * H->lo_plane <= H->row_stride),
             "hz_mlp_recurrent: the fp16-pair build runs the 4- and 8-wave shapes on ungathered or gathered fp32 rows; its lo plane "
             "lies lo_plane (a multiple of 8, at most row_stride / 2) columns behind the hi plane");
  HZ_REQUIRE(H->support_size > 0 && H->support_size <= 256 && H->off_reward % 8 == 0 && H->off_value % 8 == 0,
             "hz_mlp_recurrent: support_size must be <= 256 and the logit columns 16-B aligned");
  HZ_REQUIRE(H->row_stride % 8 == 0 && H->hidden % 8 == 0 && row_stride % 8 == 0 && plane_stride % 8 == 0 &&
                 H->state_off % 8 == 0 && H->hidden_off % 8 == 0 && H->action_table_stride % 4 == 0,
             "hz_mlp_recurrent: strides and offsets must be multiples of 8 elements");
  HZ_REQUIRE(((uintptr_t)state_src % 16) == 0 && ((uintptr_t)wstream % 16) == 0 && ((uintptr_t)hidden_out % 16) == 0 &&
                 ((uintptr_t)biases % 16) == 0 && ((uintptr_t)action_table % 16) == 0,
             "hz_mlp_recurrent: pointers must be 16-B aligned");
  HZ_REQUIRE((H->num_waves == 4 && H->tiles_per_wave == 4) || (H->num_waves == 16 && H->tiles_per_wave == 2) ||
                 (H->num_waves == 8 && H->tiles_per_wave == 4),
             "hz_mlp_recurrent: workgroup shape must be 4 waves x 4 tiles, 8 x 4 or 16 x 2 (got %d x %d)", H->num_waves,
             H->tiles_per_wave);
  for (int w = 0; w < H->num_waves; ++w)
    HZ_REQUIRE(H->wave_stream_off[w] % 8 == 0, "hz_mlp_recurrent: weight streams must start on 16-B boundaries");
  HZ_REQUIRE(H->kstep_stride >= 512 * H->tiles_per_wave && H->kstep_stride % 8 == 0,
             "hz_mlp_recurrent: kstep_stride must be a multiple of 8 and at least one k-step (512 * tiles_per_wave)");
  HZ_REQUIRE(H->in_width > 0 && H->in_width % 8 == 0, "hz_mlp_recurrent: in_width must be a positive multiple of 8");
  HZ_REQUIRE(H->num_waves != 16 || H->hidden <= 512, "hz_mlp_recurrent: the 16 x 2 shape stores a hidden state of <= 512 columns");
#ifdef HZ_MLP_PROFILE
  const size_t lds_bytes = (size_t)rows_per_wg * H->row_stride * sizeof(uint16_t) + 2048;  // + the per-wave timeline
#else
  const size_t lds_bytes = (size_t)rows_per_wg * H->row_stride * sizeof(uint16_t);
#endif
  HZ_REQUIRE(lds_bytes <= 160 * 1024, "hz_mlp_recurrent: %zu B of LDS per workgroup exceed 160 KiB", lds_bytes);
  const int grid = (num_rows + rows_per_wg - 1) / rows_per_wg;
  // (the dynamic-LDS limit is a per-device attribute of each kernel: remembered per device ordinal)
  int dev = 0;
  HZ_HIP(hipGetDevice(&dev));
  HZ_REQUIRE(dev >= 0 && dev < 64, "hz_mlp_recurrent: device ordinal %d outside [0, 64)", dev);
#define HZ_LAUNCH_EL(EL, RT, NW, NT) HZ_LAUNCH_K((k_mlp_recurrent<EL, RT, NW, NT>), NW)
#define HZ_LAUNCH_EL16(EL, RT) HZ_LAUNCH_K((k_mlp_recurrent16<EL, RT>), 16)
#define HZ_LAUNCH_K(KERNEL, NW)                                                                                 \
  do {                                                                                                               \
    static size_t configured[64];                                                                                    \
    if (lds_bytes > configured[dev]) {                                                                               \
      HZ_HIP(hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));   \
      configured[dev] = lds_bytes;                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(64 * NW), lds_bytes, (hipStream_t)stream,                            \
                       *H, jobs, (const uint16_t*)wstream, biases, action_table, (const uint16_t*)state_src,         \
                       (long long)row_stride, plane_index, (long long)plane_stride, actions, (uint16_t*)hidden_out,  \
                       out_reward, out_value, out_policy, num_rows, (const uint16_t*)state_res, (long long)res_stride); \
  } while (0)
#define HZ_LAUNCH(RT, NW, NT)                           \
  do {                                                  \
    if (H->dtype == HZ_F16) HZ_LAUNCH_EL(ElF16, RT, NW, NT); \
    else if (H->dtype == HZ_F16X2) HZ_LAUNCH_EL(ElF16x2, RT, NW, NT); \
    else HZ_LAUNCH_EL(ElBf16, RT, NW, NT);              \
  } while (0)
  if (H->num_waves == 4) {
    if (rows_per_wg == 16) HZ_LAUNCH(1, 4, 4);
    else HZ_LAUNCH(2, 4, 4);
  } else if (H->num_waves == 8) {
    if (rows_per_wg == 16) HZ_LAUNCH(1, 8, 4);
    else HZ_LAUNCH(2, 8, 4);
  } else if (H->dtype == HZ_F16) {
    if (rows_per_wg == 16) HZ_LAUNCH_EL16(ElF16, 1);
    else HZ_LAUNCH_EL16(ElF16, 2);
  } else {
    if (rows_per_wg == 16) HZ_LAUNCH_EL16(ElBf16, 1);
    else HZ_LAUNCH_EL16(ElBf16, 2);
  }
#undef HZ_LAUNCH
#undef HZ_LAUNCH_EL
#undef HZ_LAUNCH_EL16
#undef HZ_LAUNCH_K
  HZ_HIP(hipGetLastError());
  return 0;
}
