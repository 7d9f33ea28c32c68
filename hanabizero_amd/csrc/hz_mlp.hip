// hz_mlp.hip -- the whole recurrent inference of one simulation (dynamics + reward/value/policy heads + scalar
// transforms) as ONE MFMA kernel for gfx950.  See include/hz_mlp.h for what it replaces and why.
//
// Work split: workgroup = 256 threads = 4 waves = MT rows (16 or 32) of the batch; the rows' activations live in LDS
// (bf16, one image row per batch row) for the whole layer chain; per layer the 4 waves split the output columns,
// every weight element is fetched from L2/HBM exactly once per workgroup (packed so that a wave's fragment loads are
// 1 KiB-contiguous dwordx4 loads, prefetched two k-steps ahead into registers) and multiplied on the matrix cores:
//     D[n][row] += W[n][k] * X[k][row]      v_mfma_f32_16x16x32_bf16, A = weights, B = activations
// so a lane ends up with 4 consecutive output columns of one batch row -> one ds_write_b64 per tile into the next
// layer's input image.  The kernel is bound by the per-CU weight stream (the weights are shared by all rows of a
// workgroup only); MT grows with N so that the grid stays <= 256 workgroups (one per CU).
#include "hz_common.h"
#include "hz_mlp.h"
#include "hz_tree.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {  // round-to-nearest-even; NaN stays NaN
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// One layer for this wave.  TW = output tiles (16 columns) per wave, G = groups, KS = k-steps (K/32), RT = row tiles.
template <int TW, int G, int KS, int RT>
__device__ __forceinline__ void run_layer(const hz_mlp_layer_t& L, const uint16_t* __restrict__ W,
                                          const float* __restrict__ bias, uint16_t* lds, int rs, int wave, int lane) {
  constexpr int TG = TW / G;
  constexpr int D = (KS >= 4) ? 4 : 2;  // weight prefetch distance in k-steps (>= 32 KiB in flight per wave)
  f32x4 acc[TW][RT];
#pragma unroll
  for (int t = 0; t < TW; ++t)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[t][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bf16x8* wp = reinterpret_cast<const bf16x8*>(W + L.w_off) + (size_t)wave * KS * TW * 64 + lane;
  bf16x8 wf[D + 1][TW];
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < KS) {
#pragma unroll
      for (int t = 0; t < TW; ++t) wf[d][t] = wp[(d * TW + t) * 64];
    }
  const int r0 = lane & 15, kq = (lane >> 4) * 8;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (s + D < KS) {
#pragma unroll
      for (int t = 0; t < TW; ++t) wf[(s + D) % (D + 1)][t] = wp[((s + D) * TW + t) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch D k-steps ahead of its use (the scheduler would sink it)
    bf16x8 b[G][RT];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        b[g][rt] = *reinterpret_cast<const bf16x8*>(lds + (size_t)(16 * rt + r0) * rs + L.src_off + g * L.src_gstride + 32 * s + kq);
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s % (D + 1)][t], b[t / TG][rt], acc[t][rt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  // epilogue: bias (+ residual) (+ ReLU) in fp32, round to bf16, 4 consecutive columns per lane
  const int ng = L.nout / G;
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    const int g = t / TG;
    const int col = g * ng + 16 * (wave * TG + (t % TG)) + 4 * (lane >> 4);
    const float4 bv = *reinterpret_cast<const float4*>(bias + L.b_off + col);
    const bool relu = (L.relu_mask >> g) & 1;
    const bool res = L.res_off >= 0 && (L.res_group < 0 || L.res_group == g);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const size_t rowbase = (size_t)(16 * rt + r0) * rs;
      float v[4] = {acc[t][rt][0] + bv.x, acc[t][rt][1] + bv.y, acc[t][rt][2] + bv.z, acc[t][rt][3] + bv.w};
      if (res) {
        const uint2 rr = *reinterpret_cast<const uint2*>(lds + rowbase + L.res_off + col);
        v[0] += bf2f((uint16_t)(rr.x & 0xffffu)); v[1] += bf2f((uint16_t)(rr.x >> 16));
        v[2] += bf2f((uint16_t)(rr.y & 0xffffu)); v[3] += bf2f((uint16_t)(rr.y >> 16));
      }
      if (relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (v[r] > 0.0f || v[r] != v[r]) ? v[r] : 0.0f;
      }
      uint2 o;
      o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
      o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
      *reinterpret_cast<uint2*>(lds + rowbase + L.dst_off + col) = o;
    }
  }
}

template <int RT>
__device__ __forceinline__ void dispatch_layer(const hz_mlp_layer_t& L, const uint16_t* W, const float* bias,
                                               uint16_t* lds, int rs, int wave, int lane) {
  switch (L.kind) {  // (tiles per wave, groups, k-steps)
    case 0: run_layer<8, 1, 17, RT>(L, W, bias, lds, rs, wave, lane); break;   // 544 -> 512
    case 1: run_layer<8, 1, 16, RT>(L, W, bias, lds, rs, wave, lane); break;   // 512 -> 512
    case 2: run_layer<12, 1, 16, RT>(L, W, bias, lds, rs, wave, lane); break;  // 512 -> 768
    case 3: run_layer<12, 3, 8, RT>(L, W, bias, lds, rs, wave, lane); break;   // 3 x (256 -> 256)
    case 4: run_layer<1, 1, 8, RT>(L, W, bias, lds, rs, wave, lane); break;    // 256 -> 64
    case 5: run_layer<6, 1, 16, RT>(L, W, bias, lds, rs, wave, lane); break;   // 512 -> 384
    case 6: run_layer<3, 3, 4, RT>(L, W, bias, lds, rs, wave, lane); break;    // 3 x (128 -> 64)
    case 7: run_layer<8, 1, 18, RT>(L, W, bias, lds, rs, wave, lane); break;   // 576 -> 512
    default: break;
  }
}

// inverse_scalar_transform of one LDS row of bf16 logits by one wave (same maths as hz_tree.hip support_to_scalar)
__device__ __forceinline__ float lds_support_to_scalar(const uint16_t* row, int V, int support_min, int lane) {
  float m = -INFINITY;
  for (int i = lane; i < V; i += 64) m = fmaxf(m, bf2f(row[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float se = 0.0f, sw = 0.0f;
  for (int i = lane; i < V; i += 64) {
    const float e = __expf(bf2f(row[i]) - m);
    se += e;
    sw += e * (float)(support_min + i);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    se += __shfl_xor(se, o, 64);
    sw += __shfl_xor(sw, o, 64);
  }
  const float v = sw / se;
  const float eps = 0.001f;
  const float t = (sqrtf(1.0f + 4.0f * eps * (fabsf(v) + 1.0f + eps)) - 1.0f) / (2.0f * eps);
  float out = t * t - 1.0f;
  if (v < 0.0f) out = -out;
  if (out != out) out = 0.0f;
  return out;
}

template <int RT>
__global__ __launch_bounds__(256, 1) void k_mlp_recurrent(hz_mlp_program_t P, const uint16_t* __restrict__ net_in,
                                                          long long net_in_stride, const uint16_t* __restrict__ W,
                                                          const float* __restrict__ bias,
                                                          uint16_t* __restrict__ hidden_out, float* __restrict__ out_reward,
                                                          float* __restrict__ out_value, float* __restrict__ out_policy,
                                                          int n_rows) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  constexpr int MT = 16 * RT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * MT;
  const int rs = P.row_stride;
  // stage [state | one-hot | pad] rows into LDS columns [0, in_width), 16 B per thread-trip; rows past N read as zero
  {
    const int chunks = P.in_width / 8;
    for (int i = tid; i < MT * chunks; i += 256) {
      const int r = i / chunks, c = i % chunks;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (row0 + r < n_rows) v = *reinterpret_cast<const uint4*>(net_in + (size_t)(row0 + r) * net_in_stride + c * 8);
      *reinterpret_cast<uint4*>(lds + (size_t)r * rs + c * 8) = v;
    }
  }
  __syncthreads();
  for (int li = 0; li < P.n_layers; ++li) {
    const hz_mlp_layer_t& L = P.layer[li];
    dispatch_layer<RT>(L, W, bias, lds, rs, wave, lane);
    __syncthreads();
    if (L.store_hidden) {
      const int chunks = P.hidden / 8;
      for (int i = tid; i < MT * chunks; i += 256) {
        const int r = i / chunks, c = i % chunks;
        if (row0 + r < n_rows)
          *reinterpret_cast<uint4*>(hidden_out + (size_t)(row0 + r) * P.hidden + c * 8) =
              *reinterpret_cast<const uint4*>(lds + (size_t)r * rs + L.dst_off + c * 8);
      }
    }
  }
  // heads -> scalars / policy logits: one wave per row, rows round-robin over the 4 waves
  for (int r = wave; r < MT; r += 4) {
    if (row0 + r >= n_rows) break;
    const uint16_t* row = lds + (size_t)r * rs;
    const float rew = lds_support_to_scalar(row + P.off_reward, P.support_size, P.support_min, lane);
    const float val = lds_support_to_scalar(row + P.off_value, P.support_size, P.support_min, lane);
    if (lane == 0) {
      out_reward[row0 + r] = rew;
      out_value[row0 + r] = val;
    }
    if (lane < P.num_actions) {
      float x = bf2f(row[P.off_policy + lane]);
      if (x != x) x = 0.0f;  // core/mcts.py:48-49
      out_policy[(size_t)(row0 + r) * P.num_actions + lane] = x;
    }
  }
}

extern "C" int hz_mlp_recurrent(const hz_mlp_program_t* P, const void* net_in, int64_t net_in_stride, const void* weights,
                                const float* biases, void* hidden_out, float* out_reward, float* out_value,
                                float* out_policy, int num_rows, int rows_per_wg, void* stream) {
  HZ_REQUIRE(P && net_in && weights && biases && hidden_out && out_reward && out_value && out_policy,
             "hz_mlp_recurrent: NULL argument");
  HZ_REQUIRE(num_rows > 0, "hz_mlp_recurrent: num_rows must be > 0");
  HZ_REQUIRE(rows_per_wg == 16 || rows_per_wg == 32, "hz_mlp_recurrent: rows_per_wg must be 16 or 32");
  HZ_REQUIRE(P->n_layers > 0 && P->n_layers <= HZ_MLP_MAX_LAYERS, "hz_mlp_recurrent: bad layer count %d", P->n_layers);
  HZ_REQUIRE(P->row_stride % 8 == 0 && P->in_width % 8 == 0 && P->hidden % 8 == 0 && net_in_stride % 8 == 0,
             "hz_mlp_recurrent: row_stride, in_width, hidden and net_in_stride must be multiples of 8 elements");
  HZ_REQUIRE(((uintptr_t)net_in % 16) == 0 && ((uintptr_t)weights % 16) == 0 && ((uintptr_t)hidden_out % 16) == 0 &&
                 ((uintptr_t)biases % 16) == 0,
             "hz_mlp_recurrent: pointers must be 16-B aligned");
  for (int i = 0; i < P->n_layers; ++i) {
    const hz_mlp_layer_t& L = P->layer[i];
    HZ_REQUIRE(L.kind >= 0 && L.kind <= 7, "hz_mlp_recurrent: layer %d has unknown kind %d", i, L.kind);
    HZ_REQUIRE(L.src_off % 8 == 0 && L.src_gstride % 8 == 0 && L.dst_off % 4 == 0 && (L.res_off < 0 || L.res_off % 4 == 0) &&
                   L.w_off % 8 == 0 && L.b_off % 4 == 0,
               "hz_mlp_recurrent: layer %d has misaligned offsets", i);
  }
  const size_t lds_bytes = (size_t)rows_per_wg * P->row_stride * sizeof(uint16_t);
  HZ_REQUIRE(lds_bytes <= 160 * 1024, "hz_mlp_recurrent: %zu B of LDS per workgroup exceed 160 KiB", lds_bytes);
  const int grid = (num_rows + rows_per_wg - 1) / rows_per_wg;
  if (rows_per_wg == 16) {
    HZ_HIP(hipFuncSetAttribute((const void*)k_mlp_recurrent<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(k_mlp_recurrent<1>, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, *P, (const uint16_t*)net_in,
                       (long long)net_in_stride, (const uint16_t*)weights, biases, (uint16_t*)hidden_out, out_reward,
                       out_value, out_policy, num_rows);
  } else {
    HZ_HIP(hipFuncSetAttribute((const void*)k_mlp_recurrent<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(k_mlp_recurrent<2>, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, *P, (const uint16_t*)net_in,
                       (long long)net_in_stride, (const uint16_t*)weights, biases, (uint16_t*)hidden_out, out_reward,
                       out_value, out_policy, num_rows);
  }
  HZ_HIP(hipGetLastError());
  return 0;
}
