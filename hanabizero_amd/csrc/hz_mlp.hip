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

// One layer for this wave.  TW = output tiles (16 columns) per wave, G = groups, RT = row tiles; ks = K/32 at run time.
// The k-loop is ROLLED (4 k-steps per trip, a 4-slot register ring for the weight fragments: prefetch distance 3
// k-steps, >= 24 KiB in flight per wave for TW = 8) so that the whole layer chain stays instruction-cache resident;
// a fully unrolled chain is ~100 KiB of straight-line code executed once per launch and runs at I-fetch speed.
template <int TW, int G, int RT>
__device__ __forceinline__ void run_layer(const hz_mlp_layer_t& L, const uint16_t* __restrict__ W,
                                          const float* __restrict__ bias, uint16_t* lds, int rs, int wave, int lane) {
  constexpr int TG = TW / G;
  constexpr int R = 4;  // ring slots; prefetch distance R - 1
  const int ks = L.K >> 5;
  const int r0 = lane & 15, kq = (lane >> 4) * 8;
  const int ng = L.nout / G;
  // this wave's biases: issued first so their latency hides under the k-loop
  float4 bv[TW];
#pragma unroll
  for (int t = 0; t < TW; ++t)
    bv[t] = *reinterpret_cast<const float4*>(bias + L.b_off + (t / TG) * ng + 16 * (wave * TG + (t % TG)) + 4 * (lane >> 4));
  f32x4 acc[TW][RT];
#pragma unroll
  for (int t = 0; t < TW; ++t)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[t][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bf16x8* wp = reinterpret_cast<const bf16x8*>(W + L.w_off) + (size_t)wave * ks * TW * 64 + lane;
  const uint16_t* src = lds + (size_t)r0 * rs + L.src_off + kq;
  bf16x8 wf[R][TW];
#pragma unroll
  for (int d = 0; d < R - 1; ++d)
    if (d < ks) {
#pragma unroll
      for (int t = 0; t < TW; ++t) wf[d][t] = wp[(d * TW + t) * 64];
    }
  bf16x8 b[G][RT], bn[G][RT];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) b[g][rt] = *reinterpret_cast<const bf16x8*>(src + (size_t)(16 * rt) * rs + g * L.src_gstride);
  __builtin_amdgcn_sched_barrier(0);

#define HZ_MLP_STEP(S, U)                                                                                              \
  {                                                                                                                    \
    if ((S) + R - 1 < ks) {                                                                                            \
      _Pragma("unroll") for (int t = 0; t < TW; ++t) wf[((U) + R - 1) % R][t] = wp[(((S) + R - 1) * TW + t) * 64];    \
    }                                                                                                                  \
    if ((S) + 1 < ks) {                                                                                                \
      _Pragma("unroll") for (int g = 0; g < G; ++g) _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                 \
          bn[g][rt] = *reinterpret_cast<const bf16x8*>(src + (size_t)(16 * rt) * rs + g * L.src_gstride + 32 * ((S) + 1)); \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    _Pragma("unroll") for (int t = 0; t < TW; ++t) _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                  \
        acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[(U) % R][t], b[t / TG][rt], acc[t][rt], 0, 0, 0);     \
    _Pragma("unroll") for (int g = 0; g < G; ++g) _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) b[g][rt] = bn[g][rt]; \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  }

  const int ks_main = ks & ~3;
  for (int s = 0; s < ks_main; s += 4) {
    HZ_MLP_STEP(s, 0)
    HZ_MLP_STEP(s + 1, 1)
    HZ_MLP_STEP(s + 2, 2)
    HZ_MLP_STEP(s + 3, 3)
  }
  if (ks - ks_main >= 1) HZ_MLP_STEP(ks_main, 0)
  if (ks - ks_main >= 2) HZ_MLP_STEP(ks_main + 1, 1)
  if (ks - ks_main >= 3) HZ_MLP_STEP(ks_main + 2, 2)
#undef HZ_MLP_STEP

  // epilogue: bias (+ residual) (+ ReLU) in fp32, round to bf16, 4 consecutive columns per lane
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    const int g = t / TG;
    const int col = g * ng + 16 * (wave * TG + (t % TG)) + 4 * (lane >> 4);
    const bool relu = (L.relu_mask >> g) & 1;
    const bool res = L.res_off >= 0 && (L.res_group < 0 || L.res_group == g);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const size_t rowbase = (size_t)(16 * rt + r0) * rs;
      float v[4] = {acc[t][rt][0] + bv[t].x, acc[t][rt][1] + bv[t].y, acc[t][rt][2] + bv[t].z, acc[t][rt][3] + bv[t].w};
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
  switch (L.kind) {  // (tiles per wave, groups); the k-step count is a run-time value
    case 0: case 1: case 7: run_layer<8, 1, RT>(L, W, bias, lds, rs, wave, lane); break;  // 544|512|576 -> 512
    case 2: run_layer<12, 1, RT>(L, W, bias, lds, rs, wave, lane); break;                 // 512 -> 768
    case 3: run_layer<12, 3, RT>(L, W, bias, lds, rs, wave, lane); break;                 // 3 x (256 -> 256)
    case 4: run_layer<1, 1, RT>(L, W, bias, lds, rs, wave, lane); break;                  // 256 -> 64
    case 5: run_layer<6, 1, RT>(L, W, bias, lds, rs, wave, lane); break;                  // 512 -> 384
    case 6: run_layer<3, 3, RT>(L, W, bias, lds, rs, wave, lane); break;                  // 3 x (128 -> 64)
    default: break;
  }
}

// inverse_scalar_transform of LDS rows of bf16 logits, one (row, head) pair per 16-lane row of the wave (DPP
// reductions, no shuffles through LDS).  Same maths as hz_tree.hip support_to_scalar.
__device__ __forceinline__ float row16_support_to_scalar(const uint16_t* row, int V, int support_min, int l16) {
  float m = -INFINITY;
  for (int i = l16; i < V; i += 16) m = fmaxf(m, bf2f(row[i]));
  m = hz_row16_max(m);
  float se = 0.0f, sw = 0.0f;
  for (int i = l16; i < V; i += 16) {
    const float e = __expf(bf2f(row[i]) - m);
    se += e;
    sw += e * (float)(support_min + i);
  }
  se = hz_row16_sum(se);
  sw = hz_row16_sum(sw);
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
  // heads -> scalars / policy logits: 16 lanes per (row, head) pair, 16 pairs in flight per workgroup pass
  {
    const int l16 = tid & 15, slot = tid >> 4;  // 16 slots of 16 lanes
    for (int pair = slot; pair < 2 * MT; pair += 16) {
      const int r = pair >> 1, head = pair & 1;
      if (row0 + r < n_rows) {
        const uint16_t* row = lds + (size_t)r * rs;
        const float x = row16_support_to_scalar(row + (head ? P.off_value : P.off_reward), P.support_size, P.support_min, l16);
        if (l16 == 0) (head ? out_value : out_reward)[row0 + r] = x;
      }
    }
    for (int i = tid; i < MT * P.num_actions; i += 256) {
      const int r = i / P.num_actions, a = i % P.num_actions;
      if (row0 + r < n_rows) {
        float x = bf2f(lds[(size_t)r * rs + P.off_policy + a]);
        if (x != x) x = 0.0f;  // core/mcts.py:48-49
        out_policy[(size_t)(row0 + r) * P.num_actions + a] = x;
      }
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
