// hz_replay.hip -- observation windows of a learner batch expanded from the bit-packed replay (include/hz_replay.h); gfx950.
// HBM-bound byte work: one workgroup per output row, 16-B stores where the slot layout allows them.
#include "hz_common.h"
#include "hz_env.h"
#include "hz_replay.h"

// 16-bit elements, slots of a multiple of 8 elements on 16-B boundaries: a thread expands one byte of a frame word into eight
// elements = one 16-B store; the frame words of a slot are read once (44 words for Hanabi-Full 5p: L1 hits after the first lane)
__global__ __launch_bounds__(256) void k_replay_windows16(const int32_t* __restrict__ frames, int W, const int64_t* __restrict__ row0,
                                                          const int32_t* __restrict__ t, int stack, int D, uint16_t* __restrict__ out,
                                                          long long out_row_elems, int slot_elems, uint32_t one) {
  const int m = blockIdx.x;
  const int tm = t[m];
  const long long r0 = row0[m];
  uint4* orow = reinterpret_cast<uint4*>(out + (long long)m * out_row_elems);
  const int chunks = slot_elems >> 3;  // 8 elements per chunk
  for (int q = threadIdx.x; q < stack * chunks; q += blockDim.x) {
    const int j = q / chunks, c8 = q - j * chunks;  // slot, chunk inside it: elements [8 c8, 8 c8 + 8)
    uint32_t bits = 0;
    if (tm >= 0 && 8 * c8 < D) {
      const int ft = tm - (stack - 1) + j;
      const long long fr = r0 + (ft > 0 ? ft : 0);
      bits = ((uint32_t)frames[fr * W + (c8 >> 2)] >> ((c8 & 3) * 8)) & 0xffu;
      const int left = D - 8 * c8;  // (the last chunk of a frame: bits past D are padding of the word, not observation)
      if (left < 8) bits &= (1u << left) - 1u;
    }
    uint4 v;
    v.x = ((bits & 1u) ? one : 0u) | ((bits & 2u) ? one << 16 : 0u);
    v.y = ((bits & 4u) ? one : 0u) | ((bits & 8u) ? one << 16 : 0u);
    v.z = ((bits & 16u) ? one : 0u) | ((bits & 32u) ? one << 16 : 0u);
    v.w = ((bits & 64u) ? one : 0u) | ((bits & 128u) ? one << 16 : 0u);
    orow[j * chunks + c8] = v;
  }
}

// any element type, any slot width: one element per thread and trip (the learner's fp32 input rows of stack * D elements)
template <typename E>
__global__ __launch_bounds__(256) void k_replay_windows(const int32_t* __restrict__ frames, int W, const int64_t* __restrict__ row0,
                                                        const int32_t* __restrict__ t, int stack, int D, E* __restrict__ out,
                                                        long long out_row_elems, int slot_elems, E one, E zero) {
  const int m = blockIdx.x;
  const int tm = t[m];
  const long long r0 = row0[m];
  E* orow = out + (long long)m * out_row_elems;
  for (int q = threadIdx.x; q < stack * slot_elems; q += blockDim.x) {
    const int j = q / slot_elems, c = q - j * slot_elems;
    E v = zero;
    if (tm >= 0 && c < D) {
      const int ft = tm - (stack - 1) + j;
      const long long fr = r0 + (ft > 0 ? ft : 0);
      if (((uint32_t)frames[fr * W + (c >> 5)] >> (c & 31)) & 1u) v = one;
    }
    orow[q] = v;
  }
}

extern "C" int hz_replay_windows(const int32_t* frames, int packed_words, const int64_t* row0, const int32_t* t, int M, int stack, int D,
                                 void* out, int64_t out_row_elems, int64_t slot_elems, int out_dtype, void* stream) {
  HZ_REQUIRE(frames && row0 && t && out, "hz_replay_windows: null pointer");
  HZ_REQUIRE(M >= 0 && stack >= 1 && D >= 1 && packed_words == (D + 31) / 32, "hz_replay_windows: M=%d stack=%d D=%d packed_words=%d (want %d)",
             M, stack, D, packed_words, (D + 31) / 32);
  HZ_REQUIRE(slot_elems >= D && out_row_elems >= (int64_t)stack * slot_elems && slot_elems < (1 << 24),
             "hz_replay_windows: slots of %lld elements for %d-bit frames, rows of %lld for %d slots", (long long)slot_elems, D,
             (long long)out_row_elems, stack);
  HZ_REQUIRE(out_dtype == HZ_OBS_F32 || out_dtype == HZ_OBS_BF16 || out_dtype == HZ_OBS_F16, "hz_replay_windows: bad out_dtype %d", out_dtype);
  if (M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == HZ_OBS_F32) {
    hipLaunchKernelGGL(k_replay_windows<float>, dim3(M), dim3(256), 0, s, frames, packed_words, row0, t, stack, D, (float*)out,
                       (long long)out_row_elems, (int)slot_elems, 1.0f, 0.0f);
  } else {
    const uint16_t one = out_dtype == HZ_OBS_BF16 ? 0x3f80u : 0x3c00u;
    if (slot_elems % 8 == 0 && out_row_elems % 8 == 0 && ((uintptr_t)out % 16) == 0)
      hipLaunchKernelGGL(k_replay_windows16, dim3(M), dim3(256), 0, s, frames, packed_words, row0, t, stack, D, (uint16_t*)out,
                         (long long)out_row_elems, (int)slot_elems, (uint32_t)one);
    else
      hipLaunchKernelGGL(k_replay_windows<uint16_t>, dim3(M), dim3(256), 0, s, frames, packed_words, row0, t, stack, D, (uint16_t*)out,
                         (long long)out_row_elems, (int)slot_elems, one, (uint16_t)0);
  }
  HZ_HIP(hipGetLastError());
  return 0;
}
