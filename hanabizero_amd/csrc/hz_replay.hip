// hz_replay.hip -- observation windows of a learner batch expanded from the bit-packed replay (include/hz_replay.h); gfx950.
// HBM-bound byte work: one workgroup per output row, 16-B stores where the slot layout allows them.
#include "hz_common.h"
#include "hz_env.h"
#include "hz_replay.h"

// Where output row m looks: either given per row (row0[m], t[m]) or derived from the replay's position arrays (hz_replay_windows_seq):
// row m = (b, j) of B x G rows, position p = phys[b], the window ending `shift0 + j` moves behind it -- none (t = -1) past the game's end.
struct WindowIndex {
  const int64_t* row0;
  const int32_t* t;
  const int64_t* phys;      // != null: the sequential form
  const int64_t* pos_row0;
  const int32_t* pos_t;
  const int32_t* pos_T;
  int G, shift0;
  // ... and, optionally, what the re-search needs beside the windows: the legal-move row of the window's last frame and the mask
  const uint8_t* legal;     // [frame rows][A]
  uint8_t* legal_out;       // [M][A]
  uint8_t* valid_out;       // [M]
  int A;
  long long frame_rows;
};
__device__ __forceinline__ void window_index(const WindowIndex& ix, int m, int& tm, long long& r0) {
  if (ix.phys == nullptr) {
    tm = ix.t[m];
    r0 = ix.row0[m];
    return;
  }
  const int b = m / ix.G, j = m - b * ix.G;
  const long long p = ix.phys[b];
  const int t = ix.pos_t[p], shift = ix.shift0 + j;
  const bool valid = t + shift < ix.pos_T[p];
  tm = valid ? t + shift : -1;
  r0 = ix.pos_row0[p];
  if (ix.legal_out != nullptr) {  // (reanalyze_worker.py:101-144: the legal moves at the window's own position, zeros past the end)
    long long fr = r0 + t + shift;
    if (fr > ix.frame_rows - 1) fr = ix.frame_rows - 1;
    for (int a = threadIdx.x; a < ix.A; a += blockDim.x) ix.legal_out[(long long)m * ix.A + a] = valid ? ix.legal[fr * ix.A + a] : (uint8_t)0;
  }
  if (ix.valid_out != nullptr && threadIdx.x == 0) ix.valid_out[m] = valid ? 1 : 0;
}

// 16-bit elements, slots of a multiple of 8 elements on 16-B boundaries: a thread expands one byte of a frame word into eight
// elements = one 16-B store; the frame words of a slot are read once (44 words for Hanabi-Full 5p: L1 hits after the first lane)
__global__ __launch_bounds__(256) void k_replay_windows16(const int32_t* __restrict__ frames, int W, WindowIndex ix, int stack, int D,
                                                          uint16_t* __restrict__ out, long long out_row_elems, int slot_elems, uint32_t one) {
  const int m = blockIdx.x;
  int tm;
  long long r0;
  window_index(ix, m, tm, r0);
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
__global__ __launch_bounds__(256) void k_replay_windows(const int32_t* __restrict__ frames, int W, WindowIndex ix, int stack, int D,
                                                        E* __restrict__ out, long long out_row_elems, int slot_elems, E one, E zero) {
  const int m = blockIdx.x;
  int tm;
  long long r0;
  window_index(ix, m, tm, r0);
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

static int launch_windows(const char* who, const int32_t* frames, int packed_words, const WindowIndex& ix, int M, int stack, int D, void* out,
                          int64_t out_row_elems, int64_t slot_elems, int out_dtype, void* stream) {
  HZ_REQUIRE(M >= 0 && stack >= 1 && D >= 1 && packed_words == (D + 31) / 32, "%s: M=%d stack=%d D=%d packed_words=%d (want %d)", who, M, stack, D,
             packed_words, (D + 31) / 32);
  HZ_REQUIRE(slot_elems >= D && out_row_elems >= (int64_t)stack * slot_elems && slot_elems < (1 << 24),
             "%s: slots of %lld elements for %d-bit frames, rows of %lld for %d slots", who, (long long)slot_elems, D, (long long)out_row_elems, stack);
  HZ_REQUIRE(out_dtype == HZ_OBS_F32 || out_dtype == HZ_OBS_BF16 || out_dtype == HZ_OBS_F16, "%s: bad out_dtype %d", who, out_dtype);
  if (M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == HZ_OBS_F32) {
    hipLaunchKernelGGL(k_replay_windows<float>, dim3(M), dim3(256), 0, s, frames, packed_words, ix, stack, D, (float*)out, (long long)out_row_elems,
                       (int)slot_elems, 1.0f, 0.0f);
  } else {
    const uint16_t one = out_dtype == HZ_OBS_BF16 ? 0x3f80u : 0x3c00u;
    if (slot_elems % 8 == 0 && out_row_elems % 8 == 0 && ((uintptr_t)out % 16) == 0)
      hipLaunchKernelGGL(k_replay_windows16, dim3(M), dim3(256), 0, s, frames, packed_words, ix, stack, D, (uint16_t*)out, (long long)out_row_elems,
                         (int)slot_elems, (uint32_t)one);
    else
      hipLaunchKernelGGL(k_replay_windows<uint16_t>, dim3(M), dim3(256), 0, s, frames, packed_words, ix, stack, D, (uint16_t*)out,
                         (long long)out_row_elems, (int)slot_elems, one, (uint16_t)0);
  }
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_replay_windows(const int32_t* frames, int packed_words, const int64_t* row0, const int32_t* t, int M, int stack, int D,
                                 void* out, int64_t out_row_elems, int64_t slot_elems, int out_dtype, void* stream) {
  HZ_REQUIRE(frames && row0 && t && out, "hz_replay_windows: null pointer");
  WindowIndex ix = {};
  ix.row0 = row0;
  ix.t = t;
  return launch_windows("hz_replay_windows", frames, packed_words, ix, M, stack, D, out, out_row_elems, slot_elems, out_dtype, stream);
}

extern "C" int hz_replay_windows_seq(const int32_t* frames, int packed_words, const int64_t* pos_row0, const int32_t* pos_t, const int32_t* pos_T,
                                     const int64_t* phys, int B, int G, int shift0, int stack, int D, void* out, int64_t out_row_elems,
                                     int64_t slot_elems, int out_dtype, const uint8_t* legal, int num_actions, int64_t frame_rows,
                                     uint8_t* legal_out, uint8_t* valid_out, void* stream) {
  HZ_REQUIRE(frames && pos_row0 && pos_t && pos_T && phys && out, "hz_replay_windows_seq: null pointer");
  HZ_REQUIRE(B >= 0 && G >= 1 && shift0 >= 0 && (long long)B * G < (1ll << 31), "hz_replay_windows_seq: B=%d G=%d shift0=%d", B, G, shift0);
  HZ_REQUIRE(!legal_out || (legal && num_actions >= 1 && frame_rows >= 1), "hz_replay_windows_seq: legal_out without the legal rows");
  WindowIndex ix = {};
  ix.phys = phys; ix.pos_row0 = pos_row0; ix.pos_t = pos_t; ix.pos_T = pos_T; ix.G = G; ix.shift0 = shift0;
  ix.legal = legal; ix.legal_out = legal_out; ix.valid_out = valid_out; ix.A = num_actions; ix.frame_rows = frame_rows;
  return launch_windows("hz_replay_windows_seq", frames, packed_words, ix, B * G, stack, D, out, out_row_elems, slot_elems, out_dtype, stream);
}

// ---------------------------------------------------------------------------------------------------------------- a batch's targets
// One thread per (batch row, unroll position): index arithmetic and a few gathers; float64 where learner.make_batch computes in it.
__global__ __launch_bounds__(256) void k_replay_targets(const int64_t* __restrict__ phys, int B, int U, int td, int A, long long head,
                                                        const int32_t* __restrict__ pos_t, const int32_t* __restrict__ pos_T,
                                                        const int8_t* __restrict__ action, const int16_t* __restrict__ reward,
                                                        const int16_t* __restrict__ visits, const float* __restrict__ bootstrap,
                                                        const double* __restrict__ gpow, const int64_t* __restrict__ rand_actions,
                                                        int64_t* __restrict__ out_action, float* __restrict__ out_reward,
                                                        float* __restrict__ out_value, float* __restrict__ out_policy,
                                                        uint8_t* __restrict__ out_inside) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * (U + 1)) return;
  const int b = i / (U + 1), k = i - b * (U + 1);
  const long long p = phys[b];
  const int t = pos_t[p], T = pos_T[p];
  const bool inside = t + k < T;
  auto at = [&](int j) { const long long q = p + j; return q < head ? q : head - 1; };   // (clamped: never read past the live positions)
  auto rew = [&](int j) { return t + j < T ? (double)reward[at(j)] : 0.0; };
  if (k < U) {
    out_action[(long long)b * U + k] = inside ? (int64_t)action[at(k)] : rand_actions[(long long)b * U + k];
    out_reward[(long long)b * U + k] = (float)rew(k);
  }
  // td-step return: bootstrap * discount^td where a window td steps ahead exists, then the rewards in the reference's order i = 0 .. td - 1
  const bool reach = t + k + td < T;
  double v = (double)bootstrap[i] * gpow[td] * (reach ? 1.0 : 0.0);
  for (int j = 0; j < td; ++j) v = v + rew(k + j) * gpow[j];
  out_value[i] = inside ? (float)v : 0.0f;
  if (out_inside != nullptr) out_inside[i] = inside ? 1 : 0;
  const int16_t* c = visits + at(k) * A;
  double sum = 0.0;
  for (int a = 0; a < A; ++a) sum += (double)c[a];
  float* o = out_policy + (long long)i * A;
  for (int a = 0; a < A; ++a) o[a] = inside ? (float)((double)c[a] / sum) : 0.0f;
}

extern "C" int hz_replay_targets(const int64_t* phys, int B, int unroll_steps, int td_steps, int num_actions, int64_t head, const int32_t* pos_t,
                                 const int32_t* pos_T, const int8_t* action, const int16_t* reward, const int16_t* visits,
                                 const float* bootstrap, const double* discount_powers, const int64_t* rand_actions, int64_t* out_action,
                                 float* out_reward, float* out_value, float* out_policy, uint8_t* out_inside, void* stream) {
  HZ_REQUIRE(phys && pos_t && pos_T && action && reward && visits && bootstrap && discount_powers && rand_actions && out_action && out_reward &&
                 out_value && out_policy,
             "hz_replay_targets: null pointer");
  HZ_REQUIRE(B >= 0 && unroll_steps >= 1 && td_steps >= 0 && num_actions >= 1 && head >= 1 && (long long)B * (unroll_steps + 1) < (1ll << 31),
             "hz_replay_targets: B=%d unroll_steps=%d td_steps=%d num_actions=%d head=%lld", B, unroll_steps, td_steps, num_actions, (long long)head);
  if (B == 0) return 0;
  const int n = B * (unroll_steps + 1);
  hipLaunchKernelGGL(k_replay_targets, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, phys, B, unroll_steps, td_steps, num_actions,
                     (long long)head, pos_t, pos_T, action, reward, visits, bootstrap, discount_powers, rand_actions, out_action, out_reward, out_value,
                     out_policy, out_inside);
  HZ_HIP(hipGetLastError());
  return 0;
}
