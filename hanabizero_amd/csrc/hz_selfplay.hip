// hz_selfplay.hip -- batch form of the per-env Python glue of core/selfplay_worker.py:286-347 (gfx950).
#include "hz_selfplay_dev.h"

__global__ __launch_bounds__(256) void k_select_action(int N, int A, int32_t* __restrict__ counts,
                                                       const uint8_t* __restrict__ legal,
                                                       const double* __restrict__ uniform, float temperature,
                                                       int deterministic, int32_t* __restrict__ out_action,
                                                       double* __restrict__ out_entropy) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= N) return;
  double ent;
  out_action[env] = select_action_env(env, A, counts, legal, uniform, temperature, deterministic, &ent);
  if (out_entropy) out_entropy[env] = ent;
}

extern "C" int hz_select_action(int N, int A, int32_t* counts, const uint8_t* legal, const double* uniform,
                                float temperature, int deterministic, int32_t* out_action, double* out_entropy,
                                void* stream) {
  HZ_REQUIRE(N > 0 && A > 0 && A <= 64, "hz_select_action: bad sizes N=%d A=%d", N, A);
  HZ_REQUIRE(counts && legal && out_action, "hz_select_action: NULL argument");
  HZ_REQUIRE(deterministic || uniform, "hz_select_action: uniform samples required when sampling");
  HZ_REQUIRE(temperature > 0.0f, "hz_select_action: temperature must be > 0");
  hipLaunchKernelGGL(k_select_action, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, A, counts, legal,
                     uniform, temperature, deterministic, out_action, out_entropy);
  HZ_HIP(hipGetLastError());
  return 0;
}

// ---- finished-game flush: masked row scatter ------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rows_scatter(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                      long long row_bytes, const int32_t* __restrict__ slot, int n) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int s = slot[row];
  if (s < 0) return;
  const uint8_t* a = src + (size_t)row * (size_t)row_bytes;
  uint8_t* b = dst + (size_t)s * (size_t)row_bytes;
  if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)row_bytes) & 15) == 0) {
    for (long long off = (long long)lane * 16; off < row_bytes; off += 64 * 16)
      *reinterpret_cast<uint4*>(b + off) = *reinterpret_cast<const uint4*>(a + off);
  } else if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)row_bytes) & 3) == 0) {
    for (long long off = (long long)lane * 4; off < row_bytes; off += 64 * 4)
      *reinterpret_cast<uint32_t*>(b + off) = *reinterpret_cast<const uint32_t*>(a + off);
  } else {
    for (long long off = lane; off < row_bytes; off += 64) b[off] = a[off];
  }
}

extern "C" int hz_rows_scatter(const void* src, void* dst, int64_t row_bytes, const int32_t* slot, int num_rows,
                               void* stream) {
  HZ_REQUIRE(src && dst && slot, "hz_rows_scatter: NULL argument");
  HZ_REQUIRE(num_rows > 0 && row_bytes > 0, "hz_rows_scatter: num_rows and row_bytes must be positive");
  hipLaunchKernelGGL(k_rows_scatter, dim3((num_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src,
                     (uint8_t*)dst, (long long)row_bytes, slot, num_rows);
  HZ_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void k_actor_record_search(hz_actor_bufs_t b, int32_t* __restrict__ counts,
                                                             const float* __restrict__ root_values,
                                                             const uint8_t* __restrict__ legal,
                                                             const double* __restrict__ uniform, float temperature,
                                                             int deterministic, int32_t* __restrict__ out_action,
                                                             double* __restrict__ out_entropy) {
  const int lane = threadIdx.x & 63;
  const int env = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (env >= b.num_envs) return;
  const int A = b.num_actions, T = b.max_moves;
  double ent;
  int mc;
  const int action = select_action_wave(env, lane, A, counts, lane < A ? counts[(size_t)env * A + lane] : 0, lane < A ? (int)legal[(size_t)env * A + lane] : 0,
                                        deterministic ? 0.0 : uniform[env], temperature, deterministic, &ent, &mc);
  const int t = actor_t(b, env);
  if (lane < A) b.visits[((size_t)env * T + t) * A + lane] = (int16_t)mc;  // masked counts (store_search_stats gets the mutated list)
  if (lane == 0) {
    out_action[env] = action;
    if (out_entropy) out_entropy[env] = ent;
    b.action[(size_t)env * T + t] = (int8_t)action;
    b.value[(size_t)env * T + t] = root_values[env];
    b.ent_sum[env] += ent;
  }
}

// outbox slots of the games that just ended, in env order: one workgroup, ballot prefix counts
__device__ __forceinline__ void actor_slots_block(const hz_actor_bufs_t& b, const uint8_t* __restrict__ done) {
  __shared__ int wave_total[16];
  __shared__ long long base_s;
  __shared__ int moves_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    base_s = *b.out_count;
    moves_s = 0;
  }
  __syncthreads();
  long long base = base_s;
  for (int start = 0; start < b.num_envs; start += 1024) {
    const int env = start + tid;
    const bool d = env < b.num_envs && done[env] != 0;
    const uint64_t m = __ballot(d);
    const int before = __popcll((unsigned long long)(m & ((1ull << lane) - 1ull)));
    if (lane == 0) wave_total[wave] = __popcll((unsigned long long)m);
    __syncthreads();
    int off = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
      const int c = wave_total[w];
      if (w < wave) off += c;
      tot += c;
    }
    if (env < b.num_envs) b.slot[env] = d ? (int32_t)((base + off + before) % (long long)b.outbox_games) : -1;
    if (d) {
      b.finished[(int)(base - base_s) + off + before] = env;
      atomicAdd(&moves_s, actor_t(b, env) + 1);  // its length (what actor_record_step_env writes into the meta row)
    }
    base += tot;
    __syncthreads();
  }
  if (tid == 0) {
    b.out_count[0] = base;
    b.out_count[1] += moves_s;
    *b.num_finished = (int)(base - base_s);
  }
}

// hz_actor_record_step in one launch: workgroups [0, gridDim.x - 1) record 256 envs each (four threads per env), the last
// one computes the outbox slots (it needs `done` and the trajectory lengths only, nothing the other workgroups write)
__global__ __launch_bounds__(1024) void k_actor_record_step_slots(hz_actor_bufs_t b, const int32_t* __restrict__ reward,
                                                                  const uint8_t* __restrict__ done,
                                                                  const int32_t* __restrict__ score,
                                                                  const int32_t* __restrict__ status,
                                                                  const int32_t* __restrict__ packed,
                                                                  const uint8_t* __restrict__ legal_next) {
  if (blockIdx.x + 1 == gridDim.x) actor_slots_block(b, done);
  else actor_record_step_env(b, blockIdx.x * 256 + (threadIdx.x >> 2), threadIdx.x & 3, reward, score, status, packed, legal_next);
}

struct FlushTable {
  const uint8_t* src[7];
  uint8_t* dst[7];
  long long row_bytes[7];
};

// blockIdx.y = array; the workgroups of one array walk the list of finished envs, one whole workgroup per row
__global__ __launch_bounds__(256) void k_actor_flush(FlushTable ft, const int32_t* __restrict__ slot,
                                                     const int32_t* __restrict__ finished,
                                                     const int32_t* __restrict__ num_finished) {
  const int k = blockIdx.y;
  const long long rb = ft.row_bytes[k];
  const int n = *num_finished;
  for (int j = blockIdx.x; j < n; j += gridDim.x) {
    const int row = finished[j];
    const int s = slot[row];
    if (s < 0) continue;
    copy_row_block(ft.src[k] + (size_t)row * (size_t)rb, ft.dst[k] + (size_t)s * (size_t)rb, rb, threadIdx.x, blockDim.x);
  }
}

// Outbox ring rows [first, first + n) -> one packed byte buffer of seven sections, the games back to back (ragged: game j
// owns len_j rows of the per-move sections and len_j + 1 rows of the observation / legal-mask sections).
// k_actor_pack_starts: exclusive prefix sums of the games' lengths, one workgroup.
__global__ __launch_bounds__(1024) void k_actor_pack_starts(const int32_t* __restrict__ out_meta, long long first, int n, int cap,
                                                           int32_t* __restrict__ starts) {
  __shared__ int wave_total[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int j0 = 0; j0 < n; j0 += 1024) {
    const int j = j0 + tid;
    const int len = j < n ? out_meta[(size_t)((first + j) % (long long)cap) * 4] : 0;
    int incl = len;  // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wave_total[w];
    if (j < n) starts[j] = off + incl - len;
    __syncthreads();
    if (tid == 1023) base_s = off + incl;
    __syncthreads();
  }
  if (tid == 0) starts[n] = base_s;  // total moves
}

struct PackTable {
  const uint8_t* src[7];
  long long src_row_bytes[7];  // a ring row
  long long unit_bytes[7];     // bytes per move (per game for the meta section)
  long long dst_off[7];        // section start in the packed buffer
  int extra[7];                // rows beyond the game's length (1 for observations and legal masks)
};

__global__ __launch_bounds__(256) void k_actor_pack(PackTable pt, const int32_t* __restrict__ out_meta,
                                                    const int32_t* __restrict__ starts, long long first, int n, int cap,
                                                    int moves, uint8_t* __restrict__ out) {
  const int k = blockIdx.y;
  for (int j = blockIdx.x; j < n; j += gridDim.x) {
    const long long row = (first + j) % (long long)cap;
    const int len = out_meta[(size_t)row * 4], st = starts[j];
    if (len < 0 || st < 0 || st + len > moves) continue;  // (the caller's total does not cover this game: never the case)
    const long long rows_before = k == 0 ? j : (long long)st + (pt.extra[k] ? j : 0);
    const long long nrows = k == 0 ? 1 : len + pt.extra[k];
    copy_row_block(pt.src[k] + (size_t)row * (size_t)pt.src_row_bytes[k],
                   out + pt.dst_off[k] + (size_t)rows_before * (size_t)pt.unit_bytes[k], nrows * pt.unit_bytes[k], threadIdx.x,
                   blockDim.x);
  }
}

template <typename U>
__global__ __launch_bounds__(256) void k_actor_begin_move(hz_actor_bufs_t b, const uint8_t* __restrict__ done,
                                                          const int32_t* __restrict__ packed,
                                                          const uint8_t* __restrict__ legal,
                                                          const uint8_t* __restrict__ newest, long long newest_row_bytes,
                                                          uint8_t* __restrict__ stack_buf, long long stack_row_bytes,
                                                          int stack, long long obs_bytes) {
  const int lane = threadIdx.x & 63;
  const int env = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (env >= b.num_envs) return;
  actor_begin_move_wave<U>(b, env, lane, done, packed, legal, newest, newest_row_bytes, stack_buf, stack_row_bytes, stack, obs_bytes);
}

#define HZ_ACTOR_CHECK(b, who)                                                                                      \
  HZ_REQUIRE((b) != nullptr, who ": bufs is NULL");                                                                 \
  HZ_REQUIRE((b)->num_envs > 0 && (b)->num_actions > 0 && (b)->num_actions <= 64 && (b)->packed_words > 0 &&        \
                 (b)->max_moves > 0 && (b)->outbox_games > 0,                                                       \
             who ": bad sizes N=%d A=%d W=%d T=%d cap=%d", (b)->num_envs, (b)->num_actions, (b)->packed_words,      \
             (b)->max_moves, (b)->outbox_games);                                                                    \
  HZ_REQUIRE((b)->action && (b)->reward && (b)->value && (b)->visits && (b)->legal && (b)->obs && (b)->traj_len &&  \
                 (b)->ent_sum && (b)->meta && (b)->out_count && (b)->slot && (b)->finished && (b)->num_finished &&        \
                 (b)->illegal_steps,                    \
             who ": NULL buffer in bufs")

extern "C" int hz_actor_record_search(const hz_actor_bufs_t* bufs, int32_t* counts, const float* root_values,
                                      const uint8_t* legal, const double* uniform, float temperature, int deterministic,
                                      int32_t* out_action, double* out_entropy, void* stream) {
  HZ_ACTOR_CHECK(bufs, "hz_actor_record_search");
  HZ_REQUIRE(counts && root_values && legal && out_action, "hz_actor_record_search: NULL argument");
  HZ_REQUIRE(deterministic || uniform, "hz_actor_record_search: uniform samples required when sampling");
  HZ_REQUIRE(temperature > 0.0f, "hz_actor_record_search: temperature must be > 0");
  hipLaunchKernelGGL(k_actor_record_search, dim3((bufs->num_envs + 3) / 4), dim3(256), 0, (hipStream_t)stream, *bufs,
                     counts, root_values, legal, uniform, temperature, deterministic, out_action, out_entropy);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_actor_record_step(const hz_actor_bufs_t* bufs, const int32_t* reward, const uint8_t* done,
                                    const int32_t* score, const int32_t* status, const int32_t* packed,
                                    const uint8_t* legal_next, void* stream) {
  HZ_ACTOR_CHECK(bufs, "hz_actor_record_step");
  HZ_REQUIRE(reward && done && score && status && packed && legal_next, "hz_actor_record_step: NULL argument");
  hipLaunchKernelGGL(k_actor_record_step_slots, dim3((bufs->num_envs + 255) / 256 + 1), dim3(1024), 0, (hipStream_t)stream,
                     *bufs, reward, done, score, status, packed, legal_next);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_actor_flush(const hz_actor_bufs_t* bufs, void* stream) {
  HZ_ACTOR_CHECK(bufs, "hz_actor_flush");
  HZ_REQUIRE(bufs->out_action && bufs->out_reward && bufs->out_value && bufs->out_visits && bufs->out_legal &&
                 bufs->out_obs && bufs->out_meta, "hz_actor_flush: NULL outbox buffer");
  const long long T = bufs->max_moves, A = bufs->num_actions, W = bufs->packed_words;
  FlushTable ft;
  const void* src[7] = {bufs->action, bufs->reward, bufs->value, bufs->visits, bufs->legal, bufs->obs, bufs->meta};
  void* dst[7] = {bufs->out_action, bufs->out_reward, bufs->out_value, bufs->out_visits, bufs->out_legal, bufs->out_obs,
                  bufs->out_meta};
  const long long rb[7] = {T, T, 4 * T, 2 * T * A, (T + 1) * A, 4 * (T + 1) * W, 16};
  for (int k = 0; k < 7; ++k) {
    ft.src[k] = (const uint8_t*)src[k];
    ft.dst[k] = (uint8_t*)dst[k];
    ft.row_bytes[k] = rb[k];
  }
  const int gx = bufs->num_envs < 256 ? bufs->num_envs : 256;
  hipLaunchKernelGGL(k_actor_flush, dim3(gx, 7), dim3(256), 0, (hipStream_t)stream, ft, bufs->slot, bufs->finished,
                     bufs->num_finished);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_actor_flush_job(const hz_actor_bufs_t* bufs, hz_rows_job_t* job) {
  HZ_ACTOR_CHECK(bufs, "hz_actor_flush_job");
  HZ_REQUIRE(job != nullptr, "hz_actor_flush_job: NULL job");
  HZ_REQUIRE(bufs->out_action && bufs->out_reward && bufs->out_value && bufs->out_visits && bufs->out_legal &&
                 bufs->out_obs && bufs->out_meta, "hz_actor_flush_job: NULL outbox array");
  const long long A = bufs->num_actions, W = bufs->packed_words, T = bufs->max_moves;
  const void* src[7] = {bufs->action, bufs->reward, bufs->value, bufs->visits, bufs->legal, bufs->obs, bufs->meta};
  void* dst[7] = {bufs->out_action, bufs->out_reward, bufs->out_value, bufs->out_visits, bufs->out_legal, bufs->out_obs,
                  bufs->out_meta};
  const long long rb[7] = {T, T, 4 * T, 2 * T * A, (T + 1) * A, 4 * (T + 1) * W, 16};
  job->slot = bufs->slot; job->list = bufs->finished; job->count = bufs->num_finished;
  job->num_arrays = 7; job->max_rows = bufs->num_envs;
  for (int k = 0; k < 8; ++k) {
    job->src[k] = k < 7 ? src[k] : nullptr;
    job->dst[k] = k < 7 ? dst[k] : nullptr;
    job->row_bytes[k] = k < 7 ? rb[k] : 0;
  }
  return 0;
}

extern "C" int64_t hz_actor_packed_bytes(int n, int64_t moves, int num_actions, int packed_words, int64_t* offsets) {
  if (n < 0 || moves < 0 || num_actions <= 0 || packed_words <= 0) return -1;
  const long long A = num_actions, W = packed_words, m = moves, g = n;
  // meta action reward value visits legal obs
  const long long bytes[7] = {16 * g, m, m, 4 * m, 2 * m * A, (m + g) * A, 4 * (m + g) * W};
  long long off = 0;
  for (int k = 0; k < 7; ++k) {
    if (offsets) offsets[k] = off;
    off += (bytes[k] + 15) / 16 * 16;
  }
  return off;
}

extern "C" int hz_actor_pack(const hz_actor_bufs_t* bufs, int64_t first, int n, int64_t moves, int32_t* starts, void* out,
                             int64_t out_bytes, void* stream) {
  HZ_ACTOR_CHECK(bufs, "hz_actor_pack");
  HZ_REQUIRE(bufs->out_action && bufs->out_reward && bufs->out_value && bufs->out_visits && bufs->out_legal &&
                 bufs->out_obs && bufs->out_meta && out && starts, "hz_actor_pack: NULL outbox array");
  HZ_REQUIRE(first >= 0 && n >= 1 && n <= bufs->outbox_games && moves >= n && moves <= (int64_t)n * bufs->max_moves,
             "hz_actor_pack: first=%lld n=%d moves=%lld outside the outbox (capacity %d, %d moves per game)",
             (long long)first, n, (long long)moves, bufs->outbox_games, bufs->max_moves);
  int64_t offs[7];
  const int64_t total = hz_actor_packed_bytes(n, moves, bufs->num_actions, bufs->packed_words, offs);
  HZ_REQUIRE(out_bytes >= total, "hz_actor_pack: %lld B buffer, %lld B needed", (long long)out_bytes, (long long)total);
  const long long A = bufs->num_actions, W = bufs->packed_words, T = bufs->max_moves;
  PackTable pt;
  const void* src[7] = {bufs->out_meta, bufs->out_action, bufs->out_reward, bufs->out_value, bufs->out_visits, bufs->out_legal,
                        bufs->out_obs};
  const long long srb[7] = {16, T, T, 4 * T, 2 * T * A, (T + 1) * A, 4 * (T + 1) * W};
  const long long unit[7] = {16, 1, 1, 4, 2 * A, A, 4 * W};
  for (int k = 0; k < 7; ++k) {
    pt.src[k] = (const uint8_t*)src[k];
    pt.src_row_bytes[k] = srb[k];
    pt.unit_bytes[k] = unit[k];
    pt.dst_off[k] = offs[k];
    pt.extra[k] = k >= 5;
  }
  hipLaunchKernelGGL(k_actor_pack_starts, dim3(1), dim3(1024), 0, (hipStream_t)stream, bufs->out_meta, (long long)first, n,
                     bufs->outbox_games, starts);
  // (the alignment padding between sections is never read by unpack; it keeps whatever the buffer held)
  hipLaunchKernelGGL(k_actor_pack, dim3(n < 8192 ? n : 8192, 7), dim3(256), 0, (hipStream_t)stream, pt, bufs->out_meta, starts,
                     (long long)first, n, bufs->outbox_games, (int)moves, (uint8_t*)out);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_actor_begin_move(const hz_actor_bufs_t* bufs, const uint8_t* done, const int32_t* packed,
                                   const uint8_t* legal, const void* newest, int64_t newest_row_bytes, void* stack_buf,
                                   int64_t stack_row_bytes, int stack, int64_t obs_bytes, void* stream) {
  HZ_ACTOR_CHECK(bufs, "hz_actor_begin_move");
  HZ_REQUIRE(done && packed && legal && newest && stack_buf, "hz_actor_begin_move: NULL argument");
  HZ_REQUIRE(stack >= 1 && obs_bytes > 0 && stack_row_bytes >= (int64_t)stack * obs_bytes && newest_row_bytes >= obs_bytes,
             "hz_actor_begin_move: bad window geometry (stack=%d obs_bytes=%lld)", stack, (long long)obs_bytes);
  const dim3 grid((bufs->num_envs + 3) / 4), block(256);
  const uintptr_t al = (uintptr_t)newest | (uintptr_t)stack_buf | (uintptr_t)newest_row_bytes | (uintptr_t)stack_row_bytes |
                       (uintptr_t)obs_bytes;
#define HZ_BEGIN_MOVE(U)                                                                                          \
  hipLaunchKernelGGL(k_actor_begin_move<U>, grid, block, 0, (hipStream_t)stream, *bufs, done, packed, legal,      \
                     (const uint8_t*)newest, (long long)newest_row_bytes, (uint8_t*)stack_buf,                    \
                     (long long)stack_row_bytes, stack, (long long)obs_bytes)
  if ((al & 15) == 0) HZ_BEGIN_MOVE(uint4);  // slot-padded windows (InferenceEngine.pad_observations)
  else if ((al & 3) == 0) HZ_BEGIN_MOVE(uint32_t);
  else if ((al & 1) == 0) HZ_BEGIN_MOVE(uint16_t);  // bf16 observations of odd width
  else HZ_BEGIN_MOVE(uint8_t);
#undef HZ_BEGIN_MOVE
  HZ_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void k_actor_draw(uint64_t seed, long long env_id_base, long long* __restrict__ move_count,
                                                    int N, int A, double alpha, float* __restrict__ noise,
                                                    double* __restrict__ uniform) {
  __shared__ double g_s[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int env = blockIdx.x * 4 + wave;
  if (env >= N) return;
  actor_draw_wave(seed, env_id_base, move_count, env, lane, A, alpha, noise, uniform, g_s[wave]);
}

// hz_actor_begin_move and the NEXT move's hz_actor_draw in one launch: waves 0-3 of a workgroup move the windows of four
// envs while waves 4-7 draw for the same four (the two are independent and each is bound by one wave's latency).
template <typename U>
__global__ __launch_bounds__(512) void k_actor_begin_move_draw(hz_actor_bufs_t b, const uint8_t* __restrict__ done,
                                                               const int32_t* __restrict__ packed,
                                                               const uint8_t* __restrict__ legal,
                                                               const uint8_t* __restrict__ newest, long long newest_row_bytes,
                                                               uint8_t* __restrict__ stack_buf, long long stack_row_bytes,
                                                               int stack, long long obs_bytes, uint64_t seed,
                                                               long long* __restrict__ move_count, double alpha,
                                                               float* __restrict__ noise, double* __restrict__ uniform) {
  __shared__ double g_s[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int env = blockIdx.x * 4 + (wave & 3);
  if (env >= b.num_envs) return;
  if (wave < 4)
    actor_begin_move_wave<U>(b, env, lane, done, packed, legal, newest, newest_row_bytes, stack_buf, stack_row_bytes, stack,
                             obs_bytes);
  else
    actor_draw_wave(seed, (long long)b.env_id_base, move_count, env, lane, b.num_actions, alpha, noise, uniform, g_s[wave & 3]);
}

extern "C" int hz_actor_draw(uint64_t seed, int64_t env_id_base, int64_t* move_count, int num_envs, int num_actions,
                             double alpha, float* noise, double* uniform, void* stream) {
  HZ_REQUIRE(num_envs > 0 && num_actions > 0 && num_actions <= 64, "hz_actor_draw: bad sizes N=%d A=%d", num_envs,
             num_actions);
  HZ_REQUIRE(move_count && noise && uniform, "hz_actor_draw: NULL argument");
  HZ_REQUIRE(alpha > 0.0, "hz_actor_draw: alpha must be > 0");
  hipLaunchKernelGGL(k_actor_draw, dim3((num_envs + 3) / 4), dim3(256), 0, (hipStream_t)stream, seed,
                     (long long)env_id_base, (long long*)move_count, num_envs, num_actions, alpha, noise, uniform);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_actor_begin_move_draw(const hz_actor_bufs_t* bufs, const uint8_t* done, const int32_t* packed,
                                        const uint8_t* legal, const void* newest, int64_t newest_row_bytes, void* stack_buf,
                                        int64_t stack_row_bytes, int stack, int64_t obs_bytes, uint64_t seed,
                                        int64_t* move_count, double alpha, float* noise, double* uniform, void* stream) {
  HZ_ACTOR_CHECK(bufs, "hz_actor_begin_move_draw");
  HZ_REQUIRE(done && packed && legal && newest && stack_buf && move_count && noise && uniform,
             "hz_actor_begin_move_draw: NULL argument");
  HZ_REQUIRE(stack >= 1 && obs_bytes > 0 && stack_row_bytes >= (int64_t)stack * obs_bytes && newest_row_bytes >= obs_bytes,
             "hz_actor_begin_move_draw: bad window geometry (stack=%d obs_bytes=%lld)", stack, (long long)obs_bytes);
  HZ_REQUIRE(alpha > 0.0, "hz_actor_begin_move_draw: alpha must be > 0");
  const dim3 grid((bufs->num_envs + 3) / 4), block(512);
  const uintptr_t al = (uintptr_t)newest | (uintptr_t)stack_buf | (uintptr_t)newest_row_bytes | (uintptr_t)stack_row_bytes |
                       (uintptr_t)obs_bytes;
#define HZ_BEGIN_MOVE_DRAW(U)                                                                                       \
  hipLaunchKernelGGL(k_actor_begin_move_draw<U>, grid, block, 0, (hipStream_t)stream, *bufs, done, packed, legal,   \
                     (const uint8_t*)newest, (long long)newest_row_bytes, (uint8_t*)stack_buf,                      \
                     (long long)stack_row_bytes, stack, (long long)obs_bytes, seed, (long long*)move_count, alpha,  \
                     noise, uniform)
  if ((al & 15) == 0) HZ_BEGIN_MOVE_DRAW(uint4);
  else if ((al & 3) == 0) HZ_BEGIN_MOVE_DRAW(uint32_t);
  else if ((al & 1) == 0) HZ_BEGIN_MOVE_DRAW(uint16_t);
  else HZ_BEGIN_MOVE_DRAW(uint8_t);
#undef HZ_BEGIN_MOVE_DRAW
  HZ_HIP(hipGetLastError());
  return 0;
}
