// hz_env.hip -- N independent Hanabi games, bit-packed in HBM, advanced by hand-written HIP kernels (gfx950).
//
// What it replaces (reference, /root/reference): envs/hanabi/hanabi_lib/{hanabi_game,hanabi_state,hanabi_hand,
// hanabi_observation,canonical_encoders}.cc behind envs/hanabi/pyhanabi.{h,cc}, as driven by
// envs/hanabi/rl_env.py HanabiEnv.reset/step.  Written from scratch; no code shared with the reference or oracle/.
//
// HBM layout (per hz_env_t, N envs):
//   state [N][32] u32   128 B per env, one cache line:
//       w0..w24  card slot p*H+i : color[0:3) rank[3:6) color_plausible[6:11) rank_plausible[11:16)
//                                  color_hinted[16] rank_hinted[17]
//       w25,w26  deck counts   2 bits per (color*R+rank), 16 per word
//       w27,w28  discard counts 2 bits per (color*R+rank)
//       w29      fireworks 5x3 [0:15) info[15:19) life[19:21) cur_player[21:24) next_player[24:27) turns_to_play[27:30)
//       w30      hand sizes 5x3 [0:15) deck_total[15:21) has_last_move[21]
//       w31      last non-deal move: player[0:3) type[3:6) card_index[6:9) target_offset[9:12) color[12:15)
//                rank[15:18) scored[18] info_token[19] card_color[20:23) card_rank[23:26) reveal_mask[26:31)
//   mt    [625][N] u32  std::mt19937 state of every env (row 624 = position), struct-of-arrays so that the
//                       one-lane-per-env rules kernel reads/writes it coalesced
// Kernels: k_env_rules (one lane per env: reset or apply move, deal loop, reward/done/score),
//          k_env_observe (one wave per env: sets observation bits in LDS with ds_or, then streams the row out
//          coalesced as u8/f32/bf16/f16 and/or bit-packed; lane a also evaluates legal move a).
#include "hz_env_dev.h"

// ---- rules kernel: one lane per env ------------------------------------------------------------------
// mode 0: reset (rl_env.py:249-252)   mode 1: step (rl_env.py:418-442)
#define RULES_THREADS 64
__global__ __launch_bounds__(RULES_THREADS) void k_env_rules(EnvCfg g, uint32_t* __restrict__ state,
                                                             uint32_t* __restrict__ mt, int mode,
                                                             const int32_t* __restrict__ actions,
                                                             const uint8_t* __restrict__ mask,
                                                             int32_t* __restrict__ reward, uint8_t* __restrict__ done,
                                                             int32_t* __restrict__ score_out,
                                                             int32_t* __restrict__ status) {
  __shared__ uint32_t lds[32 * RULES_THREADS];
  __shared__ uint32_t lds_rng[(2 * MT_MAX_DRAWS + 1) * RULES_THREADS];
  const int env = blockIdx.x * RULES_THREADS + threadIdx.x;
  if (env >= g.N) return;
  if (mask != nullptr && mask[env] == 0) return;
  MtBatch rng;
  rng.mt = mt; rng.N = g.N; rng.env = env; rng.stride = RULES_THREADS;
  rng.w = lds_rng + threadIdx.x;
  rng.far = lds_rng + (MT_MAX_DRAWS + 1) * RULES_THREADS + threadIdx.x;
  mt_batch_begin(rng, mode == 0 ? 2 * g.P * g.H : 2);  // a reset deals every hand, a move at most one card
  St s;
  s.p = lds + threadIdx.x;
  s.stride = RULES_THREADS;
  uint4* gs = reinterpret_cast<uint4*>(state + (size_t)env * 32);
  int cur;
  if (mode == 0) {
    // HanabiState ctor (hanabi_state.cc:90-102) + HanabiDeck ctor (:53-64)
    for (int w = 0; w < 32; ++w) s.word(w) = 0;
    int total = 0;
    for (int c = 0; c < g.C; ++c)
      for (int r = 0; r < g.R; ++r) {
        s.set_deck(c * g.R + r, g.inst[r]);
        total += g.inst[r];
      }
    s.set_deck_total(total);
    s.set_info(g.max_info);
    s.set_life(g.max_life);
    s.set_next(0);  // GetSampledStartPlayer, random_start_player = false (hanabi_game.cc:138-145)
    s.set_turns(g.P);
    cur = -1;
    while (cur == -1) deal_random(g, s, cur, rng);
    s.set_cur(cur);
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint4 v = gs[q];
      s.word(4 * q + 0) = v.x; s.word(4 * q + 1) = v.y; s.word(4 * q + 2) = v.z; s.word(4 * q + 3) = v.w;
    }
    int rw, dn, sc;
    const int st = env_step_lane(g, s, rng, actions[env], rw, dn, sc);
    status[env] = st;
    reward[env] = rw;
    done[env] = (uint8_t)dn;
    score_out[env] = sc;
    if (st != HZ_ENV_OK) return;
  }
  mt_batch_commit(rng);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    uint4 v;
    v.x = s.word(4 * q + 0); v.y = s.word(4 * q + 1); v.z = s.word(4 * q + 2); v.w = s.word(4 * q + 3);
    gs[q] = v;
  }
}

__global__ __launch_bounds__(256) void k_env_reset_wave(EnvCfg g, uint32_t* __restrict__ state, uint32_t* __restrict__ mt,
                                                        const uint8_t* __restrict__ mask) {
  __shared__ uint32_t st_s[4][32];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  env_reset_wave_body(g, state, mt, mask, blockIdx.x * 4 + wave, lane, st_s[wave]);
}

// the reset plus an independent row scatter (include/hz_rows.h) in one launch: workgroups [0, reset_blocks) reset, the rest
// walk the job's row list, `per_array` workgroups per array (one whole workgroup per row)
__global__ __launch_bounds__(256) void k_env_reset_rows(EnvCfg g, uint32_t* __restrict__ state, uint32_t* __restrict__ mt,
                                                        const uint8_t* __restrict__ mask, hz_rows_job_t job, int reset_blocks,
                                                        int per_array) {
  __shared__ uint32_t st_s[4][32];
  if ((int)blockIdx.x < reset_blocks) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    env_reset_wave_body(g, state, mt, mask, blockIdx.x * 4 + wave, lane, st_s[wave]);
    return;
  }
  const int b = (int)blockIdx.x - reset_blocks;
  const int k = b / per_array;
  const long long rb = job.row_bytes[k];
  const int n = *job.count;
  for (int j = b % per_array; j < n; j += per_array) {
    const int row = job.list[j];
    const int sl = job.slot[row];
    if (sl < 0) continue;
    copy_row_block((const uint8_t*)job.src[k] + (size_t)row * (size_t)rb, (uint8_t*)job.dst[k] + (size_t)sl * (size_t)rb, rb,
                   threadIdx.x, blockDim.x);
  }
}

__global__ __launch_bounds__(256) void k_env_observe(EnvCfg g, const uint32_t* __restrict__ state, int mdp,
                                                     void* __restrict__ obs_out, int dtype, long long stride,
                                                     uint32_t* __restrict__ packed_out,
                                                     uint8_t* __restrict__ legal_out) {
  __shared__ uint32_t s_state[4][32];
  __shared__ uint32_t s_bits[4][OBS_WORDS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int env = blockIdx.x * 4 + wave;
  if (env >= g.N) return;
  uint32_t* bits = s_bits[wave];
  if (lane < 32) s_state[wave][lane] = state[(size_t)env * 32 + lane];
  if (lane < OBS_WORDS) bits[lane] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  St s;
  s.p = s_state[wave];
  s.stride = 1;
  env_observe_bits(g, s, bits, lane, mdp);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  env_observe_emit(g, s, bits, lane, env, mdp, obs_out, dtype, stride, packed_out, legal_out);
}

__global__ void k_env_probe(EnvCfg g, const uint32_t* __restrict__ state, int32_t* __restrict__ out) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= g.N) return;
  St s;
  s.p = const_cast<uint32_t*>(state) + (size_t)env * 32;
  s.stride = 1;
  int32_t* o = out + (size_t)env * HZ_ENV_PROBE_FIELDS;
  o[0] = s.cur(); o[1] = s.deck_total(); o[2] = s.info(); o[3] = s.life();
  for (int c = 0; c < 5; ++c) o[4 + c] = c < g.C ? s.fw(c) : 0;
  for (int p = 0; p < 5; ++p) o[9 + p] = p < g.P ? s.hand_n(p) : 0;
  o[14] = env_end_status(g, s);
  o[15] = env_score(g, s);
}

// ---- C ABI -----------------------------------------------------------------------------------------------
extern "C" int hz_env_create(hz_env_t** out, int N, int colors, int ranks, int players, int hand_size, int max_info,
                             int max_life, const int32_t* host_seeds, int device) {
  HZ_REQUIRE(out != nullptr && host_seeds != nullptr, "hz_env_create: NULL argument");
  HZ_REQUIRE(N > 0, "hz_env_create: num_envs must be > 0");
  HZ_REQUIRE(colors >= 1 && colors <= 5 && ranks >= 1 && ranks <= 5, "hz_env_create: colors and ranks must be in [1,5]");
  HZ_REQUIRE(players >= 2 && players <= 5, "hz_env_create: players must be in [2,5] (hanabi_game.cc:33)");
  EnvCfg g;
  memset(&g, 0, sizeof(g));
  g.N = N; g.C = colors; g.R = ranks; g.P = players;
  g.H = hand_size > 0 ? hand_size : (players < 4 ? 5 : 4);
  HZ_REQUIRE(g.H <= 5 && g.P * g.H <= 25, "hz_env_create: players*hand_size must be <= 25 and hand_size <= 5");
  HZ_REQUIRE(max_info >= 0 && max_info <= 15 && max_life >= 1 && max_life <= 3,
             "hz_env_create: max_information_tokens in [0,15], max_life_tokens in [1,3]");
  g.max_info = max_info; g.max_life = max_life;
  g.bpc = colors * ranks;
  g.per_color = 0;
  for (int r = 0; r < ranks; ++r) {
    g.inst[r] = (r == 0) ? 3 : (r == ranks - 1 ? 1 : 2);
    g.inst_prefix[r] = g.per_color;
    g.per_color += g.inst[r];
  }
  g.max_deck = g.per_color * colors;
  HZ_REQUIRE(g.H * g.P <= g.max_deck, "hz_env_create: deck too small for the hands (hanabi_game.cc:58)");
  g.num_moves = 2 * g.H + (players - 1) * colors + (players - 1) * ranks;
  HZ_REQUIRE(g.num_moves <= 64, "hz_env_create: more than 64 moves");
  const int hands = (players - 1) * g.H * g.bpc + players;
  const int board = g.max_deck - players * g.H + colors * ranks + max_info + max_life;
  const int last = players + 4 + players + colors + ranks + g.H + g.H + g.bpc + 2;
  const int know = players * g.H * (g.bpc + colors + ranks);
  g.off_board = hands;
  g.off_disc = hands + board;
  g.off_last = g.off_disc + g.max_deck;
  g.off_know = g.off_last + last;
  g.obs_len = g.off_know + know;
  g.own_len = g.H * g.bpc;
  HZ_REQUIRE(g.own_len + g.obs_len + players <= 32 * (OBS_WORDS - 1), "hz_env_create: observation too long");
  HZ_HIP(hipSetDevice(device));
  hz_env* e = new hz_env();
  e->cfg = g; e->device = device; e->bytes = 0; e->state = nullptr; e->mt = nullptr;
  hipError_t err = hipMalloc((void**)&e->state, (size_t)N * 32 * sizeof(uint32_t));
  if (err == hipSuccess) err = hipMalloc((void**)&e->mt, (size_t)N * 625 * sizeof(uint32_t));
  if (err != hipSuccess) {
    hz_env_destroy(e);
    hz_set_error("hz_env_create: hipMalloc failed: %s", hipGetErrorString(err));
    return -2;
  }
  e->bytes = (int64_t)N * (32 + 625) * 4;
  // mersenne_twister_engine::seed(value) (libstdc++ random.tcc), transposed to [625][N]
  uint32_t* host = new uint32_t[(size_t)N * 625];
  for (int i = 0; i < N; ++i) {
    uint32_t x = (uint32_t)host_seeds[i];
    host[(size_t)0 * N + i] = x;
    for (int k = 1; k < 624; ++k) {
      x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)k;
      host[(size_t)k * N + i] = x;
    }
    host[(size_t)624 * N + i] = 624u;
  }
  err = hipMemcpy(e->mt, host, (size_t)N * 625 * sizeof(uint32_t), hipMemcpyHostToDevice);
  delete[] host;
  if (err == hipSuccess) err = hipMemset(e->state, 0, (size_t)N * 32 * sizeof(uint32_t));
  if (err != hipSuccess) {
    hz_env_destroy(e);
    hz_set_error("hz_env_create: upload failed: %s", hipGetErrorString(err));
    return -2;
  }
  *out = e;
  return 0;
}

extern "C" int hz_env_destroy(hz_env_t* e) {
  if (!e) return 0;
  (void)hipSetDevice(e->device);
  (void)hipFree(e->state);
  (void)hipFree(e->mt);
  delete e;
  return 0;
}

extern "C" int hz_env_dims(const hz_env_t* e, int* num_moves, int* obs_len, int* own_len, int* players) {
  HZ_REQUIRE(e != nullptr, "hz_env_dims: NULL handle");
  if (num_moves) *num_moves = e->cfg.num_moves;
  if (obs_len) *obs_len = e->cfg.obs_len;
  if (own_len) *own_len = e->cfg.own_len;
  if (players) *players = e->cfg.P;
  return 0;
}

extern "C" int hz_env_reset(hz_env_t* e, const uint8_t* mask, void* stream) {
  HZ_REQUIRE(e != nullptr, "hz_env_reset: NULL handle");
  // one wave per env (envs not in the mask leave at once); k_env_rules mode 0 is the lane-per-env form of the same reset
  hipLaunchKernelGGL(k_env_reset_wave, dim3((e->cfg.N + 3) / 4), dim3(256), 0, (hipStream_t)stream, e->cfg, e->state,
                     e->mt, mask);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_env_reset_rows(hz_env_t* e, const uint8_t* mask, const hz_rows_job_t* rows, void* stream) {
  HZ_REQUIRE(e != nullptr && rows != nullptr, "hz_env_reset_rows: NULL argument");
  HZ_REQUIRE(rows->num_arrays >= 1 && rows->num_arrays <= 8 && rows->max_rows >= 1 && rows->slot && rows->list && rows->count,
             "hz_env_reset_rows: malformed rows job (%d arrays, %d rows)", rows->num_arrays, rows->max_rows);
  for (int k = 0; k < rows->num_arrays; ++k)
    HZ_REQUIRE(rows->src[k] && rows->dst[k] && rows->row_bytes[k] > 0, "hz_env_reset_rows: array %d of the rows job is empty", k);
  const int reset_blocks = (e->cfg.N + 3) / 4;
  const int per_array = rows->max_rows < 256 ? rows->max_rows : 256;
  hipLaunchKernelGGL(k_env_reset_rows, dim3(reset_blocks + per_array * rows->num_arrays), dim3(256), 0, (hipStream_t)stream,
                     e->cfg, e->state, e->mt, mask, *rows, reset_blocks, per_array);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_env_step(hz_env_t* e, const int32_t* actions, const uint8_t* mask, int32_t* reward, uint8_t* done,
                           int32_t* score, int32_t* status, void* stream) {
  HZ_REQUIRE(e != nullptr, "hz_env_step: NULL handle");
  HZ_REQUIRE(actions && reward && done && score && status, "hz_env_step: NULL argument");
  const int blocks = (e->cfg.N + RULES_THREADS - 1) / RULES_THREADS;
  hipLaunchKernelGGL(k_env_rules, dim3(blocks), dim3(RULES_THREADS), 0, (hipStream_t)stream, e->cfg, e->state, e->mt, 1,
                     actions, mask, reward, done, score, status);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_env_observe(hz_env_t* e, int mdp, void* obs_out, int obs_dtype, int64_t obs_stride,
                              uint32_t* packed_out, uint8_t* legal_out, void* stream) {
  HZ_REQUIRE(e != nullptr, "hz_env_observe: NULL handle");
  HZ_REQUIRE(mdp == HZ_MDP_GLOBAL || mdp == HZ_MDP_LOCAL, "hz_env_observe: bad mdp %d", mdp);
  HZ_REQUIRE(obs_dtype >= HZ_OBS_U8 && obs_dtype <= HZ_OBS_F16, "hz_env_observe: bad obs_dtype %d", obs_dtype);
  const int D = (mdp == HZ_MDP_GLOBAL ? e->cfg.own_len : 0) + e->cfg.obs_len + e->cfg.P;
  HZ_REQUIRE(obs_out == nullptr || obs_stride >= D, "hz_env_observe: obs_stride %lld < row length %d", (long long)obs_stride, D);
  hipLaunchKernelGGL(k_env_observe, dim3((e->cfg.N + 3) / 4), dim3(256), 0, (hipStream_t)stream, e->cfg, e->state, mdp,
                     obs_out, obs_dtype, (long long)obs_stride, packed_out, legal_out);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int hz_env_probe(hz_env_t* e, int32_t* out, void* stream) {
  HZ_REQUIRE(e != nullptr && out != nullptr, "hz_env_probe: NULL argument");
  hipLaunchKernelGGL(k_env_probe, dim3((e->cfg.N + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->cfg, e->state, out);
  HZ_HIP(hipGetLastError());
  return 0;
}

extern "C" int64_t hz_env_hbm_bytes(const hz_env_t* e) { return e ? e->bytes : 0; }

extern "C" int hz_env_snapshot(hz_env_t* e, void* out_state, uint32_t* out_positions, void* stream) {
  HZ_REQUIRE(e && out_state && out_positions, "hz_env_snapshot: NULL argument");
  const size_t N = (size_t)e->cfg.N;
  HZ_HIP(hipMemcpyAsync(out_state, e->state, N * 32 * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  HZ_HIP(hipMemcpyAsync(out_positions, e->mt + 624 * N, N * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

extern "C" int hz_env_restore(hz_env_t* e, const void* state, const uint32_t* positions, void* stream) {
  HZ_REQUIRE(e && state && positions, "hz_env_restore: NULL argument");
  const size_t N = (size_t)e->cfg.N;
  HZ_HIP(hipMemcpyAsync(e->state, state, N * 32 * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  HZ_HIP(hipMemcpyAsync(e->mt + 624 * N, positions, N * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}
