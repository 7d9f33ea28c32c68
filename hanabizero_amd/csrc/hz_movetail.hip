// hz_movetail.hip -- everything a lock-step does per env AFTER the search, in two launches (include/hz_movetail.h).
//
// The reference does this part as a Python loop over envs (core/selfplay_worker.py:284-347, 216-240).  Launch by launch it
// was: root read-out (cytree get_distributions / get_values) -> select_action + store_search_stats -> env.step -> observe ->
// GameHistory.append + outbox slots -> finished-game flush + masked reset -> observe -> observation windows + the next move's
// draws: eight dependent launches of per-env independent, latency-bound work, each paying a launch boundary and a round trip
// through HBM for what the next one reads.  Here one wave owns one env from the read-out to the history append (k_move_tail_a)
// and from the flush to the next move's window (k_move_tail_b): what used to travel between kernels stays in the wave's
// registers and LDS (the chosen action, the env's 128-B state, the observation's bits).  The ONE step that looks across envs --
// the outbox slots of the games that just ended are handed out in env order -- is why there are two launches, and it costs no
// extra one: every workgroup of the second launch counts the done flags of the envs in front of its own (<= N bytes from L2).
// Same bodies (hz_env_dev.h, hz_selfplay_dev.h) as the one-launch-per-phase entry points, which stay: same bits.
#include "hz_env_dev.h"
#include "hz_movetail.h"
#include "hz_selfplay_dev.h"
#include "hz_tree_dev.h"

// Diagnostic build only (-DHZ_TAIL_PROFILE, tools/tail_profile.py): s_memrealtime (100 MHz) stamps of every 64th workgroup.
#ifdef HZ_TAIL_PROFILE
__device__ unsigned long long hz_tail_prof[2][16][8][8];  // kernel, sampled workgroup, wave, stamp (7: 1 + done)
extern "C" int hz_tail_profile_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(hz_tail_prof), sizeof(hz_tail_prof));
}
#define TS(K, I) do { if ((blockIdx.x & 63) == 0 && blockIdx.x < 1024 && lane == 0) hz_tail_prof[K][blockIdx.x >> 6][wave][I] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TSV(K, I, V) do { if ((blockIdx.x & 63) == 0 && blockIdx.x < 1024 && lane == 0) hz_tail_prof[K][blockIdx.x >> 6][wave][I] = (V); } while (0)
#else
#define TS(K, I) (void)0
#define TSV(K, I, V) (void)0
#endif

struct TailA {
  int32_t* counts;        // [N][A] OUT: root child visit counts, illegal ones zeroed (what hz_actor_record_search leaves)
  float* values;          // [N] OUT: root values
  const uint8_t* legal;   // [N][A] legal moves of the position searched
  const double* uniform;  // [N] sampling uniforms of this move
  float temperature;
  const float* temperature_dev;  // [1] or NULL: read at run time, so that a captured lock-step follows a temperature schedule
  int deterministic;
  int32_t* action;        // [N] OUT
  double* entropy;        // [N] OUT or NULL
  int32_t* reward;        // [N] OUT  (the env's step outputs)
  uint8_t* done;
  int32_t* score;
  int32_t* status;
  int64_t* count_snap;    // [1] OUT: out_count[0] as this lock-step found it (k_move_tail_b's base)
  int mdp;
  uint64_t seed;          // the NEXT move's draws (hz_actor_draw), made in this launch while the wave's own loads are in flight
  long long* move_count;
  double alpha;
  float* noise;           // [N][A] OUT
  double* uniform_next;   // [N] OUT (the same buffer as `uniform`: overwritten once this move's value has been read)
};

// One wave per env, four envs per workgroup, from the root read-out to the history append.  A wave first requests what it will
// need (state, generator position, root records); while those loads are in flight the NEXT move's draws are made (hz_actor_draw:
// long fp64 code -- Marsaglia-Tsang gamma draws -- independent of everything this move does), floor(64 / A) envs side by side
// in a wave, so that at A = 20 two of the four waves run that code instead of all four at a third of their lanes; this move's
// sampling uniforms are read before the draws overwrite them.  (Tried: the draws on four partner waves per workgroup -- the
// launch of 1024 workgroups of 512 threads ramps up over 16 us; workgroups of 256 all start within 0.3 us: tools/tail_profile.py.)
__global__ __launch_bounds__(256) void k_move_tail_a(TreeView tv, EnvCfg g, uint32_t* __restrict__ state, uint32_t* __restrict__ mt,
                                                     hz_actor_bufs_t b, TailA a) {
  __shared__ uint32_t s_state[4][32];
  __shared__ uint32_t s_bits[4][OBS_WORDS];
  __shared__ uint32_t s_rng[4][8];  // MtBatch of one move: w[3] | far[2]
  __shared__ double g_s[4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int env = blockIdx.x * 4 + wave;
  TS(0, 0);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.count_snap[0] = b.out_count[0];  // (nothing in this launch changes out_count; the next one hands out slots from here)
    *b.num_finished = 0;
  }
  const int A = b.num_actions, T = b.max_moves, W = b.packed_words;
  // this move's uniforms, read by everybody before anybody's draw overwrites them
  const bool live = env < b.num_envs;
  const double u_now = (a.deterministic || !live) ? 0.0 : a.uniform[env];
  const float temperature = a.temperature_dev ? a.temperature_dev[0] : a.temperature;
  __syncthreads();
  // the next move's draws, floor(64 / A) envs to a wave: wave w draws for the workgroup's envs [w * per, (w + 1) * per)
  const int per = 64 / A;
  const int dfirst = blockIdx.x * 4 + wave * per;
  int dcount = min(per, 4 - wave * per);
  dcount = min(dcount, b.num_envs - dfirst);
  if (!live && dcount <= 0) return;
  uint32_t* st = s_state[wave];
  uint32_t* bits = s_bits[wave];
  // requests first: the env's state, the generator's position, the root's records
#define MT(k) mt[(size_t)(k) * g.N + (live ? env : 0)]
  uint32_t st_v = 0u;
  int idx0 = 0, visit = 0, rv = 0, lg_now = 0, t = 0;
  float rvs = 0.0f;
  if (live) {
    st_v = lane < 32 ? state[(size_t)env * 32 + lane] : 0u;
    idx0 = (int)MT(624);
    visit = lane < A ? (int)(__float_as_uint(tv.rec[((size_t)env * tv.S) * tv.A + lane].w) >> 16) : 0;
    rv = tv.root_visit[env];
    rvs = tv.root_vsum[env];
    lg_now = lane < A ? (int)a.legal[(size_t)env * A + lane] : 0;
    t = actor_t(b, env);
  }
  if (lane < OBS_WORDS) bits[lane] = 0;
  // (meanwhile the loads above arrive)
  actor_draw_packed(a.seed, (long long)b.env_id_base, a.move_count, dfirst, dcount, lane, A, a.alpha, a.noise, a.uniform_next, g_s[wave]);
  TS(0, 1);
  if (!live) return;
  // the raw generator words one move can need (mt_batch_begin, lanes side by side) -- used after the action is chosen
  if (idx0 >= 624) idx0 = 0;
  idx0 = __builtin_amdgcn_readfirstlane(idx0);
  const uint32_t mw = lane <= 2 ? MT(mt_wrap(idx0 + lane)) : 0u;
  const uint32_t mf = lane < 2 ? MT(mt_wrap(mt_wrap(idx0 + lane) + 397)) : 0u;
  if (lane < 32) st[lane] = st_v;
  // root read-out (cytree get_distributions / get_values: cnode.cpp:261-292) ...
  if (lane < A) a.counts[(size_t)env * A + lane] = visit;
  const float root_value = rv == 0 ? 0.0f : rvs / (float)rv;
  // ... select_action + store_search_stats (hz_actor_record_search)
  double ent;
  int mc;
  int action = select_action_wave(env, lane, A, a.counts, visit, lg_now, u_now, temperature, a.deterministic, &ent, &mc);
  action = __builtin_amdgcn_readfirstlane(action);
  TS(0, 2);
  if (lane < A) b.visits[((size_t)env * T + t) * A + lane] = (int16_t)mc;
  if (lane == 0) {
    a.values[env] = root_value;
    a.action[env] = action;
    if (a.entropy) a.entropy[env] = ent;
    b.action[(size_t)env * T + t] = (int8_t)action;
    b.value[(size_t)env * T + t] = root_value;
    b.ent_sum[env] += ent;
  }
  TS(0, 3);
  // env.step (hz_env_step): the rules are one lane's work
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  MtBatch rng;
  rng.mt = mt; rng.N = g.N; rng.env = env; rng.stride = 1;
  rng.w = s_rng[wave];
  rng.far = s_rng[wave] + 3;
  rng.idx0 = idx0; rng.n = 2; rng.pos = 0;
  if (lane <= 2) rng.w[lane] = mw;
  if (lane < 2) rng.far[lane] = mf;
#undef MT
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  St s;
  s.p = st;
  s.stride = 1;
  int rw = 0, dn = 0, sc = 0, status = 0, last_score = 0, need = 0;
  if (lane == 0) {
    status = env_step_apply(g, s, action, last_score);
    need = status == HZ_ENV_OK && env_step_needs_deal(g, s);
  }
  status = __builtin_amdgcn_readfirstlane(status);
  need = __builtin_amdgcn_readfirstlane(need);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (need) {  // the chance player's move, all lanes at work (deal_pick_wave: lane = card type)
    const int ncards = g.C * g.R;
    const int cnt = lane < ncards ? (int)((st[25 + (lane >> 4)] >> ((lane & 15) * 2)) & 3u) : 0;
    const uint64_t present = __ballot(cnt > 0);
    const int n_out = __popcll((unsigned long long)present);
    double u = 0.0;
    if (n_out >= 2) {  // generate_canonical<double,53>: two 32-bit draws, low word first (random.tcc:3348-3380)
      uint32_t lo_w = 0, hi_w = 0;
      if (lane == 0) {
        lo_w = mt_batch_next(rng);
        hi_w = mt_batch_next(rng);
      }
      const double lo = (double)(uint32_t)__builtin_amdgcn_readfirstlane((int)lo_w);
      const double hi = (double)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi_w);
      u = (lo + hi * 4294967296.0) / 18446744073709551616.0;
      if (u >= 1.0) u = 0x1.fffffffffffffp-1;
    }
    const int pick = deal_pick_wave(ncards, cnt, s.deck_total(), present, n_out, u);
    if (lane == 0) env_apply_deal(g, s, pick);
  }
  if (lane == 0) {
    if (status == HZ_ENV_OK) {
      env_step_finish(g, s, last_score, rw, dn, sc);
      mt_batch_commit(rng);
    } else {
      dn = (int)(env_end_status(g, s) != 0);
      sc = last_score;
    }
    a.reward[env] = rw;
    a.done[env] = (uint8_t)dn;
    a.score[env] = sc;
    a.status[env] = status;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  status = __builtin_amdgcn_readfirstlane(status);
  TS(0, 4);
  if (status == HZ_ENV_OK && lane < 32) state[(size_t)env * 32 + lane] = st[lane];
  // the observation after the move, terminal one included (hz_env_observe; selfplay_worker.py:308) ...
  env_observe_bits(g, s, bits, lane, a.mdp);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  TS(0, 5);
  // ... appended to the history (hz_actor_record_step, minus its outbox slots)
  int32_t* orow = b.obs + ((size_t)env * (T + 1) + t + 1) * W;
  if (lane < W) orow[lane] = (int32_t)bits[lane];
  uint8_t* lrow = b.legal + ((size_t)env * (T + 1) + t + 1) * A;
  if (lane < A) lrow[lane] = (uint8_t)move_is_legal(g, s, s.cur(), decode_move(g, lane));
  if (lane == 0) {
    b.reward[(size_t)env * T + t] = (int8_t)rw;
    if (status != 0) atomicAdd(reinterpret_cast<unsigned long long*>(b.illegal_steps), 1ull);
    int32_t* m = b.meta + (size_t)env * 4;
    m[0] = t + 1;
    m[1] = sc;
    m[2] = env + b.env_id_base;
    m[3] = (int32_t)__float_as_uint((float)b.ent_sum[env]);
  }
  TS(0, 6);
  TSV(0, 7, 1ull + (unsigned long long)need);
}

struct TailB {
  const uint8_t* done;      // [N] from k_move_tail_a
  const int64_t* count_snap;
  int mdp, obs_dtype;
  uint8_t* legal;           // [N][A] OUT: legal moves of the next position
  int32_t* packed;          // [N][W] OUT: its observation, bit-packed
  uint8_t* stack_buf;       // the model's input windows: rows of `stack` slots of slot_elems elements of obs_dtype
  long long stack_row_elems;
  int stack, slot_elems;
};

struct FlushRows {
  const uint8_t* src[7];
  uint8_t* dst[7];
  long long row_bytes[7];
};

template <typename U>
__device__ __forceinline__ U obs_one(int dtype);
template <>
__device__ __forceinline__ uint8_t obs_one<uint8_t>(int) { return 1; }
template <>
__device__ __forceinline__ uint16_t obs_one<uint16_t>(int dtype) { return dtype == HZ_OBS_BF16 ? 0x3f80u : 0x3c00u; }
template <>
__device__ __forceinline__ uint32_t obs_one<uint32_t>(int) { return 0x3f800000u; }

// One row of bytes moved by ONE wave, eight 16-B requests in flight per lane.  A finished game's rows are 160 B .. 16 KB and most
// of them only 4-byte aligned -- with an odd number of actions (Hanabi-Small: 11) the legal-mask rows not even that: gfx950
// (unaligned access mode, the ROCm default: the target feature the compiler itself relies on) takes dwordx4 accesses at any byte
// address, which the packed, 1-aligned type below makes the compiler emit (same ISA as with aligned(4)); a load -> store round
// trip per 256 B would be 60 dependent trips for the observation row alone.
struct __attribute__((packed, aligned(1))) Dwords4 {
  uint32_t x, y, z, w;
};
__device__ __forceinline__ uint4 load_dwords4(const uint8_t* p) {  // (values travel as uint4: arrays of the packed type end up in scratch)
  const Dwords4 t = *reinterpret_cast<const Dwords4*>(p);
  return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void store_dwords4(uint8_t* p, uint4 v) {
  Dwords4 t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  *reinterpret_cast<Dwords4*>(p) = t;
}
struct __attribute__((packed, aligned(1))) Dword1 {
  uint32_t x;
};
__device__ __forceinline__ uint32_t load_dword1(const uint8_t* p) { return reinterpret_cast<const Dword1*>(p)->x; }
__device__ __forceinline__ void store_dword1(uint8_t* p, uint32_t v) { reinterpret_cast<Dword1*>(p)->x = v; }
__device__ __forceinline__ void copy_row_wave(const uint8_t* a, uint8_t* b, long long n, int lane) {
  if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)n) & 3) == 0) {
    const long long n16 = n & ~15ll;
    for (long long off0 = 0; off0 < n16; off0 += 64 * 16 * 8) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long off = off0 + ((long long)u * 64 + lane) * 16;
        if (off < n16) v[u] = load_dwords4(a + off);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long off = off0 + ((long long)u * 64 + lane) * 16;
        if (off < n16) store_dwords4(b + off, v[u]);
      }
    }
    if (lane < (int)((n - n16) >> 2)) *reinterpret_cast<uint32_t*>(b + n16 + 4 * lane) = *reinterpret_cast<const uint32_t*>(a + n16 + 4 * lane);
  } else {
    for (long long off = lane; off < n; off += 64) b[off] = a[off];
  }
}

// The seven rows of a finished game moved by a whole workgroup (256 threads) with every load of every row requested before the
// first store: ONE round trip to memory for a Hanabi-Full game (16 KB of observations, 11 KB in the six other rows) instead of one
// per 2 KB.  Row BIG (the observations) has NB 16-B pieces per thread and round, the others NS; longer rows take further rounds.
// Any length, any alignment: 16-B pieces, then up to three dwords (threads 0..2), then up to three bytes (threads 64..66).
template <int BIG, int NB, int NS>
__device__ __forceinline__ void copy_rows_block(const FlushRows& fr, size_t row, size_t slot, int tid) {
  bool more = true;
  for (int r = 0; more; ++r) {
    more = false;
    uint4 vb[NB], vs[7][NS];
    uint32_t tail[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const long long n = fr.row_bytes[k], n16 = n & ~15ll;
      const uint8_t* a = fr.src[k] + row * (size_t)n;
      if (k == BIG) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const long long off = ((long long)(r * NB + u) * 256 + tid) * 16;
          if (off < n16) vb[u] = load_dwords4(a + off);
        }
        if ((long long)(r + 1) * NB * 256 * 16 < n16) more = true;
      } else {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          const long long off = ((long long)(r * NS + u) * 256 + tid) * 16;
          if (off < n16) vs[k][u] = load_dwords4(a + off);
        }
        if ((long long)(r + 1) * NS * 256 * 16 < n16) more = true;
      }
      if (r == 0 && tid < (int)((n - n16) >> 2)) tail[k] = load_dword1(a + n16 + 4 * tid);
      if (r == 0 && tid >= 64 && tid < 64 + (int)(n & 3)) tail[k] = a[(n & ~3ll) + (tid - 64)];
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const long long n = fr.row_bytes[k], n16 = n & ~15ll;
      uint8_t* bdst = fr.dst[k] + slot * (size_t)n;
      if (k == BIG) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const long long off = ((long long)(r * NB + u) * 256 + tid) * 16;
          if (off < n16) store_dwords4(bdst + off, vb[u]);
        }
      } else {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          const long long off = ((long long)(r * NS + u) * 256 + tid) * 16;
          if (off < n16) store_dwords4(bdst + off, vs[k][u]);
        }
      }
      if (r == 0 && tid < (int)((n - n16) >> 2)) store_dword1(bdst + n16 + 4 * tid, tail[k]);
      if (r == 0 && tid >= 64 && tid < 64 + (int)(n & 3)) bdst[(n & ~3ll) + (tid - 64)] = (uint8_t)tail[k];
    }
  }
}

// One wave per env, four envs per workgroup:
//   1  every thread counts done flags (the slots' prefix); each wave hands out its env's outbox slot                [barrier]
//   2  the workgroup's finished games leave for the outbox (hz_actor_flush), all four waves on each                 [barrier]
//   3  per env: the new game if the old one ended (hz_env_reset), the observation, the trajectory head, the window
template <typename U>
__global__ __launch_bounds__(256) void k_move_tail_b(EnvCfg g, uint32_t* __restrict__ state, uint32_t* __restrict__ mt,
                                                     hz_actor_bufs_t b, TailB a, FlushRows fr) {
  __shared__ uint32_t s_state[4][32];
  __shared__ uint32_t s_bits[4][OBS_WORDS];
  __shared__ int s_before, s_done[4], s_slot[4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int env0 = blockIdx.x * 4;
  const int env = env0 + wave;
  const bool live = env < b.num_envs;
  TS(1, 0);
  // how many games ended in the envs in front of this workgroup's (done flags are 0 / 1 bytes; env0 is a multiple of 4)
  if (tid == 0) s_before = 0;
  if (tid < 4) s_done[tid] = (env0 + tid < b.num_envs && a.done[env0 + tid] != 0) ? 1 : 0;
  // (requested now, used after the barriers)
  const uint32_t st_v = (live && lane < 32) ? state[(size_t)env * 32 + lane] : 0u;
  const long long base = a.count_snap[0];
  __syncthreads();
  {
    int c = 0;
    const uint32_t* d4 = reinterpret_cast<const uint32_t*>(a.done);
    for (int i = tid; i < env0 / 4; i += 256) c += __popc(d4[i] & 0x01010101u);
    for (int off = 32; off; off >>= 1) c += __shfl_xor(c, off);
    if (lane == 0 && c) atomicAdd(&s_before, c);
  }
  __syncthreads();
  const int A = b.num_actions, T = b.max_moves, W = b.packed_words;
  const bool d = live && s_done[wave] != 0;
  uint32_t* st = s_state[wave];
  uint32_t* bits = s_bits[wave];
  if (live) {
    int sl = -1;
    if (d) {  // the outbox slot of the game that just ended (env order) and the counters (hz_actor_record_step's second half)
      int before = s_before;
      for (int k = 0; k < wave; ++k) before += s_done[k];
      sl = (int)((base + before) % (long long)b.outbox_games);
      if (lane == 0) {
        b.finished[before] = env;
        atomicAdd(reinterpret_cast<unsigned long long*>(b.out_count), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long*>(b.out_count + 1), (unsigned long long)b.meta[(size_t)env * 4]);
        atomicAdd(b.num_finished, 1);
      }
    }
    if (lane == 0) {
      b.slot[env] = sl;
      s_slot[wave] = sl;
    }
    if (lane < OBS_WORDS) bits[lane] = 0;
    if (lane < 32) st[lane] = st_v;
  }
  __syncthreads();
  TS(1, 1);
  if (s_done[0] | s_done[1] | s_done[2] | s_done[3]) {  // (uniform over the workgroup)
    for (int e = 0; e < 4; ++e) {
      if (!s_done[e]) continue;
      copy_rows_block<5, 4, 2>(fr, (size_t)(env0 + e), (size_t)s_slot[e], tid);
    }
    // every row has been READ once its stores have been issued (they carry the loaded data), so a bare barrier -- no wait for
    // the stores' completion, a second round trip -- is what lets the owners overwrite the heads of their trajectories
    asm volatile("s_barrier" ::: "memory");
  }
  TS(1, 2);
  if (!live) return;
  TSV(1, 7, 1ull + (unsigned long long)d);
  if (d) env_reset_wave_body(g, state, mt, nullptr, env, lane, st);  // leaves the new game's state in `st` and in HBM
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  TS(1, 3);
  // everybody's current observation (hz_env_observe) ...
  St s;
  s.p = st;
  s.stride = 1;
  env_observe_bits(g, s, bits, lane, a.mdp);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  TS(1, 4);
  const int obs = s.cur();
  uint8_t lg = 0;
  if (lane < A) {
    lg = (uint8_t)move_is_legal(g, s, obs, decode_move(g, lane));
    a.legal[(size_t)env * A + lane] = lg;
  }
  if (lane < W) a.packed[(size_t)env * W + lane] = (int32_t)bits[lane];
  // ... heads the trajectory (hz_actor_begin_move; selfplay_worker.py:237, 326-327) ...
  const int t0 = d ? 0 : actor_t(b, env) + 1;
  if (lane == 0) {
    b.traj_len[env] = t0;
    if (d) b.ent_sum[env] = 0.0;
  }
  if (lane < W) b.obs[((size_t)env * (T + 1) + t0) * W + lane] = (int32_t)bits[lane];
  if (lane < A) b.legal[((size_t)env * (T + 1) + t0) * A + lane] = lg;
  // ... and enters the model's input window: a running game's slots move up by one and the last takes the new observation, a
  // new game's are all filled with it (the slots' pad elements stay zero)
  TS(1, 5);
  const int D = ((a.mdp == HZ_MDP_GLOBAL) ? g.own_len : 0) + g.obs_len + g.P;
  U* row = reinterpret_cast<U*>(a.stack_buf) + (size_t)env * (size_t)a.stack_row_elems;
  const U one = obs_one<U>(a.obs_dtype);
  if (!d)  // (one overlapping move towards lower addresses, ascending, every block's loads before its stores)
    copy_row_wave(reinterpret_cast<const uint8_t*>(row + a.slot_elems), reinterpret_cast<uint8_t*>(row),
                  (long long)(a.stack - 1) * a.slot_elems * (long long)sizeof(U), lane);
  for (int k = d ? 0 : a.stack - 1; k < a.stack; ++k) {
    U* dst = row + (size_t)k * a.slot_elems;
    for (int j = lane; j < D; j += 64) dst[j] = ((bits[j >> 5] >> (j & 31)) & 1u) ? one : (U)0;
  }
  TS(1, 6);
}

extern "C" int hz_actor_move_tail(hz_tree_t* tree, hz_env_t* env, const hz_actor_bufs_t* bufs, int mdp, int32_t* counts,
                                  float* root_values, uint8_t* legal, double* uniform, float temperature, const float* temperature_dev, int deterministic,
                                  int32_t* action, double* entropy, int32_t* reward, uint8_t* done, int32_t* score,
                                  int32_t* status, int32_t* packed, void* stack_buf, int64_t stack_row_bytes, int stack,
                                  int64_t slot_bytes, int obs_dtype, uint64_t seed, int64_t* move_count, double alpha,
                                  float* noise, int64_t* scratch, void* stream) {
  HZ_REQUIRE(tree && env && bufs && counts && root_values && legal && uniform && action && reward && done && score && status &&
                 packed && stack_buf && move_count && noise && scratch,
             "hz_actor_move_tail: NULL argument");
  const EnvCfg& g = env->cfg;
  HZ_REQUIRE(bufs->num_envs > 0 && bufs->num_envs == g.N && bufs->num_envs == tree->N && bufs->num_actions == g.num_moves &&
                 bufs->num_actions == tree->A && bufs->num_actions <= 64,
             "hz_actor_move_tail: tree (%d x %d), env (%d x %d) and actor buffers (%d x %d) disagree", tree->N, tree->A, g.N,
             g.num_moves, bufs->num_envs, bufs->num_actions);
  HZ_REQUIRE(mdp == HZ_MDP_GLOBAL || mdp == HZ_MDP_LOCAL, "hz_actor_move_tail: bad mdp %d", mdp);
  const int D = (mdp == HZ_MDP_GLOBAL ? g.own_len : 0) + g.obs_len + g.P;
  HZ_REQUIRE(bufs->packed_words == (D + 31) / 32 && bufs->packed_words <= 64 && bufs->max_moves > 0 && bufs->outbox_games > 0,
             "hz_actor_move_tail: packed_words %d for a %d-bit observation", bufs->packed_words, D);
  HZ_REQUIRE(bufs->action && bufs->reward && bufs->value && bufs->visits && bufs->legal && bufs->obs && bufs->traj_len &&
                 bufs->ent_sum && bufs->meta && bufs->out_count && bufs->slot && bufs->finished && bufs->num_finished &&
                 bufs->illegal_steps && bufs->out_action && bufs->out_reward && bufs->out_value && bufs->out_visits &&
                 bufs->out_legal && bufs->out_obs && bufs->out_meta,
             "hz_actor_move_tail: NULL buffer in bufs");
  HZ_REQUIRE(deterministic || uniform, "hz_actor_move_tail: uniform samples required when sampling");
  HZ_REQUIRE(temperature > 0.0f && alpha > 0.0, "hz_actor_move_tail: temperature and alpha must be > 0");
  HZ_REQUIRE(obs_dtype >= HZ_OBS_U8 && obs_dtype <= HZ_OBS_F16, "hz_actor_move_tail: bad obs_dtype %d", obs_dtype);
  const int es = obs_dtype == HZ_OBS_U8 ? 1 : (obs_dtype == HZ_OBS_F32 ? 4 : 2);
  HZ_REQUIRE(stack >= 1 && slot_bytes >= (int64_t)D * es && slot_bytes % 16 == 0 && stack_row_bytes >= (int64_t)stack * slot_bytes &&
                 stack_row_bytes % 16 == 0 && ((uintptr_t)stack_buf % 16) == 0,
             "hz_actor_move_tail: window slots must be 16-B multiples holding %d elements (stack=%d slot_bytes=%lld row_bytes=%lld)",
             D, stack, (long long)slot_bytes, (long long)stack_row_bytes);
  HZ_REQUIRE(((uintptr_t)done % 4) == 0, "hz_actor_move_tail: the done flags must be 4-B aligned");
  TailA a;
  a.counts = counts; a.values = root_values; a.legal = legal; a.uniform = uniform; a.temperature = temperature; a.temperature_dev = temperature_dev;
  a.deterministic = deterministic; a.action = action; a.entropy = entropy; a.reward = reward; a.done = done; a.score = score;
  a.status = status; a.count_snap = scratch; a.mdp = mdp;
  a.seed = seed; a.move_count = (long long*)move_count; a.alpha = alpha; a.noise = noise; a.uniform_next = uniform;
  const int N = bufs->num_envs;
  hipLaunchKernelGGL(k_move_tail_a, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, view(tree), g, env->state, env->mt, *bufs, a);
  TailB bb;
  bb.done = done; bb.count_snap = scratch; bb.mdp = mdp; bb.obs_dtype = obs_dtype; bb.legal = legal; bb.packed = packed;
  bb.stack_buf = (uint8_t*)stack_buf; bb.stack_row_elems = stack_row_bytes / es; bb.stack = stack; bb.slot_elems = (int)(slot_bytes / es);
  const long long T = bufs->max_moves, A = bufs->num_actions, W = bufs->packed_words;
  FlushRows fr;
  const void* src[7] = {bufs->action, bufs->reward, bufs->value, bufs->visits, bufs->legal, bufs->obs, bufs->meta};
  void* dst[7] = {bufs->out_action, bufs->out_reward, bufs->out_value, bufs->out_visits, bufs->out_legal, bufs->out_obs, bufs->out_meta};
  const long long rb[7] = {T, T, 4 * T, 2 * T * A, (T + 1) * A, 4 * (T + 1) * W, 16};
  for (int k = 0; k < 7; ++k) {
    fr.src[k] = (const uint8_t*)src[k];
    fr.dst[k] = (uint8_t*)dst[k];
    fr.row_bytes[k] = rb[k];
  }
  const dim3 grid((N + 3) / 4), block(256);
  if (es == 1) hipLaunchKernelGGL(k_move_tail_b<uint8_t>, grid, block, 0, (hipStream_t)stream, g, env->state, env->mt, *bufs, bb, fr);
  else if (es == 2) hipLaunchKernelGGL(k_move_tail_b<uint16_t>, grid, block, 0, (hipStream_t)stream, g, env->state, env->mt, *bufs, bb, fr);
  else hipLaunchKernelGGL(k_move_tail_b<uint32_t>, grid, block, 0, (hipStream_t)stream, g, env->state, env->mt, *bufs, bb, fr);
  HZ_HIP(hipGetLastError());
  return 0;
}
