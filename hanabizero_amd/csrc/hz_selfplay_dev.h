// hz_selfplay_dev.h -- device code of the actor's per-move bookkeeping (see hz_selfplay.hip / include/hz_selfplay.h), shared by
// the one-launch-per-phase kernels (hz_selfplay.hip) and the fused move tail (hz_movetail.hip).
#pragma once
#include <math.h>

#include "hz_common.h"
#include "hz_selfplay.h"

// one lane per env; rows are short (A <= 64) and the kernel is launch-bound, not bandwidth-bound
__device__ __forceinline__ int select_action_env(int env, int A, int32_t* __restrict__ counts,
                                                 const uint8_t* __restrict__ legal, const double* __restrict__ uniform,
                                                 float temperature, int deterministic, double* ent_out) {
  int32_t* c = counts + (size_t)env * A;
  const uint8_t* lg = legal + (size_t)env * A;
  // utils.py:282-284: zero the counts of illegal actions
  double total = 0.0;
  int best = 0, best_count = INT32_MIN;
  const bool unit_t = (temperature == 1.0f);
  const double inv_t = 1.0 / (double)temperature;
  for (int a = 0; a < A; ++a) {
    int v = c[a];
    if (lg[a] == 0 && v >= 1) {
      v = 0;
      c[a] = 0;
    }
    if (v > best_count) {  // np.argmax: first maximum
      best_count = v;
      best = a;
    }
    total += unit_t ? (double)v : pow((double)v, inv_t);  // utils.py:286-287 (Python sum, left to right)
  }
  if (!(total > 0.0)) {
    *ent_out = 0.0;
    return -1;
  }
  // np.random.choice: cdf = cumsum(p); cdf /= cdf[-1]; idx = searchsorted(cdf, u, side='right')
  double last = 0.0;
  for (int a = 0; a < A; ++a) {
    const double x = unit_t ? (double)c[a] : pow((double)c[a], inv_t);
    last += x / total;
  }
  int action = best;
  double ent = 0.0;
  if (!deterministic) {
    const double u = uniform[env];
    double acc = 0.0;
    int idx = 0;
    for (int a = 0; a < A; ++a) {
      const double x = unit_t ? (double)c[a] : pow((double)c[a], inv_t);
      acc += x / total;
      if (acc / last <= u) idx = a + 1;  // side='right': number of cdf entries <= u
    }
    action = idx < A ? idx : A - 1;
  }
  {
    // scipy.stats.entropy(pk, base=2): pk /= sum(pk); sum(-pk*log(pk)) / log(2)
    double psum = 0.0;
    for (int a = 0; a < A; ++a) psum += (unit_t ? (double)c[a] : pow((double)c[a], inv_t)) / total;
    for (int a = 0; a < A; ++a) {
      const double pk = ((unit_t ? (double)c[a] : pow((double)c[a], inv_t)) / total) / psum;
      if (pk > 0.0) ent -= pk * log(pk);
    }
    *ent_out = ent / log(2.0);
  }
  return action;
}


// ---- the actor's per-move bookkeeping (include/hz_selfplay.h) ---------------------------------------------------
__device__ __forceinline__ int actor_t(const hz_actor_bufs_t& b, int env) {
  const long long len = b.traj_len[env];  // a Hanabi game cannot outlast max_moves; the clamp keeps indices in range
  return (int)(len < (long long)(b.max_moves - 1) ? len : (long long)(b.max_moves - 1));
}

// select_action_env with one wave per env (lane = action): the same fp64 values in the same summation order -- the
// left-to-right sums walk the lanes with v_readlane -- but the divisions, pow and log of the A actions run side by side.
__device__ __forceinline__ double readlane_d(double v, int l) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffu), l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// (`u`: the env's sampling uniform; not looked at when deterministic)
// (`count`, `lg`: this lane's action's visit count and legal flag -- lanes >= A ignored; `counts`: where the masked counts go)
__device__ __forceinline__ int select_action_wave(int env, int lane, int A, int32_t* __restrict__ counts, int count, int lg,
                                                  double u, float temperature, int deterministic, double* ent_out,
                                                  int* masked_count) {
  const bool on = lane < A;
  const bool unit_t = (temperature == 1.0f);
  const double inv_t = 1.0 / (double)temperature;
  int v = on ? count : INT32_MIN;
  if (on && lg == 0 && v >= 1) {  // utils.py:282-284
    v = 0;
    counts[(size_t)env * A + lane] = 0;
  }
  *masked_count = on ? v : 0;
  const double x = on ? (unit_t ? (double)v : pow((double)v, inv_t)) : 0.0;
  int vmax = v;  // np.argmax: first maximum
  for (int off = 32; off; off >>= 1) vmax = max(vmax, __shfl_xor(vmax, off));
  const int best = __ffsll((unsigned long long)__ballot(on && v == vmax)) - 1;
  double total = 0.0;
  for (int a = 0; a < A; ++a) total += readlane_d(x, a);  // utils.py:286-287 (Python sum, left to right)
  if (!(total > 0.0)) {
    *ent_out = 0.0;
    return -1;
  }
  const double p = x / total;
  double last = 0.0, mine = 0.0;  // np.random.choice: cdf = cumsum(p); cdf /= cdf[-1]
  for (int a = 0; a < A; ++a) {
    last += readlane_d(p, a);
    if (lane == a) mine = last;
  }
  int action = best;
  if (!deterministic) {
    const uint64_t le = __ballot(on && mine / last <= u);  // searchsorted(cdf, u, side='right') = #entries <= u ...
    const int idx = le ? 64 - __clzll((unsigned long long)le) : 0;  // ... which the serial loop finds as last hit + 1
    action = idx < A ? idx : A - 1;
  }
  // scipy.stats.entropy(pk, base=2): pk /= sum(pk); sum(-pk*log(pk)) / log(2); sum(pk) is `last`
  const double pk = p / last;
  const double term = (on && pk > 0.0) ? pk * log(pk) : 0.0;
  double ent = 0.0;
  for (int a = 0; a < A; ++a) ent -= readlane_d(term, a);
  *ent_out = ent / log(2.0);
  return action;
}


// four threads per env (part = 0..3): the row copies are split among them, part 0 also writes the scalars
__device__ __forceinline__ void actor_record_step_env(const hz_actor_bufs_t& b, int env, int part,
                                                      const int32_t* __restrict__ reward, const int32_t* __restrict__ score,
                                                      const int32_t* __restrict__ status, const int32_t* __restrict__ packed,
                                                      const uint8_t* __restrict__ legal_next) {
  if (env >= b.num_envs) return;
  const int A = b.num_actions, T = b.max_moves, W = b.packed_words;
  const int t = actor_t(b, env);
  int32_t* orow = b.obs + ((size_t)env * (T + 1) + t + 1) * W;
  for (int w = part; w < W; w += 4) orow[w] = packed[(size_t)env * W + w];
  uint8_t* lrow = b.legal + ((size_t)env * (T + 1) + t + 1) * A;
  for (int a = part; a < A; a += 4) lrow[a] = legal_next[(size_t)env * A + a];
  if (part != 0) return;
  b.reward[(size_t)env * T + t] = (int8_t)reward[env];
  if (status[env] != 0) atomicAdd(reinterpret_cast<unsigned long long*>(b.illegal_steps), 1ull);
  int32_t* m = b.meta + (size_t)env * 4;
  m[0] = t + 1;
  m[1] = score[env];
  m[2] = env + b.env_id_base;
  m[3] = (int32_t)__float_as_uint((float)b.ent_sum[env]);
}


__device__ __forceinline__ void copy_row(const uint8_t* a, uint8_t* b, long long n, int lane) {
  if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)n) & 15) == 0) {
    for (long long off = (long long)lane * 16; off < n; off += 64 * 16)
      *reinterpret_cast<uint4*>(b + off) = *reinterpret_cast<const uint4*>(a + off);
  } else if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)n) & 3) == 0) {
    for (long long off = (long long)lane * 4; off < n; off += 64 * 4)
      *reinterpret_cast<uint32_t*>(b + off) = *reinterpret_cast<const uint32_t*>(a + off);
  } else {
    for (long long off = lane; off < n; off += 64) b[off] = a[off];
  }
}


// one wave per env: trajectory heads and the model's input window
template <typename U>
__device__ __forceinline__ void actor_begin_move_wave(const hz_actor_bufs_t& b, int env, int lane, const uint8_t* __restrict__ done,
                                                      const int32_t* __restrict__ packed, const uint8_t* __restrict__ legal,
                                                      const uint8_t* __restrict__ newest, long long newest_row_bytes,
                                                      uint8_t* __restrict__ stack_buf, long long stack_row_bytes, int stack,
                                                      long long obs_bytes) {
  const int A = b.num_actions, T = b.max_moves, W = b.packed_words;
  const bool d = done[env] != 0;
  const int t0 = d ? 0 : actor_t(b, env) + 1;
  if (lane == 0) {
    b.traj_len[env] = t0;
    if (d) b.ent_sum[env] = 0.0;
  }
  int32_t* orow = b.obs + ((size_t)env * (T + 1) + t0) * W;
  for (int w = lane; w < W; w += 64) orow[w] = packed[(size_t)env * W + w];
  uint8_t* lrow = b.legal + ((size_t)env * (T + 1) + t0) * A;
  for (int a = lane; a < A; a += 64) lrow[a] = legal[(size_t)env * A + a];
  // input window in units of U: slot k <- slot k+1 (running game) or <- newest (new game); last slot <- newest
  const long long n = obs_bytes / (long long)sizeof(U);
  U* row = reinterpret_cast<U*>(stack_buf + (size_t)env * (size_t)stack_row_bytes);
  const U* nw = reinterpret_cast<const U*>(newest + (size_t)env * (size_t)newest_row_bytes);
  for (int k = 0; k < stack; ++k) {
    const U* src = (d || k == stack - 1) ? nw : row + (size_t)(k + 1) * n;
    U* dst = row + (size_t)k * n;
    for (long long i = lane; i < n; i += 64) dst[i] = src[i];
  }
}


// ---- root noise + sampling uniforms (include/hz_selfplay.h) -----------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

struct CounterRng {  // stream = key, k-th output = mix64(key + k * golden)
  uint64_t key, k;
  __device__ double next() {  // (0, 1): 53 bits, never 0
    const uint64_t r = mix64(key + (++k) * 0x9E3779B97F4A7C15ull);
    return ((double)(r >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  }
};

__device__ double gamma_draw(CounterRng& g, double alpha) {  // Marsaglia & Tsang (2000); alpha < 1 via alpha + 1
  const double a = alpha < 1.0 ? alpha + 1.0 : alpha;
  const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
  double out = 0.0;
  for (int it = 0; it < 64; ++it) {  // acceptance > 95 %: the bound is never the exit in practice
    const double u1 = g.next(), u2 = g.next();
    const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);  // Box-Muller
    double v = 1.0 + c * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    const double u = g.next();
    out = d * v;
    if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) break;
  }
  if (alpha < 1.0) out *= pow(g.next(), 1.0 / alpha);
  return out;
}

// one wave per env, lane = action; g_w: 64 doubles of LDS for this wave
__device__ __forceinline__ void actor_draw_wave(uint64_t seed, long long env_id_base, long long* __restrict__ move_count, int env,
                                                int lane, int A, double alpha, float* __restrict__ noise,
                                                double* __restrict__ uniform, double* g_w) {
  const long long k = move_count[env];
  const uint64_t base = mix64(mix64(seed ^ 0x68616e616269ull) + (uint64_t)(env_id_base + env)) + (uint64_t)k * 0xD1B54A32D192ED03ull;
  CounterRng g;
  g.key = mix64(base + (uint64_t)(lane + 1));
  g.k = 0;
  double x = 0.0;
  if (lane < A) x = gamma_draw(g, alpha);
  g_w[lane] = x;
  __builtin_amdgcn_wave_barrier();
  double sum = 0.0;
  for (int a = 0; a < A; ++a) sum += g_w[a];  // same-wave LDS traffic is ordered; action order, like numpy
  if (lane < A) noise[(size_t)env * A + lane] = sum > 0.0 ? (float)(x / sum) : 1.0f / (float)A;
  if (lane == 0) {
    CounterRng gu;
    gu.key = mix64(base);
    gu.k = 0;
    const uint64_t r = mix64(gu.key + 0x9E3779B97F4A7C15ull);
    uniform[env] = (double)(r >> 11) * (1.0 / 9007199254740992.0);  // [0, 1)
    move_count[env] = k + 1;
  }
}

// The same draws with floor(64 / A) envs side by side in one wave (lane = env-in-wave * A + action): the gamma draws are long
// fp64 code that leaves two thirds of a wave's lanes idle at A = 20.  `first_env`: the wave's first env; `n_env`: how many it draws
// for (0: none); g_w: 64 doubles of LDS for this wave.  Same streams, same order of additions per env: same bits.
__device__ __forceinline__ void actor_draw_packed(uint64_t seed, long long env_id_base, long long* __restrict__ move_count,
                                                  int first_env, int n_env, int lane, int A, double alpha,
                                                  float* __restrict__ noise, double* __restrict__ uniform, double* g_w) {
  if (n_env <= 0) return;
  const int e = lane / A, a = lane - e * A;
  const bool on = e < n_env;
  const int env = first_env + (on ? e : 0);
  const long long k = move_count[env];
  const uint64_t base = mix64(mix64(seed ^ 0x68616e616269ull) + (uint64_t)(env_id_base + env)) + (uint64_t)k * 0xD1B54A32D192ED03ull;
  CounterRng g;
  g.key = mix64(base + (uint64_t)(a + 1));
  g.k = 0;
  double x = 0.0;
  if (on) x = gamma_draw(g, alpha);
  g_w[lane] = x;
  __builtin_amdgcn_wave_barrier();
  double sum = 0.0;
  for (int j = 0; j < A; ++j) sum += g_w[(on ? e : 0) * A + j];  // same-wave LDS traffic is ordered; action order, like numpy
  if (on) noise[(size_t)env * A + a] = sum > 0.0 ? (float)(x / sum) : 1.0f / (float)A;
  if (on && a == 0) {
    const uint64_t r = mix64(mix64(base) + 0x9E3779B97F4A7C15ull);
    uniform[env] = (double)(r >> 11) * (1.0 / 9007199254740992.0);  // [0, 1)
    move_count[env] = k + 1;
  }
}
