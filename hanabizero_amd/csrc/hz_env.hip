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
#include "hz_common.h"
#include "hz_env.h"

enum { MV_PLAY = 1, MV_DISCARD = 2, MV_REVEAL_COLOR = 3, MV_REVEAL_RANK = 4 };  // hanabi_move.h:34

struct EnvCfg {
  int N, C, R, P, H, max_info, max_life;
  int num_moves, obs_len, own_len, bpc, max_deck, per_color;
  int inst[5];         // NumberCardInstances per rank (hanabi_game.cc:126-136)
  int inst_prefix[5];  // offset of rank r inside one colour's discard thermometers
  int off_board, off_disc, off_last, off_know;  // section starts inside the canonical vector (hands start at 0)
};

struct hz_env {
  EnvCfg cfg;
  int device;
  uint32_t* state;
  uint32_t* mt;
  int64_t bytes;
};

// ---- bit-field access on a state held in LDS (word w of this env at p[w * stride]) ------------------
struct St {
  uint32_t* p;
  int stride;
  __device__ __forceinline__ uint32_t get(int w, int sh, int nb) const { return (p[w * stride] >> sh) & ((1u << nb) - 1u); }
  __device__ __forceinline__ void set(int w, int sh, int nb, uint32_t v) {
    const uint32_t m = ((1u << nb) - 1u) << sh;
    p[w * stride] = (p[w * stride] & ~m) | ((v << sh) & m);
  }
  __device__ __forceinline__ uint32_t& word(int w) { return p[w * stride]; }
  __device__ __forceinline__ uint32_t word(int w) const { return p[w * stride]; }
  // named fields
  __device__ __forceinline__ int deck(int idx) const { return (int)get(25 + (idx >> 4), (idx & 15) * 2, 2); }
  __device__ __forceinline__ void set_deck(int idx, int v) { set(25 + (idx >> 4), (idx & 15) * 2, 2, (uint32_t)v); }
  __device__ __forceinline__ int disc(int idx) const { return (int)get(27 + (idx >> 4), (idx & 15) * 2, 2); }
  __device__ __forceinline__ void set_disc(int idx, int v) { set(27 + (idx >> 4), (idx & 15) * 2, 2, (uint32_t)v); }
  __device__ __forceinline__ int fw(int c) const { return (int)get(29, 3 * c, 3); }
  __device__ __forceinline__ void set_fw(int c, int v) { set(29, 3 * c, 3, (uint32_t)v); }
  __device__ __forceinline__ int info() const { return (int)get(29, 15, 4); }
  __device__ __forceinline__ void set_info(int v) { set(29, 15, 4, (uint32_t)v); }
  __device__ __forceinline__ int life() const { return (int)get(29, 19, 2); }
  __device__ __forceinline__ void set_life(int v) { set(29, 19, 2, (uint32_t)v); }
  __device__ __forceinline__ int cur() const { return (int)get(29, 21, 3); }
  __device__ __forceinline__ void set_cur(int v) { set(29, 21, 3, (uint32_t)v); }
  __device__ __forceinline__ int next() const { return (int)get(29, 24, 3); }
  __device__ __forceinline__ void set_next(int v) { set(29, 24, 3, (uint32_t)v); }
  __device__ __forceinline__ int turns() const { return (int)get(29, 27, 3); }
  __device__ __forceinline__ void set_turns(int v) { set(29, 27, 3, (uint32_t)v); }
  __device__ __forceinline__ int hand_n(int pl) const { return (int)get(30, 3 * pl, 3); }
  __device__ __forceinline__ void set_hand_n(int pl, int v) { set(30, 3 * pl, 3, (uint32_t)v); }
  __device__ __forceinline__ int deck_total() const { return (int)get(30, 15, 6); }
  __device__ __forceinline__ void set_deck_total(int v) { set(30, 15, 6, (uint32_t)v); }
  __device__ __forceinline__ int has_last() const { return (int)get(30, 21, 1); }
};

__device__ __forceinline__ int card_color(uint32_t c) { return (int)(c & 7u); }
__device__ __forceinline__ int card_rank(uint32_t c) { return (int)((c >> 3) & 7u); }

__device__ __forceinline__ int env_score(const EnvCfg& g, const St& s) {  // hanabi_state.cc:359-364
  if (s.life() <= 0) return 0;
  int v = 0;
  for (int c = 0; c < g.C; ++c) v += s.fw(c);
  return v;
}

__device__ __forceinline__ int env_end_status(const EnvCfg& g, const St& s) {  // hanabi_state.cc:366-377
  if (s.life() < 1) return 1;
  if (env_score(g, s) >= g.C * g.R) return 3;
  if (s.turns() <= 0) return 2;
  return 0;
}

__device__ __forceinline__ int player_to_deal(const EnvCfg& g, const St& s) {  // hanabi_state.cc:157-164
  for (int i = 0; i < g.P; ++i)
    if (s.hand_n(i) < g.H) return i;
  return -1;
}

// ---- std::mt19937 in HBM (libstdc++ bits/random.tcc) ---------------------------------------------------
// The generator regenerates its 624 words in one block ("twist", random.tcc:395-431) whenever the position reaches
// 624, then tempers word after word.  Here the same recurrence runs one word per draw, in place: word i of the new
// block needs the OLD words i and i+1 and word i+397 -- old for i < 227, already regenerated (i - 227) afterwards;
// word 623 uses the new word 0 -- which is exactly what is in the array when draw i arrives.  Same output stream,
// no 624-step stall of one lane while 63 wait (position 624 after seeding == position 0 of the first block).
// All draws of one kernel call are prepared together: the raw words they need are requested at the start of the kernel
// (independent loads, one round trip: a reset needs 2 * players * hand_size draws, which was 40 dependent round trips
// word by word), the recurrence then runs out of LDS, and the words of the draws actually consumed are written back.
#define MT_MAX_DRAWS 50  // 2 * (players * hand_size <= 25)
struct MtBatch {
  uint32_t* mt;
  int N, env;
  uint32_t* w;    // LDS column [MT_MAX_DRAWS + 1]: words at positions idx0 .. idx0 + n (old; regenerated in place)
  uint32_t* far;  // LDS column [MT_MAX_DRAWS]: words at positions idx0 + 397 + k (mod 624)
  int stride, idx0, n, pos;
};

__device__ __forceinline__ int mt_wrap(int i) { return i >= 624 ? i - 624 : i; }

__device__ void mt_batch_begin(MtBatch& b, int n) {
#define MT(k) b.mt[(size_t)(k) * b.N + b.env]
  int idx = (int)MT(624);
  if (idx >= 624) idx = 0;
  b.idx0 = idx;
  b.n = n;
  b.pos = 0;
  // (the far words are never among those this batch regenerates: that would need n > 227)
  for (int k0 = 0; k0 <= n; k0 += 8) {
    uint32_t a[8], f[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u;
      a[u] = k <= n ? MT(mt_wrap(idx + k)) : 0u;
      f[u] = k < n ? MT(mt_wrap(mt_wrap(idx + k) + 397)) : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u;
      if (k <= n) b.w[k * b.stride] = a[u];
      if (k < n) b.far[k * b.stride] = f[u];
    }
  }
}

__device__ uint32_t mt_batch_next(MtBatch& b) {
  const int k = b.pos++;
  const uint32_t y = (b.w[k * b.stride] & 0x80000000u) | (b.w[(k + 1) * b.stride] & 0x7fffffffu);
  uint32_t z = b.far[k * b.stride] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  b.w[k * b.stride] = z;
  z ^= (z >> 11);
  z ^= (z << 7) & 0x9d2c5680u;
  z ^= (z << 15) & 0xefc60000u;
  z ^= (z >> 18);
  return z;
}

__device__ void mt_batch_commit(MtBatch& b) {
  if (b.pos == 0) return;
  for (int k = 0; k < b.pos; ++k) MT(mt_wrap(b.idx0 + k)) = b.w[k * b.stride];
  const int idx = b.idx0 + b.pos;
  MT(624) = (uint32_t)(idx > 624 ? idx - 624 : idx);
#undef MT
}

// ApplyRandomChance (hanabi_state.cc:282-286): ChanceOutcomes (:313-325) -> PickRandomChance
// (hanabi_game.cc:106-112: std::discrete_distribution over doubles count/deck_size) -> ApplyMove(kDeal) (:229-241)
__device__ void deal_random(const EnvCfg& g, St& s, int& cur, MtBatch& rng) {
  const int ncards = g.C * g.R;
  const double total = (double)s.deck_total();
  // a card type has 1..3 copies left (2-bit deck counters): its probability count/total takes three values, so the
  // 2 x 25 fp64 divisions of the two passes below collapse to 2 x 3 with bit-identical quotients
  const double q1 = 1.0 / total, q2 = 2.0 / total, q3 = 3.0 / total;
  // the two deck words (16 + 9 counters) in registers
  const uint32_t d0 = s.word(25), d1 = s.word(26);
#define DECK(uid) (int)((((uid) < 16 ? d0 : d1) >> ((((uid) & 15)) * 2)) & 3u)
  int n = 0, only = 0;
  double sum = 0.0;  // std::accumulate(probabilities, 0.0) in chance-uid order
  for (int uid = 0; uid < ncards; ++uid) {
    const int cnt = DECK(uid);
    if (cnt == 0) continue;
    sum += cnt == 1 ? q1 : (cnt == 2 ? q2 : q3);  // ChanceOutcomeProb (:277-280)
    only = uid;
    ++n;
  }
  int pick = only;
  if (n >= 2) {  // with < 2 outcomes libstdc++ returns index 0 WITHOUT drawing (random.tcc:2660-2664, 2704-2705)
    // generate_canonical<double,53>: two 32-bit draws, low word first (random.tcc:3348-3380)
    const double lo = (double)mt_batch_next(rng);
    const double hi = (double)mt_batch_next(rng);
    double u = (lo + hi * 4294967296.0) / 18446744073709551616.0;
    if (u >= 1.0) u = 0x1.fffffffffffffp-1;
    // normalise, partial_sum, last := 1.0, lower_bound (random.tcc:2666-2676, 2710-2712)
    const double p1 = q1 / sum, p2 = q2 / sum, p3 = q3 / sum;
    double acc = 0.0;
    int seen = 0;
    for (int uid = 0; uid < ncards; ++uid) {
      const int cnt = DECK(uid);
      if (cnt == 0) continue;
      const double p = cnt == 1 ? p1 : (cnt == 2 ? p2 : p3);
      acc = (seen == 0) ? p : acc + p;
      ++seen;
      const double cp = (seen == n) ? 1.0 : acc;
      if (!(cp < u)) {
        pick = uid;
        break;
      }
    }
  }
#undef DECK
  const int to = player_to_deal(g, s);
  const int slot = s.hand_n(to);
  const int color = pick / g.R, rank = pick % g.R;
  // fresh CardKnowledge: everything plausible, nothing hinted (hanabi_hand.cc:24-27, 44-45)
  s.word(to * g.H + slot) = (uint32_t)color | ((uint32_t)rank << 3) | (((1u << g.C) - 1u) << 6) | (((1u << g.R) - 1u) << 11);
  s.set_hand_n(to, slot + 1);
  s.set_deck(pick, s.deck(pick) - 1);
  s.set_deck_total(s.deck_total() - 1);
  // AdvanceToNextPlayer (hanabi_state.cc:104-111)
  if (s.deck_total() != 0 && player_to_deal(g, s) >= 0) {
    cur = -1;
  } else {
    cur = s.next();
    s.set_next((cur + 1) % g.P);
  }
}

struct Move { int type, card_index, target_offset, color, rank; };

__device__ __forceinline__ Move decode_move(const EnvCfg& g, int uid) {  // hanabi_game.cc:159-183
  Move m = {0, -1, -1, -1, -1};
  if (uid < 0 || uid >= g.num_moves) return m;
  if (uid < g.H) { m.type = MV_DISCARD; m.card_index = uid; return m; }
  uid -= g.H;
  if (uid < g.H) { m.type = MV_PLAY; m.card_index = uid; return m; }
  uid -= g.H;
  if (uid < (g.P - 1) * g.C) { m.type = MV_REVEAL_COLOR; m.target_offset = 1 + uid / g.C; m.color = uid % g.C; return m; }
  uid -= (g.P - 1) * g.C;
  m.type = MV_REVEAL_RANK; m.target_offset = 1 + uid / g.R; m.rank = uid % g.R;
  return m;
}

__device__ __forceinline__ bool move_is_legal(const EnvCfg& g, const St& s, int cur, const Move& m) {  // hanabi_state.cc:166-219
  switch (m.type) {
    case MV_DISCARD:
      return s.info() < g.max_info && m.card_index < s.hand_n(cur);
    case MV_PLAY:
      return m.card_index < s.hand_n(cur);
    case MV_REVEAL_COLOR:
    case MV_REVEAL_RANK: {
      if (s.info() <= 0) return false;
      if (m.target_offset < 1 || m.target_offset >= g.P) return false;
      const int t = (cur + m.target_offset) % g.P;
      const int n = s.hand_n(t);
      for (int i = 0; i < n; ++i) {
        const uint32_t c = s.word(t * g.H + i);
        if (m.type == MV_REVEAL_COLOR ? card_color(c) == m.color : card_rank(c) == m.rank) return true;
      }
      return false;
    }
    default:
      return false;
  }
}

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
    cur = s.cur();
    const int last_score = env_score(g, s);
    const Move m = decode_move(g, actions[env]);
    if (!move_is_legal(g, s, cur, m)) {  // reference: REQUIRE(MoveIsLegal(move)) -> abort (hanabi_state.cc:222)
      status[env] = HZ_ENV_ILLEGAL_MOVE;
      reward[env] = 0;
      done[env] = (uint8_t)(env_end_status(g, s) != 0);
      score_out[env] = last_score;
      return;
    }
    status[env] = HZ_ENV_OK;
    // ApplyMove (hanabi_state.cc:221-275)
    if (s.deck_total() == 0) s.set_turns(s.turns() > 0 ? s.turns() - 1 : 0);
    uint32_t lm = (uint32_t)cur | ((uint32_t)m.type << 3);
    const int p = cur;
    if (m.type == MV_DISCARD || m.type == MV_PLAY) {
      const uint32_t c = s.word(p * g.H + m.card_index);
      const int cc = card_color(c), cr = card_rank(c);
      lm |= ((uint32_t)m.card_index << 6) | ((uint32_t)cc << 20) | ((uint32_t)cr << 23);
      bool to_discard = true;
      if (m.type == MV_DISCARD) {
        if (s.info() < g.max_info) {  // IncrementInformationTokens (:113-120)
          s.set_info(s.info() + 1);
          lm |= 1u << 19;
        }
      } else if (cr == s.fw(cc)) {  // AddToFireworks (:132-144)
        s.set_fw(cc, cr + 1);
        lm |= 1u << 18;
        to_discard = false;
        if (cr + 1 == g.R && s.info() < g.max_info) {
          s.set_info(s.info() + 1);
          lm |= 1u << 19;
        }
      } else {
        s.set_life(s.life() - 1);
      }
      if (to_discard) s.set_disc(cc * g.R + cr, s.disc(cc * g.R + cr) + 1);
      // HanabiHand::RemoveFromHand (hanabi_hand.cc:87-94): younger cards slide down
      const int n = s.hand_n(p);
      for (int i = m.card_index; i + 1 < n; ++i) s.word(p * g.H + i) = s.word(p * g.H + i + 1);
      s.word(p * g.H + n - 1) = 0;
      s.set_hand_n(p, n - 1);
    } else {
      s.set_info(s.info() - 1);
      const int t = (p + m.target_offset) % g.P;
      const int n = s.hand_n(t);
      uint32_t reveal = 0;
      for (int i = 0; i < n; ++i) {  // HandColorBitmask/HandRankBitmask (:27-50) + RevealColor/RevealRank (hanabi_hand.cc:96-126)
        uint32_t c = s.word(t * g.H + i);
        if (m.type == MV_REVEAL_COLOR) {
          if (card_color(c) == m.color) {
            reveal |= 1u << i;
            c = (c & ~(31u << 6)) | ((1u << m.color) << 6) | (1u << 16);
          } else {
            c &= ~((1u << m.color) << 6);
          }
        } else {
          if (card_rank(c) == m.rank) {
            reveal |= 1u << i;
            c = (c & ~(31u << 11)) | ((1u << m.rank) << 11) | (1u << 17);
          } else {
            c &= ~((1u << m.rank) << 11);
          }
        }
        s.word(t * g.H + i) = c;
      }
      lm |= ((uint32_t)m.target_offset << 9) | (reveal << 26);
      if (m.type == MV_REVEAL_COLOR) lm |= (uint32_t)m.color << 12;
      else lm |= (uint32_t)m.rank << 15;
    }
    s.word(31) = lm;
    s.set(30, 21, 1, 1u);
    // AdvanceToNextPlayer, then rl_env.py:422-423: deal while the chance player is to act (also at game end)
    if (s.deck_total() != 0 && player_to_deal(g, s) >= 0) {
      cur = -1;
    } else {
      cur = s.next();
      s.set_next((cur + 1) % g.P);
    }
    while (cur == -1) deal_random(g, s, cur, rng);
    s.set_cur(cur);
    const int sc = env_score(g, s);
    reward[env] = sc - last_score;
    done[env] = (uint8_t)(env_end_status(g, s) != 0);
    score_out[env] = sc;
  }
  mt_batch_commit(rng);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    uint4 v;
    v.x = s.word(4 * q + 0); v.y = s.word(4 * q + 1); v.z = s.word(4 * q + 2); v.w = s.word(4 * q + 3);
    gs[q] = v;
  }
}

// ---- reset, one WAVE per env --------------------------------------------------------------------------------
// A reset deals players * hand_size cards one after the other; each deal is a std::discrete_distribution draw over the
// remaining card types (two passes over up to 25 fp64 probabilities, summed in order) -- ~80 us when one lane does it
// all (k_env_rules mode 0), whatever the number of envs.  Here a wave owns the env: lane uid holds the count of card
// type uid, the order-sensitive fp64 sums walk the ballot of present types with v_readlane (each deal still adds in the
// reference's order), and all mt19937 words of the 2 * players * hand_size draws are regenerated by the lanes in
// parallel (every input of the recurrence is an OLD word as long as fewer than 227 are drawn: see mt_batch_begin).
// Same state bits and the same generator state as the lane-per-env path (tests/test_hip_env.py).
__device__ __forceinline__ void env_reset_wave_body(const EnvCfg& g, uint32_t* __restrict__ state, uint32_t* __restrict__ mt,
                                                    const uint8_t* __restrict__ mask, int env, int lane, uint32_t* st) {
  if (env >= g.N) return;
  if (mask != nullptr && mask[env] == 0) return;
  const int ncards = g.C * g.R, deals = g.P * g.H, n = 2 * deals;
#define MT(k) mt[(size_t)(k) * g.N + env]
  int idx0 = (int)MT(624);
  if (idx0 >= 624) idx0 = 0;
  idx0 = __builtin_amdgcn_readfirstlane(idx0);
  const uint32_t wk = lane <= n ? MT(mt_wrap(idx0 + lane)) : 0u;
  const uint32_t fk = lane < n ? MT(mt_wrap(mt_wrap(idx0 + lane) + 397)) : 0u;
  const uint32_t wk1 = (uint32_t)__shfl_down((int)wk, 1);
  const uint32_t y = (wk & 0x80000000u) | (wk1 & 0x7fffffffu);
  const uint32_t neww = fk ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  uint32_t out = neww;  // lane k: the k-th draw of this call
  out ^= (out >> 11);
  out ^= (out << 7) & 0x9d2c5680u;
  out ^= (out << 15) & 0xefc60000u;
  out ^= (out >> 18);

  if (lane < 32) st[lane] = 0u;
  int cnt = lane < ncards ? g.inst[lane % g.R] : 0;  // HanabiDeck ctor (hanabi_state.cc:53-64): lane = color * R + rank
  int total = 0;
  for (int r = 0; r < g.R; ++r) total += g.inst[r];
  total *= g.C;
  int pos = 0;
  for (int deal = 0; deal < deals; ++deal) {
    // ApplyRandomChance (hanabi_state.cc:282-286) -> PickRandomChance (hanabi_game.cc:106-112): see deal_random
    const double tot = (double)total;
    // (2.0 / tot == 2 * (1.0 / tot) bit for bit: scaling by two commutes with the rounding of the quotient)
    const double q1 = 1.0 / tot, q2 = q1 + q1, q3 = 3.0 / tot;
    const uint64_t present = __ballot(cnt > 0);
    const int n_out = __popcll((unsigned long long)present);
    double sum = 0.0;
    int pick = 0;
    for (uint64_t m = present; m;) {
      const int uid = __ffsll((unsigned long long)m) - 1;
      m &= m - 1;
      const int c = __builtin_amdgcn_readlane(cnt, uid);
      sum += c == 1 ? q1 : (c == 2 ? q2 : q3);
      pick = uid;
    }
    if (n_out >= 2) {
      const double lo = (double)(uint32_t)__builtin_amdgcn_readlane((int)out, pos);
      const double hi = (double)(uint32_t)__builtin_amdgcn_readlane((int)out, pos + 1);
      pos += 2;
      double u = (lo + hi * 4294967296.0) / 18446744073709551616.0;
      if (u >= 1.0) u = 0x1.fffffffffffffp-1;
      const double p1 = q1 / sum, p2 = p1 + p1, p3 = q3 / sum;
      double acc = 0.0;
      int seen = 0;
      for (uint64_t m = present; m;) {
        const int uid = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
        const int c = __builtin_amdgcn_readlane(cnt, uid);
        const double p = c == 1 ? p1 : (c == 2 ? p2 : p3);
        acc = (seen == 0) ? p : acc + p;
        ++seen;
        const double cp = (seen == n_out) ? 1.0 : acc;
        if (!(cp < u)) {
          pick = uid;
          break;
        }
      }
    }
    pick = __builtin_amdgcn_readfirstlane(pick);
    // the chance player deals to the first player whose hand is not full (hanabi_state.cc:157-164): hands fill in order
    if (lane == 0)
      st[deal] = (uint32_t)(pick / g.R) | ((uint32_t)(pick % g.R) << 3) | (((1u << g.C) - 1u) << 6) |
                 (((1u << g.R) - 1u) << 11);  // fresh CardKnowledge (hanabi_hand.cc:24-27, 44-45); deal = player * H + slot
    if (lane == pick) cnt -= 1;
    total -= 1;
  }
  // deck counters (2 bits per card type, 16 per word), hand sizes, tokens, players: the bits k_env_rules mode 0 leaves
  if (lane < ncards && cnt != 0) atomicOr(&st[25 + (lane >> 4)], (uint32_t)cnt << ((lane & 15) * 2));
  if (lane == 0) {
    uint32_t w30 = (uint32_t)total << 15;
    for (int pl = 0; pl < g.P; ++pl) w30 |= (uint32_t)g.H << (3 * pl);
    st[30] = w30;
    // info, life, current player 0, next player 1 % P (GetSampledStartPlayer without random start, then
    // AdvanceToNextPlayer once the hands are full), turns_to_play = P
    st[29] = ((uint32_t)g.max_info << 15) | ((uint32_t)g.max_life << 19) | (0u << 21) | ((uint32_t)(1 % g.P) << 24) |
             ((uint32_t)g.P << 27);
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < 32) state[(size_t)env * 32 + lane] = st[lane];
  if (lane < pos) MT(mt_wrap(idx0 + lane)) = neww;
  if (lane == 0) {
    const int idx = idx0 + pos;
    MT(624) = (uint32_t)(idx > 624 ? idx - 624 : idx);
  }
#undef MT
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

// ---- observation kernel: one wave per env ----------------------------------------------------------------
__device__ __forceinline__ void or_bits(uint32_t* bits, int off, uint32_t value, int nbits) {
  if (value == 0) return;
  const int w = off >> 5, sh = off & 31;
  atomicOr(&bits[w], value << sh);
  if (sh + nbits > 32) atomicOr(&bits[w + 1], value >> (32 - sh));
}
__device__ __forceinline__ void or_ones(uint32_t* bits, int off, int n) {  // thermometer of n ones
  while (n > 0) {
    const int k = n > 32 ? 32 : n;
    or_bits(bits, off, k == 32 ? 0xffffffffu : ((1u << k) - 1u), k);
    off += k;
    n -= k;
  }
}

#define OBS_WORDS 48  // >= ceil((125 + 1280 + 5) / 32) + 1
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
  const int C = g.C, R = g.R, P = g.P, H = g.H, bpc = g.bpc;
  const int obs = s.cur();  // the observing player is the player to act (rl_env.py:253-263)
  const int base = (mdp == HZ_MDP_GLOBAL) ? g.own_len : 0;  // canonical vector starts after the own-hand block
  const int D = base + g.obs_len + P;

  if (lane < P * H) {
    // one card slot per lane: EncodeOwnHand (canonical_encoders.cc:465-486), EncodeHands (:66-109),
    // EncodeCardKnowledge (:370-423)
    const int p = lane / H, i = lane % H;
    if (i < s.hand_n(p)) {
      const uint32_t c = s.word(lane);
      const int rel = (p - obs + P) % P;  // hanabi_observation.cc:60-64
      const int cidx = card_color(c) * R + card_rank(c);
      if (rel == 0) {
        if (mdp == HZ_MDP_GLOBAL) or_bits(bits, i * bpc + cidx, 1u, 1);
      } else {
        or_bits(bits, base + ((rel - 1) * H + i) * bpc + cidx, 1u, 1);
      }
      const int ko = base + g.off_know + (rel * H + i) * (bpc + C + R);
      const uint32_t cpl = (c >> 6) & 31u, rpl = (c >> 11) & 31u;
      uint32_t grid = 0;
      for (int col = 0; col < C; ++col)
        if ((cpl >> col) & 1u) grid |= rpl << (col * R);
      or_bits(bits, ko, grid, bpc);
      if ((c >> 16) & 1u) or_bits(bits, ko + bpc + card_color(c), 1u, 1);
      if ((c >> 17) & 1u) or_bits(bits, ko + bpc + C + card_rank(c), 1u, 1);
    }
  } else if (lane == 25) {  // missing-card flags (:99-104)
    uint32_t f = 0;
    for (int rel = 0; rel < P; ++rel)
      if (s.hand_n((obs + rel) % P) < H) f |= 1u << rel;
    or_bits(bits, base + (P - 1) * H * bpc, f, P);
  } else if (lane == 26) {  // deck thermometer (:136-140)
    or_ones(bits, base + g.off_board, s.deck_total());
  } else if (lane == 27) {  // fireworks one-hot per colour (:142-151)
    const int o = base + g.off_board + (g.max_deck - H * P);
    for (int c = 0; c < C; ++c)
      if (s.fw(c) > 0) or_bits(bits, o + c * R + s.fw(c) - 1, 1u, 1);
  } else if (lane == 28) {  // info and life thermometers (:153-167)
    const int o = base + g.off_board + (g.max_deck - H * P) + C * R;
    or_ones(bits, o, s.info());
    or_ones(bits, o + g.max_info, s.life());
  } else if (lane == 29) {  // EncodeLastAction (:240-342) on the most recent non-deal move
    if (s.has_last()) {
      const uint32_t lm = s.word(31);
      const int lp = (int)(lm & 7u), type = (int)((lm >> 3) & 7u);
      const int rel_player = (lp - obs + P) % P;  // hanabi_observation.cc:33-48
      int o = base + g.off_last;
      or_bits(bits, o + rel_player, 1u, 1);
      o += P;
      or_bits(bits, o + (type == MV_PLAY ? 0 : type == MV_DISCARD ? 1 : type == MV_REVEAL_COLOR ? 2 : 3), 1u, 1);
      o += 4;
      const bool is_reveal = type == MV_REVEAL_COLOR || type == MV_REVEAL_RANK;
      const bool is_card = type == MV_PLAY || type == MV_DISCARD;
      if (is_reveal) or_bits(bits, o + (rel_player + (int)((lm >> 9) & 7u)) % P, 1u, 1);
      o += P;
      if (type == MV_REVEAL_COLOR) or_bits(bits, o + (int)((lm >> 12) & 7u), 1u, 1);
      o += C;
      if (type == MV_REVEAL_RANK) or_bits(bits, o + (int)((lm >> 15) & 7u), 1u, 1);
      o += R;
      if (is_reveal) or_bits(bits, o, (lm >> 26) & ((1u << H) - 1u), H);
      o += H;
      if (is_card) or_bits(bits, o + (int)((lm >> 6) & 7u), 1u, 1);
      o += H;
      if (is_card) or_bits(bits, o + (int)((lm >> 20) & 7u) * R + (int)((lm >> 23) & 7u), 1u, 1);
      o += bpc;
      if (type == MV_PLAY) or_bits(bits, o, (lm >> 18) & 3u, 2);
    }
  } else if (lane == 30) {  // agent_turn one-hot, absolute player id (rl_env.py:254-255)
    or_bits(bits, base + g.obs_len + obs, 1u, 1);
  } else if (lane >= 32 && lane < 32 + C * R) {  // EncodeDiscards (:192-215): one (colour, rank) per lane
    const int idx = lane - 32, c = idx / R, r = idx % R;
    or_ones(bits, base + g.off_disc + c * g.per_color + g.inst_prefix[r], s.disc(idx));
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  if (legal_out != nullptr && lane < g.num_moves)  // LegalMoves(observer) (hanabi_state.cc:288-304)
    legal_out[(size_t)env * g.num_moves + lane] = (uint8_t)move_is_legal(g, s, obs, decode_move(g, lane));
  if (packed_out != nullptr) {
    const int nw = (D + 31) >> 5;
    if (lane < nw) packed_out[(size_t)env * nw + lane] = bits[lane];
  }
  if (obs_out != nullptr) {
    for (int j = lane; j < D; j += 64) {
      const uint32_t b = (bits[j >> 5] >> (j & 31)) & 1u;
      const size_t o = (size_t)env * (size_t)stride + j;
      if (dtype == HZ_OBS_U8) ((uint8_t*)obs_out)[o] = (uint8_t)b;
      else if (dtype == HZ_OBS_F32) ((float*)obs_out)[o] = b ? 1.0f : 0.0f;
      else if (dtype == HZ_OBS_BF16) ((uint16_t*)obs_out)[o] = b ? 0x3f80u : 0u;
      else ((uint16_t*)obs_out)[o] = b ? 0x3c00u : 0u;
    }
  }
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
