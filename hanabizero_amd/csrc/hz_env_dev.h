// hz_env_dev.h -- device code of the Hanabi env (see hz_env.hip for the layout and what it replaces), shared by the stand-alone
// env kernels (hz_env.hip) and the fused move tail of the self-play actor (hz_movetail.hip).
#pragma once
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

// ---- one move of one game: rl_env.py:418-442 on the state in `s` (LDS, any stride), in three parts so that the chance outcome
// in the middle can be one lane's loop (deal_random: k_env_rules, a lane per env) or a wave's work (deal_pick_wave: the fused move
// tail, a wave per env).
// (1) ApplyMove without the deal.  Returns HZ_ENV_OK, or HZ_ENV_ILLEGAL_MOVE with the state untouched (the reference aborts).
__device__ __forceinline__ int env_step_apply(const EnvCfg& g, St& s, int action, int& last_score) {
  const int cur = s.cur();
  last_score = env_score(g, s);
  const Move m = decode_move(g, action);
  if (!move_is_legal(g, s, cur, m)) {  // reference: REQUIRE(MoveIsLegal(move)) -> abort (hanabi_state.cc:222)
    return HZ_ENV_ILLEGAL_MOVE;
  }
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
  return HZ_ENV_OK;
}
// (2) does the chance player act next (hanabi_state.cc:104-111)?  At most one card is missing after a move.
__device__ __forceinline__ bool env_step_needs_deal(const EnvCfg& g, const St& s) {
  return s.deck_total() != 0 && player_to_deal(g, s) >= 0;
}
// ... its move, given the card type drawn (ApplyMove(kDeal), hanabi_state.cc:229-241)
__device__ __forceinline__ void env_apply_deal(const EnvCfg& g, St& s, int pick) {
  const int to = player_to_deal(g, s);
  const int slot = s.hand_n(to);
  // fresh CardKnowledge: everything plausible, nothing hinted (hanabi_hand.cc:24-27, 44-45)
  s.word(to * g.H + slot) = (uint32_t)(pick / g.R) | ((uint32_t)(pick % g.R) << 3) | (((1u << g.C) - 1u) << 6) | (((1u << g.R) - 1u) << 11);
  s.set_hand_n(to, slot + 1);
  s.set_deck(pick, s.deck(pick) - 1);
  s.set_deck_total(s.deck_total() - 1);
}
// (3) the next player takes over (AdvanceToNextPlayer once nobody is to be dealt to); reward, done, score
__device__ __forceinline__ void env_step_finish(const EnvCfg& g, St& s, int last_score, int& reward, int& done, int& score) {
  const int cur = s.next();
  s.set_next((cur + 1) % g.P);
  s.set_cur(cur);
  const int sc = env_score(g, s);
  reward = sc - last_score;
  done = (int)(env_end_status(g, s) != 0);
  score = sc;
}

// one lane does it all; `rng` prepared for 2 draws
__device__ __forceinline__ int env_step_lane(const EnvCfg& g, St& s, MtBatch& rng, int action, int& reward, int& done,
                                             int& score) {
  int last_score;
  if (env_step_apply(g, s, action, last_score) != HZ_ENV_OK) {
    reward = 0;
    done = (int)(env_end_status(g, s) != 0);
    score = last_score;
    return HZ_ENV_ILLEGAL_MOVE;
  }
  // AdvanceToNextPlayer, then rl_env.py:422-423: deal while the chance player is to act (also at game end)
  int cur;
  if (env_step_needs_deal(g, s)) {
    cur = -1;
  } else {
    cur = s.next();
    s.set_next((cur + 1) % g.P);
  }
  while (cur == -1) deal_random(g, s, cur, rng);
  s.set_cur(cur);
  const int sc = env_score(g, s);
  reward = sc - last_score;
  done = (int)(env_end_status(g, s) != 0);
  score = sc;
  return HZ_ENV_OK;
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffull), l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ---- one chance outcome, a whole wave at work ------------------------------------------------------------------
// PickRandomChance (hanabi_game.cc:106-112) as deal_random does it, with lane uid holding the count of card type uid (0 beyond
// the card types): which card type is dealt.  `n_out` = the number of types present (uniform); with fewer than two libstdc++
// draws nothing (random.tcc:2660-2664) and `u` is not looked at.
// The two order-sensitive fp64 sums (std::accumulate of the probabilities; partial_sum of the normalised ones) are chains of
// 25 dependent v_add_f64 and nothing else: term j comes off the chain by v_readlane at a constant lane (fully unrolled), every
// lane adds the same terms in the same order -- the total needs no lane of its own, and lane j's partial sum is what the chain
// holds after term j.  A skipped (absent) type enters as +0.0, which changes no sum.  (A systolic DPP
// pass had three dependent instructions per term, a ballot-walking readlane loop ~12 with a scalar-to-vector hazard each: a reset
// of 10 deals took 19 us on one wave, r03 tools/tail_profile.py.)  All 64 lanes active.
__device__ __forceinline__ int deal_pick_wave(int ncards, int cnt, int total, uint64_t present, int n_out, double u) {
  int pick = 63 - __clzll((unsigned long long)present);  // the only outcome, if there is one
  if (n_out >= 2) {
    const int lane = (int)(threadIdx.x & 63);
    const double tot = (double)total;
    // (2.0 / tot == 2 * (1.0 / tot) bit for bit: scaling by two commutes with the rounding of the quotient)
    const double q1 = 1.0 / tot, q2 = q1 + q1, q3 = 3.0 / tot;
    const double q = cnt == 0 ? 0.0 : (cnt == 1 ? q1 : (cnt == 2 ? q2 : q3));  // ChanceOutcomeProb (hanabi_state.cc:277-280)
    // (25 = the most card types there are; lanes beyond the game's own hold +0.0)
    double sum = 0.0;  // std::accumulate(probabilities, 0.0) in chance-uid order
#pragma unroll
    for (int j = 0; j < 25; ++j) sum += readlane_f64(q, j);
    // normalise, partial_sum, last := 1.0, lower_bound (random.tcc:2666-2676, 2710-2712)
    const double p1 = q1 / sum, p2 = p1 + p1, p3 = q3 / sum;
    const double pr = cnt == 0 ? 0.0 : (cnt == 1 ? p1 : (cnt == 2 ? p2 : p3));
    // one chain for all lanes; lane j keeps what the chain holds after term j (the compare is made afresh per call -- an opaque
    // copy of the lane index -- or the compiler keeps 25 lane masks alive across the caller's loop and spills them)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    double run = 0.0, acc = 0.0;
#pragma unroll
    for (int j = 0; j < 25; ++j) {
      run += readlane_f64(pr, j);
      acc = (lane_o == j) ? run : acc;
    }
    const double cp = (lane == pick) ? 1.0 : acc;
    const uint64_t hit = __ballot(cnt > 0 && !(cp < u));
    pick = __ffsll((unsigned long long)hit) - 1;
  }
  return __builtin_amdgcn_readfirstlane(pick);
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
    // ApplyRandomChance (hanabi_state.cc:282-286) -> PickRandomChance (hanabi_game.cc:106-112): see deal_random / deal_pick_wave
    const uint64_t present = __ballot(cnt > 0);
    const int n_out = __popcll((unsigned long long)present);
    double u = 0.0;
    if (n_out >= 2) {  // generate_canonical<double,53>: two 32-bit draws, low word first (random.tcc:3348-3380)
      const double lo = (double)(uint32_t)__builtin_amdgcn_readlane((int)out, pos);
      const double hi = (double)(uint32_t)__builtin_amdgcn_readlane((int)out, pos + 1);
      pos += 2;
      u = (lo + hi * 4294967296.0) / 18446744073709551616.0;
      if (u >= 1.0) u = 0x1.fffffffffffffp-1;
    }
    int pick = deal_pick_wave(ncards, cnt, total, present, n_out, u);
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
// one wave, the game's state in LDS (stride 1), `bits` zeroed and visible to the wave: sets the observation's bits
__device__ __forceinline__ void env_observe_bits(const EnvCfg& g, const St& s, uint32_t* bits, int lane, int mdp) {
  const int C = g.C, R = g.R, P = g.P, H = g.H, bpc = g.bpc;
  const int obs = s.cur();  // the observing player is the player to act (rl_env.py:253-263)
  const int base = (mdp == HZ_MDP_GLOBAL) ? g.own_len : 0;  // canonical vector starts after the own-hand block

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
}

// ... and writes it out (after a wave barrier): the row as obs_dtype elements, the bit-packed row, the legal-move mask
__device__ __forceinline__ void env_observe_emit(const EnvCfg& g, const St& s, const uint32_t* bits, int lane, int env, int mdp,
                                                 void* __restrict__ obs_out, int dtype, long long stride,
                                                 uint32_t* __restrict__ packed_out, uint8_t* __restrict__ legal_out) {
  const int obs = s.cur();
  const int D = ((mdp == HZ_MDP_GLOBAL) ? g.own_len : 0) + g.obs_len + g.P;
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

