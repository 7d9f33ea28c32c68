/* env_oracle.c -- TEST INFRASTRUCTURE (oracle), NOT product code.
 *
 * Plain-C, sequential CPU restatement of the reference Hanabi environment as the self-play
 * worker sees it: envs/hanabi/rl_env.py HanabiEnv.reset/step (:148-267, :292-442) over
 * envs/hanabi/hanabi_lib/{hanabi_game,hanabi_state,hanabi_hand,hanabi_observation,
 * canonical_encoders}.cc.  Each function cites the reference lines it follows.  Card deals use
 * oracle/mt_discrete.c (libstdc++ mt19937 + discrete_distribution restated).
 *
 * Pinned against the compiled reference by tests/golden/env_*.npz (tools/gen_golden.py drives
 * oracle/_ref/libpyhanabi.so) -- see tests/test_oracle_env.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mt_discrete.h"

#define MAXC 5 /* hanabi_card.h / util.h kMaxNumColors */
#define MAXR 5
#define MAXP 5
#define MAXH 5

enum { MV_INVALID = 0, MV_PLAY = 1, MV_DISCARD = 2, MV_REVEAL_COLOR = 3, MV_REVEAL_RANK = 4, MV_DEAL = 5 }; /* hanabi_move.h:34 */

typedef struct {
  int color, rank;      /* the card */
  int cplaus[MAXC];     /* ValueKnowledge::value_plausible_ for color (hanabi_hand.h:56) */
  int rplaus[MAXR];
  int chint, rhint;     /* ValueKnowledge::value_ : -1 = not hinted */
} ocard;

typedef struct {
  hzo_mt19937 rng;      /* HanabiGame::rng_ (hanabi_game.h:114): one per env, survives reset */
  int deck[MAXC * MAXR], deck_total;
  ocard hand[MAXP][MAXH];
  int hand_n[MAXP];
  int discard[MAXC * MAXR]; /* counts; the encoder only needs counts (canonical_encoders.cc:198-201) */
  int fireworks[MAXC];
  int info, life, cur, next_player, turns_to_play;
  /* most recent non-deal HanabiHistoryItem (hanabi_history_item.h:28-58) */
  int has_last, lm_player, lm_type, lm_card_index, lm_target_offset, lm_color, lm_rank;
  int lm_scored, lm_info_token, lm_card_color, lm_card_rank, lm_reveal_mask;
} oenv;

typedef struct {
  int N, C, R, P, H, max_info, max_life;
  int num_moves, obs_len, own_len, bpc, max_deck;
  oenv* e;
} oenvs;

static int num_instances(const oenvs* g, int rank) { /* hanabi_game.cc:126-136 */
  if (rank == 0) return 3;
  if (rank == g->R - 1) return 1;
  return 2;
}

void* hzo_env_new(int N, int colors, int ranks, int players, int hand_size, int max_info, int max_life,
                  const int32_t* seeds) {
  oenvs* g = (oenvs*)calloc(1, sizeof(oenvs));
  g->N = N; g->C = colors; g->R = ranks; g->P = players;
  g->H = hand_size > 0 ? hand_size : (players < 4 ? 5 : 4); /* hanabi_game.cc:147-152 */
  g->max_info = max_info; g->max_life = max_life;
  g->bpc = colors * ranks;
  int per_color = 0;
  for (int r = 0; r < ranks; ++r) per_color += num_instances(g, r);
  g->max_deck = per_color * colors;
  /* hanabi_game.cc:69-72 MaxMoves */
  g->num_moves = 2 * g->H + (players - 1) * colors + (players - 1) * ranks;
  /* canonical_encoders.cc:52-55, 111-116, 173, 217-227, 344-347 */
  int hands = (players - 1) * g->H * g->bpc + players;
  int board = g->max_deck - players * g->H + colors * ranks + max_info + max_life;
  int disc = g->max_deck;
  int last = players + 4 + players + colors + ranks + g->H + g->H + g->bpc + 2;
  int know = players * g->H * (g->bpc + colors + ranks);
  g->obs_len = hands + board + disc + last + know;
  g->own_len = g->H * g->bpc; /* canonical_encoders.cc:57-59 */
  g->e = (oenv*)calloc(N, sizeof(oenv));
  for (int i = 0; i < N; ++i) hzo_mt_seed(&g->e[i].rng, (uint32_t)seeds[i]); /* hanabi_game.cc:48-51 */
  return g;
}

void hzo_env_free(void* h) { oenvs* g = (oenvs*)h; free(g->e); free(g); }

void hzo_env_dims(void* h, int* num_moves, int* obs_len, int* own_len, int* players) {
  oenvs* g = (oenvs*)h;
  *num_moves = g->num_moves; *obs_len = g->obs_len; *own_len = g->own_len; *players = g->P;
}

static int player_to_deal(const oenvs* g, const oenv* e) { /* hanabi_state.cc:157-164 */
  for (int i = 0; i < g->P; ++i)
    if (e->hand_n[i] < g->H) return i;
  return -1;
}

static void advance(const oenvs* g, oenv* e) { /* hanabi_state.cc:104-111 */
  if (e->deck_total != 0 && player_to_deal(g, e) >= 0) {
    e->cur = -1;
  } else {
    e->cur = e->next_player;
    e->next_player = (e->cur + 1) % g->P;
  }
}

/* hanabi_state.cc:282-286 ApplyRandomChance -> :313-325 ChanceOutcomes -> hanabi_game.cc:106-112
 * -> ApplyMove(kDeal) hanabi_state.cc:229-241 */
static void deal_random(const oenvs* g, oenv* e) {
  int uids[MAXC * MAXR];
  double probs[MAXC * MAXR];
  int n = 0;
  for (int uid = 0; uid < g->C * g->R; ++uid) {
    if (e->deck[uid] == 0) continue; /* MoveIsLegal(kDeal), hanabi_state.cc:168-175 */
    uids[n] = uid;
    probs[n] = (double)e->deck[uid] / (double)e->deck_total; /* :277-280 */
    ++n;
  }
  int pick = uids[hzo_discrete(&e->rng, probs, n)];
  /* ApplyMove: deck is not empty here, so turns_to_play_ is untouched (:223-225) */
  int to = player_to_deal(g, e);
  ocard* c = &e->hand[to][e->hand_n[to]++]; /* HanabiHand::AddCard, hanabi_hand.cc:80-85 */
  c->color = pick / g->R;
  c->rank = pick % g->R;
  for (int k = 0; k < g->C; ++k) c->cplaus[k] = 1; /* fresh CardKnowledge, hanabi_hand.cc:24-27,44-45 */
  for (int k = 0; k < g->R; ++k) c->rplaus[k] = 1;
  c->chint = c->rhint = -1;
  e->deck[pick]--;
  e->deck_total--;
  advance(g, e);
}

/* rl_env.py:249-252: new_initial_state (hanabi_state.cc:90-102) then deal until a player is to act */
static void reset_one(const oenvs* g, oenv* e) {
  e->deck_total = 0;
  for (int c = 0; c < g->C; ++c)
    for (int r = 0; r < g->R; ++r) { /* HanabiDeck ctor, hanabi_state.cc:53-64 */
      e->deck[c * g->R + r] = num_instances(g, r);
      e->deck_total += num_instances(g, r);
    }
  memset(e->hand_n, 0, sizeof(e->hand_n));
  memset(e->discard, 0, sizeof(e->discard));
  memset(e->fireworks, 0, sizeof(e->fireworks));
  e->info = g->max_info;
  e->life = g->max_life;
  e->cur = -1;
  e->next_player = 0; /* GetSampledStartPlayer with random_start_player=false, hanabi_game.cc:138-145 */
  e->turns_to_play = g->P;
  e->has_last = 0;
  while (e->cur == -1) deal_random(g, e);
}

void hzo_env_reset(void* h, const uint8_t* mask) {
  oenvs* g = (oenvs*)h;
  for (int i = 0; i < g->N; ++i)
    if (!mask || mask[i]) reset_one(g, &g->e[i]);
}

static int score(const oenvs* g, const oenv* e) { /* hanabi_state.cc:359-364 */
  if (e->life <= 0) return 0;
  int s = 0;
  for (int c = 0; c < g->C; ++c) s += e->fireworks[c];
  return s;
}

static int end_status(const oenvs* g, const oenv* e) { /* hanabi_state.cc:366-377 */
  if (e->life < 1) return 1;
  if (score(g, e) >= g->C * g->R) return 3;
  if (e->turns_to_play <= 0) return 2;
  return 0;
}

/* hanabi_game.cc:159-183 ConstructMove */
static void decode_move(const oenvs* g, int uid, int* type, int* card_index, int* target_offset, int* color,
                        int* rank) {
  *card_index = *target_offset = *color = *rank = -1;
  if (uid < 0 || uid >= g->num_moves) { *type = MV_INVALID; return; }
  if (uid < g->H) { *type = MV_DISCARD; *card_index = uid; return; }
  uid -= g->H;
  if (uid < g->H) { *type = MV_PLAY; *card_index = uid; return; }
  uid -= g->H;
  if (uid < (g->P - 1) * g->C) { *type = MV_REVEAL_COLOR; *target_offset = 1 + uid / g->C; *color = uid % g->C; return; }
  uid -= (g->P - 1) * g->C;
  *type = MV_REVEAL_RANK; *target_offset = 1 + uid / g->R; *rank = uid % g->R;
}

/* hanabi_state.cc:166-219 MoveIsLegal (player moves) */
static int move_is_legal(const oenvs* g, const oenv* e, int uid) {
  int type, ci, to, color, rank;
  decode_move(g, uid, &type, &ci, &to, &color, &rank);
  switch (type) {
    case MV_DISCARD:
      if (e->info >= g->max_info) return 0;
      if (ci >= e->hand_n[e->cur]) return 0;
      return 1;
    case MV_PLAY:
      return ci < e->hand_n[e->cur];
    case MV_REVEAL_COLOR:
    case MV_REVEAL_RANK: {
      if (e->info <= 0) return 0; /* HintingIsLegal :146-155 */
      if (to < 1 || to >= g->P) return 0;
      int t = (e->cur + to) % g->P;
      for (int i = 0; i < e->hand_n[t]; ++i)
        if (type == MV_REVEAL_COLOR ? e->hand[t][i].color == color : e->hand[t][i].rank == rank) return 1;
      return 0;
    }
    default:
      return 0;
  }
}

static void remove_from_hand(oenv* e, int p, int idx) { /* hanabi_hand.cc:87-94 */
  for (int i = idx; i + 1 < e->hand_n[p]; ++i) e->hand[p][i] = e->hand[p][i + 1];
  e->hand_n[p]--;
}

/* hanabi_state.cc:221-275 ApplyMove for a player move; returns 0 ok, -1 illegal (reference: abort) */
static int apply_move(const oenvs* g, oenv* e, int uid) {
  if (e->cur < 0 || !move_is_legal(g, e, uid)) return -1;
  int type, ci, to, color, rank;
  decode_move(g, uid, &type, &ci, &to, &color, &rank);
  if (e->deck_total == 0) --e->turns_to_play;
  e->has_last = 1;
  e->lm_player = e->cur; e->lm_type = type; e->lm_card_index = ci; e->lm_target_offset = to;
  e->lm_color = color; e->lm_rank = rank;
  e->lm_scored = 0; e->lm_info_token = 0; e->lm_card_color = -1; e->lm_card_rank = -1; e->lm_reveal_mask = 0;
  int p = e->cur;
  switch (type) {
    case MV_DISCARD: {
      if (e->info < g->max_info) { ++e->info; e->lm_info_token = 1; } /* :113-120 */
      e->lm_card_color = e->hand[p][ci].color;
      e->lm_card_rank = e->hand[p][ci].rank;
      e->discard[e->lm_card_color * g->R + e->lm_card_rank]++;
      remove_from_hand(e, p, ci);
      break;
    }
    case MV_PLAY: {
      int cc = e->hand[p][ci].color, cr = e->hand[p][ci].rank;
      e->lm_card_color = cc; e->lm_card_rank = cr;
      if (cr == e->fireworks[cc]) { /* AddToFireworks :132-144, CardPlayableOnFireworks :306-311 */
        ++e->fireworks[cc];
        e->lm_scored = 1;
        if (e->fireworks[cc] == g->R && e->info < g->max_info) { ++e->info; e->lm_info_token = 1; }
      } else {
        --e->life;
        e->discard[cc * g->R + cr]++;
      }
      remove_from_hand(e, p, ci);
      break;
    }
    case MV_REVEAL_COLOR: {
      --e->info;
      int t = (p + to) % g->P;
      for (int i = 0; i < e->hand_n[t]; ++i) { /* :27-37 bitmask; hanabi_hand.cc:96-110 RevealColor */
        ocard* c = &e->hand[t][i];
        if (c->color == color) {
          e->lm_reveal_mask |= 1 << i;
          c->chint = color; /* ApplyIsValueHint hanabi_hand.cc:29-36 */
          for (int k = 0; k < g->C; ++k) c->cplaus[k] = (k == color);
        } else {
          c->cplaus[color] = 0; /* ApplyIsNotValueHint :38-42 */
        }
      }
      break;
    }
    case MV_REVEAL_RANK: {
      --e->info;
      int t = (p + to) % g->P;
      for (int i = 0; i < e->hand_n[t]; ++i) { /* :40-50; hanabi_hand.cc:112-126 RevealRank */
        ocard* c = &e->hand[t][i];
        if (c->rank == rank) {
          e->lm_reveal_mask |= 1 << i;
          c->rhint = rank;
          for (int k = 0; k < g->R; ++k) c->rplaus[k] = (k == rank);
        } else {
          c->rplaus[rank] = 0;
        }
      }
      break;
    }
  }
  advance(g, e);
  return 0;
}

/* rl_env.py:418-442: score delta reward, deal loop (runs even at a terminal state), done flag */
int hzo_env_step(void* h, const int32_t* actions, const uint8_t* mask, int32_t* reward, uint8_t* done,
                 int32_t* score_out) {
  oenvs* g = (oenvs*)h;
  int rc = 0;
  for (int i = 0; i < g->N; ++i) {
    if (mask && !mask[i]) continue;
    oenv* e = &g->e[i];
    int last = score(g, e);
    if (apply_move(g, e, actions[i]) != 0) { rc = -(i + 1); reward[i] = 0; done[i] = 0; score_out[i] = last; continue; }
    while (e->cur == -1) deal_random(g, e);
    int s = score(g, e);
    reward[i] = s - last;
    done[i] = end_status(g, e) != 0;
    score_out[i] = s;
  }
  return rc;
}

/* hanabi_observation.cc:52-96 (observer = current player) + canonical_encoders.cc:441-486 +
 * rl_env.py:256-263 / :429-434: share_obs = own_hand ++ canonical ++ onehot(cur_player).
 * out row length = own_len + obs_len + P (uint8 0/1); legal row length = num_moves. */
static void observe_one(const oenvs* g, const oenv* e, uint8_t* out, uint8_t* legal) {
  int C = g->C, R = g->R, P = g->P, H = g->H, bpc = g->bpc;
  int obs = e->cur;
  memset(out, 0, (size_t)(g->own_len + g->obs_len + P));
  /* EncodeOwnHand canonical_encoders.cc:465-486 */
  for (int i = 0; i < e->hand_n[obs]; ++i) out[i * bpc + e->hand[obs][i].color * R + e->hand[obs][i].rank] = 1;
  uint8_t* v = out + g->own_len;
  int off = 0;
  /* EncodeHands :66-109 */
  for (int rel = 1; rel < P; ++rel) {
    int p = (obs + rel) % P;
    for (int i = 0; i < e->hand_n[p]; ++i) v[off + i * bpc + e->hand[p][i].color * R + e->hand[p][i].rank] = 1;
    off += H * bpc;
  }
  for (int rel = 0; rel < P; ++rel)
    if (e->hand_n[(obs + rel) % P] < H) v[off + rel] = 1;
  off += P;
  /* EncodeBoard :127-171 */
  for (int i = 0; i < e->deck_total; ++i) v[off + i] = 1;
  off += g->max_deck - H * P;
  for (int c = 0; c < C; ++c) {
    if (e->fireworks[c] > 0) v[off + e->fireworks[c] - 1] = 1;
    off += R;
  }
  for (int i = 0; i < e->info; ++i) v[off + i] = 1;
  off += g->max_info;
  for (int i = 0; i < e->life; ++i) v[off + i] = 1;
  off += g->max_life;
  /* EncodeDiscards :192-215 */
  for (int c = 0; c < C; ++c)
    for (int r = 0; r < R; ++r) {
      for (int i = 0; i < e->discard[c * R + r]; ++i) v[off + i] = 1;
      off += num_instances(g, r);
    }
  /* EncodeLastAction :240-342 on GetLastNonDealMove(obs.LastMoves()) :34-41 */
  if (e->has_last) {
    int rel_player = (e->lm_player - obs + P) % P; /* hanabi_observation.cc:33-48 */
    int o = off;
    v[o + rel_player] = 1;
    o += P;
    switch (e->lm_type) {
      case MV_PLAY: v[o] = 1; break;
      case MV_DISCARD: v[o + 1] = 1; break;
      case MV_REVEAL_COLOR: v[o + 2] = 1; break;
      case MV_REVEAL_RANK: v[o + 3] = 1; break;
    }
    o += 4;
    int is_reveal = e->lm_type == MV_REVEAL_COLOR || e->lm_type == MV_REVEAL_RANK;
    int is_card = e->lm_type == MV_PLAY || e->lm_type == MV_DISCARD;
    if (is_reveal) v[o + (rel_player + e->lm_target_offset) % P] = 1;
    o += P;
    if (e->lm_type == MV_REVEAL_COLOR) v[o + e->lm_color] = 1;
    o += C;
    if (e->lm_type == MV_REVEAL_RANK) v[o + e->lm_rank] = 1;
    o += R;
    if (is_reveal)
      for (int i = 0; i < H; ++i)
        if (e->lm_reveal_mask & (1 << i)) v[o + i] = 1;
    o += H;
    if (is_card) v[o + e->lm_card_index] = 1;
    o += H;
    if (is_card) v[o + e->lm_card_color * R + e->lm_card_rank] = 1;
    o += bpc;
    if (e->lm_type == MV_PLAY) {
      if (e->lm_scored) v[o] = 1;
      if (e->lm_info_token) v[o + 1] = 1;
    }
  }
  off += P + 4 + P + C + R + H + H + bpc + 2;
  /* EncodeCardKnowledge :370-423 (hands_[0] keeps the observer's own knowledge) */
  for (int rel = 0; rel < P; ++rel) {
    int p = (obs + rel) % P;
    for (int i = 0; i < e->hand_n[p]; ++i) {
      const ocard* c = &e->hand[p][i];
      int o = off + i * (bpc + C + R);
      for (int col = 0; col < C; ++col)
        if (c->cplaus[col])
          for (int rk = 0; rk < R; ++rk)
            if (c->rplaus[rk]) v[o + col * R + rk] = 1;
      if (c->chint >= 0) v[o + bpc + c->chint] = 1;
      if (c->rhint >= 0) v[o + bpc + C + c->rhint] = 1;
    }
    off += H * (bpc + C + R);
  }
  v[off + obs] = 1; /* agent_turn one-hot, absolute player id (rl_env.py:254-255) */
  /* legal moves: HanabiObservation ctor -> LegalMoves(observer) hanabi_state.cc:288-304 */
  for (int uid = 0; uid < g->num_moves; ++uid) legal[uid] = (uint8_t)move_is_legal(g, e, uid);
}

void hzo_env_observe(void* h, uint8_t* share_obs, uint8_t* legal) {
  oenvs* g = (oenvs*)h;
  size_t D = (size_t)(g->own_len + g->obs_len + g->P);
  for (int i = 0; i < g->N; ++i) observe_one(g, &g->e[i], share_obs + i * D, legal + (size_t)i * g->num_moves);
}

/* probe row: cur, deck, info, life, fireworks[5], hand_n[5], status, score  (16 ints) */
void hzo_env_probe(void* h, int32_t* out) {
  oenvs* g = (oenvs*)h;
  for (int i = 0; i < g->N; ++i) {
    const oenv* e = &g->e[i];
    int32_t* o = out + (size_t)i * 16;
    o[0] = e->cur; o[1] = e->deck_total; o[2] = e->info; o[3] = e->life;
    for (int c = 0; c < 5; ++c) o[4 + c] = c < g->C ? e->fireworks[c] : 0;
    for (int p = 0; p < 5; ++p) o[9 + p] = p < g->P ? e->hand_n[p] : 0;
    o[14] = end_status(g, e);
    o[15] = score(g, e);
  }
}

/* cards of one env: out[P][H] = color*R+rank or -1 (used by scripted test policies) */
void hzo_env_hands(void* h, int env, int32_t* out) {
  oenvs* g = (oenvs*)h;
  const oenv* e = &g->e[env];
  for (int p = 0; p < g->P; ++p)
    for (int i = 0; i < g->H; ++i)
      out[p * g->H + i] = i < e->hand_n[p] ? e->hand[p][i].color * g->R + e->hand[p][i].rank : -1;
}
