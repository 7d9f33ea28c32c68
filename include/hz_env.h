/* hz_env.h -- C ABI of the MI355X batched Hanabi environment (libhanabizero_hip.so).
 *
 * Drop-in boundary for the native side of the reference env: the per-object extern "C" API of
 * envs/hanabi/pyhanabi.h (file:line below relative to /root/reference) that envs/hanabi/pyhanabi.py binds with
 * cffi and envs/hanabi/rl_env.py drives once per env per step.  One hz_env_t holds N independent games
 * (one reference HanabiGame + HanabiState each, every game with its own std::mt19937) bit-packed in HBM and
 * advances all of them per call:
 *
 *   reference call (one env)                                              here (N envs)
 *   NewGame pyhanabi.h:151 + NewObservationEncoder :181                   hz_env_create
 *   NewState :110 + StateDealRandomCard :116 until a player is to act    hz_env_reset          (rl_env.py:249-252)
 *   GetMoveByUid :163, StateApplyMove :114, StateDealRandomCard loop,
 *     StateScore :129, StateEndOfGameStatus :125                          hz_env_step           (rl_env.py:418-442)
 *   NewObservation :168, ObsNumLegalMoves/ObsGetLegalMove :190-192,
 *     EncodeObservation :186, EncodeOwnHandObservation :192               hz_env_observe        (rl_env.py:426-434)
 *   StateCurPlayer / StateDeckSize / StateFireworks / ... getters         hz_env_probe
 *
 * Conventions: as include/hz_tree.h (0 / <0 + hz_last_error(), device pointers, explicit stream, no hidden
 * RNG: the per-env seeds are inputs, deals follow libstdc++'s mt19937 + discrete_distribution bit for bit).
 * An illegal move does not abort (reference: REQUIRE -> abort, hanabi_state.cc:222): the env is left unchanged
 * and status[i] is set to HZ_ENV_ILLEGAL_MOVE.
 */
#ifndef HZ_ENV_H
#define HZ_ENV_H

#include <stdint.h>

#include "hz_rows.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hz_env hz_env_t;

enum { HZ_OBS_U8 = 0, HZ_OBS_F32 = 1, HZ_OBS_BF16 = 2, HZ_OBS_F16 = 3 };
enum { HZ_MDP_GLOBAL = 0, HZ_MDP_LOCAL = 1 }; /* config/hanabi_control/env_wrapper.py:18-31 `mdp` */
enum { HZ_ENV_OK = 0, HZ_ENV_ILLEGAL_MOVE = 1 };
#define HZ_ENV_PROBE_FIELDS 16 /* cur_player, deck_size, info, life, fireworks[5], hand_size[5], end_status, score */

/* hand_size <= 0 selects the rule default (5 cards for 2-3 players, 4 for 4-5; hanabi_game.cc:147-152).
 * host_seeds [N] (HOST pointer): seed of env i's mt19937 (reference: one HanabiGame per env, "seed" parameter). */
int hz_env_create(hz_env_t** out, int num_envs, int colors, int ranks, int players, int hand_size,
                  int max_information_tokens, int max_life_tokens, const int32_t* host_seeds, int device);
int hz_env_destroy(hz_env_t* e);
/* MaxMoves :164, ObservationShape :184, OwnHandShape :189, NumPlayers :153 */
int hz_env_dims(const hz_env_t* e, int* num_moves, int* obs_len, int* own_hand_len, int* players);

/* mask [N] u8 (device) or NULL = all envs.  RNG state carries over, as rl_env.py:249 reuses self.game. */
int hz_env_reset(hz_env_t* e, const uint8_t* mask, void* stream);
/* hz_env_reset plus an independent row scatter (hz_rows.h; e.g. hz_actor_flush_job: the finished games' trajectories into
 * the outbox ring) executed by additional workgroups of the same launch: the reset is bound by the latency of the few
 * wavefronts that have a game to deal, the scatter rides along for free.  Same results as the two separate calls. */
int hz_env_reset_rows(hz_env_t* e, const uint8_t* mask, const hz_rows_job_t* rows, void* stream);

/* actions [N] i32 move uids; outputs reward [N] i32 (score delta, may be negative at the loss of the last life),
 * done [N] u8, score [N] i32, status [N] i32.  Envs with mask[i]==0 are untouched (their outputs too). */
int hz_env_step(hz_env_t* e, const int32_t* actions, const uint8_t* mask, int32_t* reward, uint8_t* done,
                int32_t* score, int32_t* status, void* stream);

/* Observation of the player to act, for every env:
 *   mdp GLOBAL: own_hand ++ canonical ++ onehot(cur_player)   (rl_env.py share_obs; D = own + obs + players)
 *   mdp LOCAL :             canonical ++ onehot(cur_player)   (rl_env.py obs;       D = obs + players)
 * obs_out   [N] rows of `obs_stride` elements of obs_dtype, the first D written with 0/1 (may be NULL)
 * packed_out[N][ceil(D/32)] u32, bit j of the row = element j (little-endian bit order; may be NULL)
 * legal_out [N][num_moves] u8 0/1 (may be NULL) */
int hz_env_observe(hz_env_t* e, int mdp, void* obs_out, int obs_dtype, int64_t obs_stride, uint32_t* packed_out,
                   uint8_t* legal_out, void* stream);

int hz_env_probe(hz_env_t* e, int32_t* out /* [N][HZ_ENV_PROBE_FIELDS] */, void* stream);
int64_t hz_env_hbm_bytes(const hz_env_t* e);

/* The games' bit-packed states ([N][32] u32) and generator positions ([N] u32) copied out / back in on `stream` (device
 * buffers): lets a measurement replay one move several times (bench.py's tail timing).  The generators' WORDS are not part of
 * it -- after a restore the draws are made from words the replayed moves have already regenerated once: valid games, but not
 * the games the saved position would have led to.  Not for product code. */
int hz_env_snapshot(hz_env_t* e, void* out_state, uint32_t* out_positions, void* stream);
int hz_env_restore(hz_env_t* e, const void* state, const uint32_t* positions, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HZ_ENV_H */
