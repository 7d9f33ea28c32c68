/* hz_selfplay.h -- C ABI of the per-move actor glue that the reference runs as a Python loop over envs
 * (core/selfplay_worker.py:286-347), here one kernel over the whole batch.  Conventions as include/hz_tree.h.
 */
#ifndef HZ_SELFPLAY_H
#define HZ_SELFPLAY_H

#include <stdint.h>

#include "hz_rows.h"

#ifdef __cplusplus
extern "C" {
#endif

/* core/utils.py:280-295 select_action, for N envs at once:
 *   counts  [N][A] i32  IN/OUT: visit counts of the root children; entries with legal == 0 and count >= 1 are
 *                       zeroed in place, exactly as the reference mutates its `visit_counts` list (the masked
 *                       counts are what GameHistory.store_search_stats then normalises, core/game.py:189-200)
 *   legal   [N][A] u8
 *   uniform [N] f64     one U[0,1) sample per env: the `random_sample()` np.random.choice draws
 *   temperature         visit_softmax_temperature_fn (1.0 in both Hanabi configs); p_i = count_i ** (1/T)
 *   deterministic       != 0: action = first arg-max of the masked counts (np.argmax), uniform unused
 *   out_action  [N] i32 index = searchsorted(cumsum(p)/cumsum(p)[-1], u, side='right') in fp64, the algorithm of
 *                       numpy.random.RandomState.choice(len, p=p)
 *   out_entropy [N] f64 scipy.stats.entropy(p, base=2) (may be NULL)
 * An env whose masked counts sum to 0 gets action -1 (the reference raises from np.random.choice there). */
int hz_select_action(int num_envs, int num_actions, int32_t* counts, const uint8_t* legal, const double* uniform,
                     float temperature, int deterministic, int32_t* out_action, double* out_entropy, void* stream);

/* Finished-game flush (replaces the per-game Python of selfplay_worker.py:216-228: game_over() + put() +
 * replay_buffer.save_pools.remote): copies row i of `src` ([N] rows of row_bytes) to row slot[i] of `dst` for every
 * env with slot[i] >= 0; rows with slot[i] < 0 are skipped.  One wavefront per row, 16 B per lane per trip when
 * rows are 16-byte aligned (4 B or 1 B per lane otherwise). */
int hz_rows_scatter(const void* src, void* dst, int64_t row_bytes, const int32_t* slot, int num_rows, void* stream);

/* Root exploration noise and the action-sampling uniform of one move for every env, on the device (the reference draws
 * np.random.dirichlet([alpha] * A).astype(float32) per env on the host, selfplay_worker.py:279, and select_action's
 * np.random.choice draws one uniform, core/utils.py:293):
 *   noise   [N][A] f32  Dirichlet(alpha, ..., alpha): A independent Gamma(alpha, 1) draws (Marsaglia-Tsang on
 *                       Gamma(alpha + 1) times U^(1/alpha), fp64) divided by their sum in action order
 *   uniform [N] f64     U[0, 1) with 53 random bits
 * The random stream is counter-based: draws of env i at its k-th move are a pure function of (seed, env_id_base + i,
 * k), so results do not depend on how envs are sharded over GPUs; move_count [N] i64 holds k and is incremented. */
int hz_actor_draw(uint64_t seed, int64_t env_id_base, int64_t* move_count, int num_envs, int num_actions, double alpha,
                  float* noise, double* uniform, void* stream);

/* ---- the actor's per-move bookkeeping (core/selfplay_worker.py:286-347 + GameHistory.append/store_search_stats,
 * core/game.py:170-200), one launch per phase instead of a Python loop over envs.  The caller owns every buffer. */
typedef struct {
  int32_t num_envs, num_actions, packed_words, max_moves; /* N, A, W, T */
  int32_t outbox_games;                                   /* capacity of the outbox ring */
  int32_t env_id_base;                                    /* global id of env 0 */
  /* trajectories under construction, one row per env */
  int8_t* action;    /* [N][T]      GameHistory.actions */
  int8_t* reward;    /* [N][T]      GameHistory.rewards */
  float* value;      /* [N][T]      GameHistory.root_values */
  int16_t* visits;   /* [N][T][A]   masked root child visit counts (child_visits before normalisation) */
  uint8_t* legal;    /* [N][T+1][A] legal-move masks */
  int32_t* obs;      /* [N][T+1][W] bit-packed observations */
  int64_t* traj_len; /* [N] moves recorded so far */
  double* ent_sum;   /* [N] running sum of the visit entropies (selfplay_worker.py:303) */
  int32_t* meta;     /* [N][4] len, final score, global env id, float bits of ent_sum */
  /* outbox ring of finished games: the same six arrays with `outbox_games` rows, and their meta rows */
  int8_t* out_action;
  int8_t* out_reward;
  float* out_value;
  int16_t* out_visits;
  uint8_t* out_legal;
  int32_t* out_obs;
  int32_t* out_meta;
  int64_t* out_count;     /* [2] games finished so far; total moves of the games finished since the caller last zeroed [1] */
  int32_t* slot;          /* [N] scratch: outbox row of the env's finished game, -1 while it runs */
  int32_t* finished;      /* [N] scratch: the envs whose game just ended, ascending */
  int32_t* num_finished;  /* [1] scratch: how many */
  int64_t* illegal_steps; /* [1] env steps that reported an illegal move (stays 0) */
} hz_actor_bufs_t;

/* After the search: hz_select_action on (counts, legal, uniform) as above, then at t = min(traj_len, T-1):
 * action[t] = a, visits[t] = masked counts, value[t] = root_values; ent_sum += entropy.  counts IN/OUT as above. */
int hz_actor_record_search(const hz_actor_bufs_t* bufs, int32_t* counts, const float* root_values,
                           const uint8_t* legal, const double* uniform, float temperature, int deterministic,
                           int32_t* out_action, double* out_entropy /* [N] or NULL */, void* stream);

/* After hz_env_step + hz_env_observe(packed, legal_next): reward[t] = reward, obs[t+1] / legal[t+1] = the observation
 * after the move (the terminal one included, selfplay_worker.py:308), meta row, illegal_steps += #(status != 0);
 * then the outbox slots of the games that just ended, in env order: slot = (out_count + rank among done) % capacity,
 * out_count += #done; the list of those envs goes to bufs->finished / num_finished. */
int hz_actor_record_step(const hz_actor_bufs_t* bufs, const int32_t* reward, const uint8_t* done, const int32_t* score,
                         const int32_t* status, const int32_t* packed, const uint8_t* legal_next, void* stream);

/* hz_rows_scatter of all six trajectory arrays and the meta rows by bufs->slot, one launch over bufs->finished
 * (one workgroup per row and array). */
int hz_actor_flush(const hz_actor_bufs_t* bufs, void* stream);

/* The scatter hz_actor_flush performs, as data (host side, no launch): pass it to hz_env_reset_rows to have the masked reset
 * that follows the flush in a lock-step carry it. */
int hz_actor_flush_job(const hz_actor_bufs_t* bufs, hz_rows_job_t* job);

/* Outbox ring rows [first, first + n) (mod capacity) as ONE packed byte buffer, the games back to back (ragged; `moves` =
 * the sum of their lengths, which out_count[1] accumulates): sections
 *   meta [n][4] i32 | action [moves] i8 | reward [moves] i8 | value [moves] f32 | visits [moves][A] i16 |
 *   legal [moves + n][A] u8 | obs [moves + n][W] i32
 * each starting on a 16-byte boundary; game j starts at row start_j = len_0 + .. + len_(j-1) of the per-move sections
 * and at row start_j + j of the last two (which hold len_j + 1 rows per game: the terminal observation is stored).
 * hz_actor_packed_bytes returns the total and, if `offsets` is not NULL, the seven section offsets.  `starts`:
 * [n + 1] i32 DEVICE scratch (receives the start rows and the total).  This is what travels to the replay owner
 * (ReplayBuffer.save_pools' payload, /root/reference/core/selfplay_worker.py:75-78). */
int64_t hz_actor_packed_bytes(int n, int64_t moves, int num_actions, int packed_words, int64_t* offsets /* [7] or NULL */);
int hz_actor_pack(const hz_actor_bufs_t* bufs, int64_t first, int n, int64_t moves, int32_t* starts, void* out,
                  int64_t out_bytes, void* stream);

/* After hz_env_reset(done) + hz_env_observe(newest, packed, legal): traj_len = done ? 0 : t+1, ent_sum = done ? 0 : ent_sum,
 * obs[traj_len] / legal[traj_len] = the current observation, and the model's input window (selfplay_worker.py:237,
 * 326-327): rows of `stack` observations of obs_bytes each; running games shift by one and append `newest`,
 * new games are filled with `newest`.  stack_row_bytes / newest_row_bytes: row strides. */
int hz_actor_begin_move(const hz_actor_bufs_t* bufs, const uint8_t* done, const int32_t* packed, const uint8_t* legal,
                        const void* newest, int64_t newest_row_bytes, void* stack_buf, int64_t stack_row_bytes, int stack,
                        int64_t obs_bytes, void* stream);

/* hz_actor_begin_move followed by hz_actor_draw for the NEXT move (env_id_base = bufs->env_id_base) in one launch: the two
 * are independent, each is bound by the latency of one wavefront per env, and run side by side they cost the longer of
 * the two.  Same results as the two calls. */
int hz_actor_begin_move_draw(const hz_actor_bufs_t* bufs, const uint8_t* done, const int32_t* packed, const uint8_t* legal,
                             const void* newest, int64_t newest_row_bytes, void* stack_buf, int64_t stack_row_bytes, int stack,
                             int64_t obs_bytes, uint64_t seed, int64_t* move_count, double alpha, float* noise,
                             double* uniform, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HZ_SELFPLAY_H */
