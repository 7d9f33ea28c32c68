/* hz_selfplay.h -- C ABI of the per-move actor glue that the reference runs as a Python loop over envs
 * (core/selfplay_worker.py:286-347), here one kernel over the whole batch.  Conventions as include/hz_tree.h.
 */
#ifndef HZ_SELFPLAY_H
#define HZ_SELFPLAY_H

#include <stdint.h>

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

#ifdef __cplusplus
}
#endif
#endif /* HZ_SELFPLAY_H */
