/* hz_movetail.h -- C ABI of the fused tail of a self-play lock-step (libhanabizero_hip.so).
 *
 * What it replaces: everything /root/reference/core/selfplay_worker.py does per env between the search and the next root
 * inference -- roots.get_distributions() / get_values() (:284-285), select_action (:289-292, core/utils.py:280-295),
 * env.step (:300), store_search_stats + append (:306-309, core/game.py:170-200), the finished games' hand-over and reset
 * (:216-240), the stacked-observation window (:237, 326-327) and the next move's Dirichlet noise (:279) -- i.e. the sequence
 *     hz_tree_get_root_stats, hz_actor_record_search, hz_env_step, hz_env_observe, hz_actor_record_step,
 *     hz_env_reset_rows (= hz_actor_flush + hz_env_reset), hz_env_observe, hz_actor_begin_move_draw
 * of include/hz_tree.h / hz_env.h / hz_selfplay.h in TWO launches, one wave per env, with the SAME results bit for bit
 * (tests/test_selfplay.py compares the two forms record by record).  The individual entry points remain.
 * Conventions as include/hz_tree.h.
 */
#ifndef HZ_MOVETAIL_H
#define HZ_MOVETAIL_H

#include <stdint.h>

#include "hz_env.h"
#include "hz_selfplay.h"
#include "hz_tree.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tree: searched this move (hz_search_run or the launch-per-phase calls); env: the N games; bufs: the actor's histories and outbox.
 *   counts [N][A] i32, root_values [N] f32   OUT  as hz_tree_get_root_stats + hz_actor_record_search leave them
 *   legal [N][A] u8                          IN: legal moves of the searched positions; OUT: of the next positions
 *   uniform [N] f64                          IN: this move's sampling uniforms; OUT: the next move's (hz_actor_draw)
 *   temperature, temperature_dev             visit_softmax_temperature_fn's value: the float, or -- when temperature_dev is not NULL --
 *                                            the DEVICE float it points at, read when the kernel runs (a lock-step captured in a
 *                                            hipGraph then follows a temperature schedule: core/config.py visit_softmax_temperature_fn,
 *                                            selfplay_worker.py:172-174)
 *   action [N] i32, entropy [N] f64 or NULL  OUT  hz_actor_record_search
 *   reward, done, score, status [N]          OUT  hz_env_step (done: u8, 4-byte aligned; the others i32)
 *   packed [N][W] i32                        OUT  the next positions' observations, bit-packed (hz_env_observe)
 *   stack_buf                                IN/OUT the model's input windows: rows of stack_row_bytes holding `stack` slots of
 *                                            slot_bytes (a multiple of 16, >= D elements of obs_dtype; pad elements are never
 *                                            written), updated as hz_actor_begin_move does
 *   seed, move_count [N] i64, alpha, noise [N][A] f32   the next move's draws, as hz_actor_draw with env_id_base = bufs->env_id_base
 *   scratch [2] i64                          DEVICE scratch
 * bufs->slot / finished / num_finished are filled as hz_actor_record_step fills them. */
int hz_actor_move_tail(hz_tree_t* tree, hz_env_t* env, const hz_actor_bufs_t* bufs, int mdp, int32_t* counts,
                       float* root_values, uint8_t* legal, double* uniform, float temperature, const float* temperature_dev, int deterministic,
                       int32_t* action, double* entropy, int32_t* reward, uint8_t* done, int32_t* score, int32_t* status,
                       int32_t* packed, void* stack_buf, int64_t stack_row_bytes, int stack, int64_t slot_bytes,
                       int obs_dtype, uint64_t seed, int64_t* move_count, double alpha, float* noise, int64_t* scratch,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HZ_MOVETAIL_H */
