/* hz_replay.h -- learner-batch observation windows straight from the bit-packed replay in HBM (gfx950).
 *
 * Replaces, for the device-resident replay (hanabizero_amd/device_replay.py), the host-side frame stacking of the reference's
 * batch makers: GameHistory.obs(i, extra_len, padding) (/root/reference/core/game.py:100-119: the window of
 * `stacked_observations` frames that ends at position i, the first frame repeated in front of a game's start), as used by
 *   BatchWorker_CPU.make_batch               core/reanalyze_worker.py:148-168   (the model input of every sampled position)
 *   BatchWorker_CPU._prepare_reward_value_context  :45-86, 204-222            (the bootstrap windows td_steps ahead)
 *   BatchWorker_CPU._prepare_policy_re_context     :101-144                   (the windows of the positions to re-search)
 * The reference keeps float frames in host memory and ships stacked float32 windows through Ray's object store and PCIe; here
 * the frames stay the 32-bit words the self-play actors packed (hz_actor_pack, include/hz_selfplay.h: bit c of a frame = bit
 * c % 32 of word c / 32) and a window is expanded where it is consumed.
 *
 * All pointers are DEVICE pointers.  Returns 0, or < 0 with hz_last_error() set; never aborts. */
#ifndef HZ_REPLAY_H
#define HZ_REPLAY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* out[m][j * slot_elems + c] = bit c of frame (row0[m] + max(0, t[m] - (stack - 1) + j))   for j < stack, c < D
 *                            = 0                                                          for D <= c < slot_elems
 * and the whole row 0 where t[m] < 0 (a position past the end of its game: zero_obs, reanalyze_worker.py:136-139, 219-221).
 *   frames      [*, packed_words] int32: the replay's frame rows, a game's T + 1 frames contiguous
 *   row0        [M] int64: first frame row of the game output row m looks into
 *   t           [M] int32: frame index inside that game the window ENDS at, or < 0
 *   out         [M] rows of out_row_elems elements of out_dtype (HZ_OBS_F32 / HZ_OBS_BF16 / HZ_OBS_F16, include/hz_env.h);
 *               out_row_elems >= stack * slot_elems, slot_elems >= D
 * HBM-bound: reads stack * packed_words * 4 B, writes stack * slot_elems * element size per row. */
int hz_replay_windows(const int32_t* frames, int packed_words, const int64_t* row0, const int32_t* t, int M, int stack, int D,
                      void* out, int64_t out_row_elems, int64_t slot_elems, int out_dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HZ_REPLAY_H */
