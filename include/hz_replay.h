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

/* The same windows for B x G rows addressed through the replay's own position arrays instead of per-row (row0, t) lists: output row
 * m = b * G + j looks at position p = phys[b], `shift0 + j` moves behind it:
 *     t = pos_t[p] + shift0 + j  if that is < pos_T[p] (the game has such a position), else none (a zero row);  row0 = pos_row0[p].
 * G = 1, shift0 = 0: the model inputs of a batch; G = U + 1, shift0 = td_steps: its bootstrap windows; G = U + 1, shift0 = 0: the
 * windows of the positions to re-search -- for those, optionally, legal_out [B * G][num_actions] (the legal-move row of the window's
 * last frame, from legal [frame_rows][num_actions]; zeros where there is no such position) and valid_out [B * G] (1 / 0).
 * One launch where the per-row form needs the caller to build two index arrays first (six small launches in PyTorch). */
int hz_replay_windows_seq(const int32_t* frames, int packed_words, const int64_t* pos_row0, const int32_t* pos_t, const int32_t* pos_T,
                          const int64_t* phys, int B, int G, int shift0, int stack, int D, void* out, int64_t out_row_elems,
                          int64_t slot_elems, int out_dtype, const uint8_t* legal, int num_actions, int64_t frame_rows, uint8_t* legal_out,
                          uint8_t* valid_out, void* stream);

/* The other tensors of a learner batch from the replay's arrays, as BatchWorker_CPU.make_batch + BatchWorker_GPU's target arithmetic
 * build them (/root/reference/core/reanalyze_worker.py:148-168, 249-304, 374-399), one thread per (batch row b, unroll position k):
 *   out_action [B][U] int64     action[p + k], or rand_actions[b][k] past the end of the game
 *   out_reward [B][U] f32       reward[p + k], 0 past the end
 *   out_value  [B][U + 1] f32   bootstrap[b][k] * discount^td (where position p + k + td exists) + sum_{i < td} discount^i *
 *                               reward[p + k + i] (0 past the end), float64 in the reference's order of additions, cast once; 0 where
 *                               p + k itself is past the end
 *   out_policy [B][U + 1][A] f32   visits / their sum (float64 quotient), 0 past the end
 *   out_inside [B][U + 1] u8 (or NULL)   p + k inside its game
 * phys [B] int64 positions; pos_t / pos_T / action int8 / reward int16 / visits int16 [.][A]: the replay's arrays, `head` = live
 * positions (reads are clamped below it); bootstrap [B][U + 1] f32: the target model's values of the bootstrap windows;
 * discount_powers [td + 1] f64: discount^0 .. discount^td as the HOST computes them (pow on the device may differ in the last bit). */
int hz_replay_targets(const int64_t* phys, int B, int unroll_steps, int td_steps, int num_actions, int64_t head, const int32_t* pos_t,
                      const int32_t* pos_T, const int8_t* action, const int16_t* reward, const int16_t* visits, const float* bootstrap,
                      const double* discount_powers, const int64_t* rand_actions, int64_t* out_action, float* out_reward, float* out_value,
                      float* out_policy, uint8_t* out_inside, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HZ_REPLAY_H */
