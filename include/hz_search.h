/* hz_search.h -- C ABI of the persistent search kernel (libhanabizero_hip.so).
 *
 * What it replaces: the whole simulation loop of MCTS.run_multi (/root/reference/core/mcts.py:26-58) -- for every
 * simulation multi_traverse (cytree.pyx:97-101), the hidden-state gather (mcts.py:31-36), recurrent_inference
 * (core/model.py:74-84) and multi_back_propagate (cytree.pyx:87-94) -- for all trees, as ONE kernel launch per move.
 * A workgroup owns 16 trees for the whole search: one wavefront per tree for descent / expand / backup (the code of
 * hz_tree_traverse / hz_tree_backprop), all 16 wavefronts together for the fused MFMA inference of those 16 rows (the
 * code of hz_mlp_recurrent, 16 waves x 2 tiles).  With more trees than 16 per compute unit of the device a workgroup owns
 * 32 (two per wavefront, searched one after the other between inferences of 32 rows per weight fragment).  Results are bit-identical to the launch-per-phase path
 * (hz_tree_traverse -> hz_mlp_recurrent -> hz_tree_backprop_traverse ...): same arithmetic, same order.
 * Conventions as include/hz_tree.h.
 */
#ifndef HZ_SEARCH_H
#define HZ_SEARCH_H

#include <stdint.h>

#include "hz_mlp.h"
#include "hz_tree.h"

#ifdef __cplusplus
extern "C" {
#endif

/* t                freshly prepared tree (hz_tree_prepare), parameters set (hz_tree_set_params)
 * num_simulations  simulations to run: the reference runs config.num_simulations - 1 (core/mcts.py:27-29)
 * H, jobs, wstream, biases, action_table   the MLP as for hz_mlp_recurrent, laid out for num_waves = 16, tiles = 2
 * pool             [>= num_simulations + 1][N][hidden] bf16 or fp16 (DEVICE), plane 0 = the roots' hidden states; plane k+1
 *                  receives the hidden states simulation k produces; plane_stride / row_stride in elements
 * ix, iy, la       [N] i32 scratch (DEVICE): on return the values of the last simulation
 * rewards, values  [N] f32, policy [N][num_actions] f32 (DEVICE): reserved -- the leaf outputs stay on chip (row image ->
 *                  registers of the tree's wave) and these arrays are not written; they must still be valid pointers
 * rows_per_workgroup  trees per workgroup: 0 = chosen from the tree count and the compute-unit count of the tree's device
 *                  (the product's setting), 16 or 32 = forced (tests, measurements).  With 32, when num_actions <= 32 and
 *                  hidden <= 512, a wavefront searches its TWO trees side by side in its two 32-lane halves; -32 forces them
 *                  one after the other; -16 = 16 trees per workgroup side by side on 8 of the 16 wavefronts (measured: no
 *                  faster than one per wavefront, slower on deep paths).  The results do not depend on any of it.
 *                  With num_actions <= 20, where the workgroup's LDS has room for a table of the nodes' last selections, a
 *                  tree whose descents have grown 6 levels long walks its predicted lines 16 (side by side: 8) levels at a time
 *                  (csrc/hz_tree_replay_dev.h) -- again the same bits, only sooner when the policy is sharp.
 * The MLP header's dtype (HZ_BF16 / HZ_F16) selects the element format of pool, weights and activations; HZ_F16X2 (include/hz_mlp.h:
 * fp32 numbers as fp16 pairs, the build inside 1e-3 of the reference's fp32 nets) takes an fp32 pool -- strides in fp32 elements --
 * and keeps 16 trees per workgroup at any tree count (rows_per_workgroup 0 or 16).
 * Limits (HZ_ERR otherwise -- the launch-per-phase calls of hz_tree.h / hz_mlp.h have none of them and compute the same
 * bits): fewer than 64 simulations per tree (t's S; the reference's configs have 50), hidden <= 512, support_size <= 256,
 * 160 KiB of LDS for 16 (or 32) row images + search state.
 * The calling thread's current device must be the tree's. */
int hz_search_run(hz_tree_t* t, int num_simulations, const hz_mlp_header_t* H, const hz_mlp_job_t* jobs,
                  const void* wstream, const float* biases, const float* action_table, void* pool,
                  int64_t plane_stride, int64_t row_stride, int32_t* ix, int32_t* iy, int32_t* la, float* rewards,
                  float* values, float* policy, int rows_per_workgroup, void* stream);

/* Which kernels hz_search_run launches for this tree: on (the default) = the ones whose descent walks predicted lines in trees
 * that have grown deep -- what a sharp policy needs (2.0 M against 1.67 M moves/s at 4096 envs, 3.0 M against 2.4 M at 8192);
 * off = the same kernels without that code, whose mere presence costs trees that never grow deep 0.6 % (1.0 % at 8192 envs).
 * The results are the same bits either way.  SelfPlayActor switches by the mean length of its searches' last paths. */
int hz_search_set_predicted_lines(hz_tree_t* t, int on);

/* hz_search_run's share of hz_mlp_poll_giveups (include/hz_mlp.h), which is the one to call. */
int hz_search_poll_giveups(unsigned int* count);
int hz_search_poll_giveups_async(unsigned int* host_pinned, void* stream);  /* hz_mlp_poll_giveups_async's share */

#ifdef __cplusplus
}
#endif
#endif /* HZ_SEARCH_H */
