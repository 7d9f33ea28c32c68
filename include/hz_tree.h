/* hz_tree.h -- C ABI of the MI355X search-tree engine (libhanabizero_hip.so).
 *
 * Drop-in boundary for the reference's Cython module core/ctree/cytree.pyx (file:line below are
 * relative to /root/reference).  One hz_tree_t replaces the triple the reference builds per move:
 *     cytree.Roots(root_num, action_num, tree_nodes)     cytree.pyx:37-70  -> CRoots   cnode.cpp:229-292
 *     cytree.MinMaxStatsList(num) + set_delta            cytree.pyx:17-27  -> cminimax.cpp:48-63
 *     cytree.ResultsWrapper(num)                         cytree.pyx:30-34  -> CSearchResults cnode.cpp:6-17
 * and is REUSED across moves (hz_tree_prepare resets it) instead of being re-allocated.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (hz_last_error() gives the text); nothing aborts;
 *   - every pointer argument is a DEVICE pointer (HBM) unless named host_*; row-major, contiguous;
 *   - `stream` is a hipStream_t passed as void* (pass PyTorch's current stream); calls only enqueue
 *     work, they never synchronise, allocate or free (safe under hipGraph capture);
 *   - no hidden RNG: the tie-break stream is the pure function of include/hz_tiebreak.h.
 */
#ifndef HZ_TREE_H
#define HZ_TREE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hz_tree hz_tree_t;

/* dtype of a hidden-state pool / net-input buffer element */
enum { HZ_F32 = 0, HZ_BF16 = 1, HZ_F16 = 2, HZ_F16X2 = 3 /* include/hz_mlp.h only: fp32 numbers as fp16 pairs */ };

#define HZ_MAX_ACTIONS 64      /* one lane per child; Hanabi needs at most 48 (5 players) */
#define HZ_MAX_SIMULATIONS 4096

/* Library-wide. */
const char* hz_last_error(void);
int hz_version(void);

/* Roots.__cinit__ (cytree.pyx:42-45) + MinMaxStatsList(num) (cytree.pyx:20-21) + ResultsWrapper(num).
 * Allocates the struct-of-arrays node pool for `num_trees` trees of `num_actions` children per node and
 * `num_simulations` expandable entries each, on HIP device `device`. */
int hz_tree_create(hz_tree_t** out, int num_trees, int num_actions, int num_simulations, int device);
int hz_tree_destroy(hz_tree_t* t); /* Roots.__dealloc__ cytree.pyx:65-66 */

/* The arguments the reference passes on every multi_traverse / multi_back_propagate call
 * (cytree.pyx:87,97; core/mcts.py:17,21) plus the tie-break stream definition.
 * tree_id_base: global id of tree 0 of this handle (env id offset of this GPU's shard). */
int hz_tree_set_params(hz_tree_t* t, int pb_c_base, float pb_c_init, float discount, float value_delta_max,
                       uint64_t tie_seed, uint32_t tree_id_base);

/* Roots.prepare (cytree.pyx:47-48 -> CRoots::prepare cnode.cpp:247-253) when noises != NULL,
 * Roots.prepare_no_noise (cytree.pyx:50-51 -> cnode.cpp:255-259) when noises == NULL.
 * Resets every tree, expands the roots from masked-softmax priors and mixes in the noise.
 *   noises [N][A] f32 | rewards [N] f32 | policy_logits [N][A] f32 | legal [N][A] u8 (0/1) */
int hz_tree_prepare(hz_tree_t* t, float root_exploration_fraction, const float* noises, const float* rewards,
                    const float* policy_logits, const uint8_t* legal, void* stream);

/* multi_traverse (cytree.pyx:97-101 -> cmulti_traverse cnode.cpp:407-441): one pUCT descent per tree.
 *   sim: simulation index of this call (0-based; feeds the tie-break stream)
 *   out_ix / out_iy / out_last_action [N] i32: hidden_state_index_x/y of the leaf's parent and the
 *   action taken into the leaf (the three lists the reference returns).
 * The search paths stay inside the handle for the following hz_tree_backprop (ResultsWrapper). */
int hz_tree_traverse(hz_tree_t* t, int sim, int32_t* out_ix, int32_t* out_iy, int32_t* out_last_action,
                     void* stream);

/* hz_tree_traverse fused with the hidden-state gather of core/mcts.py:31-36: additionally copies
 * pool[ix][tree][0:hidden] into net_in[tree][0:hidden] (row stride net_in_stride elements) and, when
 * action_onehot_cols > 0, writes one_hot(last_action) into net_in[tree][hidden : hidden+action_onehot_cols]
 * (the concat of MuZeroNet.dynamics, config/hanabi_control/model.py:215-219; columns >= num_actions are zeroed
 * so the caller can pad the first dynamics layer's K to a GEMM-friendly size).
 *   pool   [num_simulations][N][hidden] elements of `dtype` (entry e = hidden state written after sim e-1;
 *          entry 0 = root states), resident in HBM for the whole move. */
int hz_tree_traverse_gather(hz_tree_t* t, int sim, int32_t* out_ix, int32_t* out_iy, int32_t* out_last_action,
                            const void* pool, int hidden, int dtype, void* net_in, int net_in_stride,
                            int action_onehot_cols, void* stream);

/* multi_back_propagate (cytree.pyx:87-94 -> cmulti_back_propagate cnode.cpp:337-344): expand each leaf
 * (all-legal mask) with (hidden_state_index_x, tree) / reward / policy logits, back up `values` along the
 * stored paths, then recompute each tree's min-max statistics (update_tree_q cnode.cpp:296-315).
 *   rewards [N] f32 | values [N] f32 | policy_logits [N][A] f32 (NaN logits are the caller's to clear,
 *   core/mcts.py:48-49; hz_tree_backprop treats them exactly as the reference's expand does). */
int hz_tree_backprop(hz_tree_t* t, int hidden_state_index_x, const float* rewards, const float* values,
                     const float* policy_logits, void* stream);

/* hz_tree_backprop of simulation k fused with hz_tree_traverse of simulation k+1 (next_sim): one launch per
 * simulation; outputs as hz_tree_traverse.  Results are identical to the two separate calls. */
int hz_tree_backprop_traverse(hz_tree_t* t, int hidden_state_index_x, const float* rewards, const float* values,
                              const float* policy_logits, int next_sim, int32_t* out_ix, int32_t* out_iy,
                              int32_t* out_last_action, void* stream);

/* hz_tree_backprop fed straight from the network heads (what core/mcts.py:44-50 + core/model.py:79-80 do on the
 * host between recurrent_inference and multi_back_propagate):
 *   reward_logits / value_logits [N] rows of `support_size` categorical logits over the integers
 *       support_min .. support_min+support_size-1, row strides in elements, element type `dtype`;
 *       scalar = inverse_scalar_transform (core/config.py:210-232), NaN -> 0;
 *   policy_logits [N] rows of num_actions logits (dtype, stride), NaN -> 0 (core/mcts.py:48-49);
 *   out_rewards / out_values [N] f32 (may be NULL): the scalars that entered the tree, for inspection.
 * The scalar transform is a network output (tolerance 1e-3); the tree update that consumes it is the same exact
 * arithmetic as hz_tree_backprop. */
int hz_tree_backprop_nets(hz_tree_t* t, int hidden_state_index_x, const void* reward_logits, int64_t reward_stride,
                          const void* value_logits, int64_t value_stride, int support_size, int support_min,
                          const void* policy_logits, int64_t policy_stride, int dtype, float* out_rewards,
                          float* out_values, void* stream);

/* The scalar transform of hz_tree_backprop_nets alone (inverse_value_transform / inverse_reward_transform,
 * core/config.py:204-232): logits [num_rows] rows of support_size (stride elements, dtype) -> out [num_rows] f32.
 * Same device code as the fused kernel, hence bit-identical to what entered the tree. */
int hz_support_to_scalar(const void* logits, int64_t stride, int support_size, int support_min, int dtype, float* out,
                         int num_rows, void* stream);

/* Roots.get_distributions / get_values / get_trajectories (cytree.pyx:53-60 -> cnode.cpp:266-292). */
int hz_tree_get_distributions(hz_tree_t* t, int32_t* out /* [N][A] */, void* stream);
int hz_tree_get_values(hz_tree_t* t, float* out /* [N] */, void* stream);
/* both of the above in one launch (what the actor reads after every search, selfplay_worker.py:286-288) */
int hz_tree_get_root_stats(hz_tree_t* t, int32_t* counts /* [N][A] */, float* values /* [N] */, void* stream);
int hz_tree_get_trajectories(hz_tree_t* t, int32_t* out /* [N][max_len], -1 padded */, int max_len, void* stream);

/* Introspection used by the parity tests (the reference exposes these only inside C++). */
int hz_tree_get_minmax(hz_tree_t* t, float* out_min /* [N] */, float* out_max /* [N] */, void* stream);
int hz_tree_get_root_priors(hz_tree_t* t, float* out /* [N][A] */, void* stream);
int hz_tree_get_path_len(hz_tree_t* t, int32_t* out /* [N] nodes on the last path incl. root and leaf */, void* stream);

/* Bytes of HBM held by the handle. */
int64_t hz_tree_hbm_bytes(const hz_tree_t* t);

/* Device-to-device snapshot of a handle's whole search state and parameters into another handle of the same
 * shape (the reference can copy-construct a CRoots; used by bench.py to time one kernel on independent copies of
 * a live mid-search state). */
int hz_tree_copy(hz_tree_t* dst, const hz_tree_t* src, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HZ_TREE_H */
