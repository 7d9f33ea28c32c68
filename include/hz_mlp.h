/* hz_mlp.h -- C ABI of the fused recurrent-inference kernel of the search loop (libhanabizero_hip.so).
 *
 * What it replaces: one call of BaseMuZeroNet.recurrent_inference (/root/reference/core/model.py:74-84) as issued by
 * core/mcts.py:38-42 -- dynamics (config/hanabi_control/model.py:61-125) + reward head + prediction heads
 * (model.py:138-149 / 250-269) + inverse scalar transforms (core/config.py:204-232) + NaN clearing of the policy logits
 * (core/mcts.py:48-49) -- for all N trees at once.  In PyTorch this is a chain of ~10 small GEMMs that are bound by
 * launch latency and by re-reading activations; here ONE kernel keeps each tile of rows' activations in LDS across all
 * layers and streams the (eval-mode, BatchNorm-folded, bf16) weights once per workgroup through MFMA
 * (v_mfma_f32_16x16x32_bf16, fp32 accumulate), i.e. the matrix cores do exactly the net GEMMs and nothing else.
 * The layer sequence is data (a small "program" built by hanabizero_amd/model.py from the module), not code.
 * Numerics: same rounding points as the bf16 PyTorch path (bf16 activations between layers, fp32 accumulation and
 * bias/residual/ReLU epilogue); agreement with the fp32 reference nets is checked at the north-star tolerance.
 * Conventions as include/hz_tree.h.
 */
#ifndef HZ_MLP_H
#define HZ_MLP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HZ_MLP_MAX_LAYERS 12

/* One fused layer: for each of `groups` independent blocks g, out_g = act(in_g @ W_g^T + b_g [+ residual]).
 * All offsets are ELEMENT offsets inside one row of the workgroup's LDS activation image (bf16 elements). */
typedef struct {
  int32_t K;            /* reduction length per group, multiple of 32 */
  int32_t nout;         /* total padded outputs over all groups, multiple of 64*groups/groups... = multiple of 64 */
  int32_t groups;       /* 1 or 3 */
  int32_t src_off;      /* source columns of group g start at src_off + g*src_gstride */
  int32_t src_gstride;
  int32_t dst_off;      /* destination columns start at dst_off (group g at dst_off + g*nout/groups) */
  int32_t res_off;      /* residual columns (same indexing as dst) or -1 */
  int32_t res_group;    /* group that gets the residual, -1 = every group */
  int32_t relu_mask;    /* bit g set: ReLU on group g */
  int32_t store_hidden; /* != 0: after this layer copy dst[0:hidden] of every row to hidden_out */
  int64_t w_off;        /* element offset of this layer's packed weights */
  int32_t b_off;        /* element offset of this layer's biases (fp32, indexed by padded output column) */
  int32_t kind;         /* kernel instantiation id (tiles per wave / groups / k-steps), see hz_mlp.hip */
} hz_mlp_layer_t;

typedef struct {
  int32_t n_layers;
  int32_t row_stride;   /* LDS elements per row (multiple of 8, chosen to avoid bank conflicts) */
  int32_t in_width;     /* columns of net_in copied to LDS columns [0, in_width) (state | one-hot | pad) */
  int32_t hidden;       /* width of the hidden state */
  int32_t off_reward, off_value, off_policy; /* LDS columns of the final reward / value / policy logits */
  int32_t support_size, support_min, num_actions;
  hz_mlp_layer_t layer[HZ_MLP_MAX_LAYERS];
} hz_mlp_program_t;

/* net_in      [N][net_in_stride] bf16  rows = [state | one_hot(action) | 0] as written by hz_tree_traverse_gather
 * weights     packed bf16 (layout: hanabizero_amd/model.py::pack_mlp_weights), biases fp32
 * hidden_out  [N][hidden] bf16         next hidden state (its slot of the search's pool)
 * out_reward / out_value [N] f32       inverse_scalar_transform of the categorical heads (NaN -> 0)
 * out_policy  [N][num_actions] f32     policy logits, NaN -> 0
 * rows_per_wg: 16 or 32 (rows of one workgroup; N need not be a multiple) */
int hz_mlp_recurrent(const hz_mlp_program_t* host_program, const void* net_in, int64_t net_in_stride,
                     const void* weights, const float* biases, void* hidden_out, float* out_reward,
                     float* out_value, float* out_policy, int num_rows, int rows_per_wg, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HZ_MLP_H */
