/* hz_train.h -- the learner's Linear -> BatchNorm1d (training mode) -> (+ residual) -> ReLU block, everything behind the GEMM as
 * ONE launch forward and ONE backward (gfx950).
 *
 * Replaces, inside update_weights (/root/reference/core/train.py:59-314: 1 initial + 5 recurrent inferences forward and
 * backward per step), what PyTorch runs per block as a dozen small kernels -- batch statistics, normalisation, running-statistics
 * update, counter increment, residual add, ReLU; backward: ReLU mask, two BatchNorm kernels, bias and affine gradient reductions,
 * gradient accumulations -- for the blocks of config/hanabi_control/model.py:18-125, 131-149, 241-269
 * (nn.Linear + nn.BatchNorm1d + ReLU; ResMLP / NewResMLP / DynamicNet / NewDynamicNet skips).  At batch 256 a learner step is
 * ~1.7 k launches of a few microseconds each: launch-bound, not bandwidth- or MFMA-bound (profiles/r04_learner_kernel_stats.md).
 * The GEMMs themselves stay hipBLASLt (torch.addmm / torch.mm on bf16 operands).
 *
 * Arithmetic = torch.nn.functional.batch_norm(training=True) on a 16-bit input with fp32 parameters: batch mean and BIASED
 * variance in fp32, y = (x - mean) * rsqrt(var + eps) * gamma + beta, running_mean / running_var updated with `momentum` (the
 * running variance with the UNBIASED batch variance), output rounded once to the element format.
 *
 * All pointers are DEVICE pointers; element format: HZ_BF16 or HZ_F16 (include/hz_tree.h).  Returns 0, or < 0 with
 * hz_last_error() set. */
#ifndef HZ_TRAIN_H
#define HZ_TRAIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* out[r][c] = act(bn(x[r][c]) + res[r][c]),  act = ReLU if relu else identity;  x, res (or NULL), out: [rows][cols] 16-bit,
 * row strides in elements; gamma, beta, running_mean, running_var [cols] fp32 (running_*: updated in place);
 * save_mean, save_invstd [cols] fp32: what the backward needs. */
int hz_bn_act_forward(const void* x, int64_t x_stride, const void* res, int64_t res_stride, void* out, int64_t out_stride, int rows,
                      int cols, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                      float eps, float* save_mean, float* save_invstd, int relu, int dtype, void* stream);

/* Backward of the same: dz = relu ? dout * (out > 0) : dout;
 *   dx[r][c] = gamma[c] * invstd[c] * (dz - mean_r(dz) - xhat * mean_r(dz * xhat)),   xhat = (x - mean) * invstd
 *   dgamma[c] += sum_r dz * xhat,  dbeta[c] += sum_r dz      (ACCUMULATED into fp32 buffers: the parameters' .grad)
 *   dres = dz (16-bit) if dres is not NULL: the gradient of the residual input. */
int hz_bn_act_backward(const void* dout, int64_t dout_stride, const void* out, int64_t out_stride, const void* x, int64_t x_stride,
                       void* dx, int64_t dx_stride, void* dres, int64_t dres_stride, int rows, int cols, const float* gamma,
                       const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, int relu, int dtype, void* stream);

/* The same two over `groups` row groups of `rows` rows each, stacked in x / res / out (dout / dx / dres): every group is normalised by
 * its OWN batch statistics -- one BatchNorm call of the module per group -- in one launch (grid: column tiles x groups).  That is
 * what lets a head of the unrolled learner step run ONCE over the stacked hidden states of all its inferences ((1 + unroll) x batch
 * rows: one GEMM instead of six) with the arithmetic of six calls.  save_mean / save_invstd: [groups][cols].
 * groups > 1: what crosses the groups is NOT done here -- running_mean / running_var and dgamma / dbeta stay untouched; the groups'
 * contributions (forward: batch mean and unbiased variance; backward: the two column sums) go to scratch [groups][2][cols] fp32, and
 * hz_bn_groups_finish applies them in group order (r <- (1 - m) r + m s_g for g = 0 .. groups - 1; dbeta += sum_g, dgamma += sum_g),
 * for every layer of the step in ONE launch.  (A workgroup that waited for its column tile's other groups inside these kernels
 * paid an L2 write-back per fence on this multi-die part: 20 us instead of 5.)  groups == 1: as hz_bn_act_forward / _backward,
 * scratch may be NULL. */
int hz_bn_act_forward_groups(const void* x, int64_t x_stride, const void* res, int64_t res_stride, void* out, int64_t out_stride, int rows,
                             int groups, int cols, const float* gamma, const float* beta, float* running_mean, float* running_var,
                             float momentum, float eps, float* save_mean, float* save_invstd, float* scratch, int relu, int dtype,
                             void* stream);
int hz_bn_act_backward_groups(const void* dout, int64_t dout_stride, const void* out, int64_t out_stride, const void* x, int64_t x_stride,
                              void* dx, int64_t dx_stride, void* dres, int64_t dres_stride, int rows, int groups, int cols,
                              const float* gamma, const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                              float* scratch, int relu, int dtype, void* stream);

/* One entry per BatchNorm layer that ran over groups: forward (backward == 0) dst0 / dst1 = running_mean / running_var, backward
 * dst0 / dst1 = dbeta / dgamma (the parameters' .grad); scratch as the grouped kernels left it.  `entries`: DEVICE array. */
typedef struct {
  float* dst0;
  float* dst1;
  const float* scratch;
  int32_t cols, groups;
  float momentum;
  int32_t reserved;
} hz_bn_finish_t;
int hz_bn_groups_finish(const hz_bn_finish_t* entries, int num_entries, int max_cols, int backward, void* stream);

/* The losses of ONE inference of the unrolled learner step (initial or recurrent; core/train.py:145-168, 196-216 with
 * config/hanabi_control/__init__.py:119-123 and core/config.py:192-253) in one launch, gradients included:
 *   value / reward   cross-entropy of the categorical head (logits [rows][support_size], 16-bit or fp32) against the two-hot
 *                    phi(h(target)) of the scalar target: h(x) = sign(x) (sqrt(|x| + 1) - 1) + 0.001 x, clamped to the support,
 *                    mass x - floor(x) on ceil(x) and the rest on floor(x) (an integer puts everything on itself);
 *   policy           cross-entropy of the policy logits [rows][num_actions] against the visit distribution (rows past the end of a
 *                    game are all zero: no loss, no gradient);
 *   predictions      the heads' scalars softmax . support -> h^-1 (what the priorities compare with the targets).
 * Outputs per row: losses[row] = {policy, value, reward, weight[row] / rows * (pc policy + vc value + rc reward)} (written, not
 * accumulated), preds[row] = {value scalar, reward scalar}; gradients of the LAST of these (the weighted total) with respect to the
 * logits, in the logits' element format: d_value, d_reward, d_policy (softmax * sum(target) - target, scaled by the row's factor).
 * reward_logits == NULL: the initial inference (no reward head: reward loss 0).  Strides in elements.  One wavefront per row. */
int hz_muzero_head_losses(const void* value_logits, int64_t value_stride, const void* reward_logits, int64_t reward_stride,
                          const void* policy_logits, int64_t policy_stride, int rows, int support_size, int support_min, int num_actions,
                          int dtype /* HZ_F32 | HZ_BF16 | HZ_F16 */, const float* target_value, int64_t target_value_stride,
                          const float* target_reward, int64_t target_reward_stride, const float* target_policy,
                          int64_t target_policy_stride, const float* weights, float value_coeff, float reward_coeff, float policy_coeff,
                          void* d_value, void* d_reward, void* d_policy, float* losses /* [rows][4] */, float* preds /* [rows][2] */,
                          void* stream);

/* hz_muzero_head_losses for ALL inferences of the unrolled step in one launch: logit rows stacked inference by inference
 * (value / policy: steps x batch rows, row k * batch + b; reward: (steps - 1) x batch rows -- the initial inference has no reward
 * head), targets indexed [b][k] through a batch stride and a step stride each (target_reward's k counts from the first recurrent
 * inference), weights [batch]; a row's total is weight[b] / batch * (...), as one call per inference gives it.  losses / preds /
 * d_*: stacked like their logits.  steps == 1 is hz_muzero_head_losses. */
int hz_muzero_unrolled_losses(const void* value_logits, int64_t value_stride, const void* reward_logits, int64_t reward_stride,
                              const void* policy_logits, int64_t policy_stride, int batch, int steps, int support_size, int support_min,
                              int num_actions, int dtype, const float* target_value, int64_t target_value_stride,
                              int64_t target_value_step_stride, const float* target_reward, int64_t target_reward_stride,
                              int64_t target_reward_step_stride, const float* target_policy, int64_t target_policy_stride,
                              int64_t target_policy_step_stride, const float* weights, float value_coeff, float reward_coeff,
                              float policy_coeff, void* d_value, void* d_reward, void* d_policy, float* losses, float* preds, void* stream);

/* out[r] = [state[r] (hidden elements) | one-hot(action[r]) (num_actions elements)]: the dynamics net's input rows
 * (config/hanabi_control/model.py:215-219) in one launch (PyTorch: zeros + scatter_ + cat).  16-bit; strides in elements. */
int hz_state_action_rows(const void* state, int64_t state_stride, const int64_t* action, int64_t action_stride, int rows, int hidden,
                         int num_actions, void* out, int64_t out_stride, int dtype, void* stream);

/* out_x[r][:] = x[r][:] * round_to_dtype(factor[r]) for x = a, c (rows x width, contiguous) and b ((rows - lag) x width_b: its row r - lag
 * takes factor[r]; NULL: none): the three heads' logit gradients times the upstream gradient of their row's loss in one launch
 * (PyTorch: three `d * g.to(d.dtype).unsqueeze(1)`). */
int hz_scale_rows3(const void* a, int width_a, const void* b, int width_b, const void* c, int width_c, const float* factor, int rows, int lag,
                   void* out_a, void* out_b, void* out_c, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HZ_TRAIN_H */
