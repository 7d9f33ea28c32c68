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

#ifdef __cplusplus
}
#endif
#endif /* HZ_TRAIN_H */
