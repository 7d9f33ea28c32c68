/* hz_netglue.h -- C ABI of the elementwise glue between the network GEMMs of the search loop.
 * The GEMMs themselves stay in PyTorch-ROCm (hipBLASLt / rocBLAS, MFMA); bias and ReLU ride in their epilogues.
 * What is left between them in config/hanabi_control/model.py -- the residual add + ReLU of NewDynamicNet
 * (:123-124), DynamicNet (:81-82), NewResMLP (:54-56) -- is one kernel here instead of two eager ones.
 * Conventions as include/hz_tree.h. */
#ifndef HZ_NETGLUE_H
#define HZ_NETGLUE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* y[r][c] = max(0, y[r][c] + res[r][c]) in place, r < rows, c < cols; row strides in elements;
 * dtype: HZ_F32 / HZ_BF16 / HZ_F16 (include/hz_tree.h).  16 B per lane when cols, strides and pointers allow. */
int hz_add_relu(void* y, int64_t y_stride, const void* res, int64_t res_stride, int rows, int cols, int dtype,
                void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HZ_NETGLUE_H */
