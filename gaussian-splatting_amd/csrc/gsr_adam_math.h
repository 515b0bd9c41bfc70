// The two Adam updates of the package, one definition for adam.hip (the optimizer kernels) and preprocess.hip (the per-Gaussian
// backward that applies the SH update in place, gsr_backward_preprocess_sh_adam): the same source compiled with the same
// -ffp-contract=off gives the same bits in both kernels.
#pragma once
#include <hip/hip_runtime.h>

// torch.optim.Adam (no weight decay, no amsgrad): m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;
// p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__device__ __forceinline__ void gsr_adam1(float& p, float g, float& m, float& v, float om_b1, float b2, float om_b2, float step_size,
                                          float inv_bc2_sqrt, float eps) {
    m = m + (g - m) * om_b1;
    v = v * b2 + om_b2 * g * g;
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

// SparseGaussianAdam ([RECALLED], no bias correction): m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;  p += -lr m / (sqrt(v) + eps)
__device__ __forceinline__ void gsr_sparse_adam1(float& p, float g, float& m, float& v, float lr, float b1, float om_b1, float b2,
                                                 float om_b2, float eps) {
    m = b1 * m + om_b1 * g;
    v = b2 * v + om_b2 * g * g;
    p += -lr * m / (sqrtf(v) + eps);
}
