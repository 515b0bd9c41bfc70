// Density-control statistics (reference: GaussianModel.add_densification_stats, scene/gaussian_model.py:471-473,
// and the max_radii2D update of train.py:166) -- the per-iteration part of adaptive density control:
//     xyz_gradient_accum[visible] += || viewspace_grad[visible, :2] ||      denom[visible] += 1
//     max_radii2D[visible] = max(max_radii2D[visible], radii[visible])
// As torch ops on boolean masks this is ~25 small kernels and three host synchronisations (nonzero) PER TRAINING ITERATION:
// measured 0.9 ms per iteration at 1 M Gaussians, 40 % of a 2.3 ms step (bench.py's train_iters_per_s_densify leg, round 3).
// One streaming pass here: 12 + 4 (+1) bytes in, 3 read-modify-writes of 4 bytes per Gaussian.
#include "gsr_internal.h"

namespace {

__global__ void __launch_bounds__(256)
density_stats_kernel(int P, const float* __restrict__ grad /*[P,3]*/, const uint8_t* __restrict__ visible, const int32_t* __restrict__ radii,
                     float* __restrict__ accum, float* __restrict__ denom, float* __restrict__ max_radii) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t r = radii ? radii[i] : 0;
        const bool vis = visible ? visible[i] != 0 : r > 0;
        if (!vis) continue;
        const float gx = grad[i * 3 + 0], gy = grad[i * 3 + 1];
        accum[i] += sqrtf(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)));      // torch.norm(dim=-1) of two elements: no contraction
        denom[i] += 1.0f;
        if (radii) max_radii[i] = fmaxf(max_radii[i], (float)r);
    }
}

}  // namespace

void gsr_launch_density_stats(int P, const float* grad, const uint8_t* visible, const int32_t* radii, float* accum, float* denom,
                              float* max_radii, hipStream_t st) {
    int64_t nb = ((int64_t)P + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(density_stats_kernel, dim3((int)nb), dim3(256), 0, st, P, grad, visible, radii, accum, denom, max_radii);
}
