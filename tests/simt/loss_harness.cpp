// The fused training loss of csrc/ssim.hip -- (1 - lambda) L1 + lambda (1 - SSIM) in the marching-wave SSIM kernels' single pass, forward and
// backward -- and the plain mean-SSIM form, compiled for the HOST through the SIMT-on-CPU shim (raw buffer loads / stores with the hardware's
// out-of-range behaviour) and run through their own launchers (tests/test_simt_loss_cpu.py).  TEST INFRASTRUCTURE, never part of libgsr_hip.so.
#define __HIPCC__ 1
#include "hip/hip_runtime.h"
#include "ssim.hip"
#include "simt_runtime.h"
#include <vector>

static char g_err[256];
static int finish(const char* what) {
    if (!simt::launch_error) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, simt::launch_error);
    simt::launch_error = nullptr;
    return -1;
}

extern "C" {

const char* simt_loss_last_error(void) { return g_err; }

// loss = (1 - lambda) mean|img1 - img2| + lambda (1 - SSIM(img1, img2)); dL_dimg1 for dL/dloss = upstream
int simt_train_loss(int planes, int H, int W, const float* img1, const float* img2, float lambda, float upstream, float* loss_out, float* dL_dimg1) {
    const size_t n = (size_t)planes * H * W;
    std::vector<float> partials(2 * (size_t)gsr_ssim_partial_count_impl(planes, H, W) + 64), a(n + 16), b(n + 16), c(n + 16);      // (SSIM and L1 parts)
    gsr_launch_train_loss_forward(planes, H, W, img1, img2, lambda, partials.data(), loss_out, a.data(), b.data(), c.data(), nullptr);
    if (finish("train loss forward")) return -1;
    gsr_launch_train_loss_backward(planes, H, W, img1, img2, &upstream, lambda, a.data(), b.data(), c.data(), dL_dimg1, nullptr);
    return finish("train loss backward");
}

// mean SSIM and its gradient (fused_ssim)
int simt_ssim_mean(int planes, int H, int W, const float* img1, const float* img2, float upstream, float* mean_out, float* dL_dimg1) {
    const size_t n = (size_t)planes * H * W;
    std::vector<float> partials((size_t)gsr_ssim_partial_count_impl(planes, H, W) + 64), a(n + 16), b(n + 16), c(n + 16);
    gsr_launch_ssim_mean_forward(planes, H, W, img1, img2, partials.data(), mean_out, a.data(), b.data(), c.data(), nullptr);
    if (finish("ssim mean forward")) return -1;
    gsr_launch_ssim_mean_backward(planes, H, W, img1, img2, &upstream, a.data(), b.data(), c.data(), dL_dimg1, nullptr);
    return finish("ssim mean backward");
}

}  // extern "C"
