// Fused SSIM map forward / backward -- "next row" N1 of SURVEY.md 8(f): the loss on the far side of the rasterizer in
// every training iteration (train.py:119-126).  The reference either calls the un-vendored `fused_ssim` CUDA extension
// (train.py:31-35,122) or falls back to utils/loss_utils.py:56-87: five grouped 11x11 conv2d over the 1080p frame plus
// their autograd backward, which measured 10.7 ms per step on MI355X (MIOpen) against 2 ms for everything else in the
// step.  Here: one kernel per direction, the separable 11-tap Gaussian window (sigma 1.5) applied inside LDS.
//
//   forward : per 16x16 output tile, the 26x26 halo of img1 / img2 is staged in LDS (zero outside the image = conv2d's
//             zero padding 5), a horizontal pass produces the 5 windowed moments (x, y, x^2, y^2, xy) for the 26 halo
//             rows, a vertical pass finishes them per pixel; the SSIM value and the three partial derivatives
//             dm/dmu1, dm/dE[x^2], dm/dE[xy] (statistics held, as in the published fused-ssim scheme) are written.
//   backward: dL/dimg1 = G * (dL/dm dm/dmu1) + 2 img1 (G * (dL/dm dm/dE[x^2])) + img2 (G * (dL/dm dm/dE[xy])), the same
//             separable window applied to three maps.
// HBM traffic ~0.25 GB per direction at 3x1080x1920 -> streaming-bound by design; arithmetic is a few FMA per byte.
#include "gsr_internal.h"

namespace {

constexpr int TS = 16;            // output tile
constexpr int HALO = 5;
constexpr int TW = TS + 2 * HALO; // 26
// LDS row strides (floats).  In both passes a wave covers 4 consecutive rows x 16 columns: with a 48-float stride the
// four rows start 16 banks apart (48 r mod 64 = 0, 48, 32, 16), with 16 they are simply consecutive -- every ds_read /
// ds_write of the two passes is conflict-free (the former 27 / 17 strides measured 2.3 conflict cycles per LDS
// instruction issue cycle, profiles/r01_pmc_sq_per_kernel.csv).
constexpr int SW = 48;            // staged halo rows
constexpr int SHW = 16;           // horizontally filtered rows
constexpr float C1 = 0.01f * 0.01f;
constexpr float C2 = 0.03f * 0.03f;

// normalised 11-tap Gaussian, sigma = 1.5 (utils/loss_utils.py:43-46), rounded from fp64
__constant__ float c_win[11] = {1.0283801239e-03f, 7.5987582095e-03f, 3.6000773311e-02f, 1.0936068743e-01f,
                                2.1300552785e-01f, 2.6601171494e-01f, 2.1300552785e-01f, 1.0936068743e-01f,
                                3.6000773311e-02f, 7.5987582095e-03f, 1.0283801239e-03f};

// MEAN: instead of the SSIM map the workgroup writes the SUM of its tile's SSIM values (partials[plane][tile]); a second
// one-workgroup kernel adds the partials in fixed order -> mean (deterministic, no 25 MB map round trip, no torch reduce)
template <bool MEAN>
__global__ void __launch_bounds__(256)
ssim_fwd_kernel(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2, float* __restrict__ ssim_map,
                float* __restrict__ dm_dmu1, float* __restrict__ dm_dex2, float* __restrict__ dm_dexy) {
    __shared__ float s_part[4];
    __shared__ float s_x[TW][SW];
    __shared__ float s_y[TW][SW];
    __shared__ float s_h[5][TW][SHW];
    const int tid = threadIdx.x;
    const int plane = blockIdx.z;                       // b * C + c
    const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
    const float* p1 = img1 + (int64_t)plane * H * W;
    const float* p2 = img2 + (int64_t)plane * H * W;
    for (int i = tid; i < TW * TW; i += 256) {
        const int r = i / TW, c = i - r * TW;
        const int gy = y0 + r - HALO, gx = x0 + c - HALO;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        s_x[r][c] = in ? p1[(int64_t)gy * W + gx] : 0.f;
        s_y[r][c] = in ? p2[(int64_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < TW * TS; i += 256) {          // horizontal pass on the 26 halo rows
        const int r = i / TS, c = i - r * TS;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = c_win[k], x = s_x[r][c + k], y = s_y[r][c + k];
            a0 += w * x; a1 += w * y; a2 += w * x * x; a3 += w * y * y; a4 += w * x * y;
        }
        s_h[0][r][c] = a0; s_h[1][r][c] = a1; s_h[2][r][c] = a2; s_h[3][r][c] = a3; s_h[4][r][c] = a4;
    }
    __syncthreads();
    const int ty = tid / TS, tx = tid - ty * TS;
    const int gy = y0 + ty, gx = x0 + tx;
    float m_own = 0.f;
    if (gy < H && gx < W) {
        float mu1 = 0.f, mu2 = 0.f, ex2 = 0.f, ey2 = 0.f, exy = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = c_win[k];
            mu1 += w * s_h[0][ty + k][tx]; mu2 += w * s_h[1][ty + k][tx]; ex2 += w * s_h[2][ty + k][tx];
            ey2 += w * s_h[3][ty + k][tx]; exy += w * s_h[4][ty + k][tx];
        }
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float sigma1_sq = ex2 - mu1_sq, sigma2_sq = ey2 - mu2_sq, sigma12 = exy - mu12;
        const float A = mu1_sq + mu2_sq + C1, B = sigma1_sq + sigma2_sq + C2;
        const float Cc = 2.f * mu12 + C1, D = 2.f * sigma12 + C2;
        const float inv_AB = 1.f / (A * B);
        const float m = Cc * D * inv_AB;
        const int64_t o = (int64_t)plane * H * W + (int64_t)gy * W + gx;
        m_own = m;
        if (!MEAN) ssim_map[o] = m;
        if (dm_dmu1) {
            dm_dmu1[o] = 2.f * mu2 * (D - Cc) * inv_AB - 2.f * mu1 * m / A + 2.f * mu1 * m / B;
            dm_dex2[o] = -m / B;
            dm_dexy[o] = 2.f * Cc * inv_AB;
        }
    }
    if (MEAN) {
        float v = m_own;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((tid & 63) == 0) s_part[tid >> 6] = v;
        __syncthreads();
        if (tid == 0)
            ssim_map[((int64_t)plane * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    }
}

// one workgroup: mean = (sum of the per-tile partial sums, fixed order) / count
__global__ void __launch_bounds__(256)
ssim_mean_kernel(const float* __restrict__ partials, int n, float inv_count, float* __restrict__ mean_out) {
    __shared__ float s_part[4];
    const int tid = threadIdx.x;
    const int chunk = (n + 255) / 256;
    const int lo = min(tid * chunk, n), hi = min(lo + chunk, n);
    float v = 0.f;
    for (int i = lo; i < hi; ++i) v += partials[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0) s_part[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) mean_out[0] = ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) * inv_count;
}

// MEAN: dL/dmap is the same for every pixel, dL/dmean / count, read from one device scalar
template <bool MEAN>
__global__ void __launch_bounds__(256)
ssim_bwd_kernel(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                const float* __restrict__ dL_dmap, float inv_count, const float* __restrict__ dm_dmu1,
                const float* __restrict__ dm_dex2, const float* __restrict__ dm_dexy, float* __restrict__ dL_dimg1) {
    __shared__ float s_in[3][TW][SW];
    __shared__ float s_h[3][TW][SHW];
    const int tid = threadIdx.x;
    const int plane = blockIdx.z;
    const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
    const int64_t pbase = (int64_t)plane * H * W;
    for (int i = tid; i < TW * TW; i += 256) {
        const int r = i / TW, c = i - r * TW;
        const int gy = y0 + r - HALO, gx = x0 + c - HALO;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int64_t o = pbase + (int64_t)gy * W + gx;
        const float g = in ? (MEAN ? dL_dmap[0] * inv_count : dL_dmap[o]) : 0.f;
        s_in[0][r][c] = in ? g * dm_dmu1[o] : 0.f;
        s_in[1][r][c] = in ? g * dm_dex2[o] : 0.f;
        s_in[2][r][c] = in ? g * dm_dexy[o] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < TW * TS; i += 256) {
        const int r = i / TS, c = i - r * TS;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = c_win[k];
            a0 += w * s_in[0][r][c + k]; a1 += w * s_in[1][r][c + k]; a2 += w * s_in[2][r][c + k];
        }
        s_h[0][r][c] = a0; s_h[1][r][c] = a1; s_h[2][r][c] = a2;
    }
    __syncthreads();
    const int ty = tid / TS, tx = tid - ty * TS;
    const int gy = y0 + ty, gx = x0 + tx;
    if (gy < H && gx < W) {
        float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = c_win[k];
            a += w * s_h[0][ty + k][tx]; b += w * s_h[1][ty + k][tx]; c += w * s_h[2][ty + k][tx];
        }
        const int64_t o = pbase + (int64_t)gy * W + gx;
        dL_dimg1[o] = a + 2.f * img1[o] * b + img2[o] * c;
    }
}

}  // namespace

void gsr_launch_ssim_forward(int planes, int H, int W, const float* img1, const float* img2, float* ssim_map, float* dm_dmu1,
                             float* dm_dex2, float* dm_dexy, hipStream_t st) {
    const dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, planes);
    hipLaunchKernelGGL(ssim_fwd_kernel<false>, grid, dim3(256), 0, st, H, W, img1, img2, ssim_map, dm_dmu1, dm_dex2, dm_dexy);
}

int64_t gsr_ssim_partial_count_impl(int planes, int H, int W) {
    return (int64_t)planes * ((W + TS - 1) / TS) * ((H + TS - 1) / TS);
}

void gsr_launch_ssim_mean_forward(int planes, int H, int W, const float* img1, const float* img2, float* partials,
                                  float* mean_out, float* dm_dmu1, float* dm_dex2, float* dm_dexy, hipStream_t st) {
    const dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, planes);
    hipLaunchKernelGGL(ssim_fwd_kernel<true>, grid, dim3(256), 0, st, H, W, img1, img2, partials, dm_dmu1, dm_dex2, dm_dexy);
    const double count = (double)planes * H * W;
    hipLaunchKernelGGL(ssim_mean_kernel, dim3(1), dim3(256), 0, st, partials, (int)gsr_ssim_partial_count_impl(planes, H, W),
                       (float)(1.0 / count), mean_out);
}

void gsr_launch_ssim_mean_backward(int planes, int H, int W, const float* img1, const float* img2, const float* dL_dmean,
                                   const float* dm_dmu1, const float* dm_dex2, const float* dm_dexy, float* dL_dimg1,
                                   hipStream_t st) {
    const dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, planes);
    const double count = (double)planes * H * W;
    hipLaunchKernelGGL(ssim_bwd_kernel<true>, grid, dim3(256), 0, st, H, W, img1, img2, dL_dmean, (float)(1.0 / count), dm_dmu1,
                       dm_dex2, dm_dexy, dL_dimg1);
}

void gsr_launch_ssim_backward(int planes, int H, int W, const float* img1, const float* img2, const float* dL_dmap,
                              const float* dm_dmu1, const float* dm_dex2, const float* dm_dexy, float* dL_dimg1, hipStream_t st) {
    const dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, planes);
    hipLaunchKernelGGL(ssim_bwd_kernel<false>, grid, dim3(256), 0, st, H, W, img1, img2, dL_dmap, 0.f, dm_dmu1, dm_dex2, dm_dexy,
                       dL_dimg1);
}
