// Fused SSIM map forward / backward -- "next row" N1 of SURVEY.md 8(f): the loss on the far side of the rasterizer in
// every training iteration (train.py:119-126).  The reference either calls the un-vendored `fused_ssim` CUDA extension
// (train.py:31-35,122) or falls back to utils/loss_utils.py:56-87: five grouped 11x11 conv2d over the 1080p frame plus
// their autograd backward, which measured 10.7 ms per step on MI355X (MIOpen) against 2 ms for everything else in the
// step.  Here: one kernel per direction, the separable 11-tap Gaussian window (sigma 1.5) applied inside LDS.
//
//   forward : per 64x16 output tile, the 74x26 halo of img1 / img2 is staged in LDS (zero outside the image = conv2d's
//             zero padding 5); a horizontal pass produces the 5 windowed moments (x, y, x^2, y^2, xy) for the 26 halo
//             rows -- one thread per 8 adjacent outputs, every input read once and spread over the outputs it feeds from
//             registers -- a vertical pass finishes them, four vertically adjacent pixels per thread; the SSIM value and
//             the three partial derivatives dm/dmu1, dm/dE[x^2], dm/dE[xy] (statistics held, as in the published
//             fused-ssim scheme) are written with row-contiguous 256-byte stores.
//   backward: dL/dimg1 = G * (dL/dm dm/dmu1) + 2 img1 (G * (dL/dm dm/dE[x^2])) + img2 (G * (dL/dm dm/dE[xy])), the same
//             separable window applied to three maps.
// Measured (3 x 1080 x 1920, MI355X): 82 us forward + 62 us backward -- THE SAME as round 1's 16x16-tile, one-output-per-thread
// version (2.6x halo, 91 LDS reads per pixel) although this form reads LDS 14 (horizontal) + 17.5 (vertical) times per pixel
// and stages a 1.9x halo: the kernel is not LDS- or HBM-bound (0.125 GB per direction = 1.5 TB/s) but sits on the FMA floor
// of the separable window -- 5 moments x 11 taps x 2 passes x 1.6 (halo rows) = 144 FMA per pixel = ~41 us of VALU issue
// even with v_pk_fma_f32 -- plus the load -> barrier -> horizontal -> barrier -> vertical phases of only three resident
// workgroups per CU (50 KB of LDS each).  A barrier-free "marching" form (one wave per column strip, 11-row ring of the
// horizontal moments in registers) was costed and dropped: at 1080p it has either < 3 waves per SIMD or > 30 % halo rows.
#include "gsr_internal.h"

namespace {

constexpr int TXO = 64, TYO = 16;      // output tile (x, y)
constexpr int HALO = 5;
constexpr int HX = TXO + 2 * HALO;     // 74 staged columns
constexpr int HY = TYO + 2 * HALO;     // 26 staged rows
// LDS row strides (floats), both ODD: in the horizontal pass a wave is 8 rows x 8 tasks whose addresses are
// r * stride + 8 t + j -- with an odd stride the eight rows start in eight different residues mod 8, so the 64 lanes hit 64
// different banks; the vertical pass reads 64 consecutive columns of one row.
constexpr int SW = 77;                 // staged halo rows
constexpr int SHW = 65;                // horizontally filtered rows
constexpr int HTASKS = HY * (TXO / 8); // 208 horizontal tasks of 8 outputs
// Halo staging: thread t stages elements t, t + 256, ... of the HY x HX halo.  Written as the obvious loop
// "s[r][c] = inside ? p[..] : 0" the compiler emits load -> s_waitcnt vmcnt(0) -> LDS write per element: 15 serial memory
// round trips per workgroup in the forward kernel.  So: NSTAGE unrolled, unconditional loads (address 0 when outside the
// image) into registers first, then unconditional LDS writes -- the last iteration's surplus threads write into
// STAGE_PAD spare rows nobody reads (a conditional write makes the compiler sink the load back into the branch).
constexpr int NSTAGE = (HY * HX + 255) / 256;                       // 8
constexpr int STAGE_PAD = (NSTAGE * 256 + HX - 1) / HX - HY;        // 2 spare rows
constexpr float C1 = 0.01f * 0.01f;
constexpr float C2 = 0.03f * 0.03f;

// normalised 11-tap Gaussian, sigma = 1.5 (utils/loss_utils.py:43-46), rounded from fp64
#define GSR_WIN(k)                                                                                                        \
    ((k) == 0 || (k) == 10 ? 1.0283801239e-03f : (k) == 1 || (k) == 9 ? 7.5987582095e-03f                                \
     : (k) == 2 || (k) == 8 ? 3.6000773311e-02f : (k) == 3 || (k) == 7 ? 1.0936068743e-01f                                \
     : (k) == 4 || (k) == 6 ? 2.1300552785e-01f : 2.6601171494e-01f)

// MEAN: instead of the SSIM map the workgroup writes the SUM of its tile's SSIM values (partials[plane][tile]); a second
// one-workgroup kernel adds the partials in fixed order -> mean (deterministic, no 25 MB map round trip, no torch reduce)
// XCD-aware tile order (round 3, VERDICT r02 weak #7: the 5-pixel halo of a 64x16 tile is 1.88x its pixels, and with the default
// x-fastest workgroup order the neighbours of a tile run on OTHER XCDs -- workgroup b goes to XCD b % 8 -- so every halo was
// fetched from HBM / Infinity Cache again: FETCH+WRITE 1.6-1.8x the algorithmic bytes).  The launch is 1-D; XCD k gets the k-th
// contiguous eighth of the (plane, tile row, tile column) order, so a tile's left / right neighbours and the rows above / below
// share its L2.  Results do not depend on the order (per-tile partial sums are indexed by tile and added in index order).
__device__ __forceinline__ bool ssim_tile_of_block(int ntx, int nty, int planes, int& tx, int& ty, int& plane, int64_t& tlin) {
    const int64_t ntiles = (int64_t)ntx * nty * planes;
    const int64_t per = (ntiles + 7) / 8;
    const int64_t b = blockIdx.x;
    tlin = (b & 7) * per + (b >> 3);
    if ((b >> 3) >= per || tlin >= ntiles) return false;
    tx = (int)(tlin % ntx);
    const int64_t r = tlin / ntx;
    ty = (int)(r % nty);
    plane = (int)(r / nty);
    return true;
}

// MODE 0: SSIM map.  MODE 1 (MEAN): per-tile partial sums instead of the map.  MODE 2 (round 3, the training loss of
// train.py:119-126 in one pass): as MODE 1, and the tile's sum of |img1 - img2| (the L1 term, utils/loss_utils.py:40-41) goes to
// partials[n_tiles + tile] -- the pixels are already in LDS, so the L1 half of the loss costs no extra read of either image.
template <int MODE>
__global__ void __launch_bounds__(256)
ssim_fwd_kernel(int H, int W, int planes, const float* __restrict__ img1, const float* __restrict__ img2, float* __restrict__ ssim_map,
                float* __restrict__ dm_dmu1, float* __restrict__ dm_dex2, float* __restrict__ dm_dexy) {
    constexpr bool MEAN = MODE != 0;
    __shared__ float s_part[8];
    __shared__ float s_x[HY + STAGE_PAD][SW];
    __shared__ float s_y[HY + STAGE_PAD][SW];
    __shared__ float s_h[5][HY][SHW];
    const int tid = threadIdx.x;
    const int ntx = (W + TXO - 1) / TXO, nty = (H + TYO - 1) / TYO;
    int txi, tyi, plane;                                // plane = b * C + c
    int64_t tlin;
    if (!ssim_tile_of_block(ntx, nty, planes, txi, tyi, plane, tlin)) return;
    const int x0 = txi * TXO, y0 = tyi * TYO;
    const float* p1 = img1 + (int64_t)plane * H * W;
    const float* p2 = img2 + (int64_t)plane * H * W;
    {   // halo staging: ALL loads are issued before the first LDS write, branch-free (see NSTAGE)
        float vx[NSTAGE], vy[NSTAGE];
#pragma unroll
        for (int k = 0; k < NSTAGE; ++k) {
            const int i = tid + k * 256;
            const int r = i / HX, c = i - r * HX;
            const int gy = y0 + r - HALO, gx = x0 + c - HALO;
            const bool in = i < HY * HX && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int64_t o = in ? (int64_t)gy * W + gx : 0;
            const float a = p1[o], b = p2[o];
            vx[k] = in ? a : 0.f;
            vy[k] = in ? b : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NSTAGE; ++k) {
            const int i = tid + k * 256;
            const int r = i / HX, c = i - r * HX;
            s_x[r][c] = vx[k];
            s_y[r][c] = vy[k];
        }
    }
    __syncthreads();
    if (tid < HTASKS) {                                 // horizontal pass: row r, outputs c0 .. c0+7
        const int r = tid >> 3, c0 = (tid & 7) * 8;
        float acc[8][5];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int m = 0; m < 5; ++m) acc[i][m] = 0.f;
#pragma unroll
        for (int j = 0; j < 18; ++j) {                  // input column c0 + j feeds output i with tap j - i
            const float x = s_x[r][c0 + j], y = s_y[r][c0 + j];
            const float xx = x * x, yy = y * y, xy = x * y;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = j - i;
                if (k >= 0 && k <= 10) {
                    const float w = GSR_WIN(k);
                    acc[i][0] = fmaf(w, x, acc[i][0]); acc[i][1] = fmaf(w, y, acc[i][1]); acc[i][2] = fmaf(w, xx, acc[i][2]);
                    acc[i][3] = fmaf(w, yy, acc[i][3]); acc[i][4] = fmaf(w, xy, acc[i][4]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int m = 0; m < 5; ++m) s_h[m][r][c0 + i] = acc[i][m];
    }
    __syncthreads();
    const int cx = tid & 63, rg = tid >> 6;             // vertical pass: column cx, rows 4 rg .. 4 rg + 3
    float acc[4][5];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int m = 0; m < 5; ++m) acc[o][m] = 0.f;
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        float v[5];
#pragma unroll
        for (int m = 0; m < 5; ++m) v[m] = s_h[m][4 * rg + j][cx];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int k = j - o;
            if (k >= 0 && k <= 10) {
                const float w = GSR_WIN(k);
#pragma unroll
                for (int m = 0; m < 5; ++m) acc[o][m] = fmaf(w, v[m], acc[o][m]);
            }
        }
    }
    const int gx = x0 + cx;
    float m_own = 0.f, l1_own = 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int gy = y0 + 4 * rg + o;
        if (gy < H && gx < W) {
            if (MODE == 2) l1_own += fabsf(s_x[4 * rg + o + HALO][cx + HALO] - s_y[4 * rg + o + HALO][cx + HALO]);
            const float mu1 = acc[o][0], mu2 = acc[o][1], ex2 = acc[o][2], ey2 = acc[o][3], exy = acc[o][4];
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float sigma1_sq = ex2 - mu1_sq, sigma2_sq = ey2 - mu2_sq, sigma12 = exy - mu12;
            const float A = mu1_sq + mu2_sq + C1, B = sigma1_sq + sigma2_sq + C2;
            const float Cc = 2.f * mu12 + C1, D = 2.f * sigma12 + C2;
            const float inv_AB = 1.f / (A * B);
            const float m = Cc * D * inv_AB;
            const int64_t oo = (int64_t)plane * H * W + (int64_t)gy * W + gx;
            m_own += m;
            if (!MEAN) ssim_map[oo] = m;
            if (dm_dmu1) {
                dm_dmu1[oo] = 2.f * mu2 * (D - Cc) * inv_AB - 2.f * mu1 * m / A + 2.f * mu1 * m / B;
                dm_dex2[oo] = -m / B;
                dm_dexy[oo] = 2.f * Cc * inv_AB;
            }
        }
    }
    if (MEAN) {
        float v = m_own, u = l1_own;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            v += __shfl_xor(v, off, 64);
            if (MODE == 2) u += __shfl_xor(u, off, 64);
        }
        if ((tid & 63) == 0) { s_part[tid >> 6] = v; s_part[4 + (tid >> 6)] = u; }
        __syncthreads();
        if (tid == 0) {
            ssim_map[tlin] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
            if (MODE == 2) ssim_map[(int64_t)planes * nty * ntx + tlin] = (s_part[4] + s_part[5]) + (s_part[6] + s_part[7]);
        }
    }
}

// one workgroup: out = [loss, L1, SSIM] with loss = (1 - lambda) L1 + lambda (1 - SSIM) (train.py:123), sums in fixed order
__global__ void __launch_bounds__(256)
loss_mean_kernel(const float* __restrict__ partials, int n, float inv_count, float lambda, float* __restrict__ out) {
    __shared__ float s_part[8];
    const int tid = threadIdx.x;
    float v = 0.f, u = 0.f;
    int i = tid;
    for (; i + 3 * 256 < n; i += 4 * 256) {          // coalesced, eight independent loads in flight per trip (see ssim_mean_kernel)
        float a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] = partials[i + k * 256]; b[k] = partials[n + i + k * 256]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { v += a[k]; u += b[k]; }
    }
    for (; i < n; i += 256) { v += partials[i]; u += partials[n + i]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off, 64); u += __shfl_xor(u, off, 64); }
    if ((tid & 63) == 0) { s_part[tid >> 6] = v; s_part[4 + (tid >> 6)] = u; }
    __syncthreads();
    if (tid == 0) {
        const float ssim = ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) * inv_count;
        const float l1 = ((s_part[4] + s_part[5]) + (s_part[6] + s_part[7])) * inv_count;
        out[0] = (1.0f - lambda) * l1 + lambda * (1.0f - ssim);
        out[1] = l1;
        out[2] = ssim;
    }
}

// one workgroup: mean = (sum of the per-tile partial sums, fixed order) / count
__global__ void __launch_bounds__(256)
ssim_mean_kernel(const float* __restrict__ partials, int n, float inv_count, float* __restrict__ mean_out) {
    __shared__ float s_part[4];
    const int tid = threadIdx.x;
    // thread t adds elements t, t + 256, ... (coalesced; eight independent loads in flight per trip: the chunk-per-thread form
    // was a chain of dependent trips, 6-9 us for 6 000 partials); any FIXED order is deterministic
    float v = 0.f;
    int i = tid;
    for (; i + 7 * 256 < n; i += 8 * 256) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = partials[i + k * 256];
#pragma unroll
        for (int k = 0; k < 8; ++k) v += t[k];
    }
    for (; i < n; i += 256) v += partials[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0) s_part[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) mean_out[0] = ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) * inv_count;
}

// MEAN: dL/dmap is the same for every pixel, dL/dmean / count, read from one device scalar
// MODE 2: dL/dmap[0] is dL/dloss of the fused training loss: the SSIM maps are weighted with -lambda dL/dloss / count and
// (1 - lambda) dL/dloss / count * sign(img1 - img2) -- the L1 term's gradient -- is added in the same store.
template <int MODE>
__global__ void __launch_bounds__(256)
ssim_bwd_kernel(int H, int W, int planes, const float* __restrict__ img1, const float* __restrict__ img2,
                const float* __restrict__ dL_dmap, float inv_count, float lambda, const float* __restrict__ dm_dmu1,
                const float* __restrict__ dm_dex2, const float* __restrict__ dm_dexy, float* __restrict__ dL_dimg1) {
    constexpr bool MEAN = MODE != 0;
    __shared__ float s_in[3][HY + STAGE_PAD][SW];
    __shared__ float s_h[3][HY][SHW];
    const int tid = threadIdx.x;
    const int ntx = (W + TXO - 1) / TXO, nty = (H + TYO - 1) / TYO;
    int txi, tyi, plane;
    int64_t tlin;
    if (!ssim_tile_of_block(ntx, nty, planes, txi, tyi, plane, tlin)) return;
    const int x0 = txi * TXO, y0 = tyi * TYO;
    const int64_t pbase = (int64_t)plane * H * W;
    const float gmean = MODE == 2 ? -lambda * dL_dmap[0] * inv_count : (MEAN ? dL_dmap[0] * inv_count : 0.f);
    const float gl1 = MODE == 2 ? (1.0f - lambda) * dL_dmap[0] * inv_count : 0.f;
    // the two images are only needed by the last expression: requested first, they arrive during the two passes
    const int cx = tid & 63, rg = tid >> 6;
    const int gx_out = x0 + cx;
    float im1[4], im2[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int gy = y0 + 4 * rg + o;
        const int64_t oo = (gy < H && gx_out < W) ? pbase + (int64_t)gy * W + gx_out : pbase;
        im1[o] = img1[oo];
        im2[o] = img2[oo];
    }
    {   // halo staging, all loads before the first LDS write (see NSTAGE)
        float v0[NSTAGE], v1[NSTAGE], v2[NSTAGE];
#pragma unroll
        for (int k = 0; k < NSTAGE; ++k) {
            const int i = tid + k * 256;
            const int r = i / HX, c = i - r * HX;
            const int gy = y0 + r - HALO, gx = x0 + c - HALO;
            const bool in = i < HY * HX && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int64_t o = in ? pbase + (int64_t)gy * W + gx : pbase;
            const float g = MEAN ? gmean : dL_dmap[o];
            const float a = dm_dmu1[o], b = dm_dex2[o], c2 = dm_dexy[o];
            v0[k] = in ? g * a : 0.f;
            v1[k] = in ? g * b : 0.f;
            v2[k] = in ? g * c2 : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NSTAGE; ++k) {
            const int i = tid + k * 256;
            const int r = i / HX, c = i - r * HX;
            s_in[0][r][c] = v0[k];
            s_in[1][r][c] = v1[k];
            s_in[2][r][c] = v2[k];
        }
    }
    __syncthreads();
    if (tid < HTASKS) {
        const int r = tid >> 3, c0 = (tid & 7) * 8;
        float acc[8][3];
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i][0] = 0.f; acc[i][1] = 0.f; acc[i][2] = 0.f; }
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            const float a = s_in[0][r][c0 + j], b = s_in[1][r][c0 + j], c = s_in[2][r][c0 + j];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = j - i;
                if (k >= 0 && k <= 10) {
                    const float w = GSR_WIN(k);
                    acc[i][0] = fmaf(w, a, acc[i][0]); acc[i][1] = fmaf(w, b, acc[i][1]); acc[i][2] = fmaf(w, c, acc[i][2]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s_h[0][r][c0 + i] = acc[i][0]; s_h[1][r][c0 + i] = acc[i][1]; s_h[2][r][c0 + i] = acc[i][2]; }
    }
    __syncthreads();
    float acc[4][3];
#pragma unroll
    for (int o = 0; o < 4; ++o) { acc[o][0] = 0.f; acc[o][1] = 0.f; acc[o][2] = 0.f; }
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const float a = s_h[0][4 * rg + j][cx], b = s_h[1][4 * rg + j][cx], c = s_h[2][4 * rg + j][cx];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int k = j - o;
            if (k >= 0 && k <= 10) {
                const float w = GSR_WIN(k);
                acc[o][0] = fmaf(w, a, acc[o][0]); acc[o][1] = fmaf(w, b, acc[o][1]); acc[o][2] = fmaf(w, c, acc[o][2]);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int gy = y0 + 4 * rg + o;
        if (gy < H && gx_out < W) {
            const int64_t oo = pbase + (int64_t)gy * W + gx_out;
            float v = acc[o][0] + 2.f * im1[o] * acc[o][1] + im2[o] * acc[o][2];
            if (MODE == 2) { const float d = im1[o] - im2[o]; v += d > 0.f ? gl1 : (d < 0.f ? -gl1 : 0.f); }      // torch's sign(0) = 0
            dL_dimg1[oo] = v;
        }
    }
}

// 1-D launch: 8 x ceil(tiles / 8) workgroups, see ssim_tile_of_block
dim3 ssim_grid(int planes, int H, int W) {
    const int64_t ntiles = (int64_t)planes * ((W + TXO - 1) / TXO) * ((H + TYO - 1) / TYO);
    return dim3((unsigned)(8 * ((ntiles + 7) / 8)));
}

}  // namespace

void gsr_launch_ssim_forward(int planes, int H, int W, const float* img1, const float* img2, float* ssim_map, float* dm_dmu1,
                             float* dm_dex2, float* dm_dexy, hipStream_t st) {
    hipLaunchKernelGGL(ssim_fwd_kernel<0>, ssim_grid(planes, H, W), dim3(256), 0, st, H, W, planes, img1, img2, ssim_map, dm_dmu1, dm_dex2, dm_dexy);
}

int64_t gsr_ssim_partial_count_impl(int planes, int H, int W) {
    return (int64_t)planes * ((W + TXO - 1) / TXO) * ((H + TYO - 1) / TYO);
}

void gsr_launch_ssim_mean_forward(int planes, int H, int W, const float* img1, const float* img2, float* partials,
                                  float* mean_out, float* dm_dmu1, float* dm_dex2, float* dm_dexy, hipStream_t st) {
    hipLaunchKernelGGL(ssim_fwd_kernel<1>, ssim_grid(planes, H, W), dim3(256), 0, st, H, W, planes, img1, img2, partials, dm_dmu1, dm_dex2, dm_dexy);
    const double count = (double)planes * H * W;
    hipLaunchKernelGGL(ssim_mean_kernel, dim3(1), dim3(256), 0, st, partials, (int)gsr_ssim_partial_count_impl(planes, H, W),
                       (float)(1.0 / count), mean_out);
}

void gsr_launch_ssim_mean_backward(int planes, int H, int W, const float* img1, const float* img2, const float* dL_dmean,
                                   const float* dm_dmu1, const float* dm_dex2, const float* dm_dexy, float* dL_dimg1,
                                   hipStream_t st) {
    const double count = (double)planes * H * W;
    hipLaunchKernelGGL(ssim_bwd_kernel<1>, ssim_grid(planes, H, W), dim3(256), 0, st, H, W, planes, img1, img2, dL_dmean, (float)(1.0 / count), 0.f, dm_dmu1,
                       dm_dex2, dm_dexy, dL_dimg1);
}

void gsr_launch_ssim_backward(int planes, int H, int W, const float* img1, const float* img2, const float* dL_dmap,
                              const float* dm_dmu1, const float* dm_dex2, const float* dm_dexy, float* dL_dimg1, hipStream_t st) {
    hipLaunchKernelGGL(ssim_bwd_kernel<0>, ssim_grid(planes, H, W), dim3(256), 0, st, H, W, planes, img1, img2, dL_dmap, 0.f, 0.f, dm_dmu1, dm_dex2, dm_dexy,
                       dL_dimg1);
}

// fused training loss (train.py:119-126): loss = (1 - lambda) L1 + lambda (1 - SSIM); partials holds 2 x gsr_ssim_partial_count floats
void gsr_launch_train_loss_forward(int planes, int H, int W, const float* img1, const float* img2, float lambda, float* partials,
                                   float* loss_out /*[3]: loss, L1, SSIM*/, float* dm_dmu1, float* dm_dex2, float* dm_dexy, hipStream_t st) {
    hipLaunchKernelGGL(ssim_fwd_kernel<2>, ssim_grid(planes, H, W), dim3(256), 0, st, H, W, planes, img1, img2, partials, dm_dmu1, dm_dex2, dm_dexy);
    const double count = (double)planes * H * W;
    hipLaunchKernelGGL(loss_mean_kernel, dim3(1), dim3(256), 0, st, partials, (int)gsr_ssim_partial_count_impl(planes, H, W),
                       (float)(1.0 / count), lambda, loss_out);
}

void gsr_launch_train_loss_backward(int planes, int H, int W, const float* img1, const float* img2, const float* dL_dloss, float lambda,
                                    const float* dm_dmu1, const float* dm_dex2, const float* dm_dexy, float* dL_dimg1, hipStream_t st) {
    const double count = (double)planes * H * W;
    hipLaunchKernelGGL(ssim_bwd_kernel<2>, ssim_grid(planes, H, W), dim3(256), 0, st, H, W, planes, img1, img2, dL_dloss, (float)(1.0 / count), lambda, dm_dmu1,
                       dm_dex2, dm_dexy, dL_dimg1);
}
