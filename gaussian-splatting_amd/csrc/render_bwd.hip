// Backward of the alpha-compositing pass -- replaces renderCUDA (backward) of the un-vendored reference
// rasterizer (algorithm SURVEY.md Appendix A.5; reached through loss.backward(), train.py:142).
//
// The reference issues ~10 global float atomics per contributing (pixel, Gaussian) pair.  gfx950 design:
//   * one wave64 per 8x8 pixel block (same mapping as the forward), walking the tile's list BACK TO FRONT in
//     batches of 64 with the same exact box test as the forward (a culled entry contributed to no pixel of the
//     box, so its gradient from this box is exactly zero);
//   * the list is cut at the wave-wide maximum of n_contrib (nothing behind it contributed);
//   * per surviving entry every lane computes its pixel's 10 partial derivatives, which are summed over the 64
//     lanes with DPP row-shift / row-broadcast adds (no LDS), and ONE lane issues the atomics into a packed
//     48-byte per-Gaussian gradient record -> 64x fewer atomics than per-pair, one cache line per Gaussian.
//   * entries to which no lane of the wave contributed skip the reduction entirely.
//
// Output record layout (float[12] per Gaussian, "splat_grads"):
//   0 dL/dpx  1 dL/dpy  (pixel units)   2 dL/dA  3 dL/dB  4 dL/dC  (plain derivatives of the conic entries,
//   power = -0.5(A dx^2 + C dy^2) - B dx dy)   5 dL/d(opacity*aa)   6,7,8 dL/d(rgb)   9 dL/d(1/depth)   10,11 pad
#include "gsr_internal.h"

namespace {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}

// sum over the 64 lanes; the total is valid in lane 63 only
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0x111, 0xf>(v);   // row_shr:1
    v = dpp_add<0x112, 0xf>(v);   // row_shr:2
    v = dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add<0x118, 0xf>(v);   // row_shr:8
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ float min_q_over_box(float mx, float my, float A, float B, float C, float x0, float x1,
                                                float y0, float y1) {
    const float lx = x0 - mx, hx = x1 - mx, ly = y0 - my, hy = y1 - my;
    const bool in_x = (lx <= 0.0f) && (hx >= 0.0f);
    const bool in_y = (ly <= 0.0f) && (hy >= 0.0f);
    float q = 3.0e38f;
    if (in_x && in_y) return 0.0f;
    if (!in_x) {
        const float dx = lx > 0.0f ? lx : hx;
        const float dy = fminf(hy, fmaxf(ly, -B * dx / C));
        q = fminf(q, A * dx * dx + 2.0f * B * dx * dy + C * dy * dy);
    }
    if (!in_y) {
        const float dy = ly > 0.0f ? ly : hy;
        const float dx = fminf(hx, fmaxf(lx, -B * dy / A));
        q = fminf(q, A * dx * dx + 2.0f * B * dx * dy + C * dy * dy);
    }
    return q;
}

__device__ __forceinline__ float bcast(float v, int srclane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srclane));
}

__global__ void __launch_bounds__(64)
render_bwd_wave(GsrCamDev cam, int n_band_tiles, const uint2* __restrict__ ranges,
                const uint32_t* __restrict__ point_list, const float4* __restrict__ splats,
                const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                const float* __restrict__ dL_dpix, const float* __restrict__ dL_dinvdepth, float* __restrict__ grads) {
    const int b = blockIdx.x;
    const int grp = b >> 5, r32 = b & 31;
    const int tile_local = grp * 8 + (r32 & 7);
    const int quad = r32 >> 3;
    if (tile_local >= n_band_tiles) return;
    const int tile = cam.tile_y0 * cam.gx + tile_local;
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const int lane = threadIdx.x;
    const int bx0 = tx * GSR_TILE + (quad & 1) * 8, by0 = ty * GSR_TILE + (quad >> 1) * 8;
    if (bx0 >= cam.W || by0 >= cam.H) return;
    const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
    const bool inside = px < cam.W && py < cam.H;
    const float pxf = (float)px, pyf = (float)py;
    const float x0 = (float)bx0, x1 = (float)min(bx0 + 7, cam.W - 1);
    const float y0 = (float)by0, y1 = (float)min(by0 + 7, cam.H - 1);
    const uint2 range = ranges[tile];
    const int64_t pix = (int64_t)py * cam.W + px;
    const int64_t HW = (int64_t)cam.H * cam.W;

    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t last_contrib = inside ? n_contrib[pix] : 0u;
    const float dLr = inside ? dL_dpix[pix] : 0.f;
    const float dLg = inside ? dL_dpix[HW + pix] : 0.f;
    const float dLb = inside ? dL_dpix[2 * HW + pix] : 0.f;
    const float dLd = (inside && dL_dinvdepth) ? dL_dinvdepth[pix] : 0.f;
    const float bg_dot = cam.bg[0] * dLr + cam.bg[1] * dLg + cam.bg[2] * dLb;

    // wave-wide max of n_contrib: nothing at list position >= that contributed to any pixel of the box
    uint32_t max_contrib = last_contrib;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) max_contrib = max(max_contrib, (uint32_t)__shfl_xor((int)max_contrib, off, 64));
    const uint32_t nlist = range.y - range.x;
    const uint32_t end = min(nlist, max_contrib);
    if (end == 0) return;

    float T = T_final;
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f;
    float last_alpha = 0.f, last_r = 0.f, last_g = 0.f, last_b = 0.f, last_d = 0.f;

    for (int bstart = (int)((end - 1) & ~63u); bstart >= 0; bstart -= 64) {
        const uint32_t n = min(64u, end - (uint32_t)bstart);
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
        float colb = 0.f, invd = 0.f;
        uint32_t id = 0;
        bool keep = false;
        if ((uint32_t)lane < n) {
            id = point_list[range.x + bstart + lane];
            q0 = splats[id * 3 + 0];
            q1 = splats[id * 3 + 1];
            const float4 q2 = splats[id * 3 + 2];
            colb = q2.x;
            invd = 1.0f / q2.y;
            const float tau = 2.0f * __logf(255.0f * q1.y) + 0.01f;
            const float qmin = min_q_over_box(q0.x, q0.y, q0.z, q0.w, q1.x, x0, x1, y0, y1);
            keep = !(qmin > tau);
        }
        uint64_t mask = __ballot(keep);
        while (mask) {
            const int j = 63 - __builtin_clzll(mask);
            mask &= ~(1ull << j);
            const uint32_t pos0 = (uint32_t)bstart + (uint32_t)j;   // 0-based list position
            const float gx_ = bcast(q0.x, j), gy_ = bcast(q0.y, j), cA = bcast(q0.z, j), cB = bcast(q0.w, j);
            const float cC = bcast(q1.x, j), op = bcast(q1.y, j), cr = bcast(q1.z, j), cg = bcast(q1.w, j);
            const float cb = bcast(colb, j), idp = bcast(invd, j);
            const uint32_t gid = (uint32_t)__builtin_amdgcn_readlane((int)id, j);

            const float dx = gx_ - pxf, dy = gy_ - pyf;
            const float power = -0.5f * (cA * dx * dx + cC * dy * dy) - cB * dx * dy;
            const float G = __expf(power);
            const float alpha = fminf(GSR_ALPHA_MAX, op * G);
            const bool active = (pos0 < last_contrib) && (power <= 0.0f) && (alpha >= GSR_ALPHA_MIN);
            if (__ballot(active) == 0ull) continue;

            float g_px = 0.f, g_py = 0.f, g_A = 0.f, g_B = 0.f, g_C = 0.f, g_op = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f, g_d = 0.f;
            if (active) {
                const float inv1ma = __builtin_amdgcn_rcpf(1.0f - alpha);
                T = T * inv1ma;
                const float w = alpha * T;
                // colour
                acc_r = last_alpha * last_r + (1.f - last_alpha) * acc_r;
                acc_g = last_alpha * last_g + (1.f - last_alpha) * acc_g;
                acc_b = last_alpha * last_b + (1.f - last_alpha) * acc_b;
                acc_d = last_alpha * last_d + (1.f - last_alpha) * acc_d;
                last_r = cr; last_g = cg; last_b = cb; last_d = idp;
                float dL_dalpha = (cr - acc_r) * dLr + (cg - acc_g) * dLg + (cb - acc_b) * dLb + (idp - acc_d) * dLd;
                g_r = w * dLr; g_g = w * dLg; g_b = w * dLb; g_d = w * dLd;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final * inv1ma) * bg_dot;
                const float dL_dG = op * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                g_px = dL_dG * (-gdx * cA - gdy * cB);
                g_py = dL_dG * (-gdy * cC - gdx * cB);
                g_A = -0.5f * gdx * dx * dL_dG;
                g_B = -gdx * dy * dL_dG;
                g_C = -0.5f * gdy * dy * dL_dG;
                g_op = G * dL_dalpha;
            }
            g_px = wave_sum_to_lane63(g_px); g_py = wave_sum_to_lane63(g_py);
            g_A = wave_sum_to_lane63(g_A); g_B = wave_sum_to_lane63(g_B); g_C = wave_sum_to_lane63(g_C);
            g_op = wave_sum_to_lane63(g_op);
            g_r = wave_sum_to_lane63(g_r); g_g = wave_sum_to_lane63(g_g); g_b = wave_sum_to_lane63(g_b);
            g_d = wave_sum_to_lane63(g_d);
            if (lane == 63) {
                float* o = grads + (int64_t)gid * 12;
                atomicAdd(o + 0, g_px); atomicAdd(o + 1, g_py);
                atomicAdd(o + 2, g_A); atomicAdd(o + 3, g_B); atomicAdd(o + 4, g_C);
                atomicAdd(o + 5, g_op);
                atomicAdd(o + 6, g_r); atomicAdd(o + 7, g_g); atomicAdd(o + 8, g_b);
                atomicAdd(o + 9, g_d);
            }
        }
    }
}

}  // namespace

void gsr_launch_render_backward(const GsrCamDev& cam, const uint2* ranges, const uint32_t* point_list,
                                const float4* splats, const float* final_T, const uint32_t* n_contrib,
                                const float* dL_dpix, const float* dL_dinvdepth, float* splat_grads, int variant,
                                hipStream_t st) {
    (void)variant;
    const int n_band_tiles = cam.gx * (cam.tile_y1 - cam.tile_y0);
    if (n_band_tiles <= 0) return;
    const int groups = (n_band_tiles + 7) / 8;
    hipLaunchKernelGGL(render_bwd_wave, dim3(groups * 32), dim3(64), 0, st, cam, n_band_tiles, ranges, point_list,
                       splats, final_T, n_contrib, dL_dpix, dL_dinvdepth, splat_grads);
}
