// Host build of the product's per-Gaussian math (gaussian-splatting_amd/csrc/gsr_math.h) so that the
// arithmetic the HIP preprocess kernels execute per lane can be checked against the CPU oracle WITHOUT a GPU.
// Compiled by tests/test_host_math.py with: g++ -O2 -ffp-contract=off -shared -fPIC.
// This is a test harness for product code, not an oracle and not a fallback: nothing in the package loads it.
#include <cstring>
#include "gsr_math.h"

extern "C" {

struct HostCam {
    int W, H;
    float tanfovx, tanfovy, scale_modifier;
    int sh_degree, M, antialiasing, tile_y0, tile_y1;
    float view[16], proj[16], campos[3];
};

static GsrCam to_cam(const HostCam* h) {
    GsrCam c;
    c.W = h->W; c.H = h->H;
    c.gx = (h->W + 15) / 16; c.gy = (h->H + 15) / 16;
    c.focal_x = (float)h->W / (2.0f * h->tanfovx);
    c.focal_y = (float)h->H / (2.0f * h->tanfovy);
    c.limx = 1.3f * h->tanfovx; c.limy = 1.3f * h->tanfovy;
    c.scale_modifier = h->scale_modifier;
    c.sh_degree = h->sh_degree; c.M = h->M; c.antialiasing = h->antialiasing; c.snug = 1;
    c.tile_y0 = h->tile_y0; c.tile_y1 = h->tile_y1 <= 0 ? c.gy : h->tile_y1;
    memcpy(c.view, h->view, sizeof(c.view)); memcpy(c.proj, h->proj, sizeof(c.proj));
    memcpy(c.campos, h->campos, sizeof(c.campos));
    return c;
}

// tau = 2 ln(255 opacity) + 0.01 for n opacities (csrc/gsr_math.h gsr_tau)
void host_tau(int n, const float* opacity, float* out) {
    for (int i = 0; i < n; ++i) out[i] = gsr_tau(opacity[i]);
}

// out_f[P][12] = px,py,conA,conB,conC,opacity,r,g,b,depth,tau,0 ; out_i[P][8] = radius,minx,miny,maxx,maxy,tiles,clamped,visible
void host_preprocess(const HostCam* hc, int P, const float* means, const float* scales, const float* rots,
                     const float* cov_pre, const float* opac, const float* shs, const float* colors,
                     float* out_f, int* out_i, float* out_cov) {
    GsrCam cam = to_cam(hc);
    for (int i = 0; i < P; ++i) {
        float cov[6];
        if (cov_pre) memcpy(cov, cov_pre + 6 * i, sizeof(cov));
        else gsr_cov3d(scales + 3 * i, cam.scale_modifier, rots + 4 * i, cov);
        memcpy(out_cov + 6 * i, cov, sizeof(cov));
        GsrSplat sp;
        memset(&sp, 0, sizeof(sp));
        const bool vis = gsr_project(cam, means + 3 * i, cov, opac[i], sp);
        float rgb[3] = {0, 0, 0};
        uint32_t clamped = 0;
        if (vis) {
            if (colors) memcpy(rgb, colors + 3 * i, sizeof(rgb));
            else gsr_sh_to_rgb(cam.sh_degree, cam.M, shs + (size_t)i * cam.M * 3, means + 3 * i, cam.campos, rgb, clamped);
        }
        float* f = out_f + 12 * i;
        f[0] = sp.px; f[1] = sp.py; f[2] = sp.conA; f[3] = sp.conB; f[4] = sp.conC; f[5] = sp.opacity;
        f[6] = rgb[0]; f[7] = rgb[1]; f[8] = rgb[2]; f[9] = sp.depth; f[10] = vis ? sp.tau : 0; f[11] = 0;
        int* o = out_i + 8 * i;
        o[0] = sp.radius; o[1] = sp.minx; o[2] = sp.miny; o[3] = sp.maxx; o[4] = sp.maxy; o[5] = sp.tiles;
        o[6] = clamped; o[7] = vis ? 1 : 0;
    }
}

// grads_in[P][12] as the render backward produces them; outputs mirror gsr_rasterize_backward
void host_preprocess_backward(const HostCam* hc, int P, const float* means, const float* scales, const float* rots,
                              const float* cov_pre, const float* opac, const float* shs, const int* radii,
                              const unsigned* clamped, const float* grads_in, float* dmeans2D, float* dcolors,
                              float* dopacity, float* dmeans3D, float* dcov3D, float* dsh, float* dscales, float* drots) {
    GsrCam cam = to_cam(hc);
    for (int i = 0; i < P; ++i) {
        float dmean[3] = {0, 0, 0}, dcov[6] = {0, 0, 0, 0, 0, 0}, dscale[3] = {0, 0, 0}, drot[4] = {0, 0, 0, 0};
        float dop = 0, dm2x = 0, dm2y = 0, drgb[3] = {0, 0, 0};
        if (shs) memset(dsh + (size_t)i * cam.M * 3, 0, sizeof(float) * cam.M * 3);
        if (radii[i] > 0) {
            const float* gi = grads_in + 12 * i;
            GsrSplatGrad g;
            g.dpx = gi[0]; g.dpy = gi[1]; g.dconA = gi[2]; g.dconB = gi[3]; g.dconC = gi[4]; g.dopacity = gi[5];
            g.dr = gi[6]; g.dg = gi[7]; g.db = gi[8]; g.dinvdepth = gi[9];
            drgb[0] = g.dr; drgb[1] = g.dg; drgb[2] = g.db;
            // the product's arithmetic (csrc/preprocess.hip preprocess_bwd_kernel): covariance chain in GsrBwdReal
            GsrBwdReal cov[6], dcovr[6] = {0, 0, 0, 0, 0, 0};
            if (cov_pre) for (int k = 0; k < 6; ++k) cov[k] = cov_pre[6 * i + k];
            else gsr_cov3d_r<GsrBwdReal>(scales + 3 * i, cam.scale_modifier, rots + 4 * i, cov);
            gsr_project_backward_r<GsrBwdReal>(cam, means + 3 * i, cov, opac[i], g, dmean, dcovr, dop);
            dm2x = g.dpx * (0.5f * (float)cam.W);
            dm2y = g.dpy * (0.5f * (float)cam.H);
            if (!cov_pre) gsr_cov3d_backward_r<GsrBwdReal>(scales + 3 * i, cam.scale_modifier, rots + 4 * i, dcovr, dscale, drot);
            for (int k = 0; k < 6; ++k) dcov[k] = (float)dcovr[k];
            if (shs) {
                gsr_sh_backward(cam.sh_degree, cam.M, shs + (size_t)i * cam.M * 3, means + 3 * i, cam.campos, clamped[i], drgb,
                                dsh + (size_t)i * cam.M * 3, dmean);
            }
        }
        dmeans2D[3 * i] = dm2x; dmeans2D[3 * i + 1] = dm2y; dmeans2D[3 * i + 2] = 0;
        memcpy(dcolors + 3 * i, drgb, sizeof(drgb));
        dopacity[i] = dop;
        memcpy(dmeans3D + 3 * i, dmean, sizeof(dmean));
        memcpy(dcov3D + 6 * i, dcov, sizeof(dcov));
        if (dscales) { memcpy(dscales + 3 * i, dscale, sizeof(dscale)); memcpy(drots + 4 * i, drot, sizeof(drot)); }
    }
}

}  // extern "C"
