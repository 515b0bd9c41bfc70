// Per-Gaussian fp32 math of the rasterizer's preprocess stage (forward and backward), written once as
// host+device inline functions.  The HIP kernels in preprocess.hip call these per lane; tests/ compiles
// the same header with g++ (tests/host_math_harness.cpp) to check the arithmetic against the CPU oracle
// without a GPU.
//
// ARITHMETIC CONTRACT (matches oracle/torch_oracle.py): every expression below is evaluated in fp32,
// one IEEE-754 rounding per written operation, left to right, with NO fused multiply-add -- this file
// must be compiled with -ffp-contract=off and IEEE-correct divide/sqrt (hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt).  That is what makes radii / tiles_touched bit-exact.
//
// Algorithm: SURVEY.md Appendix A.2 (forward) and A.6 (backward); the in-tree restatements it must agree
// with are utils/sh_utils.py:57-112 (SH), utils/general_utils.py:78-110 (Sigma3D) and
// scene/cameras.py:80-89 (matrix layout) of the reference.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define GSR_HD __host__ __device__ __forceinline__
#else
#define GSR_HD inline
#endif

#define GSR_TILE 16
#define GSR_NEAR_Z 0.2f
#define GSR_LOWPASS 0.3f
#define GSR_AA_FLOOR 0.000025f
#define GSR_ALPHA_MIN (1.0f / 255.0f)
#define GSR_ALPHA_MAX 0.99f
#define GSR_T_EPS 0.0001f

// utils/sh_utils.py:26-54 rounded to fp32
#define GSR_SH_C0 0.28209479177387814f
#define GSR_SH_C1 0.4886025119029199f
#define GSR_SH_C2_0 1.0925484305920792f
#define GSR_SH_C2_1 -1.0925484305920792f
#define GSR_SH_C2_2 0.31539156525252005f
#define GSR_SH_C2_3 -1.0925484305920792f
#define GSR_SH_C2_4 0.5462742152960396f
#define GSR_SH_C3_0 -0.5900435899266435f
#define GSR_SH_C3_1 2.890611442640554f
#define GSR_SH_C3_2 -0.4570457994644658f
#define GSR_SH_C3_3 0.3731763325901154f
#define GSR_SH_C3_4 -0.4570457994644658f
#define GSR_SH_C3_5 1.445305721320277f
#define GSR_SH_C3_6 -0.5900435899266435f

struct GsrCam {
    int W, H, gx, gy;
    float focal_x, focal_y;   // W / (2 tanfovx), H / (2 tanfovy)
    float limx, limy;         // 1.3 * tanfov
    float scale_modifier;
    int sh_degree, M;
    int antialiasing;
    int snug;                 // 1 = snug tile rectangle (default), 0 = the reference's square (A/B: same outputs, longer lists)
    int tile_y0, tile_y1;     // band of tile rows that is binned
    float view[16];           // flat, as passed (transposed math matrix)
    float proj[16];
    float campos[3];
};

// ---------------------------------------------------------------------------------------------------------------------
// tau = 2 ln(255 opacity) + 0.01: a splat reaches alpha >= 1/255 only where its quadratic form q = d^T conic d is <= tau
// (alpha = opacity exp(-q/2); the 0.01 is slack for the blend's own rounding).  Two consumers: the blend kernels' box test
// (the value travels in the splat record) and the SNUG tile rectangle below.  The second one DECIDES INTEGERS (tiles_touched,
// the instance lists), so tau must come out bit-identical on the GPU, in the host build of this header and in the oracle's
// restatement (oracle/torch_oracle.py: det_log / tau_of_opacity): no libm -- `logf` differs by an ulp between implementations --
// but frexp (exact) + the atanh series in fp64, one IEEE rounding per written operation, contraction switched off locally so
// that translation units built with -ffp-contract=fast (route.hip) get the same bits.
// ---------------------------------------------------------------------------------------------------------------------
GSR_HD double gsr_log_det(double v) {            // ln(v), v finite and > 0
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    // v = m 2^e, m in [0.5, 1): frexp spelled out on the bits of a positive normal double (the library call takes a pointer, an
    // opaque access the scheduler will not move loads across)
    uint64_t bits;
    __builtin_memcpy(&bits, &v, 8);
    const int e0 = (int)((bits >> 52) & 0x7ffu) - 1022;
    const uint64_t mbits = (bits & 0x800FFFFFFFFFFFFFull) | 0x3FE0000000000000ull;
    double m0;
    __builtin_memcpy(&m0, &mbits, 8);
    const bool low = m0 < 0.70710678118654752;   // (selects, not branches: see gsr_project's snug block)
    const double m = low ? m0 * 2.0 : m0;
    const int e = low ? e0 - 1 : e0;
    const double s = (m - 1.0) / (m + 1.0);      // |s| <= 0.1716; ln m = 2 atanh s
    const double s2 = s * s;
    double p = 1.0 / 19.0;
    p = p * s2 + 1.0 / 17.0;
    p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;
    p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;
    p = p * s2 + 1.0;
    const double t1 = (double)e * 0.6931471805599453;
    const double t3 = (2.0 * s) * p;
    return t1 + t3;
}

GSR_HD float gsr_tau(float opacity) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float vf = 255.0f * opacity;
    const bool ok = vf > 0.0f && vf < 3.0e38f;
    const double t = 2.0 * gsr_log_det(ok ? (double)vf : 1.0) + 0.01;
    // as logf outside the domain: ln 0 = -inf (culled everywhere), ln(+inf) = +inf, ln(negative / NaN) = NaN
    const float bad = vf == 0.0f ? -INFINITY : (vf >= 3.0e38f ? INFINITY : NAN);
    return ok ? (float)t : bad;
}

struct GsrSplat {             // result of the forward preprocess for one Gaussian
    float px, py;             // pixel-space centre
    float conA, conB, conC;   // inverse 2-D covariance
    float opacity;            // opacity * aa
    float tau;                // gsr_tau(opacity)
    float r, g, b;
    float depth;              // view-space z
    int radius;               // 0 = not visible
    uint32_t minx, miny, maxx, maxy;  // tile rectangle, y already clamped to the band
    uint32_t tiles;           // (maxx-minx)*(maxy-miny) within the band
    uint32_t clamped;         // bit c set: colour channel c was clamped at 0
};

// R = the arithmetic type: float in the forward (the oracle's fp32 expression tree, bit for bit), double in the backward's
// recomputation (see gsr_project_backward).
template <class R>
GSR_HD void gsr_cov3d_r(const float* s, float mod, const float* q, R* cov) {
    const R s0 = (R)mod * (R)s[0], s1 = (R)mod * (R)s[1], s2 = (R)mod * (R)s[2];
    const R r = q[0], x = q[1], y = q[2], z = q[3];
    const R one = 1, two = 2;
    const R R00 = one - two * (y * y + z * z), R01 = two * (x * y - r * z), R02 = two * (x * z + r * y);
    const R R10 = two * (x * y + r * z), R11 = one - two * (x * x + z * z), R12 = two * (y * z - r * x);
    const R R20 = two * (x * z - r * y), R21 = two * (y * z + r * x), R22 = one - two * (x * x + y * y);
    const R M00 = R00 * s0, M01 = R01 * s1, M02 = R02 * s2;
    const R M10 = R10 * s0, M11 = R11 * s1, M12 = R12 * s2;
    const R M20 = R20 * s0, M21 = R21 * s1, M22 = R22 * s2;
    cov[0] = M00 * M00 + M01 * M01 + M02 * M02;
    cov[1] = M00 * M10 + M01 * M11 + M02 * M12;
    cov[2] = M00 * M20 + M01 * M21 + M02 * M22;
    cov[3] = M10 * M10 + M11 * M11 + M12 * M12;
    cov[4] = M10 * M20 + M11 * M21 + M12 * M22;
    cov[5] = M20 * M20 + M21 * M21 + M22 * M22;
}
GSR_HD void gsr_cov3d(const float* s, float mod, const float* q, float* cov) { gsr_cov3d_r<float>(s, mod, q, cov); }

// Everything that depends on the Gaussian's position / covariance.  Returns false when the Gaussian is
// culled (near plane, singular Sigma2D, empty tile rectangle); out.radius/out.tiles are 0 then.
// `cov` is Sigma3D packed [xx,xy,xz,yy,yz,zz].
GSR_HD bool gsr_project(const GsrCam& cam, const float* mean, const float* cov, float opacity_in, GsrSplat& out) {
    const float* vm = cam.view;
    const float* pm = cam.proj;
    const float x = mean[0], y = mean[1], z = mean[2];
    out.radius = 0;
    out.tiles = 0;
    out.minx = out.miny = out.maxx = out.maxy = 0;
    const float pvx = vm[0] * x + vm[4] * y + vm[8] * z + vm[12];
    const float pvy = vm[1] * x + vm[5] * y + vm[9] * z + vm[13];
    const float pvz = vm[2] * x + vm[6] * y + vm[10] * z + vm[14];
    out.depth = pvz;
    if (!(pvz > GSR_NEAR_Z)) return false;
    const float hx = pm[0] * x + pm[4] * y + pm[8] * z + pm[12];
    const float hy = pm[1] * x + pm[5] * y + pm[9] * z + pm[13];
    const float hw = pm[3] * x + pm[7] * y + pm[11] * z + pm[15];
    const float pw = 1.0f / (hw + 1e-7f);
    const float projx = hx * pw, projy = hy * pw;

    const float txtz = pvx / pvz, tytz = pvy / pvz;
    const float tx = fminf(cam.limx, fmaxf(-cam.limx, txtz)) * pvz;
    const float ty = fminf(cam.limy, fmaxf(-cam.limy, tytz)) * pvz;
    const float tz = pvz;
    const float tz2 = tz * tz;
    const float J00 = cam.focal_x / tz;
    const float J02 = -(cam.focal_x * tx) / tz2;
    const float J11 = cam.focal_y / tz;
    const float J12 = -(cam.focal_y * ty) / tz2;
    const float T00 = J00 * vm[0] + J02 * vm[2];
    const float T01 = J00 * vm[4] + J02 * vm[6];
    const float T02 = J00 * vm[8] + J02 * vm[10];
    const float T10 = J11 * vm[1] + J12 * vm[2];
    const float T11 = J11 * vm[5] + J12 * vm[6];
    const float T12 = J11 * vm[9] + J12 * vm[10];
    const float S00 = cov[0], S01 = cov[1], S02 = cov[2], S11 = cov[3], S12 = cov[4], S22 = cov[5];
    const float u0 = S00 * T00 + S01 * T01 + S02 * T02;
    const float u1 = S01 * T00 + S11 * T01 + S12 * T02;
    const float u2 = S02 * T00 + S12 * T01 + S22 * T02;
    const float v0 = S00 * T10 + S01 * T11 + S02 * T12;
    const float v1 = S01 * T10 + S11 * T11 + S12 * T12;
    const float v2 = S02 * T10 + S12 * T11 + S22 * T12;
    const float a0 = T00 * u0 + T01 * u1 + T02 * u2;
    const float b = T10 * u0 + T11 * u1 + T12 * u2;
    const float c0 = T10 * v0 + T11 * v1 + T12 * v2;

    const float det0 = a0 * c0 - b * b;
    const float a = a0 + GSR_LOWPASS;
    const float c = c0 + GSR_LOWPASS;
    const float det = a * c - b * b;
    float aa = 1.0f;
    if (cam.antialiasing) aa = sqrtf(fmaxf(det0 / det, GSR_AA_FLOOR));
    if (det == 0.0f) return false;
    const float det_inv = 1.0f / det;
    out.conA = c * det_inv;
    out.conB = -b * det_inv;
    out.conC = a * det_inv;

    const float mid = 0.5f * (a + c);
    const float disc = sqrtf(fmaxf(mid * mid - det, 0.1f));
    const float lam = fmaxf(mid + disc, mid - disc);
    const float radius_f = ceilf(3.0f * sqrtf(lam));

    const float pixx = ((projx + 1.0f) * (float)cam.W - 1.0f) * 0.5f;
    const float pixy = ((projy + 1.0f) * (float)cam.H - 1.0f) * 0.5f;
    out.px = pixx;
    out.py = pixy;

    // tile rectangle: C truncation of the float quotient, clamped to the grid in float (NaN -> 0)
    const float gxf = (float)cam.gx, gyf = (float)cam.gy;
    const float lox = fminf(fmaxf(truncf((pixx - radius_f) / 16.0f), 0.0f), gxf);
    const float hix = fminf(fmaxf(truncf((pixx + radius_f + 15.0f) / 16.0f), 0.0f), gxf);
    const float loy = fminf(fmaxf(truncf((pixy - radius_f) / 16.0f), 0.0f), gyf);
    const float hiy = fminf(fmaxf(truncf((pixy + radius_f + 15.0f) / 16.0f), 0.0f), gyf);
    const int minx = (int)lox, maxx = (int)hix, miny = (int)loy, maxy = (int)hiy;
    if ((maxx - minx) * (maxy - miny) <= 0) return false;
    if (!(radius_f < 2.0e9f)) return false;   // inf / NaN radius: treated as culled (oracle: radii = 0)
    out.radius = (int)radius_f;
    out.opacity = opacity_in * aa;
    out.tau = gsr_tau(out.opacity);
    // SNUG tile rectangle (NOT in the reference, which bins the square of radius 3 sqrt(lambda_max) whatever the shape and the
    // opacity): only tiles that the ellipse q <= tau can reach hold a pixel with alpha >= 1/255; every other (tile, Gaussian)
    // instance is skipped pixel by pixel in the reference's blend (forward.cu renderCUDA: `if (alpha < 1/255) continue`) and
    // contributes nothing, so dropping it leaves every bit of the forward outputs unchanged (the gradients: to fp32 summation
    // order) and removes 30 % (uniform scene) to 49 % (clustered) of the instances from emission, tile sort and blend.  Half extents of the ellipse of the ROUNDED conic the
    // blend evaluates: ex^2 = tau C / (A C - B^2), ey^2 = tau A / (A C - B^2) -- in fp64 from the fp32 values (the products are
    // exact, so cancellation in A C - B^2 costs nothing), inflated by 1 % + half a pixel against the blend's fp32 rounding of q.
    // `radius` (the operator's `radii` output) stays the reference's.
    int sminx = minx, smaxx = maxx, sminy = miny, smaxy = maxy;
    {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
        // (written with selects, not branches: control flow in here made hipcc serialise the split-SH loader of the forward
        //  kernel into a 14-long load -> wait chain, tools/isa_audit.py)
        const double td = (double)out.tau;
        const double Ad = (double)out.conA, Bd = (double)out.conB, Cd = (double)out.conC;
        const double detc = Ad * Cd - Bd * Bd;
        const bool on = cam.snug != 0;
        const bool dead = on && td <= 0.0;        // opacity <= 1/255 (incl. tau = -inf): alpha < 1/255 everywhere
        bool shrink = on && !dead && detc > 0.0 && td < 1.0e30;      // (NaN / +inf anywhere: keep the reference rectangle)
        const double sdet = shrink ? detc : 1.0, st = shrink ? td : 1.0;
        const double ex = sqrt(st * Cd / sdet) * 1.01 + 0.5;
        const double ey = sqrt(st * Ad / sdet) * 1.01 + 0.5;
        shrink = shrink && ex < 1.0e9 && ey < 1.0e9;
        const double cx = (double)pixx, cy = (double)pixy;
        const double big = 1.0e9;
        const double lx = fmin(fmax(floor((cx - ex) / 16.0), -big), big), hx = fmin(fmax(floor((cx + ex) / 16.0) + 1.0, -big), big);
        const double ly = fmin(fmax(floor((cy - ey) / 16.0), -big), big), hy = fmin(fmax(floor((cy + ey) / 16.0) + 1.0, -big), big);
        sminx = (shrink && lx > (double)minx) ? (lx < (double)maxx ? (int)lx : maxx) : minx;
        smaxx = (shrink && hx < (double)maxx) ? (hx > (double)sminx ? (int)hx : sminx) : maxx;
        sminy = (shrink && ly > (double)miny) ? (ly < (double)maxy ? (int)ly : maxy) : miny;
        smaxy = (shrink && hy < (double)maxy) ? (hy > (double)sminy ? (int)hy : sminy) : maxy;
        smaxx = dead ? sminx : smaxx;
        smaxy = dead ? sminy : smaxy;
    }
    const int bminy = sminy < cam.tile_y0 ? cam.tile_y0 : (sminy > cam.tile_y1 ? cam.tile_y1 : sminy);
    const int bmaxy = smaxy < cam.tile_y0 ? cam.tile_y0 : (smaxy > cam.tile_y1 ? cam.tile_y1 : smaxy);
    out.minx = (uint32_t)sminx;
    out.maxx = (uint32_t)smaxx;
    out.miny = (uint32_t)bminy;
    out.maxy = (uint32_t)bmaxy;
    out.tiles = (uint32_t)((smaxx - sminx) * (bmaxy - bminy));
    return true;
}

// 12-float group load / store used by the SH routines; nf = number of valid floats (12 except for a ragged last group)
GSR_HD void gsr_ld12(const float* p, int nf, float* dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (nf == 12 && ((uintptr_t)p & 15) == 0) {
        const float4 v0 = reinterpret_cast<const float4*>(p)[0], v1 = reinterpret_cast<const float4*>(p)[1],
                     v2 = reinterpret_cast<const float4*>(p)[2];
        dst[0] = v0.x; dst[1] = v0.y; dst[2] = v0.z; dst[3] = v0.w; dst[4] = v1.x; dst[5] = v1.y; dst[6] = v1.z; dst[7] = v1.w;
        dst[8] = v2.x; dst[9] = v2.y; dst[10] = v2.z; dst[11] = v2.w;
        return;
    }
#endif
    for (int i = 0; i < 12; ++i) dst[i] = i < nf ? p[i] : 0.0f;
}
GSR_HD void gsr_st12(float* p, int nf, const float* src) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (nf == 12 && ((uintptr_t)p & 15) == 0) {
        reinterpret_cast<float4*>(p)[0] = make_float4(src[0], src[1], src[2], src[3]);
        reinterpret_cast<float4*>(p)[1] = make_float4(src[4], src[5], src[6], src[7]);
        reinterpret_cast<float4*>(p)[2] = make_float4(src[8], src[9], src[10], src[11]);
        return;
    }
#endif
    for (int i = 0; i < 12; ++i)
        if (i < nf) p[i] = src[i];
}

// SH -> RGB in the expression order of utils/sh_utils.py:78-104, then +0.5, clamp >= 0.
// sh points at this Gaussian's [M][3] block (global or host memory); it is consumed in groups of 4 coefficients
// (12 floats = three 16-byte loads) so that only one group is live in registers.  Per channel the sum is built
// left to right exactly as sh_utils writes it: result (+/-) (C * factor ...) * sh[k].
// Where a Gaussian's SH record lives.  GsrShRow: M consecutive RGB triples (the fused [P, M, 3] tensor, or an LDS row).
// GsrShRowSplit: coefficient 0 in dc[3], coefficients 1..15 in rest[45] -- the "separate_sh" call form, whose two blocks are
// staged in LDS exactly as they lie in memory (rest rows 45 floats apart: odd stride, bank-conflict-free scalar reads).
// ld / st move the 12 floats of coefficient group k0 .. k0+3.
struct GsrShRow {
    const float* p;
    GSR_HD void ld(int k0, int nf, float* dst) const { gsr_ld12(p + k0 * 3, nf, dst); }
};
struct GsrShRowOut {
    float* p;
    GSR_HD void st(int k0, int nf, const float* src) const { gsr_st12(p + k0 * 3, nf, src); }
};
#if defined(__HIPCC__)
// Row known to be 16-byte aligned with all 16 coefficients present (the LDS tile rows of the per-Gaussian kernels): three
// 16-byte accesses per group, unconditionally.  (gsr_ld12's run-time alignment test made the compiler fall back to 4-byte
// LDS accesses, which at the tile's 52-float row stride are 4-way bank-conflicted: 52 l mod 32 takes 8 values.)
struct GsrShRowAligned {
    const float* p;
    __device__ __forceinline__ void ld(int k0, int, float* dst) const {
        const float4* q = reinterpret_cast<const float4*>(p + k0 * 3);
        const float4 v0 = q[0], v1 = q[1], v2 = q[2];
        dst[0] = v0.x; dst[1] = v0.y; dst[2] = v0.z; dst[3] = v0.w; dst[4] = v1.x; dst[5] = v1.y; dst[6] = v1.z; dst[7] = v1.w;
        dst[8] = v2.x; dst[9] = v2.y; dst[10] = v2.z; dst[11] = v2.w;
    }
};
struct GsrShRowAlignedOut {
    float* p;
    __device__ __forceinline__ void st(int k0, int, const float* src) const {
        float4* q = reinterpret_cast<float4*>(p + k0 * 3);
        q[0] = make_float4(src[0], src[1], src[2], src[3]);
        q[1] = make_float4(src[4], src[5], src[6], src[7]);
        q[2] = make_float4(src[8], src[9], src[10], src[11]);
    }
};
#endif
struct GsrShRowSplit {
    const float* dc;
    const float* rest;
    GSR_HD void ld(int k0, int, float* dst) const {
        if (k0 == 0) {
            dst[0] = dc[0]; dst[1] = dc[1]; dst[2] = dc[2];
            for (int i = 0; i < 9; ++i) dst[3 + i] = rest[i];
        } else {
            for (int i = 0; i < 12; ++i) dst[i] = rest[k0 * 3 - 3 + i];
        }
    }
};
struct GsrShRowSplitOut {
    float* dc;
    float* rest;
    GSR_HD void st(int k0, int, const float* src) const {
        if (k0 == 0) {
            dc[0] = src[0]; dc[1] = src[1]; dc[2] = src[2];
            for (int i = 0; i < 9; ++i) rest[i] = src[3 + i];
        } else {
            for (int i = 0; i < 12; ++i) rest[k0 * 3 - 3 + i] = src[i];
        }
    }
};

template <class Row>
GSR_HD void gsr_sh_to_rgb_row(int deg, int M, const Row& sh, const float* mean, const float* campos, float* rgb, uint32_t& clamped) {
    const float dx = mean[0] - campos[0], dy = mean[1] - campos[1], dz = mean[2] - campos[2];
    const float n = sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx / n, y = dy / n, z = dz / n;
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, yz = y * z, xz = x * z;
    float res[3] = {0.0f, 0.0f, 0.0f};
    float s[12];
    const int ncoef = (deg + 1) * (deg + 1);
    for (int grp = 0; grp < 4; ++grp) {
        const int k0 = grp * 4;
        if (k0 >= ncoef) break;
        sh.ld(k0, (M - k0 >= 4 ? 4 : M - k0) * 3, s);
        for (int ch = 0; ch < 3; ++ch) {
            float r = res[ch];
            if (grp == 0) {
                r = GSR_SH_C0 * s[0 + ch];
                if (deg > 0) r = r - GSR_SH_C1 * y * s[3 + ch] + GSR_SH_C1 * z * s[6 + ch] - GSR_SH_C1 * x * s[9 + ch];
            } else if (grp == 1) {
                r = r + GSR_SH_C2_0 * xy * s[0 + ch] + GSR_SH_C2_1 * yz * s[3 + ch] +
                    GSR_SH_C2_2 * (2.0f * zz - xx - yy) * s[6 + ch] + GSR_SH_C2_3 * xz * s[9 + ch];
            } else if (grp == 2) {
                r = r + GSR_SH_C2_4 * (xx - yy) * s[0 + ch];
                if (deg > 2)
                    r = r + GSR_SH_C3_0 * y * (3.0f * xx - yy) * s[3 + ch] + GSR_SH_C3_1 * xy * z * s[6 + ch] +
                        GSR_SH_C3_2 * y * (4.0f * zz - xx - yy) * s[9 + ch];
            } else {
                r = r + GSR_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * s[0 + ch] +
                    GSR_SH_C3_4 * x * (4.0f * zz - xx - yy) * s[3 + ch] + GSR_SH_C3_5 * z * (xx - yy) * s[6 + ch] +
                    GSR_SH_C3_6 * x * (xx - 3.0f * yy) * s[9 + ch];
            }
            res[ch] = r;
        }
    }
    clamped = 0;
    for (int ch = 0; ch < 3; ++ch) {
        const float result = res[ch] + 0.5f;
        if (result < 0.0f) clamped |= (1u << ch);
        rgb[ch] = fmaxf(result, 0.0f);
    }
}
GSR_HD void gsr_sh_to_rgb(int deg, int M, const float* sh, const float* mean, const float* campos, float* rgb, uint32_t& clamped) {
    gsr_sh_to_rgb_row(deg, M, GsrShRow{sh}, mean, campos, rgb, clamped);
}

// ------------------------------------------------------------------------------------------------
// Backward.  Conventions: gradients w.r.t. the *plain* conic entries
// (dA, dB, dC with B the scalar in power = -0.5(A dx^2 + C dy^2) - B dx dy); dpix in pixel units.
// ------------------------------------------------------------------------------------------------
struct GsrSplatGrad {          // what the render backward accumulates per Gaussian
    float dpx, dpy;            // dL/d(pixel-space centre)
    float dconA, dconB, dconC; // dL/d(conic)
    float dopacity;            // dL/d(opacity*aa)
    float dr, dg, db;          // dL/d(rgb after clamp)
    float dinvdepth;           // dL/d(1/depth) from the inverse-depth image
};

// Inputs: the forward inputs of this Gaussian; Outputs: accumulated into dmean[3], dcov[6], dopacity_in.
//
// R = the arithmetic type of the covariance chain (view-space position -> J -> T -> Sigma2D -> conic and back).  The product
// instantiates R = double (round 6): the chain is a sequence of cancelling sums on an anisotropic splat -- det = a c - b^2 loses
// log2(condition number) bits, the 2x2 inverse derivative subtracts terms ~ (a, b, c)^2 / det^2 that are `condition number` times
// their sum, and Sigma3D's gradient is contracted with the needle's long axis, which cancels again -- so that in fp32 a needle of
// condition number ~ 1e3 came back with a rotation gradient 2.4e-3 of max |grad| from the fp64 oracle (fuzz seed 71, frame 109,
// tests/golden/hard_frames.npz) and a scale gradient 8e-4.  In fp64 from the SAME fp32 inputs and the same fp32 per-pair sums of
// the blend backward the frame is at 6e-5.  The kernel streams 500+ bytes per Gaussian: the extra ALU time hides under the loads.
// The two frustum-clamp decisions are taken on the fp32 quotients, exactly as the forward took them.
template <class R>
GSR_HD void gsr_project_backward_r(const GsrCam& cam, const float* mean, const R* cov, float opacity_in,
                                   const GsrSplatGrad& g, float* dmean, R* dcov, float& dopacity_in) {
    const float* vm = cam.view;
    const float* pm = cam.proj;
    const float x = mean[0], y = mean[1], z = mean[2];
    // the forward's own clamp decisions (fp32, same expression tree as gsr_project)
    const float pvx32 = vm[0] * x + vm[4] * y + vm[8] * z + vm[12];
    const float pvy32 = vm[1] * x + vm[5] * y + vm[9] * z + vm[13];
    const float pvz32 = vm[2] * x + vm[6] * y + vm[10] * z + vm[14];
    const float txtz32 = pvx32 / pvz32, tytz32 = pvy32 / pvz32;
    const bool inside_x = !(txtz32 < -cam.limx || txtz32 > cam.limx);
    const bool inside_y = !(tytz32 < -cam.limy || tytz32 > cam.limy);
    const R zero = 0, one = 1, two = 2;
    const R inx = inside_x ? one : zero, iny = inside_y ? one : zero;
    const R pvx = (R)vm[0] * x + (R)vm[4] * y + (R)vm[8] * z + (R)vm[12];
    const R pvy = (R)vm[1] * x + (R)vm[5] * y + (R)vm[9] * z + (R)vm[13];
    const R pvz = (R)vm[2] * x + (R)vm[6] * y + (R)vm[10] * z + (R)vm[14];
    const R limx = cam.limx, limy = cam.limy;
    const R cx = inside_x ? pvx / pvz : (txtz32 < 0.0f ? -limx : limx);
    const R cy = inside_y ? pvy / pvz : (tytz32 < 0.0f ? -limy : limy);
    const R tx = cx * pvz;
    const R ty = cy * pvz;
    const R tz = pvz;
    const R tz2 = tz * tz;
    const R itz = one / tz, itz2 = one / tz2;
    const R fx = cam.focal_x, fy = cam.focal_y;
    const R J00 = fx * itz, J02 = -(fx * tx) * itz2, J11 = fy * itz, J12 = -(fy * ty) * itz2;
    const R W00 = vm[0], W01 = vm[4], W02 = vm[8];
    const R W10 = vm[1], W11 = vm[5], W12 = vm[9];
    const R W20 = vm[2], W21 = vm[6], W22 = vm[10];
    const R T00 = J00 * W00 + J02 * W20, T01 = J00 * W01 + J02 * W21, T02 = J00 * W02 + J02 * W22;
    const R T10 = J11 * W10 + J12 * W20, T11 = J11 * W11 + J12 * W21, T12 = J11 * W12 + J12 * W22;
    const R S00 = cov[0], S01 = cov[1], S02 = cov[2], S11 = cov[3], S12 = cov[4], S22 = cov[5];
    const R u0 = S00 * T00 + S01 * T01 + S02 * T02;
    const R u1 = S01 * T00 + S11 * T01 + S12 * T02;
    const R u2 = S02 * T00 + S12 * T01 + S22 * T02;
    const R v0 = S00 * T10 + S01 * T11 + S02 * T12;
    const R v1 = S01 * T10 + S11 * T11 + S12 * T12;
    const R v2 = S02 * T10 + S12 * T11 + S22 * T12;
    const R a0 = T00 * u0 + T01 * u1 + T02 * u2;
    const R b = T10 * u0 + T11 * u1 + T12 * u2;
    const R c0 = T10 * v0 + T11 * v1 + T12 * v2;
    const R det0 = a0 * c0 - b * b;
    const R a = a0 + (R)GSR_LOWPASS, c = c0 + (R)GSR_LOWPASS;
    const R det = a * c - b * b;

    // --- opacity * aa ---
    R da = 0, db = 0, dc = 0;   // dL/d(a,b,c) of the low-passed Sigma2D (b = off-diagonal entry)
    R aa = 1;
    if (cam.antialiasing) {
        const R ratio = det0 / det;
        const R floor_ = (R)GSR_AA_FLOOR;
        aa = sqrt(ratio > floor_ ? ratio : floor_);
        if (ratio > floor_) {
            // opacity = o * sqrt(det0/det): d/d(ratio) = o * 0.5 / aa
            const R dratio = (R)g.dopacity * (R)opacity_in * (R)0.5 / aa;
            const R ddet0 = dratio / det;
            const R ddet = -dratio * det0 / (det * det);
            // det0 = a0 c0 - b^2 ; det = a c - b^2 ; a = a0 + h, c = c0 + h
            da += ddet0 * c0 + ddet * c;
            dc += ddet0 * a0 + ddet * a;
            db += -two * b * (ddet0 + ddet);
        }
    }
    dopacity_in = (float)((R)g.dopacity * aa);

    // --- conic = inverse(Sigma2D), reference uses 1/(det^2 + 1e-7) ---
    {
        const R d2 = one / (det * det + (R)1e-7f);
        const R gA = g.dconA, gB = g.dconB, gC = g.dconC;
        da += d2 * (-c * c * gA + b * c * gB + (det - a * c) * gC);
        dc += d2 * (-a * a * gC + a * b * gB + (det - a * c) * gA);
        db += d2 * (two * b * c * gA - (det + two * b * b) * gB + two * a * b * gC);
    }

    // --- Sigma2D = T Sigma T^T  (a = T0.S.T0, b = T1.S.T0, c = T1.S.T1) ---
    // dL/dSigma (symmetric, independent entries S00,S01,S02,S11,S12,S22; off-diagonals appear twice)
    dcov[0] += T00 * T00 * da + T00 * T10 * db + T10 * T10 * dc;
    dcov[3] += T01 * T01 * da + T01 * T11 * db + T11 * T11 * dc;
    dcov[5] += T02 * T02 * da + T02 * T12 * db + T12 * T12 * dc;
    dcov[1] += two * T00 * T01 * da + (T00 * T11 + T01 * T10) * db + two * T10 * T11 * dc;
    dcov[2] += two * T00 * T02 * da + (T00 * T12 + T02 * T10) * db + two * T10 * T12 * dc;
    dcov[4] += two * T01 * T02 * da + (T01 * T12 + T02 * T11) * db + two * T11 * T12 * dc;
    // dL/dT: a = sum T0j u_j (u = S T0): da/dT0j = 2 u_j ; b = sum T1j u_j: db/dT1j = u_j, db/dT0j = v_j ; dc/dT1j = 2 v_j
    const R dT00 = two * u0 * da + v0 * db, dT01 = two * u1 * da + v1 * db, dT02 = two * u2 * da + v2 * db;
    const R dT10 = two * v0 * dc + u0 * db, dT11 = two * v1 * dc + u1 * db, dT12 = two * v2 * dc + u2 * db;
    // T0j = J00 W0j + J02 W2j ; T1j = J11 W1j + J12 W2j
    const R dJ00 = W00 * dT00 + W01 * dT01 + W02 * dT02;
    const R dJ02 = W20 * dT00 + W21 * dT01 + W22 * dT02;
    const R dJ11 = W10 * dT10 + W11 * dT11 + W12 * dT12;
    const R dJ12 = W20 * dT10 + W21 * dT11 + W22 * dT12;
    // J00 = fx/tz, J02 = -fx tx/tz^2, J11 = fy/tz, J12 = -fy ty/tz^2 ; tx, ty constant w.r.t. tz when clamped
    const R itz3 = itz2 * itz;
    const R dtx = inx * (-fx * itz2 * dJ02);
    const R dty = iny * (-fy * itz2 * dJ12);
    R dtz = -fx * itz2 * dJ00 - fy * itz2 * dJ11 + two * fx * tx * itz3 * dJ02 + two * fy * ty * itz3 * dJ12;
    // inverse depth image: D += (1/depth) w  => dL/d(depth) = -dinvdepth / tz^2
    dtz += -(R)g.dinvdepth * itz2;

    // --- pixel centre through the perspective divide of the full projection (fp32: no cancellation here) ---
    const float hx = pm[0] * x + pm[4] * y + pm[8] * z + pm[12];
    const float hy = pm[1] * x + pm[5] * y + pm[9] * z + pm[13];
    const float hw = pm[3] * x + pm[7] * y + pm[11] * z + pm[15];
    const float pw = 1.0f / (hw + 1e-7f);
    const float dprojx = g.dpx * 0.5f * (float)cam.W;   // pix = ((proj+1) W - 1)/2
    const float dprojy = g.dpy * 0.5f * (float)cam.H;
    const float dhx = dprojx * pw, dhy = dprojy * pw;
    const float dhw = -(dprojx * hx + dprojy * hy) * pw * pw;
    dmean[0] += pm[0] * dhx + pm[1] * dhy + pm[3] * dhw;
    dmean[1] += pm[4] * dhx + pm[5] * dhy + pm[7] * dhw;
    dmean[2] += pm[8] * dhx + pm[9] * dhy + pm[11] * dhw;
    // --- view-space position t = W p + t0 ---
    dmean[0] += (float)(W00 * dtx + W10 * dty + W20 * dtz);
    dmean[1] += (float)(W01 * dtx + W11 * dty + W21 * dtz);
    dmean[2] += (float)(W02 * dtx + W12 * dty + W22 * dtz);
}
// the all-fp32 form (rounds 1-5; kept for the tests that hold it against the fp32 oracle's autograd operation by operation)
GSR_HD void gsr_project_backward(const GsrCam& cam, const float* mean, const float* cov, float opacity_in,
                                 const GsrSplatGrad& g, float* dmean, float* dcov, float& dopacity_in) {
    gsr_project_backward_r<float>(cam, mean, cov, opacity_in, g, dmean, dcov, dopacity_in);
}

// SH backward, streaming in groups of 4 coefficients (12 floats = three 16-byte accesses): drgb = dL/d(rgb after clamp).
// `sh` / `dsh` point at this Gaussian's [M][3] blocks (global or host memory).  A group is loaded, used and stored
// before the next one is touched, so no 48-float per-thread arrays are needed (the first, array-based version spilled
// 208 B/lane to scratch in the fused backward kernel; storing coefficient by coefficient -- 12-byte stores -- avoided
// the spill but measured 1.6x write amplification at the L2/HBM boundary).  All M rows of dsh are written (rows above
// the active degree are zero); the view-direction term is accumulated into dmean.
struct GsrShBwdAcc {
    float drgb[3];
    float ddx, ddy, ddz;
};
GSR_HD void gsr_sh_bwd_term(GsrShBwdAcc& a, const float* s3, float* d3, float basis, float bx, float by, float bz) {
    d3[0] = basis * a.drgb[0];
    d3[1] = basis * a.drgb[1];
    d3[2] = basis * a.drgb[2];
    const float dot = s3[0] * a.drgb[0] + s3[1] * a.drgb[1] + s3[2] * a.drgb[2];
    a.ddx += bx * dot;
    a.ddy += by * dot;
    a.ddz += bz * dot;
}
template <class Row, class RowOut>
GSR_HD void gsr_sh_backward_row(int deg, int M, const Row& sh, const float* mean, const float* campos,
                                uint32_t clamped, const float* drgb_in, const RowOut& dsh, float* dmean) {
    const float ox = mean[0] - campos[0], oy = mean[1] - campos[1], oz = mean[2] - campos[2];
    const float n = sqrtf(ox * ox + oy * oy + oz * oz);
    const float x = ox / n, y = oy / n, z = oz / n;
    GsrShBwdAcc a;
    for (int ch = 0; ch < 3; ++ch) a.drgb[ch] = (clamped & (1u << ch)) ? 0.0f : drgb_in[ch];
    a.ddx = a.ddy = a.ddz = 0.0f;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    float s[12], d[12];
    for (int grp = 0; grp < 4; ++grp) {
        const int k0 = grp * 4;
        if (k0 >= M) break;
        const int nf = (M - k0 >= 4 ? 4 : M - k0) * 3;
        for (int i = 0; i < 12; ++i) d[i] = 0.0f;
        const bool live = (grp == 0) || (grp == 1 && deg > 1) || (grp >= 2 && deg > (grp == 2 ? 1 : 2));
        if (live) sh.ld(k0, nf, s);
        if (grp == 0) {
            gsr_sh_bwd_term(a, s + 0, d + 0, GSR_SH_C0, 0.0f, 0.0f, 0.0f);
            if (deg > 0) {
                gsr_sh_bwd_term(a, s + 3, d + 3, -GSR_SH_C1 * y, 0.0f, -GSR_SH_C1, 0.0f);
                gsr_sh_bwd_term(a, s + 6, d + 6, GSR_SH_C1 * z, 0.0f, 0.0f, GSR_SH_C1);
                gsr_sh_bwd_term(a, s + 9, d + 9, -GSR_SH_C1 * x, -GSR_SH_C1, 0.0f, 0.0f);
            }
        } else if (grp == 1 && deg > 1) {
            gsr_sh_bwd_term(a, s + 0, d + 0, GSR_SH_C2_0 * xy, GSR_SH_C2_0 * y, GSR_SH_C2_0 * x, 0.0f);
            gsr_sh_bwd_term(a, s + 3, d + 3, GSR_SH_C2_1 * yz, 0.0f, GSR_SH_C2_1 * z, GSR_SH_C2_1 * y);
            gsr_sh_bwd_term(a, s + 6, d + 6, GSR_SH_C2_2 * (2.0f * zz - xx - yy), GSR_SH_C2_2 * -2.0f * x, GSR_SH_C2_2 * -2.0f * y,
                            GSR_SH_C2_2 * 4.0f * z);
            gsr_sh_bwd_term(a, s + 9, d + 9, GSR_SH_C2_3 * xz, GSR_SH_C2_3 * z, 0.0f, GSR_SH_C2_3 * x);
        } else if (grp == 2 && deg > 1) {
            gsr_sh_bwd_term(a, s + 0, d + 0, GSR_SH_C2_4 * (xx - yy), GSR_SH_C2_4 * 2.0f * x, GSR_SH_C2_4 * -2.0f * y, 0.0f);
            if (deg > 2) {
                gsr_sh_bwd_term(a, s + 3, d + 3, GSR_SH_C3_0 * y * (3.0f * xx - yy), GSR_SH_C3_0 * 6.0f * xy,
                                GSR_SH_C3_0 * (3.0f * xx - 3.0f * yy), 0.0f);
                gsr_sh_bwd_term(a, s + 6, d + 6, GSR_SH_C3_1 * xy * z, GSR_SH_C3_1 * yz, GSR_SH_C3_1 * xz, GSR_SH_C3_1 * xy);
                gsr_sh_bwd_term(a, s + 9, d + 9, GSR_SH_C3_2 * y * (4.0f * zz - xx - yy), GSR_SH_C3_2 * -2.0f * xy,
                                GSR_SH_C3_2 * (4.0f * zz - xx - 3.0f * yy), GSR_SH_C3_2 * 8.0f * yz);
            }
        } else if (grp == 3 && deg > 2) {
            gsr_sh_bwd_term(a, s + 0, d + 0, GSR_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), GSR_SH_C3_3 * -6.0f * xz,
                            GSR_SH_C3_3 * -6.0f * yz, GSR_SH_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy));
            gsr_sh_bwd_term(a, s + 3, d + 3, GSR_SH_C3_4 * x * (4.0f * zz - xx - yy), GSR_SH_C3_4 * (4.0f * zz - 3.0f * xx - yy),
                            GSR_SH_C3_4 * -2.0f * xy, GSR_SH_C3_4 * 8.0f * xz);
            gsr_sh_bwd_term(a, s + 6, d + 6, GSR_SH_C3_5 * z * (xx - yy), GSR_SH_C3_5 * 2.0f * xz, GSR_SH_C3_5 * -2.0f * yz,
                            GSR_SH_C3_5 * (xx - yy));
            gsr_sh_bwd_term(a, s + 9, d + 9, GSR_SH_C3_6 * x * (xx - 3.0f * yy), GSR_SH_C3_6 * (3.0f * xx - 3.0f * yy),
                            GSR_SH_C3_6 * -6.0f * xy, 0.0f);
        }
        dsh.st(k0, nf, d);
    }
    // d = o / |o|  =>  dL/do = (dL/dd - d (d . dL/dd)) / |o|
    const float dd = x * a.ddx + y * a.ddy + z * a.ddz;
    const float inv = 1.0f / n;
    dmean[0] += (a.ddx - x * dd) * inv;
    dmean[1] += (a.ddy - y * dd) * inv;
    dmean[2] += (a.ddz - z * dd) * inv;
}
GSR_HD void gsr_sh_backward(int deg, int M, const float* sh, const float* mean, const float* campos,
                            uint32_t clamped, const float* drgb_in, float* dsh, float* dmean) {
    gsr_sh_backward_row(deg, M, GsrShRow{sh}, mean, campos, clamped, drgb_in, GsrShRowOut{dsh}, dmean);
}

// Sigma3D = (R S)(R S)^T backward: dcov (independent packed entries) -> dscale[3], drot[4]
// (w.r.t. the quaternion as given, no normalisation backward).
template <class R>
GSR_HD void gsr_cov3d_backward_r(const float* s, float mod, const float* q, const R* dcov, float* dscale, float* drot) {
    const R one = 1, two = 2, four = 4, half = (R)0.5;
    const R s0 = (R)mod * (R)s[0], s1 = (R)mod * (R)s[1], s2 = (R)mod * (R)s[2];
    const R r = q[0], x = q[1], y = q[2], z = q[3];
    const R Rm[3][3] = {{one - two * (y * y + z * z), two * (x * y - r * z), two * (x * z + r * y)},
                        {two * (x * y + r * z), one - two * (x * x + z * z), two * (y * z - r * x)},
                        {two * (x * z - r * y), two * (y * z + r * x), one - two * (x * x + y * y)}};
    const R sv[3] = {s0, s1, s2};
    R Mm[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Mm[i][j] = Rm[i][j] * sv[j];
    // full symmetric gradient matrix G with G_ij = dL/dSigma_ij counting each off-diagonal independent
    // entry once split over both positions: Sigma = M M^T => dL/dM = (G + G^T) M with G upper-packed/2
    const R G[3][3] = {{dcov[0], half * dcov[1], half * dcov[2]},
                       {half * dcov[1], dcov[3], half * dcov[4]},
                       {half * dcov[2], half * dcov[4], dcov[5]}};
    R dM[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            R acc = 0;
            for (int k = 0; k < 3; ++k) acc += two * G[i][k] * Mm[k][j];
            dM[i][j] = acc;
        }
    R dR[3][3];
    for (int j = 0; j < 3; ++j) {
        R acc = 0;
        for (int i = 0; i < 3; ++i) {
            acc += dM[i][j] * Rm[i][j];
            dR[i][j] = dM[i][j] * sv[j];
        }
        dscale[j] = (float)(acc * (R)mod);
    }
    drot[0] = (float)(two * (z * (dR[1][0] - dR[0][1]) + y * (dR[0][2] - dR[2][0]) + x * (dR[2][1] - dR[1][2])));
    drot[1] = (float)(two * (y * (dR[0][1] + dR[1][0]) + z * (dR[0][2] + dR[2][0]) + r * (dR[2][1] - dR[1][2])) -
                      four * x * (dR[1][1] + dR[2][2]));
    drot[2] = (float)(two * (x * (dR[0][1] + dR[1][0]) + r * (dR[0][2] - dR[2][0]) + z * (dR[1][2] + dR[2][1])) -
                      four * y * (dR[0][0] + dR[2][2]));
    drot[3] = (float)(two * (r * (dR[1][0] - dR[0][1]) + x * (dR[0][2] + dR[2][0]) + y * (dR[1][2] + dR[2][1])) -
                      four * z * (dR[0][0] + dR[1][1]));
}
GSR_HD void gsr_cov3d_backward(const float* s, float mod, const float* q, const float* dcov, float* dscale, float* drot) {
    gsr_cov3d_backward_r<float>(s, mod, q, dcov, dscale, drot);
}

// The arithmetic type of the per-Gaussian backward's covariance chain in the product (gsr_project_backward_r's header comment).
#ifndef GSR_BWD_REAL
#define GSR_BWD_REAL double      /* A/B builds: GSR_EXTRA_FLAGS=-DGSR_BWD_REAL=float is rounds 1-5's all-fp32 chain */
#endif
typedef GSR_BWD_REAL GsrBwdReal;
