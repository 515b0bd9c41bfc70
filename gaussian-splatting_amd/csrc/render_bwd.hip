// Backward of the alpha-compositing pass -- replaces renderCUDA (backward) of the un-vendored reference
// rasterizer (algorithm SURVEY.md Appendix A.5; reached through loss.backward(), train.py:142).
//
// The reference issues ~10 global float atomics per contributing (pixel, Gaussian) pair.  On MI355X device-scope
// float atomics are resolved at the memory side (8 non-coherent XCD L2s): one atomic set per (8x8 block, Gaussian) -- the
// first design here, kept as A/B variant 1 -- measured 2.46 ms on the 1 M / 1080p frame, 7x the forward blend.  The
// default design has NO atomics of any kind, no barriers, and is BIT-REPRODUCIBLE:
//
//  render_bwd_quad     one wave64 per 8x8 pixel QUADRANT of a tile, fully independent of the other three quadrants (no
//                      workgroup barrier: round 1's workgroup-per-tile kernel, A/B variant 4, spent 35 % of its wave time
//                      waiting for the slowest quadrant at every super-batch, and its four waves added into one LDS table
//                      in arrival order).  The quadrant's list is walked BACK TO FRONT from its own last contributor in
//                      batches of 64 entries: the lanes first act as Gaussian lanes (coalesced id read, 64-byte record
//                      gather -- ids two batches ahead, records one batch ahead, as in the forward --, the same exact
//                      box test as the forward: a culled entry contributed to no pixel of the box, so its gradient from
//                      this box is zero), park the batch in the wave's private LDS, then walk the survivors (s_flbit);
//                      every lane computes its pixel's 10 per-pair values (five MOMENTS of the pixel offset, see
//                      bwd_step, plus opacity / colour / inverse-depth terms), and the 10 values are summed over the 64
//                      lanes with a transpose-reduce: v_permlane32_swap + v_permlane16_swap butterflies fold four values
//                      into one register (one value per 16-lane row), DPP row shifts finish each row -- ~27 VALU ops for
//                      10 values instead of 60.  Lanes 15/31/47/63 store the row totals into the batch's LDS record
//                      table (plain stores: a (quadrant, entry) pair is visited exactly once).  After the batch every
//                      touched entry's 48-byte record goes to the quadrant's SLOT of the instance's EMISSION index
//                      k = goffset[g] + (ty-miny)*w + (tx-minx) (rectangle and goffset ride in the 4th quad of the
//                      64-byte splat record), plus one flag byte: slot_grads[quadrant][k], flags[k] byte `quadrant`.
//  bwd_reduce_instances  in emission order a Gaussian's instances are contiguous: streams the flag words, adds the up
//                      to four quadrant records of every instance in FIXED order, sums the runs of each Gaussian with a
//                      segmented scan, turns the summed moments into derivatives with the Gaussian's conic, and writes
//                      the per-Gaussian 2-D gradient record ("splat_grads") that preprocess.hip's fused per-Gaussian
//                      backward consumes.  Every sum has a fixed association order -> two runs agree bit for bit.
//
// Per-Gaussian record layout ("splat_grads", float[12]; also variant 1 and the multi-GPU exchange): 0 dL/dpx 1 dL/dpy
// (pixel units) 2 dL/dA 3 dL/dB 4 dL/dC (plain derivatives of the conic entries, power = -0.5(A dx^2 + C dy^2) - B dx dy)
// 5 dL/d(opacity*aa) 6,7,8 dL/d(rgb) 9 dL/d(1/depth) 10,11 pad.  Instance records hold the raw moments
// (sum m dx, sum m dy, sum m dx^2, sum m dx dy, sum m dy^2) in slots 0..4 instead.
#include "gsr_internal.h"
#include <algorithm>
#include "gsr_wave.h"

namespace {

#ifdef GSR_SIMT_SHIM      // (tests/simt/: the kernel source compiled for the host, where the two swap builtins are functions of the shim)
typedef uint2 uint2v;
#else
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
#endif

// Every cross-lane add of the walk is ONE v_add_f32_dpp: the DPP moves use the full row mask and bound_ctrl (a lane without a source reads 0; the rows a
// partial row mask would protect only hold partial sums nobody reads: row_bcast:15 results are consumed in lanes 31 / 63, row_bcast:31 in lane 63), and
// the sum is pinned in a register before the branch that consumes it.  (Rounds 2-4 left four of them per step as v_mov 0 + v_mov_dpp + v_add: the
// compiler sinks the add of the last row_shr into the "(lane & 15) == 15" branch -- a DPP move cannot follow it there -- and it cannot fuse a move with a
// partial row mask into a float add, the kept lanes would need -0 + 0 = -0.)  103 -> 95 VALU per step; the consumed lanes add the same values in the same
// order, so the gradients are the bits of the earlier form (tests/test_simt_forward_cpu.py ran both); measured on the GPU, same box, interleaved:
// blend backward 0.342 -> 0.326 ms (profiles/r05_ab_candidates.json).
// dpp_pin: no instruction -- the sum exists in all lanes here, so its add stays next to its DPP move instead of sinking into the consumer's branch, and
// the pins of one stage keep their order
__device__ __forceinline__ void dpp_pin(float& r) {
#if !defined(GSR_SIMT_SHIM)
    asm volatile("" : "+v"(r));
#endif
}
template <int CTRL, int ROW_MASK /*documents which rows consume the result; the move itself takes all rows*/>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(moved);
}
template <int CTRL>
__device__ __forceinline__ void dpp_stage(float& c, float& a, float& b) {
    c = dpp_add<CTRL, 0xf>(c); dpp_pin(c);
    a = dpp_add<CTRL, 0xf>(a); dpp_pin(a);
    b = dpp_add<CTRL, 0xf>(b); dpp_pin(b);
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

// one step of a segmented inclusive scan over the wave: lanes whose DPP source lane carries the same segment id add its values
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_scan_step(int seg, float (&v)[10]) {
    const int src_seg = __builtin_amdgcn_update_dpp(-1, seg, CTRL, ROW_MASK, 0xf, false);
    const bool take = src_seg == seg;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const float t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), CTRL, ROW_MASK, 0xf, false));
        if (take) v[i] += t;
    }
}

// [a.lo+a.hi | b.lo+b.hi] : lanes 0-31 hold 32 partial sums of a, lanes 32-63 of b
__device__ __forceinline__ float fold32(float a, float b) {
    const uint2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// rows (16 lanes): [p.row0+p.row1 | q.row0+q.row1 | p.row2+p.row3 | q.row2+q.row3]
__device__ __forceinline__ float fold16(float p, float q) {
    const uint2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(q), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// Four values summed over the wave at once.  Result: lane 15 -> sum(a), lane 31 -> sum(c), lane 47 -> sum(b),
// lane 63 -> sum(d) (other lanes hold partial sums).
__device__ __forceinline__ float reduce4(float a, float b, float c, float d) {
    float v = fold16(fold32(a, b), fold32(c, d));
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    return v;
}

// Two values summed over the wave: lane 31 -> sum(a), lane 63 -> sum(b)
__device__ __forceinline__ float reduce2(float a, float b) {
    float v = fold32(a, b);
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
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
        const float dy = fminf(hy, fmaxf(ly, -B * dx * __builtin_amdgcn_rcpf(C)));
        q = fminf(q, A * dx * dx + 2.0f * B * dx * dy + C * dy * dy);
    }
    if (!in_y) {
        const float dy = ly > 0.0f ? ly : hy;
        const float dx = fminf(hx, fmaxf(lx, -B * dy * __builtin_amdgcn_rcpf(A)));
        q = fminf(q, A * dx * dx + 2.0f * B * dx * dy + C * dy * dy);
    }
    return q;
}

__device__ __forceinline__ float bcast(float v, int srclane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srclane));
}

// per-pixel running state of the back-to-front walk (Appendix A.5).  The reference keeps one running colour per channel
// ("accum_rec") and dots (c - accum) with dL/dpixel afterwards; both are linear in the channel, so the walk here carries
// the already-dotted scalars: accD = <accum, dL/dpix>, lastD = <last colour, dL/dpix> -- 2 VALU ops per step instead of 12.
struct BwdPix {
    float T, accD, lastD, last_alpha;
};

// One (pixel, Gaussian) step.  Outputs are zero when the pixel does not take part (hard masks: behind n_contrib,
// power > 0, alpha < 1/255).  The five geometric outputs are the raw MOMENTS of m = dL/dG * G over the pixel offsets,
//     mx = m dx, my = m dy, mxx = m dx^2, mxy = m dx dy, myy = m dy^2,
// not the derivatives themselves: the derivatives are linear in the moments with per-Gaussian coefficients
//     dL/dpx = -A mx - B my,  dL/dpy = -C my - B mx,  dL/dA = -mxx/2,  dL/dB = -mxy,  dL/dC = -myy/2
// so that multiplication is done ONCE per Gaussian after all sums (bwd_reduce_instances) instead of per pair.
// Pairs of per-pixel values (the two pixels of a lane).  Plain scalars on purpose: written on ext_vector_type(2) floats the
// compiler emits v_pk_{mul,add,fma}_f32, and on this part a packed op costs MORE than the two scalar ops it replaces
// (measured on one box, whole kernel: 0.4476 ms packed, 0.4428 ms scalar; the forward blend likewise 0.1481 -> 0.1435 ms
// once the SLP vectoriser's automatic packing was switched off) -- this file and render_fwd.hip are built -fno-slp-vectorize.
struct v2f { float x, y; };
__device__ __forceinline__ v2f operator+(v2f a, v2f b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ v2f operator-(v2f a, v2f b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ v2f operator*(v2f a, v2f b) { return {a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ v2f operator*(float a, v2f b) { return {a * b.x, a * b.y}; }
__device__ __forceinline__ v2f operator-(v2f a) { return {-a.x, -a.y}; }

__device__ __forceinline__ bool bwd_step(BwdPix& s, bool take, float pxf, float pyf, float Tf_bg, float dLr, float dLg,
                                         float dLb, float dLd, float gx_, float gy_, float a2, float b2, float c2, float op,
                                         float cr, float cg, float cb, float idp, float& mx, float& my, float& mxx,
                                         float& mxy, float& myy, float& g_op, float& g_r, float& g_g, float& g_b,
                                         float& g_d) {
    const v2f d = (v2f){gx_, gy_} - (v2f){pxf, pyf};
    const float dx = d.x, dy = d.y;
    const float t = fmaf(b2, dy, a2 * dx);
    const float p2 = fmaf(dx, t, (c2 * dy) * dy);
    const float G = __builtin_amdgcn_exp2f(p2);
    const float alpha = fminf(GSR_ALPHA_MAX, op * G);
    const bool active = take & (p2 <= 0.0f) & (alpha >= GSR_ALPHA_MIN);
    const v2f dL01 = {dLr, dLg}, dL23 = {dLb, dLd};
    const v2f cd = (v2f){cr, cg} * dL01 + (v2f){cb, idp} * dL23;
    const float cD = cd.x + cd.y;
    float w = 0.0f, dL_dalpha = 0.0f;
    if (active) {
        const float inv1ma = __builtin_amdgcn_rcpf(1.0f - alpha);
        s.T = s.T * inv1ma;
        w = alpha * s.T;
        s.accD = fmaf(s.last_alpha, s.lastD - s.accD, s.accD);
        s.lastD = cD;
        s.last_alpha = alpha;
        dL_dalpha = fmaf(cD - s.accD, s.T, Tf_bg * inv1ma);      // Tf_bg = -T_final * <bg, dL/dpix>
    }
    const v2f g01 = w * dL01, g23 = w * dL23;
    g_r = g01.x; g_g = g01.y; g_b = g23.x; g_d = g23.y;
    const float m = op * (G * dL_dalpha);
    g_op = m;                            // zeroth moment (see moments_to_grads)
    const v2f md = m * d;                // (m dx, m dy)
    const v2f mxd = md.x * d;            // (m dx^2, m dx dy)
    mx = md.x; my = md.y;
    mxx = mxd.x; mxy = mxd.y;
    myy = md.y * dy;
    return active;
}

// per-Gaussian conversion of the summed moments into the derivatives (record layout of the file header)
// Slot 5 of an instance record carries the ZEROTH moment sum(m) = opacity * sum(G dL/dalpha) (round 3: the blend kernel needs m
// anyway, so dL/dopacity = sum(m) / opacity is one division per Gaussian here instead of two multiplies and an add per step there;
// a Gaussian with opacity 0 never passes alpha >= 1/255, its sum is exactly 0).
__device__ __forceinline__ void moments_to_grads(float A, float B, float C, float opacity, float4& u0, float4& u1) {
    const float mx = u0.x, my = u0.y, mxx = u0.z, mxy = u0.w, myy = u1.x;
    u0.x = -A * mx - B * my;
    u0.y = -C * my - B * mx;
    u0.z = -0.5f * mxx;
    u0.w = -mxy;
    u1.x = -0.5f * myy;
    u1.y = opacity > 0.0f ? u1.y / opacity : 0.0f;
}

constexpr float LOG2E = 1.4426950408889634f;

// q3 of the splat record = (rect.x bits, rect.y bits, first emission index bits, tiles bits), see preprocess.hip / binning.hip
__device__ __forceinline__ uint32_t emission_index(const float4 q3, uint32_t tx, uint32_t ty) {
    const uint32_t rx = __float_as_uint(q3.x), ry = __float_as_uint(q3.y), goff = __float_as_uint(q3.z);
    const uint32_t minx = rx & 0xFFFFu, w = (rx >> 16) - minx, miny = ry & 0xFFFFu;
    return goff + (ty - miny) * w + (tx - minx);
}
constexpr int REC_STRIDE = 3;      // float4 per staged entry (48 B: 12-word stride, 3 coprime to 16 -> per-lane ds_read_b128 is conflict-free)

// ------------------------------------------------------------------------------------------------
// default: one wave per 16x8 HALF tile, two pixels per lane (same row, 8 columns apart), no atomics, no barriers.
// The blend backward is VALU-issue bound (SQ counters), so the lever is instructions per (pixel, entry) pair:
//  * the two pixels of a lane share dy, c2 dy^2, every wave-uniform operand, the LDS record reads and the loop's scalar work
//    (their per-pixel arithmetic is written on pairs, v2f -- scalar pairs: packed v_pk_*_f32 forms measured slower);
//  * their ten gradient terms are added per lane BEFORE the cross-lane transpose-reduce, so the ~27-instruction reduction is
//    paid once per (half tile, entry) instead of once per (quadrant, entry);
//  * an instance gets at most two records (one per half) instead of four, which halves the reduce kernel's stream.
// Price: the survivor list of a wave is the UNION of its two quadrants' lists (measured on the bench frame: 0.58 of their sum).
// p2 is computed by the forward's exact operation sequence (mul, fma, fma), so the hard masks agree with the forward's.
// ------------------------------------------------------------------------------------------------
// Round 3: the walk carries TWO values per pixel, (T, acc), instead of four.  The reference's lazy pair (last_alpha,
// last_color) is folded eagerly: after an entry has used acc (= <running colour behind it, dL/dpix>) the entry itself is folded
// in at once, acc <- fma(alpha, cD - acc, acc) -- the very expression (same operands, same rounding) that the lazy form
// evaluates one active entry later, so results are bit-identical to round 2's kernel.  An entry that a pixel skips (behind
// its last contributor, power > 0, alpha < 1/255) takes part with alpha = 0 and G = 0: then 1 - alpha = 1, rcp(1) = 1,
// T * 1 = T, fma(0, x, acc) = acc and every gradient term is 0 * finite = 0 -- the state needs no select at all, and the
// per-pixel select count drops from six (T, acc, last colour, last alpha, w, dL/dalpha) to two (alpha, G).
// HAS_DEPTH = false (no gradient arrives for the inverse-depth image: train.py without depth supervision) drops the
// 1/depth term of cD, the tenth gradient value and one cross-lane reduction.
struct BwdPix2 {
    v2f T, acc;
};

// occupancy attribute of the walk kernel (tuning builds: GSR_EXTRA_FLAGS='-DGSR_BWD_OCC=__attribute__((amdgpu_waves_per_eu(8,8)))')
#ifndef GSR_BWD_OCC
#define GSR_BWD_OCC
#endif
template <bool HAS_DEPTH>
__global__ void __launch_bounds__(64) GSR_BWD_OCC
render_bwd_half(GsrCamDev cam, int n_band_tiles, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                const float4* __restrict__ splats, const float* __restrict__ final_T,
                const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                const float* __restrict__ dL_dinvdepth, float4* __restrict__ slot_grads /*[4][R] records of 3 float4*/,
                uint8_t* __restrict__ slot_flags /*[R][4]*/, int64_t R, const uint32_t* __restrict__ tile_order /*NULL: index order*/,
                int wave_order /*tile_order lists (tile << 1 | half) per WAVE, heaviest first*/, unsigned long long* __restrict__ counters) {
    __shared__ float4 s_rec[64 * REC_STRIDE];
    __shared__ float s_grad[64 * 12];
    // the two halves of a tile get workgroup ids b and b + 16 -> same XCD -> they share the gathered records in L2;
    // workgroups are dispatched in id order, and tile_order lists the heaviest tiles first (bwd_plan_kernel below)
    const int b = blockIdx.x;
    int tile_local, half;
    if (wave_order) {
        // every half tile has its own place in the launch: the waves that start last are the lightest ones of the whole frame, so the
        // launch drains in the time of a light wave (plan kernel below; the two halves of a tile weigh about the same and still start
        // within a few hundred workgroups of each other)
        if (b >= 2 * n_band_tiles) return;
        const uint32_t e = tile_order[b];
        tile_local = (int)(e >> 1);
        half = (int)(e & 1u);
    } else {
        const int grp = b >> 5, r32 = b & 31;
        const int turn = grp * 16 + (r32 & 15);
        half = r32 >> 4;
        if (turn >= n_band_tiles) return;
        tile_local = tile_order ? (int)tile_order[turn] : turn;
    }
    const unsigned long long t_start = counters ? wall_clock64() : 0ull;      // (measurement only: per-wave trace, gsr_profile_trace)
    const bool tracing = counters && gsr_trace_mode(counters);
    unsigned long long t_mark = t_start, t_walk = 0ull, t_prep = 0ull, t_store = 0ull;
    const int tile = cam.tile_y0 * cam.gx + tile_local;
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const int lane = threadIdx.x;
    const int bx0 = tx * GSR_TILE, by0 = ty * GSR_TILE + half * 8;
    if (by0 >= cam.H) return;
    const int pxA = bx0 + (lane & 7), pxB = pxA + 8, py = by0 + (lane >> 3);
    const bool inA = pxA < cam.W && py < cam.H, inB = pxB < cam.W && py < cam.H;
    const float pxfA = (float)pxA, pxfB = (float)pxB;
    const float pyf = (float)py;
    // boxes of the existing pixels of the two quadrants (the right one may be empty at the image border)
    const float y0 = (float)by0, y1 = (float)min(by0 + 7, cam.H - 1);
    const float xa0 = (float)bx0, xa1 = (float)min(bx0 + 7, cam.W - 1);
    const float xb0 = (float)(bx0 + 8), xb1 = (float)min(bx0 + 15, cam.W - 1);
    const bool quadB_alive = bx0 + 8 < cam.W;
    const uint2 range = ranges[tile];
    const int64_t pixA = (int64_t)py * cam.W + pxA, pixB = pixA + 8;
    const int64_t HW = (int64_t)cam.H * cam.W;
    const v2f T_final = {inA ? final_T[pixA] : 0.f, inB ? final_T[pixB] : 0.f};
    const uint32_t lastA = inA ? n_contrib[pixA] : 0u, lastB = inB ? n_contrib[pixB] : 0u;
    const v2f dLr = {inA ? dL_dpix[pixA] : 0.f, inB ? dL_dpix[pixB] : 0.f};
    const v2f dLg = {inA ? dL_dpix[HW + pixA] : 0.f, inB ? dL_dpix[HW + pixB] : 0.f};
    const v2f dLb = {inA ? dL_dpix[2 * HW + pixA] : 0.f, inB ? dL_dpix[2 * HW + pixB] : 0.f};
    v2f dLd = {0.f, 0.f};
    if (HAS_DEPTH) dLd = (v2f){inA ? dL_dinvdepth[pixA] : 0.f, inB ? dL_dinvdepth[pixB] : 0.f};
    const v2f Tf_bg = -T_final * (cam.bg[0] * dLr + cam.bg[1] * dLg + cam.bg[2] * dLb);
    uint32_t mxA = lastA, mxB = lastB;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mxA = max(mxA, (uint32_t)__shfl_xor((int)mxA, off, 64));
        mxB = max(mxB, (uint32_t)__shfl_xor((int)mxB, off, 64));
    }
    // (the butterfly leaves the same maximum in every lane, which the compiler cannot know: as a scalar the batch counter, the list position of a step and
    // the loop tests move to the SALU)
    const uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)min(range.y - range.x, max(mxA, mxB)), 0);
    if (end == 0) return;

    BwdPix2 s = {T_final, {0.f, 0.f}};
    float4* s_grad4 = reinterpret_cast<float4*>(s_grad);
    float4* slot = slot_grads + (int64_t)half * R * 3;
    const int nbatch = (int)((end + 63u) >> 6);
    auto load_id = [&](int bi) -> uint32_t {
        const uint32_t e = (uint32_t)bi * 64u + (uint32_t)lane;
        return (bi >= 0 && e < end) ? point_list[range.x + e] : 0xFFFFFFFFu;
    };
    uint32_t id_n = load_id(nbatch - 1);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 n0 = zero4, n1 = zero4, n2 = zero4, n3 = zero4;
    if (id_n != 0xFFFFFFFFu) { n0 = splats[id_n * 4 + 0]; n1 = splats[id_n * 4 + 1]; n2 = splats[id_n * 4 + 2]; n3 = splats[id_n * 4 + 3]; }
    id_n = load_id(nbatch - 2);
    uint32_t nsteps = 0;
    for (int bi = nbatch - 1; bi >= 0; --bi) {
        const uint32_t base = (uint32_t)bi * 64u;
        const uint32_t n = min(64u, end - base);
        const float4 q0 = n0, q1 = n1, q2 = n2, q3 = n3;
        if (id_n != 0xFFFFFFFFu) { n0 = splats[id_n * 4 + 0]; n1 = splats[id_n * 4 + 1]; n2 = splats[id_n * 4 + 2]; n3 = splats[id_n * 4 + 3]; }
        id_n = load_id(bi - 2);
        bool keepA = false, keepB = false;
        uint32_t k_emit = 0;
        if ((uint32_t)lane < n) {
            const uint32_t pos = base + (uint32_t)lane;
            // an entry at or behind a quadrant's own last contributor touches none of its pixels
            keepA = pos < mxA && !(min_q_over_box(q0.x, q0.y, q0.z, q0.w, q1.x, xa0, xa1, y0, y1) > q2.z);
            keepB = quadB_alive && pos < mxB && !(min_q_over_box(q0.x, q0.y, q0.z, q0.w, q1.x, xb0, xb1, y0, y1) > q2.z);
            k_emit = emission_index(q3, (uint32_t)tx, (uint32_t)ty);
            s_rec[lane * REC_STRIDE + 0] = make_float4(q0.x, q0.y, -0.5f * LOG2E * q0.z, -LOG2E * q0.w);
            s_rec[lane * REC_STRIDE + 1] = make_float4(-0.5f * LOG2E * q1.x, q1.y, q1.z, q1.w);
            s_rec[lane * REC_STRIDE + 2] = make_float4(q2.x, q2.w, 0.f, 0.f);
        }
        const uint64_t maskA = __ballot(keepA), maskB = __ballot(keepB);
        uint64_t mask = maskA | maskB;
        nsteps += (uint32_t)__popcll(mask);
        uint64_t touched = 0ull;
        if (tracing) { const unsigned long long t = wall_clock64(); t_prep += t - t_mark; t_mark = t; }
        while (mask) {
            const int j = 63 - __builtin_clzll(mask);
            mask &= ~(1ull << j);
            const uint32_t pos0 = base + (uint32_t)j;
            const bool doA = (maskA >> j) & 1ull, doB = (maskB >> j) & 1ull;      // wave-uniform
            const float4 r0 = s_rec[j * REC_STRIDE + 0];
            const float4 r1 = s_rec[j * REC_STRIDE + 1];
            float r2x, r2y = 0.f;
            if (HAS_DEPTH) { const float2 r2 = *reinterpret_cast<const float2*>(&s_rec[j * REC_STRIDE + 2]); r2x = r2.x; r2y = r2.y; }
            else r2x = s_rec[j * REC_STRIDE + 2].x;
            const float a2 = r0.z, b2 = r0.w, c2 = r1.x, op = r1.y;
            // ---- geometry: the forward's operation sequence per pixel (mul, fma, fma; blend_step_bf in render_fwd.hip), so
            // that the hard masks agree with the forward's bit for bit ----
            const float dxA = r0.x - pxfA, dxB = r0.x - pxfB;
            const float dy = r0.y - pyf;
            const float u = (c2 * dy) * dy;
            const float tA = fmaf(b2, dy, a2 * dxA), tB = fmaf(b2, dy, a2 * dxB);
            const float p2A = fmaf(dxA, tA, u), p2B = fmaf(dxB, tB, u);
            const float GA = __builtin_amdgcn_exp2f(p2A), GB = __builtin_amdgcn_exp2f(p2B);
            const float ogA = op * GA, ogB = op * GB;
            const float alA = fminf(GSR_ALPHA_MAX, ogA), alB = fminf(GSR_ALPHA_MAX, ogB);
            const bool actA = doA & (pos0 < lastA) & (p2A <= 0.0f) & (alA >= GSR_ALPHA_MIN);
            const bool actB = doB & (pos0 < lastB) & (p2B <= 0.0f) & (alB >= GSR_ALPHA_MIN);
            if (__builtin_amdgcn_ballot_w64(actA | actB) == 0ull) continue;
            // ---- the only selects of the step: a skipped pixel takes part with alpha = 0 and opacity * G = 0 ----
            const v2f alpha = {actA ? alA : 0.0f, actB ? alB : 0.0f};
            const v2f opG = {actA ? ogA : 0.0f, actB ? ogB : 0.0f};
            const v2f dx = {dxA, dxB};
            // ---- recurrences (Appendix A.5) ----
            v2f cD = r1.z * dLr + r1.w * dLg + r2x * dLb;
            if (HAS_DEPTH) cD = cD + r2y * dLd;
            const v2f one_m = (v2f){1.0f, 1.0f} - alpha;
            const v2f inv1ma = {__builtin_amdgcn_rcpf(one_m.x), __builtin_amdgcn_rcpf(one_m.y)};
            s.T = s.T * inv1ma;
            const v2f diff = cD - s.acc;
            const v2f dL_dalpha = diff * s.T + Tf_bg * inv1ma;
            s.acc = (v2f){fmaf(alpha.x, diff.x, s.acc.x), fmaf(alpha.y, diff.y, s.acc.y)};
            const v2f w = alpha * s.T;
            // ---- the per-pair terms, the two pixels added per lane ----
            const v2f gr2 = w * dLr, gg2 = w * dLg, gb2 = w * dLb;
            const v2f m = opG * dL_dalpha;          // m = dL/dG * G = opacity G dL/dalpha (straight-through the 0.99 cap, Appendix A.5)
            const v2f mdx = m * dx;                 // m dx
            const v2f mdxx = mdx * dx;              // m dx^2
            const float msum = m.x + m.y, mdxsum = mdx.x + mdx.y;
            const float g_px = mdxsum, g_py = msum * dy, g_A = mdxx.x + mdxx.y, g_B = mdxsum * dy, g_C = (msum * dy) * dy;
            const float g_op = msum, g_r = gr2.x + gr2.y, g_g = gg2.x + gg2.y, g_b = gb2.x + gb2.y;      // g_op: the zeroth moment
            // the three reductions stage by stage (same adds per chain as reduce4 / reduce2 / wave_sum_to_lane63 above, which the measurement build's variants still call): between two DPP adds of one chain
            // stand the other two chains' adds -- the two wait states a DPP read of a fresh VALU result needs, filled with work instead of s_nop
            float v0 = fold16(fold32(g_px, g_A), fold32(g_py, g_B));     // -> slots 0,1,2,3
            float v1 = fold16(fold32(g_C, g_r), fold32(g_op, g_g));      // -> slots 4,5,6,7
            float v2 = g_b;                                              // -> slot 8 (lane 63) | slots 8 (lane 31), 9 (lane 63)
            if (HAS_DEPTH) {
                const v2f gd2 = w * dLd;
                v2 = fold32(g_b, gd2.x + gd2.y);
            }
            dpp_stage<0x111>(v2, v0, v1);
            dpp_stage<0x112>(v2, v0, v1);
            dpp_stage<0x114>(v2, v0, v1);
            dpp_stage<0x118>(v2, v0, v1);
            v2 = dpp_add<0x142, 0xa>(v2);
            if (!HAS_DEPTH) v2 = dpp_add<0x143, 0xc>(v2);
            dpp_pin(v2);
            if ((lane & 15) == 15) {
                float* o = s_grad + j * 12 + (lane >> 4);
                o[0] = v0;
                o[4] = v1;
                if (HAS_DEPTH) {
                    if (lane & 16) s_grad[j * 12 + 8 + (lane >> 5)] = v2;
                } else if (lane == 63) {
                    *reinterpret_cast<float2*>(&s_grad[j * 12 + 8]) = make_float2(v2, 0.0f);
                }
            }
            touched |= 1ull << j;
        }
        if (tracing) { const unsigned long long t = wall_clock64(); t_walk += t - t_mark; t_mark = t; }
        if (touched) {
            __builtin_amdgcn_wave_barrier();
            if ((touched >> lane) & 1ull) {
                float4* dst = slot + (int64_t)k_emit * 3;
                dst[0] = s_grad4[lane * 3 + 0];
                dst[1] = s_grad4[lane * 3 + 1];
                dst[2] = s_grad4[lane * 3 + 2];
                slot_flags[(int64_t)k_emit * 4 + half] = 1;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (tracing) { const unsigned long long t = wall_clock64(); t_store += t - t_mark; t_mark = t; }
    }
    if (tracing) {
        if (lane == 0) gsr_trace_wave(counters, t_start, (uint32_t)blockIdx.x, 2u, gsr_trace_pack(nsteps, t_walk, t_prep, t_store));
    } else if (counters && lane == 0) {
        atomicAdd(counters + 2, (unsigned long long)nsteps);
        atomicAdd(counters + 3, (unsigned long long)nbatch);
        atomicMax(counters + 5, (unsigned long long)nsteps);      // the heaviest wave
    }
}

#ifdef GSR_AB_VARIANTS
#include "render_bwd_quad_superbatch.inc"      // measured-and-rejected variants (tools/ab_variants/): measurement build only
#endif  // GSR_AB_VARIANTS

// splat_grads[g] = sum of the instance records of Gaussian g.  In emission (= depth) order a Gaussian's records are one
// contiguous run.  Round 3: the stream of R records is cut into UNITS of 256 records, one wave per unit, whatever the
// Gaussians look like -- round 2 gave one workgroup to every 64 Gaussians of the depth order, and a splat that covers the
// whole frame (8 160 records; trained-looking scenes have many of hundreds) made its workgroup the kernel's tail: 0.09 ms on
// the uniform bench frame but 0.45 ms on the clustered stand-in, 41 % of its backward.
//   (memset)          all rows zero: Gaussians without records, and runs inside units that hold no record at all, are never written
//   reduce_prepare    per Gaussian: the units whose first record it owns get (owner, owner's first record) -- the only
//                     thing a unit needs to find all its owners
//   bwd_reduce_units  one wave per unit: the starts of the next <= 256 Gaussians are fetched in one batch and marked in LDS,
//                     a running max-scan over the marks gives every record its owner; 64 records per step (lane = record,
//                     fully coalesced; flag words up front, records one step ahead), the <= 2 slot records of an instance
//                     added in fixed order, a segmented scan (DPP) sums the runs; a run that begins and ends inside the unit is
//                     finished on the spot (moments -> derivatives with the Gaussian's conic), the run that enters from the
//                     previous unit becomes the unit's HEAD piece, the run that is still open at the unit's end its TAIL piece
//   reduce_stitch     per unit with a tail piece: tail + the head pieces of the following units while they belong to the same
//                     Gaussian (fixed order) -> the finished row
// Every sum has a fixed association order -> two runs agree bit for bit.
constexpr int RU = 256;                  // (1024 measured slower on regular scenes: 16 dependent steps per wave are a latency chain)
constexpr int RU_STEPS = RU / 64;

__device__ __forceinline__ void finish_row(const uint32_t* __restrict__ order, const float4* __restrict__ splats,
                                           float4* __restrict__ splat_grads, uint32_t j, const float (&v)[10]) {
    const uint32_t g = order[j];
    float4 t0 = make_float4(v[0], v[1], v[2], v[3]), t1 = make_float4(v[4], v[5], v[6], v[7]);
    const float4 q0 = splats[(int64_t)g * 4 + 0];
    const float4 q1 = splats[(int64_t)g * 4 + 1];
    moments_to_grads(q0.z, q0.w, q1.x, q1.y, t0, t1);
    splat_grads[(int64_t)g * 3 + 0] = t0;
    splat_grads[(int64_t)g * 3 + 1] = t1;
    splat_grads[(int64_t)g * 3 + 2] = make_float4(v[8], v[9], 0.f, 0.f);
}
// the same with the owner's (Gaussian id, conic) already staged in LDS by the unit's prologue: three stores, no dependent load
__device__ __forceinline__ void finish_row_staged(const float4 ow /*(id bits, A, B, C)*/, float opacity, float4* __restrict__ splat_grads,
                                                  const float (&v)[10]) {
    const uint32_t g = __float_as_uint(ow.x);
    float4 t0 = make_float4(v[0], v[1], v[2], v[3]), t1 = make_float4(v[4], v[5], v[6], v[7]);
    moments_to_grads(ow.y, ow.z, ow.w, opacity, t0, t1);
    splat_grads[(int64_t)g * 3 + 0] = t0;
    splat_grads[(int64_t)g * 3 + 1] = t1;
    splat_grads[(int64_t)g * 3 + 2] = make_float4(v[8], v[9], 0.f, 0.f);
}
__device__ __forceinline__ void write_piece(float4* __restrict__ piece, uint32_t j, const float (&v)[10], bool valid) {
    piece[0] = make_float4(v[0], v[1], v[2], v[3]);
    piece[1] = make_float4(v[4], v[5], v[6], v[7]);
    piece[2] = make_float4(v[8], v[9], __uint_as_float(j), __uint_as_float(valid ? 1u : 0u));
}

__global__ void __launch_bounds__(256)
reduce_prepare(int P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets, uint2* __restrict__ unit_first,
               float4* __restrict__ splat_grads) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < P; j += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t incl = offsets[j], excl = j ? offsets[j - 1] : 0u;
        if (incl == excl) continue;            // no record (the rows were zeroed by the launcher's memset)
        for (uint32_t k = (excl + RU - 1u) / RU; k <= (incl - 1u) / RU; ++k) unit_first[k] = make_uint2((uint32_t)j, excl);
    }
}

__global__ void __launch_bounds__(64)
bwd_reduce_units(int P, int64_t R, const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
                 const uint2* __restrict__ unit_first, const float4* __restrict__ slot_grads /*[SLOTS][R] records*/,
                 const uint32_t* __restrict__ inst_flag /*[R], byte q = slot q*/, const float4* __restrict__ splats,
                 float4* __restrict__ splat_grads, float4* __restrict__ unit_piece /*[units][2][3]*/) {
    __shared__ uint32_t s_mark[RU];
    __shared__ float4 s_own[RU + 1];              // per owner of the unit: (Gaussian id bits, conic A, B, C)
    __shared__ float s_own_op[RU + 1];            // ... and its opacity
    const int lane = threadIdx.x;
    const int64_t u = blockIdx.x;
    const uint32_t c0 = (uint32_t)(u * RU);
    const uint32_t n = (uint32_t)((R - (int64_t)c0) < (int64_t)RU ? (R - (int64_t)c0) : RU);
    const uint2 uf = unit_first[u];
    const uint32_t j0 = uf.x;
    const bool head_case = uf.y < c0;                 // owner 0's run began in an earlier unit
    // ---- every independent load of the unit up front: the starts of the next RU Gaussians, the flag words of all steps ----
    uint32_t st[RU_STEPS], fl[RU_STEPS];
#pragma unroll
    for (int t = 0; t < RU_STEPS; ++t) {
        const int64_t jj = (int64_t)j0 + 1 + t * 64 + lane;          // Gaussian jj starts at offsets[jj - 1]
        st[t] = offsets[jj <= (int64_t)P - 1 ? jj - 1 : (int64_t)P - 1];
        if (jj > (int64_t)P - 1) st[t] = 0xFFFFFFFFu;
        const uint32_t r = (uint32_t)t * 64u + (uint32_t)lane;
        fl[t] = inst_flag[(int64_t)c0 + (r < n ? r : 0u)];
        if (r >= n) fl[t] = 0u;
    }
#pragma unroll
    for (int t = 0; t < RU_STEPS; ++t) s_mark[t * 64 + lane] = 0u;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < RU_STEPS; ++t)
        if (st[t] >= c0 && st[t] - c0 < n) s_mark[st[t] - c0] = (uint32_t)(1 + t * 64 + lane);      // (Gaussians without records start at R)
    __builtin_amdgcn_wave_barrier();

    uint32_t mymax = 0u;
#pragma unroll
    for (int t = 0; t < RU_STEPS; ++t) mymax = max(mymax, s_mark[t * 64 + lane]);
    const uint32_t nown = (uint32_t)__builtin_amdgcn_readlane((int)gsrw::wave_incl_max_u32(mymax), 63);
    float4* piece = unit_piece + u * 6;
    {   // A unit without a single record (most of the stream: instances of far Gaussians that every pixel terminated in front of):
        // all its sums are zero and the rows are already zero -- only the two pieces are written (valid zeros, so that a run
        // that crosses this unit keeps its chain for reduce_stitch).
        uint32_t anyf = 0u;
#pragma unroll
        for (int t = 0; t < RU_STEPS; ++t) anyf |= fl[t];
        if (__ballot(anyf != 0u) == 0ull) {
            if (lane == 0) {
                float zero[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) zero[i] = 0.f;
                write_piece(piece, j0, zero, head_case);
                write_piece(piece + 3, j0 + nown, zero, !(nown == 0u && head_case));
            }
            return;
        }
    }
    // owners of the unit = local indices 0 .. nown: their Gaussian id and conic are fetched NOW (two dependent gathers, once per
    // unit, in flight beside the record loads) and parked in LDS -- fetched at the moment a run completes they were two
    // serial memory round trips in every step of the walk
    for (uint32_t k = 0; k * 64u <= nown; ++k) {          // wave-uniform trip count (usually one)
        const uint32_t o = min(k * 64u + (uint32_t)lane, nown);
        const uint32_t g = order[(int64_t)j0 + o];
        const float4 q0 = splats[(int64_t)g * 4 + 0];
        const float4 q1 = splats[(int64_t)g * 4 + 1];
        s_own[o] = make_float4(__uint_as_float(g), q0.z, q0.w, q1.x);
        s_own_op[o] = q1.y;
    }
    __builtin_amdgcn_wave_barrier();

    const float4* stream = slot_grads + (int64_t)c0 * 3;
    auto load_recs = [&](int t, uint32_t flags, float4 (&rec)[GSR_BWD_SLOTS][3]) {
        const uint32_t r = (uint32_t)t * 64u + (uint32_t)lane;
#pragma unroll
        for (int q = 0; q < GSR_BWD_SLOTS; ++q) {
            if ((flags >> (8 * q)) & 0xFFu) {
                const float4* p = stream + ((int64_t)q * R + (int64_t)r) * 3;
                rec[q][0] = p[0]; rec[q][1] = p[1]; rec[q][2] = p[2];
            } else {
                rec[q][0] = rec[q][1] = rec[q][2] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    float4 rec_cur[GSR_BWD_SLOTS][3], rec_nxt[GSR_BWD_SLOTS][3];
    load_recs(0, fl[0], rec_cur);
    uint32_t run_owner = 0u;                      // max-scan carry: owner (local index) of the last record seen
    bool have_open = false, head_written = false;
    uint32_t open_owner = 0u;
    float open_v[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) open_v[i] = 0.f;
    const int nsteps = (int)((n + 63u) >> 6);
#pragma unroll 1
    for (int t = 0; t < nsteps; ++t) {
        const uint32_t r = (uint32_t)t * 64u + (uint32_t)lane;
        const bool valid = r < n;
        if (t + 1 < nsteps) {
            uint32_t fnext = 0u;
#pragma unroll
            for (int k = 0; k < RU_STEPS; ++k) fnext = (k == t + 1) ? fl[k] : fnext;      // (register array: selected by unrolled compare)
            load_recs(t + 1, fnext, rec_nxt);
        }
        const uint32_t m = s_mark[r];
        const uint32_t own = max(run_owner, gsrw::wave_incl_max_u32(m));
        run_owner = (uint32_t)__builtin_amdgcn_readlane((int)own, 63);
        float v[10];
        {
            float4 a0 = rec_cur[0][0], a1 = rec_cur[0][1], a2 = rec_cur[0][2];
#pragma unroll
            for (int q = 1; q < GSR_BWD_SLOTS; ++q) {
                a0.x += rec_cur[q][0].x; a0.y += rec_cur[q][0].y; a0.z += rec_cur[q][0].z; a0.w += rec_cur[q][0].w;
                a1.x += rec_cur[q][1].x; a1.y += rec_cur[q][1].y; a1.z += rec_cur[q][1].z; a1.w += rec_cur[q][1].w;
                a2.x += rec_cur[q][2].x; a2.y += rec_cur[q][2].y;
            }
            v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
            v[8] = a2.x; v[9] = a2.y;
        }
#pragma unroll
        for (int q = 0; q < GSR_BWD_SLOTS; ++q) { rec_cur[q][0] = rec_nxt[q][0]; rec_cur[q][1] = rec_nxt[q][1]; rec_cur[q][2] = rec_nxt[q][2]; }
        // the run left open by the previous step: it continues into lane 0, or it is complete
        const uint32_t own0 = (uint32_t)__builtin_amdgcn_readlane((int)own, 0);
        if (have_open) {
            if (own0 == open_owner) {
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 10; ++i) v[i] += open_v[i];
                }
            } else if (lane == 0) {
                if (open_owner == 0u && head_case) write_piece(piece, j0, open_v, true);
                else finish_row_staged(s_own[open_owner], s_own_op[open_owner], splat_grads, open_v);
            }
            if (own0 != open_owner && open_owner == 0u && head_case) head_written = true;
        }
        const int seg = valid ? (int)own : 0x40000000 + lane;          // invalid lanes never merge
        uint32_t fcur = 0u;
#pragma unroll
        for (int k = 0; k < RU_STEPS; ++k) fcur = (k == t) ? fl[k] : fcur;
        if (__ballot(fcur != 0u) != 0ull) {
            seg_scan_step<0x111, 0xf>(seg, v);      // row_shr:1
            seg_scan_step<0x112, 0xf>(seg, v);      // row_shr:2
            seg_scan_step<0x114, 0xf>(seg, v);      // row_shr:4
            seg_scan_step<0x118, 0xf>(seg, v);      // row_shr:8
            seg_scan_step<0x142, 0xa>(seg, v);      // row_bcast:15 -> rows 1, 3
            seg_scan_step<0x143, 0xc>(seg, v);      // row_bcast:31 -> rows 2, 3
        } else {
            // no record in these 64 instances (far Gaussians hidden behind nearer ones: most of the stream): every sum is zero
            // except the run that continues the open one -- the scan's result without the scan
            const bool cont = have_open && own0 == open_owner && own == own0 && valid;
#pragma unroll
            for (int i = 0; i < 10; ++i) v[i] = cont ? open_v[i] : 0.0f;
        }
        const int last_lane = (int)min(63u, n - 1u - (uint32_t)t * 64u);
        const int seg_next = __builtin_amdgcn_update_dpp(-2, seg, 0x130, 0xf, 0xf, false);      // wave_shl:1 = the next lane's segment (DPP, no LDS round trip)
        const bool complete = valid && lane < last_lane && seg_next != seg;      // a run that ends strictly inside the step
        if (complete) {
            if (own == 0u && head_case) write_piece(piece, j0, v, true);
            else finish_row_staged(s_own[own], s_own_op[own], splat_grads, v);
        }
        if (__ballot(complete && own == 0u && head_case) != 0ull) head_written = true;
        have_open = true;
        open_owner = (uint32_t)__builtin_amdgcn_readlane((int)own, last_lane);
#pragma unroll
        for (int i = 0; i < 10; ++i) open_v[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[i]), last_lane));
    }
    // ---- the run still open at the unit's end: head piece (it entered from the previous unit and covers the whole unit) or tail ----
    if (lane == 0) {
        const bool open_is_head = have_open && open_owner == 0u && head_case;
        float zero[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) zero[i] = 0.f;
        if (open_is_head) write_piece(piece, j0, open_v, true);
        else if (!head_written) write_piece(piece, 0u, zero, false);
        if (have_open && !open_is_head) write_piece(piece + 3, j0 + open_owner, open_v, true);
        else write_piece(piece + 3, 0u, zero, false);
    }
}

__global__ void __launch_bounds__(256)
reduce_stitch(int64_t nunits, const uint32_t* __restrict__ order, const float4* __restrict__ splats,
              const float4* __restrict__ unit_piece, float4* __restrict__ splat_grads) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nunits) return;
    const float4* tp = unit_piece + u * 6 + 3;
    const float4 t2 = tp[2];
    if (__float_as_uint(t2.w) == 0u) return;
    const uint32_t j = __float_as_uint(t2.z);
    const float4 t0 = tp[0], t1 = tp[1];
    float v[10] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x, t2.y};
    for (int64_t k = u + 1; k < nunits; ++k) {           // the head pieces of the following units, in order, while they are this Gaussian's
        const float4* hp = unit_piece + k * 6;
        const float4 h2 = hp[2];
        if (__float_as_uint(h2.w) == 0u || __float_as_uint(h2.z) != j) break;
        const float4 h0 = hp[0], h1 = hp[1];
        v[0] += h0.x; v[1] += h0.y; v[2] += h0.z; v[3] += h0.w; v[4] += h1.x; v[5] += h1.y; v[6] += h1.z; v[7] += h1.w;
        v[8] += h2.x; v[9] += h2.y;
    }
    finish_row(order, splats, splat_grads, j, v);
}

#ifdef GSR_AB_VARIANTS
#include "render_bwd_atomics.inc"      // measured-and-rejected variants (tools/ab_variants/): measurement build only
#endif  // GSR_AB_VARIANTS

// ------------------------------------------------------------------------------------------------
// Heaviest tiles first.  A wave of the blend backward lives for a third of the kernel, the heaviest does 1.8-2x the mean
// number of steps, and workgroups start in id order: a heavy tile that starts in the last round IS the kernel's tail.  The
// forward has already counted, per 8x8 block, the entries it blended before the block's last pixel terminated -- the very
// survivors the backward walks back from the same point -- so the cost of every tile is known before the launch.  One
// workgroup counting-sorts the band's tiles by that cost, descending (512 bins, LDS atomics: the order inside a bin varies
// from run to run, the gradients do not -- every wave writes its own slots).
// ------------------------------------------------------------------------------------------------
constexpr int PLAN_THREADS = 1024, PLAN_WAVES = PLAN_THREADS / 64, PLAN_BINS = 512;
__device__ __forceinline__ int plan_bin(uint32_t w) {       // monotone; 8-step bins where frames live, 64-step bins above
    return w < 2048u ? (int)(w >> 3) : min(PLAN_BINS - 1, 256 + (int)((w - 2048u) >> 6));
}

// Every wave counts into its OWN copy of the histogram (a frame's tiles crowd into a few dozen bins).  The kernel takes ~9 us --
// one workgroup, five dependent phases -- whichever way its loads and atomics are arranged (one shared copy 9.1, per-wave copies
// 8.9, loads batched eight deep 12.6 on a slower box): more than half of what the order saves.  Filling work bins from the
// tail of the forward's waves instead (one returning atomic per block) was measured too: it takes the kernel out of the backward
// and puts its cost, and a little more, into the tracking forward.
// The same launch clears the instance flags (R bytes per record slot: 32 MB on the bench frame): workgroup 0 plans, the others
// fill -- the plan's 9 us of dependent phases and the fill's 8 us run side by side instead of one after the other, one launch
// boundary less.
// mode 1: tiles by the sum of their four blocks' steps (round 3); 2: tiles by their heaviest half; 3: HALF TILES (= waves) by their own
// steps, max of the two blocks a half covers -- measured against the backward's own step counts: correlation 0.98-0.998, the union of the
// two blocks' survivors is 1.00-1.06 x the larger block (tools/gpu_wave_trace.py).  Finer bins than round 3's (a half tile of the bench
// frame blends ~110 entries: eight-step bins put a third of the frame into one bin).
__device__ __forceinline__ int plan_bin_fine(uint32_t w) {       // monotone: 1-step bins below 384, 64-step bins above
    return w < 384u ? (int)w : min(PLAN_BINS - 1, 384 + (int)((w - 384u) >> 6));
}
__device__ __forceinline__ int plan_key(const uint4& v, int mode, int half) {
    if (mode == 1) return plan_bin(v.x + v.y + v.z + v.w);
    if (mode == 2) return plan_bin_fine(max(max(v.x, v.y), max(v.z, v.w)));
    return plan_bin_fine(half ? max(v.z, v.w) : max(v.x, v.y));
}
__global__ void __launch_bounds__(PLAN_THREADS)
bwd_plan_kernel(int tile0, int n_band_tiles, const uint4* __restrict__ block_steps, uint32_t* __restrict__ tile_order,
                uint4* __restrict__ flags16, int64_t n16, int mode) {
    if (blockIdx.x > 0) {
        const int64_t stride = (int64_t)(gridDim.x - 1) * PLAN_THREADS;
        for (int64_t i = (int64_t)(blockIdx.x - 1) * PLAN_THREADS + threadIdx.x; i < n16; i += stride) flags16[i] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    __shared__ uint32_t s_bin[PLAN_WAVES][PLAN_BINS];       // 32 KB
    __shared__ uint32_t s_wsum[PLAN_WAVES];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int n_items = mode == 3 ? 2 * n_band_tiles : n_band_tiles;
    const int sh = mode == 3 ? 1 : 0;
    for (int k = t; k < PLAN_WAVES * PLAN_BINS; k += PLAN_THREADS) (&s_bin[0][0])[k] = 0u;
    __syncthreads();
    for (int i = t; i < n_items; i += PLAN_THREADS) {
        const uint4 v = block_steps[tile0 + (i >> sh)];
        atomicAdd(&s_bin[wv][plan_key(v, mode, i & sh)], 1u);
    }
    __syncthreads();
    // thread t < PLAN_BINS owns the t-th heaviest bin: its total over the copies, then (after the scan over bins) a cursor per copy
    const int b = PLAN_BINS - 1 - (t & (PLAN_BINS - 1));
    uint32_t c = 0u;
    if (t < PLAN_BINS) {
#pragma unroll
        for (int k = 0; k < PLAN_WAVES; ++k) c += s_bin[k][b];
    }
    const uint32_t incl = gsrw::wave_incl_scan_u32(c, lane);
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    uint32_t base = 0u;
#pragma unroll
    for (int k = 0; k < PLAN_WAVES; ++k)
        if (k < wv) base += s_wsum[k];
    if (t < PLAN_BINS) {
        uint32_t run = base + incl - c;
#pragma unroll
        for (int k = 0; k < PLAN_WAVES; ++k) {
            const uint32_t n = s_bin[k][b];
            s_bin[k][b] = run;
            run += n;
        }
    }
    __syncthreads();
    for (int i = t; i < n_items; i += PLAN_THREADS) {      // (a wave sees the items it counted)
        const uint4 v = block_steps[tile0 + (i >> sh)];
        tile_order[atomicAdd(&s_bin[wv][plan_key(v, mode, i & sh)], 1u)] = (uint32_t)i;
    }
}

}  // namespace

int gsr_render_backward_variant_available(int variant) {
#ifdef GSR_AB_VARIANTS
    return variant == 0 || variant == 1 || variant == 4 || variant == 5;
#else
    return variant == 0;
#endif
}

void gsr_launch_render_backward(const GsrCamDev& cam, const uint2* ranges, const uint32_t* point_list,
                                const float4* splats, const float* final_T, const uint32_t* n_contrib,
                                const uint32_t* block_steps, uint32_t* tile_order,
                                const float* dL_dpix, const float* dL_dinvdepth, float* splat_grads, float* inst_grads,
                                uint32_t* inst_flag, int64_t R, int variant, int order_mode, unsigned long long* counters, hipStream_t st) {
    const int n_band_tiles = cam.gx * (cam.tile_y1 - cam.tile_y0);
    if (n_band_tiles <= 0) return;
    const int groups = (n_band_tiles + 7) / 8;
#ifdef GSR_AB_VARIANTS
    if (variant == 1) {
        hipLaunchKernelGGL(render_bwd_wave, dim3(groups * 32), dim3(64), 0, st, cam, n_band_tiles, ranges, point_list,
                           splats, final_T, n_contrib, dL_dpix, dL_dinvdepth, splat_grads);
        return;
    }
    if (variant == 4 || variant == 5) (void)hipMemsetAsync(inst_flag, 0, (size_t)R * 4, st);
    if (variant == 4) {
        hipLaunchKernelGGL(render_bwd_tile<256>, dim3(n_band_tiles), dim3(256), 0, st, cam, ranges, point_list, splats,
                           final_T, n_contrib, dL_dpix, dL_dinvdepth, reinterpret_cast<float4*>(inst_grads),
                           reinterpret_cast<uint8_t*>(inst_flag));
        return;
    }
    if (variant == 5) {
        hipLaunchKernelGGL(render_bwd_quad, dim3(groups * 32), dim3(64), 0, st, cam, n_band_tiles, ranges, point_list, splats,
                           final_T, n_contrib, dL_dpix, dL_dinvdepth, reinterpret_cast<float4*>(inst_grads),
                           reinterpret_cast<uint8_t*>(inst_flag), R, counters);
        return;
    }
#endif
    (void)variant; (void)splat_grads; (void)groups;
    const int groups16 = (n_band_tiles + 15) / 16;
    // instances that contributed nowhere get no record: only their flag words are cleared (the caller's scratch is carved in
    // 128-byte steps, so whole 16-byte words belong to the flag array)
    const int64_t n16 = (R * 4 + 15) / 16;
    if (tile_order && block_steps) {
        const int fill_blocks = (int)std::min<int64_t>(2048, (n16 + PLAN_THREADS - 1) / PLAN_THREADS);
        hipLaunchKernelGGL(bwd_plan_kernel, dim3(1 + fill_blocks), dim3(PLAN_THREADS), 0, st, cam.tile_y0 * cam.gx, n_band_tiles,
                           reinterpret_cast<const uint4*>(block_steps), tile_order, reinterpret_cast<uint4*>(inst_flag), n16, order_mode);
    } else {
        tile_order = nullptr;
        (void)hipMemsetAsync(inst_flag, 0, (size_t)R * 4, st);
    }
    const int wave_order = (tile_order && order_mode == 3) ? 1 : 0;
    if (dL_dinvdepth)
        hipLaunchKernelGGL(render_bwd_half<true>, dim3(groups16 * 32), dim3(64), 0, st, cam, n_band_tiles, ranges, point_list, splats,
                           final_T, n_contrib, dL_dpix, dL_dinvdepth, reinterpret_cast<float4*>(inst_grads),
                           reinterpret_cast<uint8_t*>(inst_flag), R, tile_order, wave_order, counters);
    else
        hipLaunchKernelGGL(render_bwd_half<false>, dim3(groups16 * 32), dim3(64), 0, st, cam, n_band_tiles, ranges, point_list, splats,
                           final_T, n_contrib, dL_dpix, dL_dinvdepth, reinterpret_cast<float4*>(inst_grads),
                           reinterpret_cast<uint8_t*>(inst_flag), R, tile_order, wave_order, counters);
}

size_t gsr_reduce_units(int64_t R) { return (size_t)((R + RU - 1) / RU); }

void gsr_launch_reduce_instances(int P, int64_t R, const uint32_t* order, const uint32_t* offsets, const float4* splats,
                                 const float* inst_grads, const uint32_t* inst_flag, float* splat_grads, uint2* unit_first,
                                 float* unit_piece, hipStream_t st) {
    const int64_t nunits = (R + RU - 1) / RU;
    int64_t nb = ((int64_t)P + 255) / 256;
    if (nb > 4096) nb = 4096;
    (void)hipMemsetAsync(splat_grads, 0, (size_t)P * 48, st);
    hipLaunchKernelGGL(reduce_prepare, dim3((int)nb), dim3(256), 0, st, P, order, offsets, unit_first, reinterpret_cast<float4*>(splat_grads));
    hipLaunchKernelGGL(bwd_reduce_units, dim3((int)nunits), dim3(64), 0, st, P, R, order, offsets, unit_first,
                       reinterpret_cast<const float4*>(inst_grads), inst_flag, splats, reinterpret_cast<float4*>(splat_grads),
                       reinterpret_cast<float4*>(unit_piece));
    hipLaunchKernelGGL(reduce_stitch, dim3((int)((nunits + 255) / 256)), dim3(256), 0, st, nunits, order, splats,
                       reinterpret_cast<const float4*>(unit_piece), reinterpret_cast<float4*>(splat_grads));
}
