// Forward alpha-compositing of the per-tile depth-sorted splat lists -- replaces renderCUDA (forward) of the
// un-vendored reference rasterizer (algorithm SURVEY.md Appendix A.4; caller:
// gaussian_renderer/__init__.py:91-110).
//
// gfx950 designs behind one launcher (variant chosen by gsr_set_option("render_fwd_variant")):
//
//  variant 0  "wave/LDS" : (default) one wave64 per 8x8 pixel block, no barriers.  The 64 lanes first act as 64
//                        *Gaussian* lanes: each loads one list entry (coalesced index read + 48-byte record gather)
//                        and tests it against the wave's 8x8 pixel box with an exact min-of-quadratic-over-a-box
//                        test -- entries that cannot reach alpha >= 1/255 anywhere in the box are dropped.  A
//                        64-bit ballot gives the survivor mask; the survivors are parked COMPACTED, in list order, in the
//                        wave's private 3 KB of LDS; the lanes then switch to *pixel* lanes, each blending the survivors
//                        read back one after the other with wave-uniform ds_read_b128 (LDS broadcast) through a
//                        BRANCH-FREE body.  Dropping is exact: a dropped entry would have been skipped by every
//                        pixel of the box anyway (alpha < 1/255), and skipped entries leave no trace in any output
//                        (contributor numbering is by list position).
//  variant 1  "block"    : one 256-thread workgroup per 16x16 tile, list staged through LDS 256 entries at a time,
//                        every lane evaluates every entry, workgroup-wide "all done" vote.  The classic structure;
//                        A/B baseline (measured 0.408 ms vs 0.151 ms for variant 0 on the 1 M / 1080p frame).
//  (variant 2, rounds 1-4: the record broadcast with v_readlane into SGPRs, 0.291 ms -- removed in round 5)
//  variant 3  as variant 0 in 256-thread workgroups of four independent waves (0.165-0.173 ms).
// The wave kernels are additionally templated on TRACK: the contributor index and final transmittance are produced
// only when the caller will run the backward (inference launches drop two VALU instructions per entry).
//
// Numerics: fp32, FMA contraction allowed, exp through v_exp_f32 (__expf).  Image parity with the oracle is
// <= 1e-5 except at pixels where a hard threshold (alpha<1/255, T<1e-4, power>0) is within rounding noise
// (tests/ use the oracle's "fragile" mask for those).
#include "gsr_internal.h"

namespace {

#ifdef GSR_AB_VARIANTS
#include "render_fwd_block.inc"      // measured-and-rejected variants (tools/ab_variants/): measurement build only
#endif  // GSR_AB_VARIANTS

// ------------------------------------------------------------------------------------------------
// wave per 8x8 pixel block: helpers (exact box test, cross-lane broadcast)
// ------------------------------------------------------------------------------------------------
// Smallest value of q(d) = A dx^2 + 2 B dx dy + C dy^2 over the pixel box [x0,x1]x[y0,y1] for a Gaussian centred
// at (mx,my).  Exact for positive-definite (A,B,C): the minimiser is the centre if it is inside, otherwise it
// lies on an edge facing the centre, where q restricted to the edge is a 1-D parabola with a clamped optimum.
__device__ __forceinline__ float min_q_over_box(float mx, float my, float A, float B, float C, float x0, float x1,
                                                float y0, float y1) {
    const float lx = x0 - mx, hx = x1 - mx, ly = y0 - my, hy = y1 - my;   // box in centre-relative coords
    const bool in_x = (lx <= 0.0f) && (hx >= 0.0f);
    const bool in_y = (ly <= 0.0f) && (hy >= 0.0f);
    float q = 3.0e38f;
    if (in_x && in_y) return 0.0f;
    if (!in_x) {
        const float dx = lx > 0.0f ? lx : hx;                 // facing vertical edge
        const float dy = fminf(hy, fmaxf(ly, -B * dx * __builtin_amdgcn_rcpf(C)));   // clamped optimum along it (tau carries a 0.01 margin: v_rcp_f32's ulp is harmless)
        q = fminf(q, A * dx * dx + 2.0f * B * dx * dy + C * dy * dy);
    }
    if (!in_y) {
        const float dy = ly > 0.0f ? ly : hy;
        const float dx = fminf(hx, fmaxf(lx, -B * dy * __builtin_amdgcn_rcpf(A)));
        q = fminf(q, A * dx * dx + 2.0f * B * dx * dy + C * dy * dy);
    }
    return q;
}


// ------------------------------------------------------------------------------------------------
// variants 0 / 3: wave per 8x8 pixel block with a BRANCH-FREE blend body.
// The blend loop issues VALU and SALU instructions at comparable rates, so exec-mask branches (s_and_saveexec /
// s_cbranch / s_or per skip condition) cost as much as the arithmetic they skip.  Here every surviving entry is
// evaluated by all 64 lanes with the three hard conditions folded into selects (weight 0 when skipped), the conic
// is pre-scaled to log2 units by the Gaussian lanes (power -> one v_exp_f32, no multiply), and tau / 1/depth come
// precomputed from the splat record.  The batch's survivors sit in the wave's private 3 KB of LDS and are re-read
// with wave-uniform ds_read_b128 (broadcast).
// ------------------------------------------------------------------------------------------------
struct PixAcc {
    float T, C0, C1, C2, D;
    uint32_t last;
};

// One (pixel, Gaussian) step of the wave kernels, branch-free.  Instead of a separate "done" flag the lane carries TWO
// transmittances: s.T, the value the outputs need (frozen at termination, composites the background), and Tl, the LIVE
// one that the recurrence uses.  Tl simply follows every valid entry: a terminated lane keeps a value below GSR_T_EPS (it only
// shrinks; a lane outside the image starts at 0), so every later valid entry is classified as the terminator again and
// contributes nothing -- the three conditions need no "not done yet" term and termination costs no select at all ("live" is
// Tl >= GSR_T_EPS in the wave's exit ballots).  Arithmetic on the contributing path: w = alpha * T, T' = T - alpha * T.
// Round 5: replaces the form that zeroed Tl at termination (one select more per step); same bits in every output
// (tests/test_simt_forward_cpu.py ran both forms lane by lane), blend 0.128 -> 0.122 ms together with the compacted walk
// (profiles/r05_ab_candidates.json).
// TRACK: the contributor index is only needed by the backward (n_contrib); inference builds drop it.
#define GSR_FWD_LIVE(Tl_) ((Tl_) >= GSR_T_EPS)
template <bool TRACK>
__device__ __forceinline__ void blend_step_bf(PixAcc& s, float& Tl, float pxf, float pyf, float gx_, float gy_, float a2,
                                              float b2, float c2, float op, float r, float g, float b, float invd,
                                              uint32_t pos) {
    const float dx = gx_ - pxf, dy = gy_ - pyf;
    const float t = fmaf(b2, dy, a2 * dx);
    const float p2 = fmaf(dx, t, (c2 * dy) * dy);           // log2(e) * power
    const float alpha = fminf(GSR_ALPHA_MAX, op * __builtin_amdgcn_exp2f(p2));
    const bool valid = (p2 <= 0.0f) & (alpha >= GSR_ALPHA_MIN);
    const float testT = fmaf(-alpha, Tl, Tl);                // T (1 - alpha)
    const bool term = valid & (testT < GSR_T_EPS);
    const bool contrib = valid & (!term);
    const float w = contrib ? alpha * Tl : 0.0f;
    s.C0 = fmaf(r, w, s.C0);
    s.C1 = fmaf(g, w, s.C1);
    s.C2 = fmaf(b, w, s.C2);
    s.D = fmaf(invd, w, s.D);
    s.T = contrib ? testT : s.T;
    Tl = valid ? testT : Tl;
    if (TRACK) s.last = contrib ? pos : s.last;
}

// WPB = waves per workgroup: 1 -> one 64-thread workgroup per 8x8 block (workgroup ids arranged so that the four blocks
// of a tile land on one XCD); 4 -> one 256-thread workgroup per tile whose four waves run independently (no barriers),
// which lifts the resident-wave count when the per-CU workgroup limit, not registers/LDS, caps occupancy.
template <bool USE_LDS, int WPB, bool TRACK>
__global__ void __launch_bounds__(64 * WPB)
render_fwd_wave_bf(GsrCamDev cam, int tile_off, int n_band_tiles /*tiles [tile_off, n_band_tiles) of the band*/, const uint2* __restrict__ ranges,
                   const uint32_t* __restrict__ point_list, const float4* __restrict__ splats,
                   float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ block_steps,
                   float* __restrict__ out_color, float* __restrict__ out_invdepth,
                   unsigned long long* __restrict__ counters /*NULL unless profiling*/) {
    __shared__ float4 s_rec_all[USE_LDS ? WPB * 64 * 3 : 1];
    float4* s_rec = s_rec_all + (USE_LDS ? (threadIdx.x >> 6) * 64 * 3 : 0);
    int tile_local, quad;
    if (WPB == 1) {
        // XCD-aware mapping: workgroup b runs on XCD b % 8 (observed).  The four 8x8 blocks of a tile share one splat
        // list, so they get ids b, b+8, b+16, b+24 -> same XCD -> same L2.
        const int b = blockIdx.x;
        const int grp = b >> 5, r32 = b & 31;
        tile_local = tile_off + grp * 8 + (r32 & 7);
        quad = r32 >> 3;
    } else {
        tile_local = tile_off + blockIdx.x;
        quad = threadIdx.x >> 6;
    }
    if (tile_local >= n_band_tiles) return;
    const unsigned long long t_start = counters ? wall_clock64() : 0ull;      // (measurement only: per-wave trace, gsr_profile_trace)
    const bool tracing = counters && gsr_trace_mode(counters);
    unsigned long long t_mark = t_start, t_walk = 0ull, t_prep = 0ull;
    const int tile = cam.tile_y0 * cam.gx + tile_local;
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const int lane = threadIdx.x & 63;
    const int bx0 = tx * GSR_TILE + (quad & 1) * 8, by0 = ty * GSR_TILE + (quad >> 1) * 8;
    if (bx0 >= cam.W || by0 >= cam.H) {
        if (TRACK && lane == 0) block_steps[tile * 4 + quad] = 0u;
        return;
    }
    const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
    const bool inside = px < cam.W && py < cam.H;
    const float pxf = (float)px, pyf = (float)py;
    const float x0 = (float)bx0, x1 = (float)min(bx0 + 7, cam.W - 1);
    const float y0 = (float)by0, y1 = (float)min(by0 + 7, cam.H - 1);
    const uint2 range = ranges[tile];
    PixAcc s = {1.0f, 0.f, 0.f, 0.f, 0.f, 0u};
    float Tl = inside ? 1.0f : 0.0f;          // live transmittance (0 = this lane takes no further entries)
    constexpr float LOG2E = 1.4426950408889634f;

    // Software pipeline over the batches of 64 list entries: every batch needs two dependent global loads (list id ->
    // 64-byte record gather), ~1-2 us under load, while a wave walks only a handful of batches before its pixels
    // terminate -- un-pipelined the kernel was latency-bound on that chain.  Ids are fetched two batches ahead and
    // records one batch ahead; the survivor loop in between touches only LDS, so the loads stay in flight across it.
    uint32_t nsteps = 0, nbatches = 0;      // wave-uniform work counters (SGPRs), reported only while profiling
    auto load_id = [&](uint32_t b) -> uint32_t { return (b + lane < range.y) ? point_list[b + lane] : 0xFFFFFFFFu; };
    uint32_t id_n1 = load_id(range.x);                    // ids of the batch whose records are fetched next
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
    if (id_n1 != 0xFFFFFFFFu) { n0 = splats[id_n1 * 4 + 0]; n1 = splats[id_n1 * 4 + 1]; n2 = splats[id_n1 * 4 + 2]; }
    id_n1 = load_id(range.x + 64);
    for (uint32_t base = range.x; base < range.y; base += 64) {
        const uint32_t n = min(64u, range.y - base);
        float4 q0 = n0, q1 = n1;
        const float4 q2 = n2;
        // issue the next batch's record gather and the id fetch of the batch after it
        if (id_n1 != 0xFFFFFFFFu) { n0 = splats[id_n1 * 4 + 0]; n1 = splats[id_n1 * 4 + 1]; n2 = splats[id_n1 * 4 + 2]; }
        id_n1 = load_id(base + 128);
        float colb = 0.f, invd = 0.f;
        bool keep = false;
        if ((uint32_t)lane < n) {
            colb = q2.x;
            invd = q2.w;
            const float qmin = min_q_over_box(q0.x, q0.y, q0.z, q0.w, q1.x, x0, x1, y0, y1);
            keep = !(qmin > q2.z);                 // q2.z = 2 ln(255 opacity) + 0.01, written by the preprocess
            q0.z *= -0.5f * LOG2E;                  // conic -> log2 units, sign folded in
            q0.w *= -LOG2E;
            q1.x *= -0.5f * LOG2E;
        }
        // The survivors are parked COMPACTED, in list order (slot = survivors on lower lanes), with their list position in the record's spare
        // word.  The walk reads consecutive records -- constant ds_read offsets from one base register inside the eight-deep unrolled body, a
        // counter instead of popping a 64-bit mask (round 4's form: s_ff1 / s_mul / v_mov per step): 22 VALU and 4 SALU per step in the
        // inference build (round 4: 24 and 11; tools/isa_audit.py forward_walk_step).
        // All 64 pixels may terminate in the middle of a batch; the wave would then blend the batch's remaining survivors into
        // nothing (half a batch per wave on average: ~15 % of the steps on the bench frame).  The walk is unrolled eight deep and the
        // termination ballot is taken once per eight steps: measured (round 3) 0.139 ms without the check, 0.1295 with one per four steps,
        // 0.127 per eight, 0.142 / 0.143 per twelve / sixteen (the unrolled body outgrows what the scheduler handles well).
        static_assert(USE_LDS, "the compacted walk reads the batch from LDS");
        const uint64_t mask = __ballot(keep);
        if (keep) {
            const int slot = (int)__popcll(mask & ((1ull << lane) - 1ull));
            s_rec[slot * 3 + 0] = q0;
            s_rec[slot * 3 + 1] = q1;
            s_rec[slot * 3 + 2] = make_float4(colb, invd, __uint_as_float((uint32_t)lane), 0.f);
        }
        __builtin_amdgcn_wave_barrier();      // (no instruction: the wave's LDS accesses stay in program order; the lanes read each other's records)
        uint32_t left = (uint32_t)__popcll(mask);
        ++nbatches;
        if (tracing) { const unsigned long long t = wall_clock64(); t_prep += t - t_mark; t_mark = t; }
        const uint32_t pos_base = base - range.x + 1;
        const float4* rp = s_rec;
        auto step_at = [&](int u) {
            const float4 r0 = rp[u * 3 + 0];
            const float4 r1 = rp[u * 3 + 1];
            if (TRACK) {
                const float4 r2 = rp[u * 3 + 2];
                blend_step_bf<TRACK>(s, Tl, pxf, pyf, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, pos_base + __float_as_uint(r2.z));
            } else {
                const float2 r2 = *reinterpret_cast<const float2*>(&rp[u * 3 + 2]);
                blend_step_bf<TRACK>(s, Tl, pxf, pyf, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, 0u);
            }
        };
#ifndef GSR_FWD_CHECK_EVERY
#define GSR_FWD_CHECK_EVERY 8
#endif
        while (left) {
            const uint32_t todo = min(left, (uint32_t)GSR_FWD_CHECK_EVERY);
            step_at(0);
#pragma unroll
            for (int u = 1; u < GSR_FWD_CHECK_EVERY; ++u) {
                if ((uint32_t)u >= todo) break;
                step_at(u);
            }
            nsteps += todo;
            left -= todo;
            rp += 3 * GSR_FWD_CHECK_EVERY;
            if (__ballot(GSR_FWD_LIVE(Tl)) == 0ull) break;
        }
        if (tracing) { const unsigned long long t = wall_clock64(); t_walk += t - t_mark; t_mark = t; }
        if (__ballot(GSR_FWD_LIVE(Tl)) == 0ull) break;
    }
    if (tracing) {
        if (lane == 0) gsr_trace_wave(counters, t_start, (uint32_t)(blockIdx.x * WPB + (threadIdx.x >> 6)), 1u, gsr_trace_pack(nsteps, t_walk, t_prep, 0ull));
    } else if (counters && lane == 0) {     // [0] (8x8 block, entry) pairs blended by all 64 lanes, [1] batches of 64 entries box-tested
        atomicAdd(counters + 0, (unsigned long long)nsteps);
        atomicAdd(counters + 1, (unsigned long long)nbatches);
        atomicMax(counters + 4, (unsigned long long)nsteps);      // the heaviest wave (tail of the launch: max / mean)
    }
    // the entries this block blended until its last pixel terminated = the steps its quadrant costs the blend backward,
    // which walks the same survivors back from the same point: the backward's plan kernel orders the tiles by it
    if (TRACK && lane == 0) block_steps[tile * 4 + quad] = nsteps;
    if (inside) {
        const int64_t pix = (int64_t)py * cam.W + px;
        const int64_t HW = (int64_t)cam.H * cam.W;
        if (TRACK) {        // inference builds (final_T == NULL) do not track: only the backward reads these
            final_T[pix] = s.T;
            n_contrib[pix] = s.last;
        }
        out_color[pix] = s.C0 + s.T * cam.bg[0];
        out_color[HW + pix] = s.C1 + s.T * cam.bg[1];
        out_color[2 * HW + pix] = s.C2 + s.T * cam.bg[2];
        if (out_invdepth) out_invdepth[pix] = s.D;
    }
}

}  // namespace

// measurement build's option render_fwd_lds_pad: bytes of dynamic LDS added to every workgroup of the wave kernels -- caps the resident waves per CU
// (3 KB static per wave: 8 per SIMD fit; 10 KB -> 4 per SIMD).  Measured (profiles/r05_ab_fwd_bands_occupancy.json): blend 0.122 ms at 8 waves per SIMD,
// 0.133 / 0.149 / 0.183 at fewer -- shorter-lived waves do not shrink the launch's drain.
#ifdef GSR_AB_VARIANTS
int g_render_fwd_lds_pad = 0;
void gsr_set_render_fwd_lds_pad(int bytes) { g_render_fwd_lds_pad = bytes < 0 ? 0 : bytes; }
#else
constexpr int g_render_fwd_lds_pad = 0;
#endif

int gsr_render_forward_variant_available(int variant) {
#ifdef GSR_AB_VARIANTS
    return variant == 0 || variant == 1 || variant == 3;
#else
    return variant == 0;
#endif
}

void gsr_launch_render_forward(const GsrCamDev& cam, const uint2* ranges, const uint32_t* point_list,
                               const float4* splats, float* final_T, uint32_t* n_contrib, uint32_t* block_steps,
                               float* out_color, float* out_invdepth, int variant, unsigned long long* counters, hipStream_t st,
                               int tile_off, int tile_cnt) {
    int n_band_tiles = cam.gx * (cam.tile_y1 - cam.tile_y0);
    if (tile_off < 0 || tile_cnt < 0) { tile_off = 0; tile_cnt = n_band_tiles; }      // the whole band in one launch
    if (tile_off + tile_cnt < n_band_tiles) n_band_tiles = tile_off + tile_cnt;
    const int n_launch = n_band_tiles - tile_off;
    if (n_launch <= 0) return;
    const int groups = (n_launch + 7) / 8;
    const bool track = final_T != nullptr && n_contrib != nullptr && block_steps != nullptr;
#define GSR_LAUNCH_BF(USE_LDS_, WPB_, GRID_, BLOCK_)                                                                              \
    do {                                                                                                                          \
        if (track)                                                                                                                \
            hipLaunchKernelGGL((render_fwd_wave_bf<USE_LDS_, WPB_, true>), dim3(GRID_), dim3(BLOCK_), g_render_fwd_lds_pad, st, cam, tile_off, n_band_tiles,    \
                               ranges, point_list, splats, final_T, n_contrib, block_steps, out_color, out_invdepth, counters);  \
        else                                                                                                                      \
            hipLaunchKernelGGL((render_fwd_wave_bf<USE_LDS_, WPB_, false>), dim3(GRID_), dim3(BLOCK_), g_render_fwd_lds_pad, st, cam, tile_off, n_band_tiles,   \
                               ranges, point_list, splats, final_T, n_contrib, block_steps, out_color, out_invdepth, counters);  \
    } while (0)
#ifdef GSR_AB_VARIANTS
    if (variant == 1) {
        if (track) (void)hipMemsetAsync(block_steps + (size_t)cam.tile_y0 * cam.gx * 4, 0, (size_t)n_band_tiles * 16, st);   // no estimate: any order
        hipLaunchKernelGGL(render_fwd_block, dim3(n_band_tiles), dim3(256), 0, st, cam, ranges, point_list, splats,
                           final_T, n_contrib, out_color, out_invdepth);
        return;
    }
    if (variant == 3) { GSR_LAUNCH_BF(true, 4, n_launch, 256); return; }
#endif
    (void)variant;
    GSR_LAUNCH_BF(true, 1, groups * 32, 64);
#undef GSR_LAUNCH_BF
}
