// The default FORWARD (and, on request, BACKWARD) of the operator from the product's kernel source, on the CPU: per-Gaussian kernel (csrc/preprocess.hip, with its frame
// statistics), bucket depth sort (depthsort.hip), fused emission + two-level tile sort (tilesort.hip), blend (render_fwd.hip) -- eleven launches --, then the blend backward (render_bwd.hip: plan, walk, the three reduce kernels) and the fused
// per-Gaussian backward (preprocess.hip), in
// the order of gsr_rasterize_forward / bin_and_render (csrc/gsr_api.cpp), every one through the product's own launcher, every lane a fiber of the
// SIMT shim (tests/simt/).  Built with g++ -ffp-contract=off (the flag preprocess.hip ships with; the blend then runs without FMA contraction and
// with libm's exp2f instead of v_exp_f32: the image is compared within the parity suite's tolerance, the integers bit for bit).
// TEST INFRASTRUCTURE (tests/test_simt_forward_cpu.py): a checker of the kernel source, never part of libgsr_hip.so.
#define __HIPCC__ 1
#include "hip/hip_runtime.h"
#include "preprocess.hip"
#include "depthsort.hip"
#include "tilesort.hip"
#include "sort.hip"
#include "route.hip"
#include "render_fwd.hip"
namespace tu_bwd {      // (render_fwd.hip and render_bwd.hip both define min_q_over_box / bcast in their anonymous namespaces)
#include "render_bwd.hip"
}  // namespace tu_bwd
#include "simt_runtime.h"
#include <vector>

static char g_err[256];

extern "C" {

const char* simt_fwd_last_error(void) { return g_err; }

// the reference's separate_sh call form (GsrRasterSettings.sh_dc / dL_dsh_dc): coefficient 0 as [P,1,3]; `shs` / dL_dsh of the next simt_forward
// call then hold coefficients 1..15 as [P,15,3] (M stays 16).  NULL switches back to the fused [P,16,3] tensor.
static const float* g_sh_dc = nullptr;
static float* g_dL_dsh_dc = nullptr;
void simt_set_split_sh(const float* dc, float* dL_ddc) { g_sh_dc = dc; g_dL_dsh_dc = dL_ddc; }

// settings as GsrRasterSettings (include/gsr.h) with HOST pointers; fused [P,16,3] SH tensor or colors_precomp; scales + rotations.
// Outputs: radii[P], tiles[P], out_color[3HW], out_invdepth[HW]; point_list (capacity r_cap), ranges[gx*gy]; track != 0: final_T[HW], n_contrib[HW].
// Returns R (>= 0) or -1.
int64_t simt_forward(const GsrRasterSettings* s, int snug, int P, int M, const float* means3D, const float* shs, const float* colors_precomp,
                     const float* opacities, const float* scales, const float* rotations, int32_t* radii, uint32_t* tiles_out, float* out_color,
                     float* out_invdepth, uint32_t* point_list, int64_t r_cap, uint2* ranges, int track, float* final_T, uint32_t* n_contrib,
                     // backward (dL_dcolor != NULL, track != 0): blend backward (plan + walk), reduce, per-Gaussian backward -- the launches of
                     // gsr_backward_blend / gsr_backward_preprocess.  dL_dinvdepth may be NULL (no depth supervision).  dL_dsh: [P,M,3] or, with colors_precomp, [P,3]
                     const float* dL_dcolor, const float* dL_dinvdepth, float* dL_dmeans2D, float* dL_dmeans3D, float* dL_dopacity, float* dL_dsh,
                     float* dL_dscales, float* dL_drotations) {
    GsrCamDev c;
    c.W = s->image_width; c.H = s->image_height;
    c.gx = (c.W + GSR_TILE - 1) / GSR_TILE; c.gy = (c.H + GSR_TILE - 1) / GSR_TILE;
    c.focal_x = (float)c.W / (2.0f * s->tanfovx); c.focal_y = (float)c.H / (2.0f * s->tanfovy);      // (make_cam, csrc/gsr_api.cpp)
    c.limx = 1.3f * s->tanfovx; c.limy = 1.3f * s->tanfovy;
    c.scale_modifier = s->scale_modifier; c.sh_degree = s->sh_degree; c.M = M; c.antialiasing = s->antialiasing ? 1 : 0; c.snug = snug;
    c.tile_y0 = 0; c.tile_y1 = c.gy;
    c.view = s->viewmatrix; c.proj = s->projmatrix; c.campos = s->campos; c.bg = s->bg; c.sh_dc = g_sh_dc; c.dL_dsh_dc = g_dL_dsh_dc;
    const int n_tiles = c.gx * c.gy;
    const size_t n = (size_t)P + 64;
    std::vector<float4> splats(4 * n);
    std::vector<uint2> rect(n), rect_sorted(n), wg_range(GSR_FRAME_MAX_GROUPS + 1), pairs0(n), pairs1(n);
    std::vector<uint32_t> tiles(n), keys0(n), keys1(n), vals0(n), vals1(n), offsets(n), frame(64, 0u), state(16, 0u);
    const size_t nblocks = gsr_depth_bucket_blocks(P), nseg = gsr_depth_bucket_segments(P);
    std::vector<uint32_t> cnt_tab(nblocks * GSR_DS_BUCKETS), tile_tab(nblocks * GSR_DS_BUCKETS), cnt_total(GSR_DS_BUCKETS), tile_total(GSR_DS_BUCKETS), plan(nseg * GSR_DS_PLAN_WORDS + 16);
    GsrGeom g{};
    g.splats = splats.data(); g.rect = rect.data(); g.tiles = tiles.data(); g.clamped = nullptr;
    g.keys[0] = keys0.data(); g.keys[1] = keys1.data(); g.vals[0] = vals0.data(); g.vals[1] = vals1.data();
    g.rect_sorted = rect_sorted.data(); g.offsets = offsets.data(); g.num_rendered = frame.data(); g.wg_range = wg_range.data();
    g.ds.pairs[0] = pairs0.data(); g.ds.pairs[1] = pairs1.data(); g.ds.cnt_tab = cnt_tab.data(); g.ds.tile_tab = tile_tab.data();
    g.ds.cnt_total = cnt_total.data(); g.ds.tile_total = tile_total.data(); g.ds.plan = plan.data();
    std::vector<uint32_t> eq_tab(GSR_EQ_TAB_WORDS);
    std::vector<uint16_t> bucket_of(n + 64);
    g.ds.eq_tab = eq_tab.data(); g.ds.bucket_of = bucket_of.data();
    GsrFrameStatsDev fs;
    fs.state = state.data(); fs.frame = frame.data(); fs.wg_range = wg_range.data(); fs.host_word = nullptr; fs.seq = 1;
    auto bail = [&]() -> int64_t { snprintf(g_err, sizeof(g_err), "%s", simt::launch_error ? simt::launch_error : "?"); simt::launch_error = nullptr; return -1; };
    const int n_range = gsr_launch_preprocess(c, P, means3D, shs, colors_precomp, opacities, scales, rotations, nullptr, g, radii, fs, nullptr);
    if (simt::launch_error) return bail();
    const uint64_t R64 = ((uint64_t)frame[1] << 32) | frame[0];
    if (state[0] != 0u || state[1] != 0u) { snprintf(g_err, sizeof(g_err), "the frame-statistics counter was not reset by the last workgroup"); return -1; }
    if ((int64_t)R64 > r_cap) { snprintf(g_err, sizeof(g_err), "R = %llu exceeds the caller's capacity", (unsigned long long)R64); return -1; }
    for (int i = 0; i < P; ++i) tiles_out[i] = tiles[i];
    const uint32_t R = (uint32_t)R64;
    const uint32_t bf_cap = (uint32_t)gsr_block_first_cap(P);
    const int64_t nblk = ((int64_t)R + GSR_TS_ITEMS - 1) / GSR_TS_ITEMS;
    std::vector<uint2> block_first(std::max<size_t>(bf_cap, (size_t)nblk + 2));
    gsr_launch_depth_bucket_sort(P, g.keys[0], g.tiles, g.rect, frame.data(), wg_range.data(), n_range, g.ds, g.vals[1], g.rect_sorted, g.offsets,
                                 block_first.data(), bf_cap, nullptr, nullptr);
    if (simt::launch_error) return bail();
    std::vector<uint64_t> words((size_t)R + 16);
    if (R > 0) {
        GsrTileSortPlan tp;
        gsr_tile_sort_plan(n_tiles, P, &tp);
        if ((uint64_t)nblk + 1 > (uint64_t)bf_cap) gsr_launch_fill_block_first(P, g.offsets, block_first.data(), (uint32_t)(nblk + 2), nullptr);
        std::vector<uint32_t> hist1((size_t)256 * (nblk + 1)), digit_total(256), bucket_base(257), blk2_start(257), hist2((size_t)(nblk + 512) * 256), tile_base(65536);
        gsr_launch_tile_sort_level1(tp, R, c.gx, block_first.data(), g.offsets, g.rect_sorted, g.vals[1], words.data(), hist1.data(), digit_total.data(),
                                    bucket_base.data(), blk2_start.data(), track ? g.splats : nullptr, nullptr);
        if (simt::launch_error) return bail();
        gsr_launch_tile_sort_level2(tp, R, n_tiles, words.data(), point_list, bucket_base.data(), blk2_start.data(), hist2.data(), tile_base.data(), ranges, nullptr);
        if (simt::launch_error) return bail();
    } else {
        for (int t = 0; t < n_tiles; ++t) ranges[t] = make_uint2(0u, 0u);
    }
    std::vector<uint32_t> block_steps((size_t)n_tiles * 4 + 16);
    gsr_launch_render_forward(c, ranges, point_list, g.splats, track ? final_T : nullptr, track ? n_contrib : nullptr, track ? block_steps.data() : nullptr,
                              out_color, out_invdepth, 0, nullptr, nullptr);
    if (simt::launch_error) return bail();
    if (dL_dcolor && track) {
        const size_t nr = (size_t)(R > 0 ? R : 1), nu = tu_bwd::gsr_reduce_units((int64_t)nr);
        std::vector<float> splat_grads((size_t)P * 12 + 16, 0.f), inst_grads(nr * 12 * GSR_BWD_SLOTS + 16), unit_piece(nu * 24 + 32);
        std::vector<uint32_t> inst_flag(nr + 16), tile_order((size_t)n_tiles * 2 + 16);
        std::vector<uint2> unit_first(nu + 16);
        if (R > 0) {
            tu_bwd::gsr_launch_render_backward(c, ranges, point_list, g.splats, final_T, n_contrib, block_steps.data(), tile_order.data(), dL_dcolor, dL_dinvdepth,
                                               nullptr, inst_grads.data(), inst_flag.data(), (int64_t)R, 0, 2, nullptr, nullptr);
            if (simt::launch_error) return bail();
            tu_bwd::gsr_launch_reduce_instances(P, (int64_t)R, g.vals[1], g.offsets, g.splats, inst_grads.data(), inst_flag.data(), splat_grads.data(),
                                                unit_first.data(), unit_piece.data(), nullptr);
            if (simt::launch_error) return bail();
        }
        gsr_launch_preprocess_backward(c, P, means3D, shs, colors_precomp, opacities, scales, rotations, nullptr, radii, g, splat_grads.data(), dL_dmeans2D,
                                       colors_precomp ? dL_dsh : nullptr, dL_dopacity, dL_dmeans3D, nullptr, shs ? dL_dsh : nullptr, dL_dscales, dL_drotations, nullptr);
        if (simt::launch_error) return bail();
    }
    return (int64_t)R;
}

// ---- mode C of the multi-GPU renderer (parallel.py: Gaussians AND tile rows sharded), every rank's kernels run here one after the other ----
// G shards of contiguous Gaussians = G bands of tile rows [bounds[b], bounds[b+1]).  Per shard: per-Gaussian kernel with full-frame rectangles and
// no frame statistics (gsr_preprocess_forward), route_count + scan, route_pack (48-byte records, stable, grouped by band).  Per band: the records of
// all shards in rank order (what the all-to-all delivers), ingest_packed (+ frame statistics), depth sort, tile sort, blend of the band's rows.
// counts_out[g * G + b] = records shard g sends to band b.  Returns the sum of the bands' R or -1.
int64_t simt_forward_sharded(const GsrRasterSettings* s, int snug, int G, const int32_t* bounds, int P, int M, const float* means3D, const float* shs,
                             const float* opacities, const float* scales, const float* rotations, int32_t* radii, float* out_color, float* out_invdepth,
                             uint32_t* counts_out) {
    GsrCamDev c;
    c.W = s->image_width; c.H = s->image_height;
    c.gx = (c.W + GSR_TILE - 1) / GSR_TILE; c.gy = (c.H + GSR_TILE - 1) / GSR_TILE;
    c.focal_x = (float)c.W / (2.0f * s->tanfovx); c.focal_y = (float)c.H / (2.0f * s->tanfovy);
    c.limx = 1.3f * s->tanfovx; c.limy = 1.3f * s->tanfovy;
    c.scale_modifier = s->scale_modifier; c.sh_degree = s->sh_degree; c.M = M; c.antialiasing = s->antialiasing ? 1 : 0; c.snug = snug;
    c.view = s->viewmatrix; c.proj = s->projmatrix; c.campos = s->campos; c.bg = s->bg; c.sh_dc = nullptr; c.dL_dsh_dc = nullptr;
    const int n_tiles = c.gx * c.gy;
    auto bail = [&]() -> int64_t { snprintf(g_err, sizeof(g_err), "%s", simt::launch_error ? simt::launch_error : "?"); simt::launch_error = nullptr; return -1; };
    // ---- every shard: project, count, pack ----
    std::vector<std::vector<float>> packed(G);
    std::vector<std::vector<int64_t>> offs(G);
    for (int gi = 0; gi < G; ++gi) {
        const int lo = (int)((int64_t)P * gi / G), hi = (int)((int64_t)P * (gi + 1) / G), Pg = hi - lo;
        offs[gi].assign(G + 1, 0);
        if (Pg == 0) continue;
        const size_t n = (size_t)Pg + 64;
        std::vector<float4> records(4 * n);
        std::vector<uint2> rect(n);
        std::vector<uint32_t> tiles(n), keys0(n), vals0(n);
        GsrGeom g{};
        g.splats = records.data(); g.rect = rect.data(); g.tiles = tiles.data(); g.keys[0] = keys0.data(); g.vals[0] = vals0.data();
        GsrFrameStatsDev none;
        none.state = nullptr; none.frame = nullptr; none.wg_range = nullptr; none.host_word = nullptr; none.seq = 0;
        GsrCamDev cs = c;
        cs.tile_y0 = 0; cs.tile_y1 = c.gy;      // the records leave the rank: rectangles of the FULL frame
        gsr_launch_preprocess(cs, Pg, means3D + (size_t)lo * 3, shs + (size_t)lo * M * 3, nullptr, opacities + lo, scales + (size_t)lo * 3, rotations + (size_t)lo * 4,
                              nullptr, g, radii + lo, none, nullptr);
        if (simt::launch_error) return bail();
        std::vector<char> scratch(gsr_route_scratch_bytes_impl(Pg, G) + 256);
        std::vector<uint32_t> band_counts(GSR_MAX_BANDS, 0u);
        gsr_launch_route_count(Pg, reinterpret_cast<const float*>(records.data()), G, bounds, reinterpret_cast<uint32_t*>(scratch.data()), band_counts.data(), nullptr);
        if (simt::launch_error) return bail();
        for (int b = 0; b < G; ++b) { counts_out[gi * G + b] = band_counts[b]; offs[gi][b + 1] = offs[gi][b] + band_counts[b]; }
        packed[gi].assign((size_t)offs[gi][G] * 12 + 16, 0.f);
        std::vector<int32_t> send_ids((size_t)offs[gi][G] + 16);
        if (offs[gi][G] > 0)
            gsr_launch_route_pack(Pg, reinterpret_cast<const float*>(records.data()), G, bounds, offs[gi].data(), reinterpret_cast<const uint32_t*>(scratch.data()),
                                  packed[gi].data(), send_ids.data(), 0xFFFFFFFFu, nullptr, nullptr);
        if (simt::launch_error) return bail();
        for (int b = 0; b < G; ++b)      // stable pack: shard-local indices ascend inside every band segment
            for (int64_t r = offs[gi][b]; r < offs[gi][b + 1]; ++r)
                if (send_ids[r] < 0 || send_ids[r] >= Pg || (r > offs[gi][b] && send_ids[r] <= send_ids[r - 1])) {
                    snprintf(g_err, sizeof(g_err), "shard %d band %d: send ids are not ascending (row %lld)", gi, b, (long long)r);
                    return -1;
                }
    }
    // ---- every band: ingest what arrived, bin, blend its rows ----
    for (size_t i = 0; i < (size_t)c.W * c.H * 3; ++i) out_color[i] = 0.f;
    for (size_t i = 0; i < (size_t)c.W * c.H; ++i) out_invdepth[i] = 0.f;
    int64_t R_total = 0;
    for (int b = 0; b < G; ++b) {
        GsrCamDev cb = c;
        cb.tile_y0 = bounds[b]; cb.tile_y1 = bounds[b + 1];
        if (cb.tile_y1 <= cb.tile_y0) continue;
        int64_t Pr = 0;
        for (int gi = 0; gi < G; ++gi) Pr += offs[gi][b + 1] - offs[gi][b];
        std::vector<uint2> ranges((size_t)n_tiles + 16, make_uint2(0u, 0u));
        std::vector<uint32_t> point_list(16);
        std::vector<float4> splats(16);
        if (Pr > 0) {
            std::vector<float> recv((size_t)Pr * 12 + 16);
            int64_t at = 0;
            for (int gi = 0; gi < G; ++gi) {
                const int64_t cnt = offs[gi][b + 1] - offs[gi][b];
                if (cnt) memcpy(recv.data() + at * 12, packed[gi].data() + offs[gi][b] * 12, (size_t)cnt * 48);
                at += cnt;
            }
            const int Pb = (int)Pr;
            const size_t n = (size_t)Pb + 64;
            splats.assign(4 * n, make_float4(0.f, 0.f, 0.f, 0.f));
            std::vector<uint2> rect(n), rect_sorted(n), wg_range(GSR_FRAME_MAX_GROUPS + 1), pairs0(n), pairs1(n);
            std::vector<uint32_t> tiles(n), keys0(n), keys1(n), vals0(n), vals1(n), offsets(n), frame(64, 0u), state(16, 0u);
            const size_t nblocks = gsr_depth_bucket_blocks(Pb), nseg = gsr_depth_bucket_segments(Pb);
            std::vector<uint32_t> cnt_tab(nblocks * GSR_DS_BUCKETS), tile_tab(nblocks * GSR_DS_BUCKETS), cnt_total(GSR_DS_BUCKETS), tile_total(GSR_DS_BUCKETS), plan(nseg * GSR_DS_PLAN_WORDS + 16);
            GsrDepthSortBufs ds;
            ds.pairs[0] = pairs0.data(); ds.pairs[1] = pairs1.data(); ds.cnt_tab = cnt_tab.data(); ds.tile_tab = tile_tab.data();
            ds.cnt_total = cnt_total.data(); ds.tile_total = tile_total.data(); ds.plan = plan.data();
            std::vector<uint32_t> eq_tab(GSR_EQ_TAB_WORDS);
            std::vector<uint16_t> bucket_of(n + 64);
            ds.eq_tab = eq_tab.data(); ds.bucket_of = bucket_of.data();
            GsrFrameStatsDev fs;
            fs.state = state.data(); fs.frame = frame.data(); fs.wg_range = wg_range.data(); fs.host_word = nullptr; fs.seq = 1;
            const int n_range = gsr_launch_ingest_packed(Pb, recv.data(), cb.tile_y0, cb.tile_y1, splats.data(), rect.data(), tiles.data(), keys0.data(), vals0.data(), fs, 0, nullptr);
            if (simt::launch_error) return bail();
            const uint32_t R = frame[0];
            R_total += R;
            const uint32_t bf_cap = (uint32_t)gsr_block_first_cap(Pb);
            const int64_t nblk = ((int64_t)R + GSR_TS_ITEMS - 1) / GSR_TS_ITEMS;
            std::vector<uint2> block_first(std::max<size_t>(bf_cap, (size_t)nblk + 2));
            gsr_launch_depth_bucket_sort(Pb, keys0.data(), tiles.data(), rect.data(), frame.data(), wg_range.data(), n_range, ds, vals1.data(), rect_sorted.data(), offsets.data(),
                                         block_first.data(), bf_cap, nullptr, nullptr);
            if (simt::launch_error) return bail();
            if (R > 0) {
                GsrTileSortPlan tp;
                gsr_tile_sort_plan(n_tiles, Pb, &tp);
                if ((uint64_t)nblk + 1 > (uint64_t)bf_cap) gsr_launch_fill_block_first(Pb, offsets.data(), block_first.data(), (uint32_t)(nblk + 2), nullptr);
                std::vector<uint64_t> words((size_t)R + 16);
                std::vector<uint32_t> hist1((size_t)256 * (nblk + 1)), digit_total(256), bucket_base(257), blk2_start(257), hist2((size_t)(nblk + 512) * 256), tile_base(65536);
                point_list.assign((size_t)R + 16, 0u);
                gsr_launch_tile_sort_level1(tp, R, c.gx, block_first.data(), offsets.data(), rect_sorted.data(), vals1.data(), words.data(), hist1.data(), digit_total.data(),
                                            bucket_base.data(), blk2_start.data(), nullptr, nullptr);
                if (simt::launch_error) return bail();
                gsr_launch_tile_sort_level2(tp, R, n_tiles, words.data(), point_list.data(), bucket_base.data(), blk2_start.data(), hist2.data(), tile_base.data(), ranges.data(), nullptr);
                if (simt::launch_error) return bail();
            }
        }
        // (a band that received nothing still blends: empty ranges -> the background)
        gsr_launch_render_forward(cb, ranges.data(), point_list.data(), splats.data(), nullptr, nullptr, nullptr, out_color, out_invdepth, 0, nullptr, nullptr);
        if (simt::launch_error) return bail();
    }
    return R_total;
}

}  // extern "C"
