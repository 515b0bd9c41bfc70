// The default forward's whole BINNING CHAIN after the per-Gaussian kernel -- the bucket depth sort (csrc/depthsort.hip: 4 launches) and the fused
// emission + two-level tile sort (csrc/tilesort.hip: 5-6 launches) -- compiled for the host through the SIMT-on-CPU shim and run through the
// product's own launchers, in the order of bin_and_render (csrc/gsr_api.cpp).  TEST INFRASTRUCTURE (tests/test_simt_chain_cpu.py).
#include "hip/hip_runtime.h"
#include "depthsort.hip"
#include "tilesort.hip"
#include "simt_runtime.h"
#include <vector>

void gsr_launch_rs_scan(uint32_t* block_hist, int nblocks, int ndigits, uint32_t* digit_total, hipStream_t) {      // sort.hip's rs_scan, restated
    for (int d = 0; d < ndigits; ++d) {
        uint32_t run = 0;
        for (int b = 0; b < nblocks; ++b) {
            const uint32_t c = block_hist[(size_t)d * nblocks + b];
            block_hist[(size_t)d * nblocks + b] = run;
            run += c;
        }
        digit_total[d] = run;
    }
}

static char g_err[256];

extern "C" {

const char* simt_chain_last_error(void) { return g_err; }

// keys / tiles / rect of P Gaussians (what the preprocess leaves) -> depth order, sorted point list [R], tile ranges [n_tiles]; returns R or -1
int64_t simt_bin(int P, int gx, int gy, const uint32_t* keys, const uint32_t* tiles, const uint2* rect, const uint2* wg_range, int n_range, uint32_t R,
                 uint32_t* order, uint32_t* point_list, uint2* ranges) {
    const int n_tiles = gx * gy;
    const size_t nblocks = gsr_depth_bucket_blocks(P), nseg = gsr_depth_bucket_segments(P);
    std::vector<uint2> pairs0((size_t)P + 16), pairs1((size_t)P + 16), rect_sorted((size_t)P + 16);
    std::vector<uint32_t> cnt_tab(nblocks * GSR_DS_BUCKETS), tile_tab(nblocks * GSR_DS_BUCKETS), cnt_total(GSR_DS_BUCKETS), tile_total(GSR_DS_BUCKETS), plan(nseg * GSR_DS_PLAN_WORDS + 16),
        offsets((size_t)P + 16), frame(64, 0u);
    const uint32_t bf_cap = (uint32_t)gsr_block_first_cap(P);
    const int64_t nblk = ((int64_t)R + GSR_TS_ITEMS - 1) / GSR_TS_ITEMS;
    std::vector<uint2> block_first(std::max<size_t>(bf_cap, (size_t)nblk + 2));
    frame[0] = R;
    GsrDepthSortBufs b;
    b.pairs[0] = pairs0.data(); b.pairs[1] = pairs1.data();
    b.cnt_tab = cnt_tab.data(); b.tile_tab = tile_tab.data(); b.cnt_total = cnt_total.data(); b.tile_total = tile_total.data(); b.plan = plan.data();
    std::vector<uint32_t> eq_tab(GSR_EQ_TAB_WORDS);
    std::vector<uint16_t> bucket_of((size_t)P + 64);
    b.eq_tab = eq_tab.data(); b.bucket_of = bucket_of.data();
    gsr_launch_depth_bucket_sort(P, keys, tiles, rect, frame.data(), wg_range, n_range, b, order, rect_sorted.data(), offsets.data(), block_first.data(), bf_cap, nullptr, nullptr);
    if (R > 0 && !simt::launch_error) {
        GsrTileSortPlan tp;
        gsr_tile_sort_plan(n_tiles, P, &tp);
        const int nb1 = 1 << tp.hb;
        if ((uint64_t)nblk + 1 > (uint64_t)bf_cap) gsr_launch_fill_block_first(P, offsets.data(), block_first.data(), (uint32_t)(nblk + 2), nullptr);      // (bin_and_render does the same)
        std::vector<uint64_t> words((size_t)R + 16);
        std::vector<uint32_t> hist1((size_t)256 * (nblk + 1)), digit_total(256), bucket_base(257), blk2_start(257), hist2((size_t)(nblk + 256 + 256) * 256), tile_base(65536);
        gsr_launch_tile_sort_level1(tp, R, gx, block_first.data(), offsets.data(), rect_sorted.data(), order, words.data(), hist1.data(), digit_total.data(),
                                    bucket_base.data(), blk2_start.data(), nullptr, nullptr);
        if (!simt::launch_error)
            gsr_launch_tile_sort_level2(tp, R, n_tiles, words.data(), point_list, bucket_base.data(), blk2_start.data(), hist2.data(), tile_base.data(), ranges, nullptr);
        (void)nb1;
    }
    if (simt::launch_error) {
        snprintf(g_err, sizeof(g_err), "%s", simt::launch_error);
        simt::launch_error = nullptr;
        return -1;
    }
    return (int64_t)R;
}

}  // extern "C"
