// csrc/depthsort.hip (the bucket depth sort: ds_hist, ds_scan, ds_scatter, ds_segsort) compiled for the HOST through the SIMT-on-CPU shim and
// run through its own launcher, gsr_launch_depth_bucket_sort (tests/test_simt_depthsort_cpu.py).  TEST INFRASTRUCTURE, never part of libgsr_hip.so.
#include "hip/hip_runtime.h"
#include "depthsort.hip"
#include "simt_runtime.h"
#include <vector>

static char g_err[256];

extern "C" {

const char* simt_ds_last_error(void) { return g_err; }

// keys[P] (27-bit depth keys, 2^27 - 1 = no tile), tiles[P], rect[P], wg_range[n_range] = (~smallest, largest key) per workgroup of the
// key-producing kernel -> order, rect_sorted, offsets (inclusive scan of the tile counts in depth order), block_first; frame[0..1] = R.
// slow_word: set to 1 by a segment that had to go through global memory.
int simt_depth_bucket_sort(int P, const uint32_t* keys, const uint32_t* tiles, const uint2* rect, uint32_t* frame, const uint2* wg_range, int n_range,
                           uint32_t* order, uint2* rect_sorted, uint32_t* offsets, uint2* block_first, uint32_t bf_cap, uint32_t* slow_word) {
    const size_t nblocks = gsr_depth_bucket_blocks(P), nseg = gsr_depth_bucket_segments(P);
    std::vector<uint2> pairs0((size_t)P + 16), pairs1((size_t)P + 16);
    std::vector<uint32_t> cnt_tab(nblocks * DS_NB), tile_tab(nblocks * DS_NB), cnt_total(DS_NB), tile_total(DS_NB), plan(nseg * GSR_DS_PLAN_WORDS + 16);
    std::vector<uint32_t> eq_tab(GSR_EQ_TAB_WORDS);
    GsrDepthSortBufs b;
    b.pairs[0] = pairs0.data(); b.pairs[1] = pairs1.data();
    b.cnt_tab = cnt_tab.data(); b.tile_tab = tile_tab.data(); b.cnt_total = cnt_total.data(); b.tile_total = tile_total.data(); b.plan = plan.data();
    std::vector<uint16_t> bucket_of((size_t)P + 64);
    b.eq_tab = eq_tab.data(); b.bucket_of = bucket_of.data();
    gsr_launch_depth_bucket_sort(P, keys, tiles, rect, frame, wg_range, n_range, b, order, rect_sorted, offsets, block_first, bf_cap, slow_word, nullptr);
    if (!simt::launch_error) return 0;
    snprintf(g_err, sizeof(g_err), "bucket depth sort: %s", simt::launch_error);
    simt::launch_error = nullptr;
    return -1;
}

}  // extern "C"
