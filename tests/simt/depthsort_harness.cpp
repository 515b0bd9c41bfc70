// csrc/depthsort.hip (the bucket depth sort: ds_hist, ds_scan, ds_scatter, ds_segsort) compiled for the HOST through the SIMT-on-CPU shim and
// run launch by launch, workgroup by workgroup (tests/test_simt_depthsort_cpu.py).  TEST INFRASTRUCTURE, never part of libgsr_hip.so.
#include "hip/hip_runtime.h"
#include "depthsort.hip"
#include "simt_runtime.h"
#include <vector>

static char g_err[256];

extern "C" {

const char* simt_ds_last_error(void) { return g_err; }

// keys[P] (27-bit depth keys, 2^27 - 1 = no tile), tiles[P], rect[P], wg_range[n_range] = (~smallest, largest key) per workgroup of the
// key-producing kernel -> order, rect_sorted, offsets (inclusive scan of the tile counts in depth order), block_first; frame[0..1] = R.
// slow_word: set to 1 by a segment that had to go through global memory.  Same four launches as gsr_launch_depth_bucket_sort.
int simt_depth_bucket_sort(int P, const uint32_t* keys, const uint32_t* tiles, const uint2* rect, uint32_t* frame, const uint2* wg_range, int n_range,
                           uint32_t* order, uint2* rect_sorted, uint32_t* offsets, uint2* block_first, uint32_t bf_cap, uint32_t* slow_word) {
    const int nblocks = (int)gsr_depth_bucket_blocks(P), nseg_cap = (int)gsr_depth_bucket_segments(P);
    std::vector<uint2> pairs0((size_t)P + 16), pairs1((size_t)P + 16);
    std::vector<uint32_t> cnt_tab((size_t)nblocks * DS_NB), tile_tab((size_t)nblocks * DS_NB), cnt_total(DS_NB), tile_total(DS_NB), plan((size_t)nseg_cap * 8 + 8);
    auto fail = [&](const char* what, int b) { snprintf(g_err, sizeof(g_err), "%s block %d: %s", what, b, simt::g.error ? simt::g.error : "?"); return -1; };
    for (int b = 0; b < nblocks; ++b)
        if (!simt::run_block((unsigned)b, (unsigned)nblocks, DS_THREADS, [&] { ds_hist(P, keys, tiles, frame, wg_range, n_range, cnt_tab.data(), tile_tab.data()); })) return fail("ds_hist", b);
    for (int b = 0; b < DS_NB / 16; ++b)
        if (!simt::run_block((unsigned)b, (unsigned)(DS_NB / 16), DS_THREADS, [&] { ds_scan(nblocks, cnt_tab.data(), tile_tab.data(), cnt_total.data(), tile_total.data()); })) return fail("ds_scan", b);
    for (int b = 0; b < nblocks + 1; ++b)
        if (!simt::run_block((unsigned)b, (unsigned)(nblocks + 1), S3_THREADS, [&] { ds_scatter(P, nblocks, keys, frame, cnt_tab.data(), cnt_total.data(), tile_total.data(), pairs0.data(), order, offsets, rect_sorted, plan.data(), nseg_cap); })) return fail("ds_scatter", b);
    for (int b = 0; b < nseg_cap; ++b)
        if (!simt::run_block((unsigned)b, (unsigned)nseg_cap, SG_THREADS, [&] { ds_segsort(plan.data(), frame, pairs0.data(), pairs1.data(), rect, order, rect_sorted, offsets, block_first, bf_cap, slow_word); })) return fail("ds_segsort", b);
    return 0;
}

}  // extern "C"
