// csrc/tilesort.hip compiled for the HOST through the SIMT-on-CPU shim
// (tests/simt/hip/hip_runtime.h) and driven through ITS OWN LAUNCHERS from Python (tests/test_simt_tilesort_cpu.py).  TEST INFRASTRUCTURE:
// built with g++ into tests/_build/, never part of libgsr_hip.so.
#include "hip/hip_runtime.h"
#define GSR_AB_VARIANTS 1
#include "tilesort.hip"
#include "simt_runtime.h"

// rs_scan of sort.hip (another translation unit), restated for the host: hist[d * nblocks + b] <- entries of earlier blocks, totals per digit
void gsr_launch_rs_scan(uint32_t* block_hist, int nblocks, int ndigits, uint32_t* digit_total, hipStream_t) {
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
static int finish(const char* what) {
    if (!simt::launch_error) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, simt::launch_error);
    simt::launch_error = nullptr;
    return -1;
}

extern "C" {

const char* simt_last_error(void) { return g_err; }

void simt_tile_sort_plan(int n_tiles, int P, int* lb, int* hb, int* word64) {
    GsrTileSortPlan plan;
    gsr_tile_sort_plan(n_tiles, P, &plan);
    *lb = plan.lb; *hb = plan.hb; *word64 = plan.word64 ? 1 : 0;
}

// the per-block table of first Gaussians from the finished offsets (gsr_launch_fill_block_first)
int simt_fill_block_first(int P, const uint32_t* offsets, uint2* block_first, uint32_t cap) {
    gsr_launch_fill_block_first(P, offsets, block_first, cap, nullptr);
    return finish("fill_block_first");
}

// level 1 through gsr_launch_tile_sort_level1: emit_hist, the scan, the scatter.  hist1: [nb1 * nblk] (left holding the scanned table), digit_total: [nb1].
int simt_level1(int word64, int64_t R, int gx, int lb, int hb, const uint2* block_first, const uint32_t* offsets, const uint2* rect_sorted,
                const uint32_t* order, void* words, uint32_t* hist1, uint32_t* digit_total, uint32_t* bucket_base, uint32_t* blk2_start, float* splats) {
    GsrTileSortPlan plan{true, lb, hb, word64 != 0};
    gsr_launch_tile_sort_level1(plan, R, gx, block_first, offsets, rect_sorted, order, words, hist1, digit_total, bucket_base, blk2_start,
                                reinterpret_cast<float4*>(splats), nullptr);
    return finish("level 1");
}

// level 2 through gsr_launch_tile_sort_level2 (scan_mode: 1 = separate bucket_scan launch, 2 = folded into bucket_scatter, 0 = the launcher's choice)
int simt_level2(int word64, int scan_mode, int64_t R, int n_tiles, int lb, int hb, const void* words, const uint32_t* bucket_base, const uint32_t* blk2_start,
                uint32_t* hist2, uint32_t* tile_base, uint32_t* point_list, uint2* ranges) {
    GsrTileSortPlan plan{true, lb, hb, word64 != 0};
    gsr_set_level2_scan_mode(scan_mode);
    gsr_launch_tile_sort_level2(plan, R, n_tiles, words, point_list, bucket_base, blk2_start, hist2, tile_base, ranges, nullptr);
    gsr_set_level2_scan_mode(0);
    return finish("level 2");
}

}  // extern "C"
