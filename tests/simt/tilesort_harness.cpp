// csrc/tilesort.hip (+ the measurement build's csrc/ab/emit_scatter_segments.inc) compiled for the HOST through the SIMT-on-CPU shim
// (tests/simt/hip/hip_runtime.h) and driven workgroup by workgroup from Python (tests/test_simt_tilesort_cpu.py).  TEST INFRASTRUCTURE:
// built with g++ into tests/_build/, never part of libgsr_hip.so.
#include "hip/hip_runtime.h"
#define GSR_AB_VARIANTS 1
#include "tilesort.hip"
#include "simt_runtime.h"

// the one launcher of another translation unit that tilesort.hip's (unused) host functions name
void gsr_launch_rs_scan(uint32_t*, int, int, uint32_t*, hipStream_t) {}

static char g_err[256];

extern "C" {

const char* simt_last_error(void) { return g_err; }

// level-1 histogram of every block (csrc/tilesort.hip emit_hist): hist[d * nblk + block]
int simt_emit_hist(uint32_t R, int gx, int lb, int hb, const uint2* block_first, const uint32_t* offsets, const uint2* rect_sorted, uint32_t* hist) {
    const int nblk = (int)((R + TS_ITEMS - 1) / TS_ITEMS), nb1 = 1 << hb;
    for (int b = 0; b < nblk; ++b)
        if (!simt::run_block((unsigned)b, (unsigned)nblk, WG_THREADS, [&] { emit_hist(R, gx, lb, nb1, block_first, offsets, rect_sorted, hist, nblk); })) {
            snprintf(g_err, sizeof(g_err), "emit_hist block %d: %s", b, simt::g.error ? simt::g.error : "?");
            return -1;
        }
    return 0;
}

// level-1 scatter of every block; mode 0 = emit_scatter (the shipped kernel), 1 = emit_scatter_seg (row pieces, measurement build).
// `hist` holds, per (bucket, block), the instances of EARLIER blocks (what rs_scan leaves), digit_total the bucket totals.
int simt_emit_scatter(int mode, int word64, uint32_t R, int gx, int lb, int hb, const uint2* block_first, const uint32_t* offsets,
                      const uint2* rect_sorted, const uint32_t* order, const uint32_t* hist, const uint32_t* digit_total, void* words_out,
                      uint32_t* bucket_base, uint32_t* blk2_start, float* splats) {
    const int nblk = (int)((R + TS_ITEMS - 1) / TS_ITEMS);
    float4* sp = reinterpret_cast<float4*>(splats);
    for (int b = 0; b < nblk; ++b) {
        bool ok;
        if (word64) {
            uint64_t* w = (uint64_t*)words_out;
            ok = mode ? simt::run_block((unsigned)b, (unsigned)nblk, WG_THREADS, [&] { emit_scatter_seg<uint64_t>(R, gx, lb, hb, block_first, offsets, rect_sorted, order, hist, digit_total, nblk, w, bucket_base, blk2_start, sp); })
                      : simt::run_block((unsigned)b, (unsigned)nblk, WG_THREADS, [&] { emit_scatter<uint64_t>(R, gx, lb, hb, block_first, offsets, rect_sorted, order, hist, digit_total, nblk, w, bucket_base, blk2_start, sp); });
        } else {
            uint32_t* w = (uint32_t*)words_out;
            ok = mode ? simt::run_block((unsigned)b, (unsigned)nblk, WG_THREADS, [&] { emit_scatter_seg<uint32_t>(R, gx, lb, hb, block_first, offsets, rect_sorted, order, hist, digit_total, nblk, w, bucket_base, blk2_start, sp); })
                      : simt::run_block((unsigned)b, (unsigned)nblk, WG_THREADS, [&] { emit_scatter<uint32_t>(R, gx, lb, hb, block_first, offsets, rect_sorted, order, hist, digit_total, nblk, w, bucket_base, blk2_start, sp); });
        }
        if (!ok) {
            snprintf(g_err, sizeof(g_err), "emit_scatter mode %d block %d: %s", mode, b, simt::g.error ? simt::g.error : "?");
            return -1;
        }
    }
    return 0;
}

// level 2 (csrc/tilesort.hip bucket_hist [+ bucket_scan] + bucket_scatter): the sorted point list and the tile ranges from the level-1 words.
// fused = 1: the scan folded into the scatter (what the host picks while a bucket has few workgroups), 0: the separate scan launch.
int simt_level2(int word64, int fused, uint32_t R, int n_tiles, int lb, int hb, const void* words, const uint32_t* bucket_base, const uint32_t* blk2_start,
                uint32_t* hist2, uint32_t* tile_base, uint32_t* point_list, uint2* ranges) {
    const int nblk = (int)((R + TS_ITEMS - 1) / TS_ITEMS), nb1 = 1 << hb, nblk2 = nblk + nb1;
    uint2* roe = fused ? ranges : nullptr;
    for (int b = 0; b < nblk2; ++b) {
        const bool ok = word64 ? simt::run_block((unsigned)b, (unsigned)nblk2, WG_THREADS, [&] { bucket_hist<uint64_t>(lb, hb, (const uint64_t*)words, bucket_base, blk2_start, hist2, roe, n_tiles); })
                               : simt::run_block((unsigned)b, (unsigned)nblk2, WG_THREADS, [&] { bucket_hist<uint32_t>(lb, hb, (const uint32_t*)words, bucket_base, blk2_start, hist2, roe, n_tiles); });
        if (!ok) { snprintf(g_err, sizeof(g_err), "bucket_hist block %d: %s", b, simt::g.error ? simt::g.error : "?"); return -1; }
    }
    if (!fused)
        for (int b = 0; b < nb1; ++b)
            if (!simt::run_block((unsigned)b, (unsigned)nb1, WG_THREADS, [&] { bucket_scan(lb, n_tiles, bucket_base, blk2_start, hist2, tile_base, ranges); })) {
                snprintf(g_err, sizeof(g_err), "bucket_scan block %d: %s", b, simt::g.error ? simt::g.error : "?");
                return -1;
            }
    for (int b = 0; b < nblk2; ++b) {
        bool ok;
#define SIMT_L2(WORD_, FUSED_) simt::run_block((unsigned)b, (unsigned)nblk2, WG_THREADS, [&] { bucket_scatter<WORD_, FUSED_>(lb, hb, (const WORD_*)words, bucket_base, blk2_start, hist2, tile_base, point_list, ranges, n_tiles); })
        if (word64) ok = fused ? SIMT_L2(uint64_t, true) : SIMT_L2(uint64_t, false);
        else ok = fused ? SIMT_L2(uint32_t, true) : SIMT_L2(uint32_t, false);
#undef SIMT_L2
        if (!ok) { snprintf(g_err, sizeof(g_err), "bucket_scatter block %d: %s", b, simt::g.error ? simt::g.error : "?"); return -1; }
    }
    return 0;
}

// the per-block table of first Gaussians from the finished offsets (csrc/tilesort.hip fill_block_first)
int simt_fill_block_first(int P, const uint32_t* offsets, uint2* block_first, uint32_t cap) {
    const int nb = (P + WG_THREADS - 1) / WG_THREADS;
    for (int b = 0; b < nb; ++b)
        if (!simt::run_block((unsigned)b, (unsigned)nb, WG_THREADS, [&] { fill_block_first(P, offsets, block_first, cap); })) {
            snprintf(g_err, sizeof(g_err), "fill_block_first block %d: %s", b, simt::g.error ? simt::g.error : "?");
            return -1;
        }
    return 0;
}

}  // extern "C"
