// What the sampling workgroups of a key-producing kernel leave in fs.sample_hist (csrc/gsr_frame.h gsr_frame_stats_commit), restated
// for the harnesses that are handed bare key arrays: workgroup w of a grid of `n_range` 256-thread workgroups owns the keys w * 256 + t + k * n_range * 256;
// per workgroup a row of 1024 coarse counts (key >> 17) + 1024 counts of (key >> 7) & 1023 over all its keys (the sub-bins of all coarse bins folded).
// TEST INFRASTRUCTURE.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "gsr_internal.h"

static inline std::vector<uint16_t> simt_sample_hist(const uint32_t* keys, int64_t P, int n_range) {
    std::vector<uint16_t> h(GSR_EQ_SAMPLE_BYTES / 2, 0);
    for (int wg = 0; wg < n_range; ++wg) {
        const int w = gsr_frame_sampler_row(P, (unsigned)n_range, (unsigned)wg);      // (the product's own choice of sampling workgroups, csrc/gsr_frame.h)
        if (w < 0) continue;
        std::vector<uint32_t> c(GSR_EQ_BINS, 0u), f(GSR_EQ_BINS, 0u);
        for (int64_t i0 = (int64_t)wg * 256; i0 < P; i0 += (int64_t)n_range * 256)
            for (int64_t i = i0; i < i0 + 256 && i < P; ++i)
                if (keys[i] != GSR_DEPTH_KEY_CULLED) {
                    ++c[keys[i] >> GSR_EQ_SHIFT];
                    ++f[(keys[i] >> GSR_EQ_SHIFT2) & (GSR_EQ_BINS - 1)];
                }
        for (int b = 0; b < GSR_EQ_BINS; ++b) {
            h[(size_t)w * GSR_EQ_SAMPLE_ROW + b] = (uint16_t)(c[b] < 65535u ? c[b] : 65535u);
            h[(size_t)w * GSR_EQ_SAMPLE_ROW + GSR_EQ_BINS + b] = (uint16_t)(f[b] < 65535u ? f[b] : 65535u);
        }
    }
    return h;
}
