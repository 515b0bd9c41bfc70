// What the first GSR_EQ_SAMPLE_WGS workgroups of a key-producing kernel leave in fs.sample_hist (csrc/gsr_frame.h gsr_frame_stats_commit), restated
// for the harnesses that are handed bare key arrays: workgroup w of a grid of `n_range` 256-thread workgroups owns the keys w * 256 + t + k * n_range * 256;
// per workgroup a row of 1024 coarse counts (key >> 17) + 1024 counts of the sub-bins ((key >> 7) & 1023) of ITS fullest coarse bin, then that bin.
// TEST INFRASTRUCTURE.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "gsr_internal.h"

static inline std::vector<uint16_t> simt_sample_hist(const uint32_t* keys, int64_t P, int n_range) {
    std::vector<uint16_t> h(GSR_EQ_SAMPLE_BYTES / 2, 0);
    uint32_t* hot_of = reinterpret_cast<uint32_t*>(h.data() + (size_t)GSR_EQ_SAMPLE_WGS * GSR_EQ_SAMPLE_ROW);
    for (int w = 0; w < GSR_EQ_SAMPLE_WGS && w < n_range; ++w) {
        std::vector<uint32_t> c(GSR_EQ_BINS, 0u), f(GSR_EQ_BINS, 0u);
        auto each = [&](auto&& fn) {
            for (int64_t i0 = (int64_t)w * 256; i0 < P; i0 += (int64_t)n_range * 256)
                for (int64_t i = i0; i < i0 + 256 && i < P; ++i)
                    if (keys[i] != GSR_DEPTH_KEY_CULLED) fn(keys[i]);
        };
        each([&](uint32_t k) { ++c[k >> GSR_EQ_SHIFT]; });
        uint32_t hot = 0;
        for (uint32_t b = 1; b < (uint32_t)GSR_EQ_BINS; ++b)
            if (c[b] > c[hot]) hot = b;      // (the lowest bin on ties)
        each([&](uint32_t k) { if ((k >> GSR_EQ_SHIFT) == hot) ++f[(k >> GSR_EQ_SHIFT2) & (GSR_EQ_BINS - 1)]; });
        for (int b = 0; b < GSR_EQ_BINS; ++b) {
            h[(size_t)w * GSR_EQ_SAMPLE_ROW + b] = (uint16_t)(c[b] < 65535u ? c[b] : 65535u);
            h[(size_t)w * GSR_EQ_SAMPLE_ROW + GSR_EQ_BINS + b] = (uint16_t)(f[b] < 65535u ? f[b] : 65535u);
        }
        hot_of[w] = hot;
    }
    return h;
}
