// Frame statistics of the key-producing kernel (preprocess / splat ingest): R = sum of the tile counts and the range of
// the depth keys, available the moment that kernel ends -- BEFORE the depth sort -- instead of after the scan.
//
// Why: R sizes the binning buffer and the emission grids, so the host must learn it mid-pipeline (the reference has the
// same read-back after its scan).  R does not depend on the depth order; summed here, the host reads
// it while the GPU is still sorting, and the read-back leaves the critical path (rounds 1-3: 5-7 us of GPU idle per frame).
// The key range makes the depth sort's bucket mapping adaptive (depthsort.hip).
//
// Protocol: every workgroup adds ONE packed 64-bit word to a device-memory counter with a relaxed agent-scope atomic -- its tile
// count, "one of my depth keys overflowed" and a ticket -- and leaves the range of its depth keys in a per-workgroup table
// (plain stores; the depth sort's first kernel reduces the table after the kernel boundary, depthsort.hip).  The returned
// value tells the workgroup that drew the last ticket so, and hands it the totals: it copies R to the geometry buffer,
// publishes it to the mapped host word (value first, sequence number last, system-scope release) and RESETS the counter, so
// the next lease finds it zeroed.  The counter belongs to the host-word lease (gsr_api.cpp), never to two frames at once.
// One atomic per workgroup, nothing to order, no spinning.  (First version: five atomics and an acq_rel ticket -- the
// agent-scope release / acquire is an L2 write-back / invalidate on this multi-XCD part and doubled the preprocess kernel.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gsr_wave.h"

// Histogram-equalised bucket mapping of the depth sort (round 6, depthsort.hip): the 27-bit key space is cut into GSR_EQ_BINS coarse bins of
// 2^GSR_EQ_SHIFT keys (64 per octave of depth); GSR_EQ_SAMPLE_WGS workgroups of the key-producing kernel (gsr_frame_sampler_row) leave the coarse histogram
// of THEIR keys -- a regular sample of the frame, every workgroup's loop strides over the whole array -- and ds_hist turns the summed sample
// into a (first bucket, buckets) table per coarse bin: one bucket for every coarse bin inside the frame's key range, the rest of the 2046
// handed out in proportion to the sampled mass, so that the buckets hold about the same number of keys whatever the depth distribution is.
#define GSR_EQ_SHIFT 17
#define GSR_EQ_BINS 1024         // 2^(27 - GSR_EQ_SHIFT)
#ifndef GSR_EQ_SAMPLE_WGS
#define GSR_EQ_SAMPLE_WGS 16      /* (A/B builds: -DGSR_EQ_SAMPLE_WGS=8) */
#endif
#define GSR_FRAME_KEY_CULLED ((1u << 27) - 1u)      // == GSR_DEPTH_KEY_CULLED (gsr_internal.h, checked there)
// Second level, for ONE coarse bin.  A sample workgroup also leaves the histogram of (key >> 7) & 1023 over ALL its keys: the 1024 sub-bins (128 keys each) of
// every coarse bin folded onto one another.  When one coarse bin ("hot") holds an eighth of the sample or more -- a wall seen head-on, a cluster within a
// fraction of a percent of one depth, thousands of equal depths -- the folded histogram IS that bin's sub-bin histogram plus a flat background from the other
// bins ((C - c_hot) / 1024 per sub-bin, subtracted), and ds_hist spreads the hot bin's buckets over its sub-bins in proportion to it, so that a
// concentration 128 keys wide -- or a run of ties -- still gets buckets of its own.  One pass over the keys, no agreement between workgroups needed.
#define GSR_EQ_SHIFT2 7          // GSR_EQ_SHIFT - log2(GSR_EQ_BINS)
#define GSR_EQ_NO_HOT 0xFFFFFFFFu
// sample buffer: per sample workgroup a row of 2 * GSR_EQ_BINS 16-bit counts (coarse | folded sub-bins)
#define GSR_EQ_SAMPLE_ROW (2 * GSR_EQ_BINS)
#define GSR_EQ_SAMPLE_BYTES ((size_t)GSR_EQ_SAMPLE_WGS * GSR_EQ_SAMPLE_ROW * 2)
// table buffer (ds_hist -> ds_scatter): GSR_EQ_BINS words level 1, GSR_EQ_BINS words level 2, then the hot bin (GSR_EQ_NO_HOT: no second level)
#define GSR_EQ_TAB_WORDS (2 * GSR_EQ_BINS + 16)

// counter (device memory, u64, zero between frames): bits [0, 42) sum of the tile counts (a workgroup's part saturates at
// 2^31: any total >= 2^31 is refused by the host anyway), [42, 53) workgroups with a key overflow, [53, 64) tickets
// -> the key-producing kernels run at most GSR_FRAME_MAX_GROUPS workgroups
// frame words (geometry buffer, GsrGeom::num_rendered): [0] R low, [1] R high; written by ds_hist: [2] kmin, [3] kmax of the ROBUST key
// range the depth buckets span, [6] / [7] the true extremes (depthsort.hip); [8] "a depth key needed more than 27 bits" (a copy of host word [3])
// host word (mapped): [0] R low, [1] sequence number, [2] R high, [3] "a depth key needed more than 27 bits"
#define GSR_FRAME_MAX_GROUPS 2047
struct GsrFrameStatsDev {
    uint32_t* state;       // NULL: the kernel keeps no statistics (shard projection without binning)
    uint32_t* frame;
    uint2* wg_range;       // [workgroups] (~smallest, largest) depth key of the workgroup's listed Gaussians; (0, 0): none
    uint16_t* sample_hist; // GSR_EQ_SAMPLE_BYTES (layout above) or NULL: key histograms of the GSR_EQ_SAMPLE_WGS sampling workgroups
    uint32_t* host_word;   // may be NULL (then only `frame` is written)
    uint32_t seq;
};

// The sampling workgroups of a key-producing kernel of `grid` 256-thread workgroups over P keys (loop: i = wg * 256 + t; i < P; i += grid * 256):
// sampler r of n = min(GSR_EQ_SAMPLE_WGS, grid) is workgroup lo + ((2 r + 1) (hi - lo)) / (2 n), with [lo, hi) the workgroups that run the fewest
// iterations if there are at least n of them, else the whole grid.  Returns the row of workgroup `wg`, or -1.  (Host code restates it: tests/simt/sample_hist.h.)
static inline __host__ __device__ int gsr_frame_sampler_row(int64_t P, unsigned grid, unsigned wg) {
    const unsigned n = grid < (unsigned)GSR_EQ_SAMPLE_WGS ? grid : (unsigned)GSR_EQ_SAMPLE_WGS;
    const int64_t stride = (int64_t)grid * 256;
    const int64_t rem = P > 0 ? P % stride : 0;
    const unsigned full = rem == 0 ? grid : (unsigned)((rem + 255) / 256);      // workgroups [0, full) run one more iteration than [full, grid)
    unsigned lo = 0, hi = grid;
    if (grid - full >= n) lo = full;
    for (unsigned r = 0; r < n; ++r)
        if (lo + ((2u * r + 1u) * (hi - lo)) / (2u * n) == wg) return (int)r;
    return -1;
}

#ifdef __HIPCC__
// Called by EVERY thread of EVERY workgroup of a 256-thread kernel, after its streaming loop.  tiles_sum / kmin / kmax /
// key_ovf are the thread's own partial results (kmin = 0xFFFFFFFF, kmax = 0 when it listed nothing).
// keys / P: the depth-key array the kernel has just written with the loop `for (i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)`
// -- the sampling workgroups (gsr_frame_sampler_row) read THEIR OWN keys back (every thread the ones it stored itself) and leave their coarse
// histogram in fs.sample_hist: a regular sample of the frame's depth distribution for the depth sort's bucket mapping (depthsort.hip).
// lds: >= 2 * GSR_EQ_BINS words of LDS that nothing else uses after the loop.
__device__ __forceinline__ void gsr_frame_stats_commit(const GsrFrameStatsDev& fs, uint64_t tiles_sum, uint32_t kmin, uint32_t kmax,
                                                       bool key_ovf, const uint32_t* keys, int64_t P, uint32_t* lds) {
    if (!fs.state) return;
    __shared__ uint64_t s_sum[4];
    __shared__ uint32_t s_nmin[4], s_max[4], s_ovf[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t wsum = gsrw::wave_incl_scan_u64(tiles_sum, lane);
    const uint32_t wnmin = gsrw::wave_incl_max_u32(~kmin), wmax = gsrw::wave_incl_max_u32(kmax);
    const uint64_t ovf = __ballot(key_ovf);
    if (lane == 63) { s_sum[w] = wsum; s_nmin[w] = wnmin; s_max[w] = wmax; s_ovf[w] = ovf ? 1u : 0u; }
    __syncthreads();
    // Which workgroups sample (gsr_frame_sampler below): spread evenly over the workgroups with the FEWEST loop iterations -- with a grid-stride
    // loop the last ones of the grid have one fewer unless P is a multiple of the stride -- so that their extra pass ends before the kernel does
    // (taken by the first workgroups it delayed the second round of workgroups that inherit their slots: +2.5 us on the projection kernel,
    // measured) and spread, not adjacent, so that an index-ordered array (a Morton-sorted model) is sampled at 16 places, not at one.
    const int my_row = gsr_frame_sampler_row(P, gridDim.x, blockIdx.x);
    if (fs.sample_hist && my_row >= 0) {      // (workgroup-uniform; every wave is past its loop: `lds` is free)
        uint32_t* h1 = lds;                    // coarse bins: key >> 17
        uint32_t* h2 = lds + GSR_EQ_BINS;      // sub-bins of ALL coarse bins folded onto one another: (key >> 7) & 1023
#pragma unroll
        for (int b = 0; b < 2 * GSR_EQ_BINS / 256; ++b) lds[b * 256 + threadIdx.x] = 0u;
        __syncthreads();
        // (four keys per trip, loaded unconditionally from a clamped index: one memory round trip per four keys, not per key)
        const int64_t first = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
        for (int64_t i0 = first; i0 < P; i0 += 4 * stride) {
            uint32_t kk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) kk[u] = keys[i0 + u * stride < P ? i0 + u * stride : P - 1];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * stride < P && kk[u] != GSR_FRAME_KEY_CULLED) {
                    atomicAdd(&h1[kk[u] >> GSR_EQ_SHIFT], 1u);
                    atomicAdd(&h2[(kk[u] >> GSR_EQ_SHIFT2) & ((uint32_t)GSR_EQ_BINS - 1u)], 1u);
                }
        }
        __syncthreads();
        uint16_t* row = fs.sample_hist + (size_t)my_row * GSR_EQ_SAMPLE_ROW;
#pragma unroll
        for (int b = 0; b < 2 * GSR_EQ_BINS / 256; ++b) {
            const uint32_t c = lds[b * 256 + threadIdx.x];
            row[b * 256 + threadIdx.x] = (uint16_t)(c < 65535u ? c : 65535u);
        }
    }
    if (threadIdx.x != 0) return;
    uint64_t sum = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    if (sum > 0x80000000ull) sum = 0x80000000ull;
    const uint32_t nmin = max(max(s_nmin[0], s_nmin[1]), max(s_nmin[2], s_nmin[3]));
    const uint32_t kmx = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    const uint64_t any_ovf = (s_ovf[0] | s_ovf[1] | s_ovf[2] | s_ovf[3]) ? 1ull : 0ull;
    fs.wg_range[blockIdx.x] = make_uint2(nmin, kmx);
    unsigned long long* ctr = reinterpret_cast<unsigned long long*>(fs.state);
    const unsigned long long mine = sum | (any_ovf << 42) | (1ull << 53);
    const unsigned long long old = __hip_atomic_fetch_add(ctr, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(old >> 53) != gridDim.x - 1u) return;
    // last workgroup
    const unsigned long long tot = old + mine;
    const unsigned long long R = tot & ((1ull << 42) - 1ull);
    const uint32_t n_ovf = (uint32_t)(tot >> 42) & 0x7FFu;
    fs.frame[0] = (uint32_t)R;
    fs.frame[1] = (uint32_t)(R >> 32);
    fs.frame[8] = n_ovf ? 1u : 0u;      // (read by the host only when the mapped word did not arrive: gsr_api.cpp wait_for_R)
    __hip_atomic_store(ctr, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (read next by a later kernel)
    if (fs.host_word) {      // value first, then the sequence number (system-scope release)
        __hip_atomic_store(fs.host_word + 0, (uint32_t)R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fs.host_word + 2, (uint32_t)(R >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fs.host_word + 3, n_ovf ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fs.host_word + 1, fs.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the thread-side accumulation, one call per Gaussian
struct GsrFrameAcc {
    uint64_t tiles = 0;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    bool ovf = false;
    __device__ __forceinline__ void add(uint32_t key, uint32_t n_tiles) {
        if (n_tiles) {
            tiles += n_tiles;
            kmin = min(kmin, key);
            kmax = max(kmax, key);
        }
    }
};
#endif
