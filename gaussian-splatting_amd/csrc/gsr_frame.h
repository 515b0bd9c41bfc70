// Frame statistics of the key-producing kernel (preprocess / splat ingest): R = sum of the tile counts and the range of
// the depth keys, available the moment that kernel ends -- BEFORE the depth sort -- instead of after the scan.
//
// Why: R sizes the binning buffer and the emission grids, so the host must learn it mid-pipeline (the reference has the
// same read-back after its scan).  R does not depend on the depth order; summed here, the host reads
// it while the GPU is still sorting, and the read-back leaves the critical path (rounds 1-3: 5-7 us of GPU idle per frame).
// The key range makes the depth sort's bucket mapping adaptive (depthsort.hip).
//
// Protocol: every workgroup folds its partial results into a small device-memory block with agent-scope atomics and takes a
// ticket; the workgroup that draws the last ticket reads the totals, copies them to the geometry buffer (for the kernels
// that follow), publishes them to the mapped host word (value first, sequence number last, system-scope release) and
// RESETS the block, so the next lease finds it zeroed.  The block belongs to the host-word lease (gsr_api.cpp), never to
// two frames at once.  No spinning anywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gsr_wave.h"

// state block (device memory, 16 words, zero between frames):
//   [0..1] sum of tile counts (u64)   [2] tickets   [3] max of ~key over listed Gaussians (= ~kmin)   [4] max key   [5] flags
// frame words (geometry buffer, GsrGeom::num_rendered): [0] R low, [1] R high, [2] kmin, [3] kmax   (kmin > kmax: nothing listed)
// host word (mapped): [0] R low, [1] sequence number, [2] R high, [3] "a depth key needed more than 27 bits"
struct GsrFrameStatsDev {
    uint32_t* state;       // NULL: the kernel keeps no statistics (shard projection without binning)
    uint32_t* frame;
    uint32_t* host_word;   // may be NULL (then only `frame` is written)
    uint32_t seq;
};

#define GSR_FRAME_FLAG_KEY_OVERFLOW 1u

#ifdef __HIPCC__
// Called by EVERY thread of EVERY workgroup of a 256-thread kernel, after its streaming loop.  tiles_sum / kmin / kmax /
// key_ovf are the thread's own partial results (kmin = 0xFFFFFFFF, kmax = 0 when it listed nothing).
__device__ __forceinline__ void gsr_frame_stats_commit(const GsrFrameStatsDev& fs, uint64_t tiles_sum, uint32_t kmin, uint32_t kmax,
                                                       bool key_ovf) {
    if (!fs.state) return;
    __shared__ uint64_t s_sum[4];
    __shared__ uint32_t s_nmin[4], s_max[4], s_ovf[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t wsum = gsrw::wave_incl_scan_u64(tiles_sum, lane);
    const uint32_t wnmin = gsrw::wave_incl_max_u32(~kmin), wmax = gsrw::wave_incl_max_u32(kmax);
    const uint64_t ovf = __ballot(key_ovf);
    if (lane == 63) { s_sum[w] = wsum; s_nmin[w] = wnmin; s_max[w] = wmax; s_ovf[w] = ovf ? 1u : 0u; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const uint64_t sum = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    const uint32_t nmin = max(max(s_nmin[0], s_nmin[1]), max(s_nmin[2], s_nmin[3]));
    const uint32_t kmx = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    const uint32_t flags = (s_ovf[0] | s_ovf[1] | s_ovf[2] | s_ovf[3]) ? GSR_FRAME_FLAG_KEY_OVERFLOW : 0u;
    unsigned long long* st_sum = reinterpret_cast<unsigned long long*>(fs.state);
    if (sum) __hip_atomic_fetch_add(st_sum, (unsigned long long)sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nmin) __hip_atomic_fetch_max(fs.state + 3, nmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (kmx) __hip_atomic_fetch_max(fs.state + 4, kmx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (flags) __hip_atomic_fetch_or(fs.state + 5, flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t ticket = __hip_atomic_fetch_add(fs.state + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket != gridDim.x - 1u) return;
    // last workgroup: every other one has folded its part in (their atomics precede their tickets)
    const unsigned long long R = __hip_atomic_load(st_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t t_nmin = __hip_atomic_load(fs.state + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t t_max = __hip_atomic_load(fs.state + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t t_flags = __hip_atomic_load(fs.state + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fs.frame[0] = (uint32_t)R;
    fs.frame[1] = (uint32_t)(R >> 32);
    fs.frame[2] = ~t_nmin;
    fs.frame[3] = t_max;
    __hip_atomic_store(st_sum, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(fs.state + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(fs.state + 4, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(fs.state + 5, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(fs.state + 2, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (fs.host_word) {      // value first, then the sequence number (system-scope release)
        __hip_atomic_store(fs.host_word + 0, (uint32_t)R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fs.host_word + 2, (uint32_t)(R >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(fs.host_word + 3, (t_flags & GSR_FRAME_FLAG_KEY_OVERFLOW) ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
