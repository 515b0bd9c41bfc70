// Frame statistics of the key-producing kernel (preprocess / splat ingest): R = sum of the tile counts and the range of
// the depth keys, available the moment that kernel ends -- BEFORE the depth sort -- instead of after the scan.
//
// Why: R sizes the binning buffer and the emission grids, so the host must learn it mid-pipeline (the reference has the
// same read-back after its scan).  R does not depend on the depth order; summed here, the host reads
// it while the GPU is still sorting, and the read-back leaves the critical path (rounds 1-3: 5-7 us of GPU idle per frame).
// The key range bounds the depth sort's bucket tables (depthsort.hip).
//
// Protocol: every workgroup adds ONE packed 64-bit word to a device-memory counter with a relaxed agent-scope atomic -- its tile
// count, "one of my depth keys overflowed" and a ticket -- and leaves the range of its depth keys in a per-workgroup table
// (plain stores; the depth sort's first kernel reduces the table after the kernel boundary, depthsort.hip).  The returned
// value tells the workgroup that drew the last ticket so, and hands it the totals: it copies R to the geometry buffer,
// publishes it to the mapped host word (value first, sequence number last, system-scope release) and RESETS the counter, so
// the next lease finds it zeroed.  The counter belongs to the host-word lease (gsr_api.cpp), never to two frames at once.
// A slot holds TWO counters and its frames alternate between them; the last workgroup also clears the one it did not use (round 6: a
// counter that does not start at zero publishes a partial R early and is left non-zero again by the workgroups that come after -- for
// ever; with the second counter such a state lasts one frame).
// One atomic per workgroup, nothing to order, no spinning.  (First version: five atomics and an acq_rel ticket -- the
// agent-scope release / acquire is an L2 write-back / invalidate on this multi-XCD part and doubled the preprocess kernel.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gsr_wave.h"

// counter (device memory, u64, zero between frames): bits [0, 42) sum of the tile counts (a workgroup's part saturates at
// 2^31: any total >= 2^31 is refused by the host anyway), [42, 53) workgroups with a key overflow, [53, 64) tickets
// -> the key-producing kernels run at most GSR_FRAME_MAX_GROUPS workgroups
// frame words (geometry buffer, GsrGeom::num_rendered): [0] R low, [1] R high; written by ds_hist: [2] / [3] and [6] / [7] the smallest / largest depth key of a listed
// Gaussian (depthsort.hip); [8] "a depth key needed more than 27 bits" (a copy of host word [3]); [9] the sequence number (a copy of host word [1])
// host word (mapped): [0] R low, [1] sequence number, [2] R high, [3] "a depth key needed more than 27 bits"
#define GSR_FRAME_MAX_GROUPS 2047
struct GsrFrameStatsDev {
    uint32_t* state;       // NULL: the kernel keeps no statistics (shard projection without binning)
    uint32_t* state_other = nullptr;      // the slot's SECOND counter (the frames of a slot alternate between two): cleared by this frame's last workgroup too, so
                           // that a counter that was ever left dirty costs one frame, not every frame after it; may be NULL
    uint32_t* frame;
    uint2* wg_range;       // [workgroups] (~smallest, largest) depth key of the workgroup's listed Gaussians; (0, 0): none
    uint32_t* host_word;   // may be NULL (then only `frame` is written)
    uint32_t seq;
};

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
    fs.frame[8] = n_ovf ? 1u : 0u;      // ([8], [9]: read by the host only when the mapped word did not arrive: gsr_api.cpp wait_for_R)
    fs.frame[9] = fs.seq;
    __hip_atomic_store(ctr, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (read next by a later kernel)
    if (fs.state_other)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(fs.state_other), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
