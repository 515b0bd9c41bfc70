// Instance emission + stable sort by tile for frames with <= 65536 tiles -- replaces duplicateWithKeys, the
// 64-bit-key cub::DeviceRadixSort::SortPairs call and identifyTileRanges of the un-vendored reference rasterizer
// (algorithm SURVEY.md Appendix A.3; reached from gaussian_renderer/__init__.py:91-110).
//
// Input: the Gaussians in DEPTH ORDER (depth sort on P keys, sort.hip) with the inclusive scan of their tile counts
// (binning.hip).  The (Gaussian, tile) instances in emission order -- depth order, then y-major / x-fastest inside the
// Gaussian's rectangle -- only need a STABLE sort by tile id to reach the reference's (tile, depth, index) order.
// That sort is a two-level MSD radix sort on the tile id split into a high digit ("bucket", ~ one tile row) and a low
// digit, with the emission FUSED into the first level:
//
//   level 1  emit_hist     block = 4096 consecutive instances (found through a per-block table of first Gaussians that
//                          the scan kernel writes); instances are generated in registers, the block's bucket
//                          histogram is written                                    (no instance ever stored unsorted)
//            rs_scan       (sort.hip) bucket totals + per-block offsets
//            emit_scatter  instances generated again, ranked with wave64 ballot matching (stable, no atomics), staged
//                          in LDS in bucket order and written as ONE packed word per instance
//                          (id << LB | low tile digit; 4 bytes when P << LB fits 32 bits, else 8)
//   level 2  bucket_hist   blocks are assigned per bucket (table written by emit_scatter's block 0): histogram of the
//                          low digit
//            bucket_scan   one workgroup per bucket: per-block offsets, and -- because (bucket, low digit) IS the tile --
//                          the per-tile [start, end) RANGES directly (no range kernel, no sorted key array)
//            bucket_scatter stable ranking by low digit, ids written in final order = the reference's point_list
//
// HBM traffic: 4R (level-1 write) + 4R + 4R (level-2 hist / scatter reads) + 4R (point list) = 16 R bytes, against
// 6R (emit) + 2 x (2R + 6R + 6R) (two LSD passes on u16 key / u32 id pairs) + 2R (ranges) = 36 R for the previous
// design (sort.hip, still used above 65536 tiles) and >= 144 R for a 64-bit-key LSD sort.  8 -> 6 launches.
#include "gsr_internal.h"
#include "gsr_wave.h"

using namespace gsrw;

namespace {

// (Both ranking kernels run at four waves per SIMD -- 126 / 107 VGPRs -- and spend half of their wave-cycles waiting (SQ_WAIT_ANY).  Forcing five / six
// with amdgpu_waves_per_eu spills 6-25 registers inside the ranking loops: emission 0.053 -> 0.076 ms, tile sort 0.042 -> 0.051 / 0.059 ms,
// profiles/r05_ab_ranking_occupancy.json.)
constexpr int TS_ITEMS = GSR_TS_ITEMS;              // instances per workgroup
constexpr int TS_IPT = TS_ITEMS / WG_THREADS;       // 16 per thread
constexpr int TS_MAXBINS = 256;

template <typename WordT> struct Pack;
template <> struct Pack<uint32_t> {
    static __device__ __forceinline__ uint32_t make(uint32_t id, uint32_t lo, int lb) { return (id << lb) | lo; }
    static __device__ __forceinline__ uint32_t id(uint32_t w, int lb) { return w >> lb; }
    static __device__ __forceinline__ uint32_t lo(uint32_t w, uint32_t mask) { return w & mask; }
};
template <> struct Pack<uint64_t> {
    static __device__ __forceinline__ uint64_t make(uint32_t id, uint32_t lo, int) { return ((uint64_t)id << 32) | lo; }
    static __device__ __forceinline__ uint32_t id(uint64_t w, int) { return (uint32_t)(w >> 32); }
    static __device__ __forceinline__ uint32_t lo(uint64_t w, uint32_t mask) { return (uint32_t)w & mask; }
};

// ---------------------------------------------------------------------------------------------------------------
// Generation of the workgroup's 4096 instances (a load-balanced expansion of the block's Gaussians).
// Wave w owns the contiguous run [w*1024, w*1024+1024) of the block's slots, item r of a lane is slot
// w*1024 + r*64 + lane (so ranking in (r, lane) order is emission order).
// The block's Gaussians [j_lo, j_hi] come from the per-block table the scan kernel writes (first Gaussian + the number of
// instances emitted before it).  Every Gaussian marks the slot of its first instance with its local index; a workgroup
// MAX-SCAN over the 4096 slots then gives every slot its owner (the first version ran a 12-step binary search per
// instance: 960 VALU instructions per lane against ~100 here -- the level-1 kernels were VALU-bound on it).  The owner's
// (first slot, rectangle, id) record is staged in LDS with coalesced loads, so the only dependent global round trips of
// the kernel are table -> staging.
// Returns in tile[r] the tile id, in id[r] the Gaussian id (WANT_ID), vmask bit r = slot < R.
// ---------------------------------------------------------------------------------------------------------------
constexpr int TS_NGCAP = 1024;      // Gaussians per block whose record is staged in LDS (beyond: global fetch)

template <bool WANT_ID>
__device__ __forceinline__ void generate_instances(uint32_t R, int gx, const uint2* __restrict__ block_first,
                                                   const uint32_t* __restrict__ offsets, const uint2* __restrict__ rect_sorted,
                                                   const uint32_t* __restrict__ order, uint16_t* s_own /*[TS_ITEMS]*/,
                                                   uint4* s_g4 /*[TS_NGCAP]*/, uint32_t* wsum /*[4]*/, float4* splats,
                                                   uint32_t (&tile)[TS_IPT], uint32_t (&id)[TS_IPT], uint32_t& vmask) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t b = blockIdx.x;
    const uint2 d0 = block_first[b];
    const uint32_t j_lo = d0.x, base_excl = d0.y, j_hi = block_first[b + 1].x;
    const int nG = (int)(j_hi - j_lo) + 1;                       // <= TS_ITEMS + 1
    const uint32_t b0 = b * (uint32_t)TS_ITEMS;
    uint4* own4 = reinterpret_cast<uint4*>(s_own);               // thread t scans slots [16t, 16t+16) = own4[2t], own4[2t+1]
    own4[2 * tid] = make_uint4(0u, 0u, 0u, 0u);
    own4[2 * tid + 1] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    for (int i = tid; i < nG; i += WG_THREADS) {
        const uint32_t excl = i ? offsets[j_lo + i - 1] : base_excl;
        const uint2 rc = rect_sorted[j_lo + i];
        const uint32_t gid = (WANT_ID || splats) ? order[j_lo + i] : 0u;
        if (i < TS_NGCAP) s_g4[i] = make_uint4(excl, rc.x, rc.y, gid);
        const uint32_t pos = excl > b0 ? excl - b0 : 0u;         // only the first Gaussian can start before the block
        if (pos < (uint32_t)TS_ITEMS) {
            s_own[pos] = (uint16_t)i;
            // first emission index of every Gaussian whose first instance lies in this block -> 4th quad of its splat
            // record (the blend backward writes its per-instance gradient records at emission indices, render_bwd.hip)
            if (splats && excl >= b0) reinterpret_cast<uint32_t*>(splats + (int64_t)gid * 4 + 3)[2] = excl;
        }
    }
    __syncthreads();
    {   // inclusive max-scan of the owner marks over the block's 4096 slots
        const uint4 a0 = own4[2 * tid], a1 = own4[2 * tid + 1];
        const uint32_t a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        uint32_t m[16];
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            run = max(run, a[k] & 0xFFFFu); m[2 * k] = run;
            run = max(run, a[k] >> 16); m[2 * k + 1] = run;
        }
        const uint32_t incl = wave_incl_max_u32(run);
        if (lane == 63) wsum[w] = incl;
        uint32_t pre = dpp_src_u32<0x138, 0xf>(incl);      // wave_shr:1 -- the previous lane's inclusive maximum, 0 for lane 0
        __syncthreads();
#pragma unroll
        for (int k = 0; k < WG_WAVES; ++k)
            if (k < w) pre = max(pre, wsum[k]);
        uint32_t o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = max(pre, m[2 * k]) | (max(pre, m[2 * k + 1]) << 16);
        own4[2 * tid] = make_uint4(o[0], o[1], o[2], o[3]);
        own4[2 * tid + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
    __syncthreads();
    const uint32_t s0 = (uint32_t)w * (64u * TS_IPT) + (uint32_t)lane;     // slot of item 0 inside the block
    vmask = 0;
    auto finish = [&](int r, uint32_t k, const uint4& g) {
        if (WANT_ID) id[r] = g.w;
        const uint32_t minx = g.y & 0xFFFFu, wd = (g.y >> 16) - minx, miny = g.z & 0xFFFFu;
        uint32_t rx;
        const uint32_t ry = div_small((k < R ? k : R - 1u) - g.x, wd ? wd : 1u, rx);
        tile[r] = (miny + ry) * (uint32_t)gx + minx + rx;
    };
    // Two separate loops on a workgroup-uniform condition.  Written as one loop with "i < TS_NGCAP ? LDS record : global
    // fetch" per item, the compiler merged the two sources into a pointer select and FOUR flat_load_dword per item (the LDS
    // record read through the flat aperture, ~15 address instructions and a full s_waitcnt per item, 16 items per thread).
    if (nG <= TS_NGCAP) {
        uint32_t own[TS_IPT];
#pragma unroll
        for (int r = 0; r < TS_IPT; ++r) own[r] = s_own[s0 + (uint32_t)r * 64u];
#pragma unroll
        for (int r = 0; r < TS_IPT; ++r) {
            const uint32_t k = b0 + s0 + (uint32_t)r * 64u;
            if (k < R) vmask |= 1u << r;
            finish(r, k, s_g4[own[r]]);
        }
    } else {            // (more than 1024 Gaussians in 4096 instances: fewer than four tiles each)
#pragma unroll 1
        for (int r = 0; r < TS_IPT; ++r) {
            const uint32_t slot = s0 + (uint32_t)r * 64u;
            const uint32_t k = b0 + slot;
            if (k < R) vmask |= 1u << r;
            const uint32_t i = s_own[slot];
            uint4 g;
            if (i < (uint32_t)TS_NGCAP) {
                g = s_g4[i];
            } else {
                const uint2 rc = rect_sorted[j_lo + i];
                g = make_uint4(offsets[j_lo + i - 1], rc.x, rc.y, WANT_ID ? order[j_lo + i] : 0u);
            }
            finish(r, k, g);
        }
    }
}

// level 1 histogram: hist[d * nblk + block] = instances of the block whose tile id >> lb == d.
// Counted per rectangle ROW, not per instance: the part of a Gaussian's instance run that falls into the block is a
// partial first row, whole rows and a partial last row of its rectangle; every row is a run of consecutive tile ids that
// is cut at the bucket boundaries (usually not at all) and added with one LDS atomic per piece -- ~3.6x fewer items than
// instances on the bench frame and no owner expansion (the instance-wise version spent 23 us, all VALU).
__global__ void __launch_bounds__(WG_THREADS)
emit_hist(uint32_t R, int gx, int lb, int nb1, const uint2* __restrict__ block_first, const uint32_t* __restrict__ offsets,
          const uint2* __restrict__ rect_sorted, uint32_t* __restrict__ hist, int nblk) {
    __shared__ uint32_t h[TS_MAXBINS];
    const int tid = threadIdx.x;
    if (tid < nb1) h[tid] = 0;
    const uint32_t b = blockIdx.x;
    const uint2 d0 = block_first[b];
    const uint32_t j_lo = d0.x, base_excl = d0.y, j_hi = block_first[b + 1].x;
    const int nG = (int)(j_hi - j_lo) + 1;
    const uint32_t b0 = b * (uint32_t)TS_ITEMS;
    const uint32_t b1 = R - b0 < (uint32_t)TS_ITEMS ? R : b0 + (uint32_t)TS_ITEMS;
    __syncthreads();
    for (int i = tid; i < nG; i += WG_THREADS) {
        const uint32_t excl = i ? offsets[j_lo + i - 1] : base_excl, incl = offsets[j_lo + i];
        const uint2 rc = rect_sorted[j_lo + i];
        const uint32_t minx = rc.x & 0xFFFFu, wd = (rc.x >> 16) - minx, miny = rc.y & 0xFFFFu;
        const uint32_t lo = excl > b0 ? excl : b0, hi = incl < b1 ? incl : b1;
        if (hi <= lo || wd == 0u) continue;
        const uint32_t ka = lo - excl, kb = hi - excl;          // the Gaussian's local instances [ka, kb) lie in this block
        uint32_t xa, xl;
        const uint32_t ra = div_small(ka, wd, xa), rb = div_small(kb - 1u, wd, xl);
        for (uint32_t r = ra; r <= rb; ++r) {
            const uint32_t x0 = r == ra ? xa : 0u, x1 = r == rb ? xl + 1u : wd;
            uint32_t t0 = (miny + r) * (uint32_t)gx + minx + x0;
            const uint32_t t1 = t0 + (x1 - x0);
            while (t0 < t1) {                                   // cut the row at bucket boundaries
                const uint32_t d = t0 >> lb, tend = min(t1, (d + 1u) << lb);
                atomicAdd(&h[d], tend - t0);
                t0 = tend;
            }
        }
    }
    __syncthreads();
    if (tid < nb1) hist[(int64_t)tid * nblk + blockIdx.x] = h[tid];
}

// Stable scatter of the workgroup's items by `digit` (< nbins <= 256, `bits` significant bits): returns through LDS
// the items in workgroup-local sorted order (s_word / s_dig) and the global base of every digit (digit_base, already
// offset so that global position = digit_base[d] + local index).  ranks: wave64 ballot matching, no atomics.
template <typename WordT>
__device__ __forceinline__ void local_stable_sort(const WordT (&word)[TS_IPT], uint32_t (&digit)[TS_IPT] /*clobbered*/, uint32_t vmask,
                                                  int bits, int nbins, uint32_t my_digit_base /*thread d: global base of digit d*/,
                                                  uint32_t (*wave_cnt)[TS_MAXBINS], uint32_t* digit_base, uint32_t* wsum,
                                                  WordT* s_word, uint8_t* s_dig) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < nbins) {
        digit_base[tid] = my_digit_base;
#pragma unroll
        for (int k = 0; k < WG_WAVES; ++k) wave_cnt[k][tid] = 0;
    }
    __syncthreads();
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < TS_IPT; ++r) {
        const bool valid = (vmask >> r) & 1u;
        const uint32_t d = digit[r];
        const uint64_t mask = match_digit(d, bits, __ballot(valid));
        const uint32_t prior = valid ? wave_cnt[w][d] : 0u;
        // the item's rank among the wave's items of its digit rides in the upper half of the digit word (rank < 4096)
        digit[r] = d | ((prior + (uint32_t)__popcll(mask & lt_mask)) << 16);
        if (valid && (mask & lt_mask) == 0ull) wave_cnt[w][d] = prior + (uint32_t)__popcll(mask);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {
        uint32_t tot[1] = {0u};
        if (tid < nbins) {
#pragma unroll
            for (int k = 0; k < WG_WAVES; ++k) tot[0] += wave_cnt[k][tid];
        }
        const uint32_t lbase = block_excl_scan<1>(tot, wsum, lane, w);
        if (tid < nbins) {
            uint32_t run = lbase;
#pragma unroll
            for (int k = 0; k < WG_WAVES; ++k) {
                const uint32_t t = wave_cnt[k][tid];
                wave_cnt[k][tid] = run;
                run += t;
            }
            digit_base[tid] -= lbase;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TS_IPT; ++r) {
        if ((vmask >> r) & 1u) {
            const uint32_t d = digit[r] & 0xFFFFu;
            const uint32_t lp = wave_cnt[w][d] + (digit[r] >> 16);
            s_word[lp] = word[r];
            s_dig[lp] = (uint8_t)d;
        }
    }
    __syncthreads();
}

// level 1 scatter (+ first-emission indices for the backward, + the bucket tables of level 2 from block 0)
template <typename WordT>
__global__ void __launch_bounds__(WG_THREADS)
emit_scatter(uint32_t R, int gx, int lb, int hb, const uint2* __restrict__ block_first, const uint32_t* __restrict__ offsets,
             const uint2* __restrict__ rect_sorted, const uint32_t* __restrict__ order, const uint32_t* __restrict__ hist,
             const uint32_t* __restrict__ digit_total, int nblk, WordT* __restrict__ words_out,
             uint32_t* __restrict__ bucket_base /*[nb1+1]*/, uint32_t* __restrict__ blk2_start /*[nb1+1]*/,
             float4* __restrict__ splats /*NULL: inference, no first-emission write*/) {
    // LDS: the generation phase (owner marks, records of the block's Gaussians) and the sorting phase (words staged in
    // bucket order) never overlap in time, so they share one region
    constexpr int GEN_BYTES = TS_ITEMS * 2 + TS_NGCAP * 16;
    constexpr int SORT_BYTES = TS_ITEMS * (int)sizeof(WordT) + TS_ITEMS;
    constexpr int SMEM_BYTES = GEN_BYTES > SORT_BYTES ? GEN_BYTES : SORT_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
    __shared__ uint32_t wave_cnt[WG_WAVES][TS_MAXBINS];
    __shared__ uint32_t digit_base[TS_MAXBINS];
    __shared__ uint32_t wsum[WG_WAVES];
    uint16_t* s_own = reinterpret_cast<uint16_t*>(smem);
    uint4* s_g4 = reinterpret_cast<uint4*>(smem + TS_ITEMS * 2);
    WordT* s_word = reinterpret_cast<WordT*>(smem);
    uint8_t* s_dig = smem + TS_ITEMS * sizeof(WordT);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nb1 = 1 << hb;
    // independent of everything else: requested first
    const uint32_t my_total = tid < nb1 ? digit_total[tid] : 0u;
    const uint32_t my_hist = tid < nb1 ? hist[(int64_t)tid * nblk + blockIdx.x] : 0u;
    uint32_t tile[TS_IPT], id[TS_IPT], vmask;
    generate_instances<true>(R, gx, block_first, offsets, rect_sorted, order, s_own, s_g4, wsum, splats, tile, id, vmask);
    // global base of bucket d for this block = exclusive scan of the bucket totals + instances of earlier blocks
    uint32_t tot[1] = {my_total};
    const uint32_t bbase = block_excl_scan<1>(tot, wsum, lane, w);
    if (blockIdx.x == 0) {
        // level-2 plan: bucket d occupies [bucket_base[d], bucket_base[d+1]) of the word array and gets
        // ceil(size / TS_ITEMS) workgroups starting at blk2_start[d]
        uint32_t nb[1] = {(tot[0] + (uint32_t)TS_ITEMS - 1u) / (uint32_t)TS_ITEMS};
        const uint32_t bstart = block_excl_scan<1>(nb, wsum, lane, w);
        if (tid < nb1) {
            bucket_base[tid] = bbase;
            blk2_start[tid] = bstart;
            if (tid == nb1 - 1) {
                bucket_base[nb1] = bbase + tot[0];
                blk2_start[nb1] = bstart + nb[0];
            }
        }
    }
    WordT word[TS_IPT];
    uint32_t digit[TS_IPT];
    const uint32_t lomask = (1u << lb) - 1u;
#pragma unroll
    for (int r = 0; r < TS_IPT; ++r) {
        const bool valid = (vmask >> r) & 1u;
        word[r] = Pack<WordT>::make(valid ? id[r] : 0u, tile[r] & lomask, lb);
        digit[r] = valid ? (tile[r] >> lb) : 0u;
    }
    // (the barriers at the top of local_stable_sort separate the last read of the generation arrays from the first write
    // of the word staging that shares their LDS)
    local_stable_sort<WordT>(word, digit, vmask, hb, nb1, bbase + my_hist, wave_cnt, digit_base, wsum, s_word, s_dig);
    const uint32_t b0 = blockIdx.x * (uint32_t)TS_ITEMS;
    const uint32_t nvalid = R - b0 < (uint32_t)TS_ITEMS ? R - b0 : (uint32_t)TS_ITEMS;
#pragma unroll
    for (int r = 0; r < TS_IPT; ++r) {
        const uint32_t i = (uint32_t)r * WG_THREADS + tid;
        if (i < nvalid) words_out[digit_base[s_dig[i]] + i] = s_word[i];
    }
}


// level 2: workgroup B2 -> (bucket h, chunk c).  Returns false when B2 is past the last planned workgroup.
// One memory round trip: thread t < nb1 fetches both table rows of bucket t and the (single) matching thread publishes the
// bucket, its word range and its first workgroup through LDS (the first version looked the bucket up, then read the two
// tables again for the match: two dependent trips in front of every level-2 kernel's own loads).
// A launch covers the buckets [h0, h1) (the whole table, or one band of a band-pipelined frame: gsr_api.cpp): its workgroup 0 is the first
// workgroup of bucket h0, B2 = blockIdx.x + blk2_start[h0] is the workgroup's index in the frame's level-2 plan (hist2 rows are indexed by it).
__device__ __forceinline__ bool bucket_block(int nb1, int h0, int h1, const uint32_t* __restrict__ bucket_base, const uint32_t* __restrict__ blk2_start,
                                             int* s_h /*[5]*/, uint32_t& h, uint32_t& begin, uint32_t& end, uint32_t& B2) {
    const int tid = threadIdx.x;
    B2 = blockIdx.x + blk2_start[h0];      // (wave-uniform load beside the table rows below)
    const int t = tid < nb1 ? tid : 0;
    const uint32_t b0 = blk2_start[t], b1 = blk2_start[t + 1], w0 = bucket_base[t], w1 = bucket_base[t + 1];
    if (tid == 0) s_h[0] = -1;
    __syncthreads();
    if (tid >= h0 && tid < h1 && b0 <= B2 && B2 < b1) {      // at most one bucket matches
        s_h[0] = tid; s_h[1] = (int)b0; s_h[2] = (int)w0; s_h[3] = (int)w1; s_h[4] = (int)b1;
    }
    __syncthreads();
    const int hh = s_h[0];
    if (hh < 0) return false;
    h = (uint32_t)hh;
    const uint32_t c = B2 - (uint32_t)s_h[1];
    begin = (uint32_t)s_h[2] + c * (uint32_t)TS_ITEMS;
    const uint32_t bend = (uint32_t)s_h[3];
    end = bend - begin < (uint32_t)TS_ITEMS ? bend : begin + (uint32_t)TS_ITEMS;
    return true;
}

// ranges_of_empty != NULL (level 2 without its scan kernel): workgroup 0 also writes the (0, 0) ranges of the buckets that hold no
// instance at all -- nobody else visits them
template <typename WordT>
__global__ void __launch_bounds__(WG_THREADS)
bucket_hist(int lb, int hb, int h0, int h1, const WordT* __restrict__ words, const uint32_t* __restrict__ bucket_base,
            const uint32_t* __restrict__ blk2_start, uint32_t* __restrict__ hist2 /*[blocks][nb2]*/, uint2* __restrict__ ranges_of_empty,
            int n_tiles) {
    __shared__ uint32_t hcnt[WG_WAVES][TS_MAXBINS];
    __shared__ int s_h[5];
    const int tid = threadIdx.x, w = tid >> 6;
    const int nb1 = 1 << hb, nb2 = 1 << lb;
    if (ranges_of_empty && blockIdx.x == 0 && tid >= h0 && tid < h1 && blk2_start[tid] == blk2_start[tid + 1]) {
        for (int t = 0; t < nb2; ++t) {
            const int tile = (tid << lb) | t;
            if (tile < n_tiles) ranges_of_empty[tile] = make_uint2(0u, 0u);
        }
    }
    uint32_t h, begin, end, B2;
    if (!bucket_block(nb1, h0, h1, bucket_base, blk2_start, s_h, h, begin, end, B2)) return;
    if (tid < nb2) {
#pragma unroll
        for (int k = 0; k < WG_WAVES; ++k) hcnt[k][tid] = 0;
    }
    __syncthreads();
    const uint32_t lomask = (uint32_t)nb2 - 1u;
    // inside a bucket consecutive words belong to unrelated tiles: plain LDS atomics, one table per wave
    WordT wd[TS_IPT];
#pragma unroll
    for (int r = 0; r < TS_IPT; ++r) {          // all loads first: 16 independent requests in flight per lane
        const uint32_t i = begin + (uint32_t)r * WG_THREADS + tid;
        wd[r] = i < end ? words[i] : (WordT)0;
    }
#pragma unroll
    for (int r = 0; r < TS_IPT; ++r) {
        const uint32_t i = begin + (uint32_t)r * WG_THREADS + tid;
        if (i < end) atomicAdd(&hcnt[w][Pack<WordT>::lo(wd[r], lomask)], 1u);
    }
    __syncthreads();
    if (tid < nb2) hist2[(int64_t)B2 * nb2 + tid] = hcnt[0][tid] + hcnt[1][tid] + hcnt[2][tid] + hcnt[3][tid];
}

// one workgroup per bucket: per-block exclusive offsets of every low digit (in place), the per-tile bases and the
// tile ranges (tile = bucket << lb | low digit)
__global__ void __launch_bounds__(WG_THREADS)
bucket_scan(int lb, int h0, int n_tiles, const uint32_t* __restrict__ bucket_base, const uint32_t* __restrict__ blk2_start,
            uint32_t* __restrict__ hist2, uint32_t* __restrict__ tile_base /*[nb1 * nb2]*/, uint2* __restrict__ ranges) {
    __shared__ uint32_t wsum[WG_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nb2 = 1 << lb;
    const uint32_t h = blockIdx.x + (uint32_t)h0;
    const uint32_t s = blk2_start[h], e = blk2_start[h + 1];
    uint32_t tot[1] = {0u};
    if (tid < nb2) {
        uint32_t run = 0;
        uint32_t* col = hist2 + tid;
        uint32_t b = s;
        for (; b + 8 <= e; b += 8) {        // eight independent loads in flight, then the eight prefixes
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = col[(int64_t)(b + k) * nb2];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                col[(int64_t)(b + k) * nb2] = run;
                run += v[k];
            }
        }
        for (; b < e; ++b) {
            const uint32_t v = col[(int64_t)b * nb2];
            col[(int64_t)b * nb2] = run;
            run += v;
        }
        tot[0] = run;
    }
    const uint32_t excl = block_excl_scan<1>(tot, wsum, lane, w);
    if (tid < nb2) {
        const uint32_t base = bucket_base[h] + excl;
        tile_base[h * (uint32_t)nb2 + tid] = base;
        const uint32_t tile = (h << lb) | (uint32_t)tid;
        if (tile < (uint32_t)n_tiles) ranges[tile] = tot[0] ? make_uint2(base, base + tot[0]) : make_uint2(0u, 0u);
    }
}

// FUSED_SCAN: there was no bucket_scan launch -- hist2 holds the raw per-workgroup counts and every workgroup sums, for each low
// digit, the counts of its bucket's workgroups (all of them: the per-tile bases; those before it: its own offsets) in one round
// of independent loads.  The bucket's first workgroup writes the tile ranges.  Chosen by the host while a bucket has few
// workgroups (the slab a workgroup reads is workgroups-per-bucket x nb2 x 4 bytes).
template <typename WordT, bool FUSED_SCAN>
__global__ void __launch_bounds__(WG_THREADS)
bucket_scatter(int lb, int hb, int h0, int h1, const WordT* __restrict__ words, const uint32_t* __restrict__ bucket_base,
               const uint32_t* __restrict__ blk2_start, const uint32_t* __restrict__ hist2,
               const uint32_t* __restrict__ tile_base, uint32_t* __restrict__ point_list, uint2* __restrict__ ranges, int n_tiles) {
    __shared__ uint32_t wave_cnt[WG_WAVES][TS_MAXBINS];
    __shared__ uint32_t digit_base[TS_MAXBINS];
    __shared__ uint32_t wsum[WG_WAVES];
    __shared__ uint32_t s_id[TS_ITEMS];
    __shared__ uint8_t s_dig[TS_ITEMS];
    __shared__ int s_h[5];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nb1 = 1 << hb, nb2 = 1 << lb;
    uint32_t h, begin, end, B2;
    if (!bucket_block(nb1, h0, h1, bucket_base, blk2_start, s_h, h, begin, end, B2)) return;
    const uint32_t lomask = (uint32_t)nb2 - 1u;
    uint32_t id[TS_IPT], digit[TS_IPT], vmask = 0;
    const uint32_t wbase = begin + (uint32_t)w * (64u * TS_IPT) + (uint32_t)lane;
#pragma unroll
    for (int r = 0; r < TS_IPT; ++r) {
        const uint32_t i = wbase + (uint32_t)r * 64u;
        const bool valid = i < end;
        const WordT wd = valid ? words[i] : (WordT)0;
        id[r] = Pack<WordT>::id(wd, lb);
        digit[r] = Pack<WordT>::lo(wd, lomask);
        if (valid) vmask |= 1u << r;
    }
    uint32_t my_base = 0u;
    if (FUSED_SCAN) {
        const uint32_t bs = (uint32_t)s_h[1], be = (uint32_t)s_h[4];
        const int G = WG_THREADS >> lb;                      // thread = (low digit t, 1 / G of the bucket's workgroups)
        const int t = tid & (nb2 - 1), g = tid >> lb;
        uint32_t tot = 0, pre = 0;
        uint32_t b = bs + (uint32_t)g;
        for (; b + 3u * (uint32_t)G < be; b += 4u * (uint32_t)G) {      // four independent loads in flight
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = hist2[(int64_t)(b + (uint32_t)(k * G)) * nb2 + t];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tot += v[k];
                if (b + (uint32_t)(k * G) < B2) pre += v[k];
            }
        }
        for (; b < be; b += (uint32_t)G) {
            const uint32_t v = hist2[(int64_t)b * nb2 + t];
            tot += v;
            if (b < B2) pre += v;
        }
        uint32_t* s_tot = wave_cnt[0];      // scratch until local_stable_sort takes the table over (behind its own barrier)
        uint32_t* s_pre = wave_cnt[1];
        s_tot[tid] = tot;
        s_pre[tid] = pre;
        __syncthreads();
        uint32_t tt[1] = {0u};
        uint32_t pp = 0;
        if (tid < nb2) {
            for (int gg = 0; gg < G; ++gg) {
                tt[0] += s_tot[gg * nb2 + tid];
                pp += s_pre[gg * nb2 + tid];
            }
        }
        const uint32_t excl = block_excl_scan<1>(tt, wsum, lane, w);
        if (tid < nb2) {
            const uint32_t base = (uint32_t)s_h[2] + excl;      // s_h[2] = bucket_base[h]
            my_base = base + pp;
            const uint32_t tile = (h << lb) | (uint32_t)tid;
            if (B2 == bs && tile < (uint32_t)n_tiles) ranges[tile] = tt[0] ? make_uint2(base, base + tt[0]) : make_uint2(0u, 0u);
        }
    } else {
        my_base = tid < nb2 ? tile_base[h * (uint32_t)nb2 + tid] + hist2[(int64_t)B2 * nb2 + tid] : 0u;
    }
    local_stable_sort<uint32_t>(id, digit, vmask, lb, nb2, my_base, wave_cnt, digit_base, wsum, s_id, s_dig);
    const uint32_t nvalid = end - begin;
#pragma unroll
    for (int r = 0; r < TS_IPT; ++r) {
        const uint32_t i = (uint32_t)r * WG_THREADS + tid;
        if (i < nvalid) point_list[digit_base[s_dig[i]] + i] = s_id[i];
    }
}

// Fallback for frames whose instance count exceeds the capacity of the per-block table the scan kernel fills
// (more than 64 tiles per Gaussian on average): the same table, sized by R, from the finished offsets.
__global__ void __launch_bounds__(WG_THREADS)
fill_block_first(int P, const uint32_t* __restrict__ offsets, uint2* __restrict__ block_first, uint32_t cap) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < P; j += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t incl = offsets[j], excl = j ? offsets[j - 1] : 0u;
        if (incl == excl) continue;
        for (uint32_t k = (excl + TS_ITEMS - 1u) / TS_ITEMS; k <= (incl - 1u) / TS_ITEMS; ++k)
            if (k < cap) block_first[k] = make_uint2((uint32_t)j, excl);
        if (j == (int64_t)P - 1 || offsets[j + 1] == incl) {
            const uint32_t k = (incl + TS_ITEMS - 1u) / TS_ITEMS;
            if (k < cap) block_first[k] = make_uint2((uint32_t)j, excl);
        }
    }
}

}  // namespace

void gsr_tile_sort_plan(int n_tiles, int P, GsrTileSortPlan* plan) {
    int nbits = 1;
    while (nbits < 32 && (1ll << nbits) < (long long)n_tiles) ++nbits;
    plan->fused = n_tiles <= 65536;
    plan->lb = (nbits + 1) / 2;
    plan->hb = nbits - plan->lb;
    plan->word64 = ((unsigned long long)(P > 0 ? P : 1) << plan->lb) > (1ull << 32);
}

void gsr_launch_fill_block_first(int P, const uint32_t* offsets, uint2* block_first, uint32_t cap, hipStream_t st) {
    int64_t nb = ((int64_t)P + WG_THREADS - 1) / WG_THREADS;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(fill_block_first, dim3((int)nb), dim3(WG_THREADS), 0, st, P, offsets, block_first, cap);
}

void gsr_launch_tile_sort_level1(const GsrTileSortPlan& plan, int64_t R, int gx, const uint2* block_first,
                                 const uint32_t* offsets, const uint2* rect_sorted, const uint32_t* order, void* words,
                                 uint32_t* hist1, uint32_t* digit_total, uint32_t* bucket_base, uint32_t* blk2_start,
                                 float4* splats, hipStream_t st) {
    const int nblk = (int)((R + TS_ITEMS - 1) / TS_ITEMS);
    const int nb1 = 1 << plan.hb;
    const uint32_t R32 = (uint32_t)R;
    hipLaunchKernelGGL(emit_hist, dim3(nblk), dim3(WG_THREADS), 0, st, R32, gx, plan.lb, nb1, block_first, offsets, rect_sorted,
                       hist1, nblk);
    gsr_launch_rs_scan(hist1, nblk, nb1, digit_total, st);
    if (plan.word64)
        hipLaunchKernelGGL(emit_scatter<uint64_t>, dim3(nblk), dim3(WG_THREADS), 0, st, R32, gx, plan.lb, plan.hb, block_first,
                           offsets, rect_sorted, order, hist1, digit_total, nblk, (uint64_t*)words, bucket_base, blk2_start, splats);
    else
        hipLaunchKernelGGL(emit_scatter<uint32_t>, dim3(nblk), dim3(WG_THREADS), 0, st, R32, gx, plan.lb, plan.hb, block_first,
                           offsets, rect_sorted, order, hist1, digit_total, nblk, (uint32_t*)words, bucket_base, blk2_start, splats);
}

int g_level2_scan_mode = 0;      // option level2_scan_mode: 0 = automatic, 1 = separate bucket_scan launch, 2 = folded into bucket_scatter

void gsr_set_level2_scan_mode(int v) { g_level2_scan_mode = v; }

void gsr_launch_tile_sort_level2(const GsrTileSortPlan& plan, int64_t R, int n_tiles, const void* words, uint32_t* point_list,
                                 const uint32_t* bucket_base, const uint32_t* blk2_start, uint32_t* hist2, uint32_t* tile_base,
                                 uint2* ranges, hipStream_t st, int h0, int h1) {
    const int nblk = (int)((R + TS_ITEMS - 1) / TS_ITEMS);
    const int nb1 = 1 << plan.hb;
    if (h1 <= 0 || h1 > nb1) h1 = nb1;      // default: every bucket
    if (h0 < 0) h0 = 0;
    if (h0 >= h1) return;
    // upper bound of the launch's level-2 workgroups (every bucket rounds up once); a band's share of the instances is not known to the host,
    // so a band launch keeps the frame's bound and its surplus workgroups -- the LAST of its grid -- leave after one load
    const int nblk2 = nblk + (h1 - h0);
    // without the scan kernel every level-2 workgroup reads its bucket's slab of the count table: fine while a bucket has a few
    // dozen workgroups (bench frame: 30), too much traffic when it has hundreds (6 M Gaussians: 180)
    const bool fused = g_level2_scan_mode == 2 || (g_level2_scan_mode == 0 && nblk <= 64 * nb1);
    uint2* roe = fused ? ranges : nullptr;
    if (plan.word64)
        hipLaunchKernelGGL(bucket_hist<uint64_t>, dim3(nblk2), dim3(WG_THREADS), 0, st, plan.lb, plan.hb, h0, h1, (const uint64_t*)words,
                           bucket_base, blk2_start, hist2, roe, n_tiles);
    else
        hipLaunchKernelGGL(bucket_hist<uint32_t>, dim3(nblk2), dim3(WG_THREADS), 0, st, plan.lb, plan.hb, h0, h1, (const uint32_t*)words,
                           bucket_base, blk2_start, hist2, roe, n_tiles);
    if (!fused)
        hipLaunchKernelGGL(bucket_scan, dim3(h1 - h0), dim3(WG_THREADS), 0, st, plan.lb, h0, n_tiles, bucket_base, blk2_start, hist2, tile_base,
                           ranges);
#define GSR_L2_SCATTER(WORD_, FUSED_)                                                                                               \
    hipLaunchKernelGGL((bucket_scatter<WORD_, FUSED_>), dim3(nblk2), dim3(WG_THREADS), 0, st, plan.lb, plan.hb, h0, h1, (const WORD_*)words,  \
                       bucket_base, blk2_start, hist2, tile_base, point_list, ranges, n_tiles)
    if (plan.word64) { if (fused) GSR_L2_SCATTER(uint64_t, true); else GSR_L2_SCATTER(uint64_t, false); }
    else { if (fused) GSR_L2_SCATTER(uint32_t, true); else GSR_L2_SCATTER(uint32_t, false); }
#undef GSR_L2_SCATTER
}
