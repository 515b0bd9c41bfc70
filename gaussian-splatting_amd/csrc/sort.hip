// Stable LSD radix sort of (u32 key, u32 value) pairs for gfx950 -- replaces the CUB
// DeviceRadixSort::SortPairs call of the reference rasterizer.
//
// How it is used (binning design): instead of one 64-bit (tile|depth) sort over all R
// instances (6+ eight-bit passes over 12-byte pairs), the P Gaussians are sorted by 32-bit depth once
// (4 passes over P pairs), instances are emitted in depth order, and the R instances are then sorted
// STABLY by tile id only (2 passes for <= 65536 tiles).  Stability makes the final order identical to
// the reference's (tile, depth, emission order).
//
// One pass = three launches:
//   rs_hist    : per-workgroup 256-bin digit histogram                  (reads keys)
//   rs_scan    : one workgroup per digit scans its row over workgroups  (tiny)
//   rs_scatter : wave64 ballot-match ranking, stable, direct scatter    (reads keys+vals, writes both)
// Ranking idiom: each wave owns a contiguous run of the workgroup's items and walks it 64 at a time; the
// 64-bit match mask of a lane's digit comes from 8 ballots; rank = running per-wave digit count (LDS) +
// popcount(mask & lanes_below); the lowest matching lane bumps the count.  No atomics, deterministic.
#include "gsr_internal.h"

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;

template <int IPT>
__global__ void __launch_bounds__(RS_THREADS)
rs_hist(const uint32_t* __restrict__ keys, int64_t n, int shift, uint32_t* __restrict__ block_hist, int nblocks) {
    __shared__ uint32_t h[RS_WAVES][256];
    const int tid = threadIdx.x, w = tid >> 6;
#pragma unroll
    for (int i = 0; i < RS_WAVES; ++i) h[i][tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * (RS_THREADS * IPT);
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int64_t idx = base + i * RS_THREADS + tid;
        if (idx < n) atomicAdd(&h[w][(keys[idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < RS_WAVES; ++i) s += h[i][tid];
    block_hist[(int64_t)tid * nblocks + blockIdx.x] = s;
}

// grid = 256 (one workgroup per digit).  In place: row[b] <- sum_{b' < b} row[b'];  digit_total[d] = row sum.
__global__ void __launch_bounds__(RS_THREADS)
rs_scan(uint32_t* __restrict__ block_hist, int nblocks, uint32_t* __restrict__ digit_total) {
    __shared__ uint32_t wsum[RS_WAVES];
    __shared__ uint32_t carry_s;
    uint32_t* row = block_hist + (int64_t)blockIdx.x * nblocks;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += RS_THREADS) {
        const int i = base + tid;
        const uint32_t v = i < nblocks ? row[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t wbase = 0;
#pragma unroll
        for (int k = 0; k < RS_WAVES; ++k)
            if (k < w) wbase += wsum[k];
        const uint32_t carry = carry_s;
        if (i < nblocks) row[i] = carry + wbase + incl - v;
        __syncthreads();
        if (tid == RS_THREADS - 1) carry_s = carry + wbase + incl;
        __syncthreads();
    }
    if (tid == 0) digit_total[blockIdx.x] = carry_s;
}

template <int IPT>
__global__ void __launch_bounds__(RS_THREADS)
rs_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
           uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int64_t n, int shift,
           const uint32_t* __restrict__ block_hist, const uint32_t* __restrict__ digit_total, int nblocks) {
    __shared__ uint32_t wave_cnt[RS_WAVES][256];
    __shared__ uint32_t digit_base[256];
    __shared__ uint32_t wsum[RS_WAVES];
    __shared__ uint32_t s_key[RS_THREADS * IPT];
    __shared__ uint32_t s_val[RS_THREADS * IPT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

    // digit_base[d] = (exclusive scan of digit totals)[d] + (keys with digit d in earlier workgroups)
    {
        const uint32_t v = digit_total[tid];
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[w] = incl;
#pragma unroll
        for (int i = 0; i < RS_WAVES; ++i) wave_cnt[i][tid] = 0;
        __syncthreads();
        uint32_t wbase = 0;
#pragma unroll
        for (int k = 0; k < RS_WAVES; ++k)
            if (k < w) wbase += wsum[k];
        digit_base[tid] = wbase + incl - v + block_hist[(int64_t)tid * nblocks + blockIdx.x];
    }
    __syncthreads();

    const int64_t wave_base = (int64_t)blockIdx.x * (RS_THREADS * IPT) + (int64_t)w * (64 * IPT);
    uint32_t key[IPT], val[IPT], rank[IPT];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int64_t idx = wave_base + r * 64 + lane;
        const bool valid = idx < n;
        key[r] = valid ? keys_in[idx] : 0u;
        val[r] = valid ? vals_in[idx] : 0u;
    }
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int64_t idx = wave_base + r * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = (key[r] >> shift) & 255u;
        uint64_t mask = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            mask &= bit ? bal : ~bal;
        }
        const uint32_t prior = wave_cnt[w][d];
        rank[r] = prior + (uint32_t)__popcll(mask & lt_mask);
        if (valid && (mask & lt_mask) == 0ull) wave_cnt[w][d] = prior + (uint32_t)__popcll(mask);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // Local (in-workgroup) sorted position of every item: exclusive prefix over digits of the workgroup's digit
    // counts, then over waves within a digit.  Items are first scattered into LDS in that order and then written
    // out by consecutive threads, so every digit's run leaves the CU as one contiguous, coalesced store burst
    // instead of 4-byte stores scattered over 256 destinations (measured 104 us -> see profiles/ for this pass).
    {
        uint32_t tot = 0;
#pragma unroll
        for (int k = 0; k < RS_WAVES; ++k) tot += wave_cnt[k][tid];
        uint32_t incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        __syncthreads();                 // wsum is re-used (its first use finished before the previous barrier)
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t wbase = 0;
#pragma unroll
        for (int k = 0; k < RS_WAVES; ++k)
            if (k < w) wbase += wsum[k];
        const uint32_t lbase = wbase + incl - tot;      // first local slot of digit `tid`
        uint32_t run = lbase;
#pragma unroll
        for (int k = 0; k < RS_WAVES; ++k) {
            const uint32_t t = wave_cnt[k][tid];
            wave_cnt[k][tid] = run;
            run += t;
        }
        digit_base[tid] -= lbase;                       // global position = digit_base[d] + local slot
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int64_t idx = wave_base + r * 64 + lane;
        if (idx < n) {
            const uint32_t d = (key[r] >> shift) & 255u;
            const uint32_t lp = wave_cnt[w][d] + rank[r];
            s_key[lp] = key[r];
            s_val[lp] = val[r];
        }
    }
    __syncthreads();
    const int64_t block_base = (int64_t)blockIdx.x * (RS_THREADS * IPT);
    const uint32_t nvalid = (uint32_t)((n - block_base) < (int64_t)(RS_THREADS * IPT) ? (n - block_base) : (RS_THREADS * IPT));
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const uint32_t i = (uint32_t)r * RS_THREADS + tid;
        if (i < nvalid) {
            const uint32_t k = s_key[i];
            const uint32_t pos = digit_base[(k >> shift) & 255u] + i;
            keys_out[pos] = k;
            vals_out[pos] = s_val[i];
        }
    }
}

template <int IPT>
void sort_pass(uint32_t* kin, uint32_t* vin, uint32_t* kout, uint32_t* vout, int64_t n, int shift, uint32_t* hist,
               uint32_t* digit_total, int nblocks, hipStream_t st) {
    hipLaunchKernelGGL(rs_hist<IPT>, dim3(nblocks), dim3(RS_THREADS), 0, st, kin, n, shift, hist, nblocks);
    hipLaunchKernelGGL(rs_scan, dim3(256), dim3(RS_THREADS), 0, st, hist, nblocks, digit_total);
    hipLaunchKernelGGL(rs_scatter<IPT>, dim3(nblocks), dim3(RS_THREADS), 0, st, kin, vin, kout, vout, n, shift, hist,
                       digit_total, nblocks);
}

}  // namespace

int gsr_radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], int64_t n, int nbits, uint32_t* hist,
                         uint32_t* digit_total, bool small_blocks, hipStream_t st) {
    int cur = 0;
    if (n <= 0) return cur;
    const int nblocks = (int)gsr_sort_blocks(n, small_blocks);
    for (int shift = 0; shift < nbits; shift += 8) {
        if (small_blocks)
            sort_pass<GSR_SORT_ITEMS_SMALL / RS_THREADS>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, hist,
                                                         digit_total, nblocks, st);
        else
            sort_pass<GSR_SORT_ITEMS / RS_THREADS>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, hist,
                                                   digit_total, nblocks, st);
        cur ^= 1;
    }
    return cur;
}
