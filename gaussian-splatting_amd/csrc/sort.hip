// Stable LSD radix sort of (u32 key, u32 value) pairs for gfx950 -- replaces the CUB
// DeviceRadixSort::SortPairs call of the reference rasterizer.
//
// How it is used (binning design): instead of one 64-bit (tile|depth) sort over all R
// instances (6+ eight-bit passes over 12-byte pairs), the P Gaussians are sorted by 32-bit depth once
// (4 passes of 8 bits over P pairs; 3 x 11 bits is implemented but measured slower), instances are emitted in depth
// order, and the R instances are then sorted STABLY by tile id only (2 passes of ceil(bits/2) bits for <= 65536 tiles).  Stability makes the
// final order identical to the reference's (tile, depth, emission order).
//
// One pass = three launches (a chained-scan "onesweep" pass was built and rejected in round 2 -- 31-50 us per pass against
// 22 us: a cross-workgroup hop costs 1-3 us under load on this part and the look-back chain is serial; git history):
//   rs_hist    : per-workgroup digit histogram                          (reads keys)
//   rs_scan    : one workgroup per digit scans its row over workgroups  (tiny)
//   rs_scatter : wave64 ballot-match ranking, stable; items are re-ordered in LDS first so that every
//                digit's run leaves the CU as one contiguous store burst (reads keys+vals, writes both)
// Ranking idiom: each wave owns a contiguous run of the workgroup's items and walks it 64 at a time; the
// 64-bit match mask of a lane's digit comes from BITS ballots; rank = running per-wave digit count (LDS) +
// popcount(mask & lanes_below); the lowest matching lane bumps the count.  No atomics, deterministic.
#include "gsr_internal.h"
#include "gsr_wave.h"      // (match_digit)

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;

// DPP scan (see gsr_wave.h): row_shr 1/2/4/8 inside the 16-lane rows, row_bcast:15 / :31 across them; no ds_bpermute
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_src_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int /*lane*/) {
    v += dpp_src_u32<0x111, 0xf>(v);
    v += dpp_src_u32<0x112, 0xf>(v);
    v += dpp_src_u32<0x114, 0xf>(v);
    v += dpp_src_u32<0x118, 0xf>(v);
    v += dpp_src_u32<0x142, 0xa>(v);
    v += dpp_src_u32<0x143, 0xc>(v);
    return v;
}

template <typename KeyT, int IPT, int BITS>
__global__ void __launch_bounds__(RS_THREADS)
rs_hist(const KeyT* __restrict__ keys, int64_t n, int shift, uint32_t* __restrict__ block_hist, int nblocks) {
    constexpr int NB = 1 << BITS;
    constexpr int NCOPY = (NB <= 512) ? RS_WAVES : 1;     // per-wave copies only when they are cheap (<= 8 KB)
    __shared__ uint32_t h[NCOPY][NB];
    const int tid = threadIdx.x, w = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * (RS_THREADS * IPT);
    // a histogram does not care about item order: every thread takes its keys in 16-byte (or IPT-key) vectors, so one
    // wave-wide load moves 1 KB instead of the 128 B of a 16-bit scalar load
    constexpr int VB = (IPT * (int)sizeof(KeyT) >= 16) ? 16 : IPT * (int)sizeof(KeyT);
    constexpr int VEC = VB / (int)sizeof(KeyT), NV = IPT / VEC, NW = VB / 4;
    static_assert(VB >= 4 && NV * VEC == IPT, "vector layout");
    uint32_t word[NV][NW];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int64_t e0 = base + ((int64_t)v * RS_THREADS + tid) * VEC;
        if (e0 + VEC <= n) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(keys + e0);
            if (NW == 4) {
                const uint4 t = *reinterpret_cast<const uint4*>(src);
                word[v][0] = t.x; word[v][1] = t.y; word[v][2 % NW] = t.z; word[v][3 % NW] = t.w;
            } else if (NW == 2) {
                const uint2 t = *reinterpret_cast<const uint2*>(src);
                word[v][0] = t.x; word[v][1 % NW] = t.y;
            } else {
                word[v][0] = src[0];
            }
        } else {
#pragma unroll
            for (int j = 0; j < NW; ++j) word[v][j] = 0u;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (e0 + j < n) {
                    const uint32_t k = (uint32_t)keys[e0 + j];
                    if (sizeof(KeyT) == 2) word[v][j / 2] |= k << (16 * (j & 1));
                    else word[v][j % NW] = k;
                }
            }
        }
    }
    // (the keys are requested before the histogram is cleared: their trip overlaps the clear and its barrier)
    for (int d = tid; d < NB; d += RS_THREADS)
#pragma unroll
        for (int i = 0; i < NCOPY; ++i) h[i][d] = 0;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int64_t e0 = base + ((int64_t)v * RS_THREADS + tid) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const bool valid = e0 + j < n;
            const uint32_t k = sizeof(KeyT) == 2 ? ((word[v][j / 2] >> (16 * (j & 1))) & 0xFFFFu) : word[v][j % NW];
            const uint32_t d = valid ? ((k >> shift) & (NB - 1)) : 0u;
            // plain LDS atomics: matching equal digits across the wave with ballots first (one add per group) measured
            // SLOWER (25 vs 14 us on 11.3 M 16-bit keys) -- the kernel waits on memory, not on the LDS adder
            if (valid) atomicAdd(&h[NCOPY == 1 ? 0 : w][d], 1u);
        }
    }
    __syncthreads();
    for (int d = tid; d < NB; d += RS_THREADS) {
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < NCOPY; ++i) s += h[i][d];
        block_hist[(int64_t)d * nblocks + blockIdx.x] = s;
    }
}

// grid = number of digits, one workgroup per digit row (one count per workgroup of the pass).  Every thread owns a
// CONTIGUOUS chunk of the row: chunk sums -> one workgroup scan of the 256 sums -> second sweep writes the exclusive
// prefixes.  Two rounds of independent loads instead of nblocks/256 dependent trips (the trip version measured 6.4 us
// per launch, almost all of it load latency).  In place: row[b] <- sum_{b' < b} row[b'];  digit_total[d] = row sum.
__global__ void __launch_bounds__(RS_THREADS)
rs_scan(uint32_t* __restrict__ block_hist, int nblocks, uint32_t* __restrict__ digit_total) {
    __shared__ uint32_t wsum[RS_WAVES];
    uint32_t* row = block_hist + (int64_t)blockIdx.x * nblocks;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int chunk = (nblocks + RS_THREADS - 1) / RS_THREADS;
    const int lo = min(tid * chunk, nblocks), hi = min(lo + chunk, nblocks);
    // rows of up to 4096 workgroups (the depth sort of <= 4 M Gaussians, 16 M instances at 4096 per workgroup) stay in
    // registers between the two sweeps: ONE round of loads instead of two (each round is a ~1 us trip, the counts were
    // just written by other CUs)
    constexpr int KEEP = 16;
    const bool keep = chunk <= KEEP;
    uint32_t v[KEEP];
    uint32_t sum = 0;
    if (keep) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) v[k] = (lo + k < hi) ? row[lo + k] : 0u;
#pragma unroll
        for (int k = 0; k < KEEP; ++k) sum += v[k];
    } else {
#pragma unroll 4
        for (int i = lo; i < hi; ++i) sum += row[i];
    }
    const uint32_t incl = wave_incl_scan_u32(sum, lane);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k)
        if (k < w) run += wsum[k];
    if (tid == RS_THREADS - 1) digit_total[blockIdx.x] = run + sum;
    if (keep) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            if (lo + k < hi) row[lo + k] = run;
            run += v[k];
        }
    } else {
#pragma unroll 4
        for (int i = lo; i < hi; ++i) {
            const uint32_t t = row[i];
            row[i] = run;
            run += t;
        }
    }
}

// Exclusive scan of NB values held DPT per thread (thread t owns digits t*DPT .. t*DPT+DPT-1); returns the exclusive
// prefix of the thread's first digit.  wsum is a 4-entry LDS scratch; contains two barriers.
template <int DPT>
__device__ __forceinline__ uint32_t block_excl_scan(const uint32_t (&v)[DPT], uint32_t* wsum, int lane, int w) {
    uint32_t tsum = 0;
#pragma unroll
    for (int i = 0; i < DPT; ++i) tsum += v[i];
    const uint32_t incl = wave_incl_scan_u32(tsum, lane);
    __syncthreads();
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t wbase = 0;
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k)
        if (k < w) wbase += wsum[k];
    return wbase + incl - tsum;
}

// rect / rect_sorted (depth sort's last pass only, else NULL): the value is a Gaussian id whose tile rectangle is gathered
// into the sorted order here -- the 8-byte random gather the scan kernel would otherwise run as a pass of its own.
template <typename KeyT, int IPT, int BITS>
__global__ void __launch_bounds__(RS_THREADS)
rs_scatter(const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
           KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int64_t n, int shift,
           const uint32_t* __restrict__ block_hist, const uint32_t* __restrict__ digit_total, int nblocks,
           const uint2* __restrict__ rect, uint2* __restrict__ rect_sorted) {
    constexpr int NB = 1 << BITS;
    constexpr int DPT = (NB + RS_THREADS - 1) / RS_THREADS;   // digits per thread (1 for <= 256 bins, 8 for 2048)
    __shared__ uint32_t wave_cnt[RS_WAVES][NB];
    __shared__ uint32_t digit_base[NB];
    __shared__ uint32_t wsum[RS_WAVES];
    __shared__ KeyT s_key[RS_THREADS * IPT];
    __shared__ uint32_t s_val[RS_THREADS * IPT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

    // the workgroup's items are requested first: their trip overlaps the digit-base prologue (loads + two barriers) below
    const int64_t wave_base = (int64_t)blockIdx.x * (RS_THREADS * IPT) + (int64_t)w * (64 * IPT);
    uint32_t key[IPT], val[IPT], rank[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int64_t idx = wave_base + r * 64 + lane;
        const bool valid = idx < n;
        key[r] = valid ? (uint32_t)keys_in[idx] : 0u;
        val[r] = valid ? vals_in[idx] : 0u;
    }

    // digit_base[d] = (exclusive scan of digit totals)[d] + (keys with digit d in earlier workgroups)
    {
        uint32_t v[DPT], bh[DPT];
#pragma unroll
        for (int i = 0; i < DPT; ++i) {
            const int d = tid * DPT + i;
            v[i] = d < NB ? digit_total[d] : 0u;
            bh[i] = block_hist[(int64_t)(d < NB ? d : 0) * nblocks + blockIdx.x];      // same trip as the totals, not one after the scan
        }
        uint32_t run = block_excl_scan<DPT>(v, wsum, lane, w);
#pragma unroll
        for (int i = 0; i < DPT; ++i) {
            const int d = tid * DPT + i;
            if (d < NB) {
                digit_base[d] = run + bh[i];
#pragma unroll
                for (int k = 0; k < RS_WAVES; ++k) wave_cnt[k][d] = 0;
            }
            run += v[i];
        }
    }
    __syncthreads();

    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int64_t idx = wave_base + r * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = (key[r] >> shift) & (NB - 1);
        const uint64_t mask = gsrw::match_digit(d, BITS, __ballot(valid));
        const uint32_t prior = wave_cnt[w][d];
        rank[r] = prior + (uint32_t)__popcll(mask & lt_mask);
        if (valid && (mask & lt_mask) == 0ull) wave_cnt[w][d] = prior + (uint32_t)__popcll(mask);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // Local (in-workgroup) sorted position of every item: exclusive prefix over digits of the workgroup's digit
    // counts, then over waves within a digit.  Items are scattered into LDS in that order and then written out by
    // consecutive threads, so every digit's run leaves the CU as one contiguous, coalesced store burst instead of
    // 4-byte stores scattered over all destinations (measured 104 us -> 55 us per pass on 11.3 M pairs).
    {
        uint32_t tot[DPT];
#pragma unroll
        for (int i = 0; i < DPT; ++i) {
            const int d = tid * DPT + i;
            uint32_t t = 0;
            if (d < NB) {
#pragma unroll
                for (int k = 0; k < RS_WAVES; ++k) t += wave_cnt[k][d];
            }
            tot[i] = t;
        }
        uint32_t lbase = block_excl_scan<DPT>(tot, wsum, lane, w);
#pragma unroll
        for (int i = 0; i < DPT; ++i) {
            const int d = tid * DPT + i;
            if (d < NB) {
                uint32_t run = lbase;
#pragma unroll
                for (int k = 0; k < RS_WAVES; ++k) {
                    const uint32_t t = wave_cnt[k][d];
                    wave_cnt[k][d] = run;
                    run += t;
                }
                digit_base[d] -= lbase;                  // global position = digit_base[d] + local slot
            }
            lbase += tot[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int64_t idx = wave_base + r * 64 + lane;
        if (idx < n) {
            const uint32_t d = (key[r] >> shift) & (NB - 1);
            const uint32_t lp = wave_cnt[w][d] + rank[r];
            s_key[lp] = (KeyT)key[r];
            s_val[lp] = val[r];
        }
    }
    __syncthreads();
    const int64_t block_base = (int64_t)blockIdx.x * (RS_THREADS * IPT);
    const uint32_t nvalid = (uint32_t)((n - block_base) < (int64_t)(RS_THREADS * IPT) ? (n - block_base) : (RS_THREADS * IPT));
    uint32_t ok_[IPT], ov_[IPT], opos_[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const uint32_t i = (uint32_t)r * RS_THREADS + tid;
        ok_[r] = (uint32_t)s_key[i];
        ov_[r] = s_val[i];
        opos_[r] = digit_base[(ok_[r] >> shift) & (NB - 1)] + i;
    }
    if (rect_sorted) {
        // last pass of the depth sort: the rectangle gather.  All IPT gathers are issued before the first store -- written
        // as "rect_sorted[pos] = rect[v]" inside the store loop each gather was waited for on its own (IPT serial round trips).
        uint2 rc[IPT];
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const uint32_t i = (uint32_t)r * RS_THREADS + tid;
            rc[r] = rect[i < nvalid ? ov_[r] : 0u];
        }
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const uint32_t i = (uint32_t)r * RS_THREADS + tid;
            if (i < nvalid) rect_sorted[opos_[r]] = rc[r];
        }
    }
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const uint32_t i = (uint32_t)r * RS_THREADS + tid;
        if (i < nvalid) {
            keys_out[opos_[r]] = (KeyT)ok_[r];
            vals_out[opos_[r]] = ov_[r];
        }
    }
}


template <typename KeyT, int IPT, int BITS>
void sort_pass(KeyT* kin, uint32_t* vin, KeyT* kout, uint32_t* vout, int64_t n, int shift, uint32_t* hist,
               uint32_t* digit_total, int nblocks, const uint2* rect, uint2* rect_sorted, hipStream_t st) {
    hipLaunchKernelGGL((rs_hist<KeyT, IPT, BITS>), dim3(nblocks), dim3(RS_THREADS), 0, st, kin, n, shift, hist, nblocks);
    hipLaunchKernelGGL(rs_scan, dim3(1 << BITS), dim3(RS_THREADS), 0, st, hist, nblocks, digit_total);
    hipLaunchKernelGGL((rs_scatter<KeyT, IPT, BITS>), dim3(nblocks), dim3(RS_THREADS), 0, st, kin, vin, kout, vout, n, shift, hist,
                       digit_total, nblocks, rect, rect_sorted);
}

template <typename KeyT, int IPT>
void sort_pass_bits(int bits, KeyT* kin, uint32_t* vin, KeyT* kout, uint32_t* vout, int64_t n, int shift,
                    uint32_t* hist, uint32_t* digit_total, int nblocks, const uint2* rect, uint2* rect_sorted, hipStream_t st) {
    switch (bits) {
        case 11: sort_pass<KeyT, IPT, 11>(kin, vin, kout, vout, n, shift, hist, digit_total, nblocks, rect, rect_sorted, st); break;
        case 10: sort_pass<KeyT, IPT, 10>(kin, vin, kout, vout, n, shift, hist, digit_total, nblocks, rect, rect_sorted, st); break;
        case 9: sort_pass<KeyT, IPT, 9>(kin, vin, kout, vout, n, shift, hist, digit_total, nblocks, rect, rect_sorted, st); break;
        case 8: sort_pass<KeyT, IPT, 8>(kin, vin, kout, vout, n, shift, hist, digit_total, nblocks, rect, rect_sorted, st); break;
        case 7: sort_pass<KeyT, IPT, 7>(kin, vin, kout, vout, n, shift, hist, digit_total, nblocks, rect, rect_sorted, st); break;
        case 6: sort_pass<KeyT, IPT, 6>(kin, vin, kout, vout, n, shift, hist, digit_total, nblocks, rect, rect_sorted, st); break;
        case 5: sort_pass<KeyT, IPT, 5>(kin, vin, kout, vout, n, shift, hist, digit_total, nblocks, rect, rect_sorted, st); break;
        default: sort_pass<KeyT, IPT, 4>(kin, vin, kout, vout, n, shift, hist, digit_total, nblocks, rect, rect_sorted, st); break;
    }
}

template <typename KeyT>
int sort_pairs_t(KeyT* keys[2], uint32_t* vals[2], int64_t n, int nbits, int max_digit_bits, uint32_t* hist,
                 uint32_t* digit_total, int items, const uint2* rect, uint2* rect_sorted, hipStream_t st) {
    int cur = 0;
    if (n <= 0) return cur;
    const int nblocks = (int)((n + items - 1) / items);
    int pass_bits[8];
    const int passes = gsr_sort_plan(nbits, max_digit_bits, pass_bits);
    int shift = 0;
    for (int p = 0; p < passes; ++p) {
        const uint2* rc = p == passes - 1 ? rect : nullptr;          // the gather rides on the last pass only
        uint2* rcs = p == passes - 1 ? rect_sorted : nullptr;
        if (items == 1024)
            sort_pass_bits<KeyT, 1024 / RS_THREADS>(pass_bits[p], keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, hist,
                                                    digit_total, nblocks, rc, rcs, st);
        else if (items == 2048)
            sort_pass_bits<KeyT, 2048 / RS_THREADS>(pass_bits[p], keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, hist,
                                                    digit_total, nblocks, rc, rcs, st);
        else if (items == 8192)
            sort_pass_bits<KeyT, 8192 / RS_THREADS>(pass_bits[p], keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, hist,
                                                    digit_total, nblocks, rc, rcs, st);
        else
            sort_pass_bits<KeyT, 4096 / RS_THREADS>(pass_bits[p], keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, hist,
                                                    digit_total, nblocks, rc, rcs, st);
        shift += pass_bits[p];
        cur ^= 1;
    }
    return cur;
}

}  // namespace


void gsr_launch_rs_scan(uint32_t* block_hist, int nblocks, int ndigits, uint32_t* digit_total, hipStream_t st) {
    hipLaunchKernelGGL(rs_scan, dim3(ndigits), dim3(RS_THREADS), 0, st, block_hist, nblocks, digit_total);
}

int gsr_sort_plan(int nbits, int max_digit_bits, int* pass_bits) {
    // number of passes and bits per pass: ceil(nbits / max) passes of (almost) equal width, each in {4..11}
    int passes = (nbits + max_digit_bits - 1) / max_digit_bits;
    if (passes < 1) passes = 1;
    int left = nbits;
    for (int i = 0; i < passes; ++i) {
        int b = (left + (passes - i) - 1) / (passes - i);
        if (b > 11) b = 11;
        if (b < 4) b = 4;
        pass_bits[i] = b;
        left -= b;
        if (left < 0) left = 0;
    }
    return passes;
}

int gsr_radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], int64_t n, int nbits, int max_digit_bits, uint32_t* hist,
                         uint32_t* digit_total, int items, hipStream_t st, const uint2* rect, uint2* rect_sorted) {
    return sort_pairs_t<uint32_t>(keys, vals, n, nbits, max_digit_bits, hist, digit_total, items, rect, rect_sorted, st);
}

// 16-bit keys (tile ids when the frame has <= 65536 tiles): 25 % less traffic per pass than 32-bit keys
int gsr_radix_sort_pairs_k16(uint16_t* keys[2], uint32_t* vals[2], int64_t n, int nbits, int max_digit_bits, uint32_t* hist,
                             uint32_t* digit_total, int items, hipStream_t st) {
    return sort_pairs_t<uint16_t>(keys, vals, n, nbits, max_digit_bits, hist, digit_total, items, nullptr, nullptr, st);
}
