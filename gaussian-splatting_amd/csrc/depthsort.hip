// Depth order of the P Gaussians in FOUR launches: one bucket pass over the keys, then every bucket finished in LDS.
// Replaces (for P <= GSR_DS_MAX_P) the nine launches of the 3 x 9-bit LSD radix sort (sort.hip) AND the two launches of the
// tile-count scan (binning.hip) -- the reference's cub::DeviceRadixSort / InclusiveSum pair of SURVEY Appendix A.3, whose
// (tile | depth) order this pipeline reaches as "depth order of the Gaussians, then a stable sort of the instances by tile".
//
// Why: at 1 M keys every radix kernel is a 5-14 us latency chain (76.8 us for 80 MB = 0.13 of HBM
// peak, 12 us more for the scan).  Bytes are irrelevant; the number of dependent launches is the cost.
//
//   ds_hist     per workgroup of 4096 keys: histogram over 2048 depth buckets of (a) the keys, (b) their tile counts.
//               The bucket of a key comes from a HISTOGRAM-EQUALISED table (round 6): the key space is cut into 1024 coarse bins
//               (64 per octave of depth); every workgroup histograms the same 4096 sample keys (256 windows of 16 keys spread over
//               the array) over them; every coarse bin inside the frame's true key range gets one bucket and
//               the rest of the 2046 usable ones are handed out in proportion to the sampled mass; inside a coarse bin the buckets
//               are uniform.  Buckets then hold about P / 2046 keys whatever the depth distribution is -- floaters 100x behind the
//               scene, a wall, a cluster (rounds 4-5: uniform buckets over the key range, then over a "robust" range; a trained
//               scene still crowded a bucket on every frame and lived on the LSD fallback).  Gaussians without a tile take
//               bucket 2047 and end up behind everything else, as in the LSD sort.
//   ds_scan     per bucket: exclusive prefix of the counts over the workgroups (in place) + the bucket totals of both tables
//   ds_scatter  stable scatter (wave64 ballot ranking, no atomics) of (key, id) pairs into bucket order; the tile-less
//               Gaussians go straight to the tail of the final arrays.  One extra workgroup cuts the bucket-ordered array
//               into SEGMENTS: the buckets whose first element lies in the same window of 2048 elements (a segment is a whole
//               number of buckets, usually 2-4 K elements) and records, per segment, its element range, its bucket range and
//               the number of tile instances in front of it (exclusive scan of the bucket tile totals).
//   ds_segsort  one workgroup per segment: stable LSD radix sort of the segment in LDS on (key - first key of the segment's
//               first bucket), 2-3 passes of <= 9 bits; then, in sorted order: Gaussian id, its tile rectangle (gathered),
//               the INCLUSIVE SCAN of the tile counts (segment base + scan inside the segment) and the per-block table of the
//               emission -- everything the depth sort's last pass and the two scan kernels used to produce.
//               A segment beyond the LDS capacity (it ends with a bucket of more than 2048 keys) is cut into groups of whole buckets
//               that fit and sorted group by group; a SINGLE bucket beyond the capacity -- after both levels of the equalised table:
//               thousands of equal depths -- that the table knows to hold ONE key value is written out by as many workgroups as it has
//               chunks (the workgroups of the windows it covers, which own no bucket); otherwise its key span is measured: equal keys need
//               no pass, anything else goes through global memory.  That, or more than three capacities of equal keys the table did not
//               know about, is reported to the host, which prefers the LSD sort for a while (gsr_api.cpp).
//
// Order: (key, Gaussian index) ascending -- ds_scatter is stable and the segment sort is stable, so equal depths keep index
// order exactly as the LSD sort (and the reference's stable 64-bit-key sort) leaves them.  No atomics on global memory, no
// spinning, bit-reproducible.
#include "gsr_internal.h"
#include "gsr_wave.h"

using namespace gsrw;

namespace {

constexpr int DS_THREADS = 256;
constexpr int DS_ITEMS = GSR_DS_ITEMS;
constexpr int DS_IPT = DS_ITEMS / DS_THREADS;          // 16
constexpr int DS_NB = GSR_DS_BUCKETS;                  // 2048
constexpr uint32_t DS_CULL = DS_NB - 1u;
constexpr int DS_DPT = DS_NB / DS_THREADS;             // 8 buckets per thread
constexpr int DS_SEG = GSR_DS_SEG;
constexpr int DS_CAP = GSR_DS_CAP;
constexpr int DS_PASS_BITS = 9;
constexpr int DS_SLOW_CHUNKS = 3;                      // a single bucket of equal keys beyond this many LDS capacities that the TABLE does not know to be one key value is reported (ds_segsort)
constexpr int DS_PASS_BINS = 1 << DS_PASS_BITS;
static_assert(DS_IPT == 16 && DS_DPT == 8, "layout");

// eq[b] = first bucket | buckets << 16 of coarse bin b; eq2[j] = the same for sub-bin j of the HOT coarse bin `hot` (GSR_EQ_NO_HOT: no second
// level); LDS.  Monotone in the key; every key inside the frame's true range lies in a coarse bin that owns at least one bucket, and the
// sub-bins of the hot bin share its buckets (a sub-bin without a bucket of its own maps to the first bucket at or behind its position).
struct EqView {
    const uint32_t* eq;
    const uint32_t* eq2;
    uint32_t hot;
};
__device__ __forceinline__ uint32_t ds_bucket(uint32_t key, const EqView& v) {
    if (key == GSR_DEPTH_KEY_CULLED) return DS_CULL;
    const uint32_t b = key >> GSR_EQ_SHIFT;
    const uint32_t e = v.eq[b];
    if (b == v.hot) {
        const uint32_t e2 = v.eq2[(key >> GSR_EQ_SHIFT2) & ((uint32_t)GSR_EQ_BINS - 1u)], nb2 = e2 >> 16, st2 = e2 & 0xFFFFu;
        if (nb2) return st2 + (((key & ((1u << GSR_EQ_SHIFT2) - 1u)) * nb2) >> GSR_EQ_SHIFT2);
        return min(st2, (e & 0xFFFFu) + (e >> 16) - 1u);
    }
    return (e & 0xFFFFu) + (((key & ((1u << GSR_EQ_SHIFT) - 1u)) * (e >> 16)) >> GSR_EQ_SHIFT);
}
// smallest key that maps to a bucket >= d (d below the number of buckets in use): binary search for the coarse bin that owns d, then --
// inside the hot bin -- for the first sub-bin whose largest bucket reaches d
__device__ __forceinline__ uint32_t ds_bucket_first_key(uint32_t d, const EqView& v) {
    uint32_t lo = 0, hi = GSR_EQ_BINS;      // first coarse bin whose first bucket is > d
#pragma unroll 1
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((v.eq[mid] & 0xFFFFu) > d) hi = mid; else lo = mid + 1;
    }
    const uint32_t b = lo - 1u;             // (the last bin that starts at or before d: it owns buckets, the empty ones behind it start later)
    const uint32_t e = v.eq[b], nb = e >> 16, st = e & 0xFFFFu;
    if (b == v.hot) {
        const uint32_t last = st + nb - 1u;
        lo = 0; hi = GSR_EQ_BINS - 1;       // first sub-bin whose largest bucket is >= d (the last sub-bin's is `last` >= d)
#pragma unroll 1
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint32_t e2 = v.eq2[mid], nb2 = e2 >> 16, st2 = e2 & 0xFFFFu;
            const uint32_t top = nb2 ? st2 + nb2 - 1u : min(st2, last);
            if (top >= d) hi = mid; else lo = mid + 1;
        }
        const uint32_t e2 = v.eq2[lo], nb2 = e2 >> 16, st2 = e2 & 0xFFFFu;
        const uint32_t x = (nb2 && st2 < d) ? (((d - st2) << GSR_EQ_SHIFT2) + nb2 - 1u) / nb2 : 0u;
        return (b << GSR_EQ_SHIFT) + (lo << GSR_EQ_SHIFT2) + x;
    }
    // smallest x with (x * nb) >> SHIFT >= d - st
    const uint32_t x = nb ? (((d - st) << GSR_EQ_SHIFT) + nb - 1u) / nb : 0u;
    return (b << GSR_EQ_SHIFT) + x;
}

// four consecutive words, vector load when whole (p + e0 is 16-byte aligned: e0 is a multiple of 4, arrays are 128-byte aligned)
__device__ __forceinline__ void load4(const uint32_t* __restrict__ p, int64_t e0, int64_t n, uint32_t fill, uint32_t (&o)[4]) {
    if (e0 + 4 <= n) {
        const uint4 t = *reinterpret_cast<const uint4*>(p + e0);
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = e0 + j < n ? p[e0 + j] : fill;
    }
}

// ---- D1 ------------------------------------------------------------------------------------------------------------
// Every workgroup first reduces the per-workgroup key ranges the key-producing kernel left (gsr_frame.h; <= 2047 entries of
// 8 bytes) to the frame's true key range and histograms the SAME 4096 sample keys (256 windows spread over the array: sixteen
// loads per thread in the shadow of its own key loads) into the equalised bucket tables -- the same integers in every
// workgroup; workgroup 0 stores range and tables for the kernels that follow.
__global__ void __launch_bounds__(DS_THREADS)
ds_hist(int P, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ tiles, uint32_t* __restrict__ frame,
        const uint2* __restrict__ wg_range, int n_range, uint32_t* __restrict__ eq_tab,
        uint16_t* __restrict__ bucket_of, uint32_t* __restrict__ cnt_tab, uint32_t* __restrict__ tile_tab) {
    __shared__ uint32_t h_cnt[DS_NB], h_tile[DS_NB];
    __shared__ __attribute__((aligned(16))) uint32_t s_eq[GSR_EQ_BINS], s_eq2[GSR_EQ_BINS];      // the tables; before that: the sample's two histograms
    __shared__ uint32_t s_nmin[WG_WAVES], s_max[WG_WAVES], s_w[WG_WAVES], s_w2[WG_WAVES], s_hotp[WG_WAVES], s_w3[WG_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * DS_ITEMS;
    static_assert(GSR_EQ_BINS == 4 * DS_THREADS, "thread t owns the coarse bins 4t .. 4t+3");
    static_assert(GSR_EQ_SAMPLE == 16 * DS_THREADS, "sixteen sample keys per thread");
    uint32_t k[4][4], t[4][4];
    uint2 rg[8];
    uint32_t sk[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) rg[j] = wg_range[min(j * DS_THREADS + tid, n_range - 1)];      // (clamped: duplicates do not change a max)
    // THE SAMPLE: 256 windows of 16 consecutive keys (one 64-byte line each), window q at ((2 q + 1) P / 512) rounded down to a multiple of 16 -- the
    // same 4096 keys in every workgroup (L2 hits; a wave's load touches four lines), so every workgroup builds the same tables; an index past P
    // reads the last key again.  Many short windows: neighbours in the array are often neighbours in space (clones sit next to each other).
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int64_t q = (int64_t)j * 16 + (tid >> 4);
        const int64_t i = ((((2 * q + 1) * (int64_t)P) >> 9) & ~(int64_t)15) + (tid & 15);
        sk[j] = keys[i < P ? i : (int64_t)P - 1];
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {      // all loads first, the LDS clear rides in their shadow
        const int64_t e0 = base + ((int64_t)v * DS_THREADS + tid) * 4;
        load4(keys, e0, P, GSR_DEPTH_KEY_CULLED, k[v]);
        load4(tiles, e0, P, 0u, t[v]);
    }
#pragma unroll
    for (int i = 0; i < DS_DPT; ++i) {
        h_cnt[i * DS_THREADS + tid] = 0u;
        h_tile[i * DS_THREADS + tid] = 0u;
    }
    reinterpret_cast<uint4*>(s_eq)[tid] = make_uint4(0u, 0u, 0u, 0u);
    reinterpret_cast<uint4*>(s_eq2)[tid] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    // the sample's histograms: s_eq[key >> 17] (coarse bins), s_eq2[(key >> 7) & 1023] (the 1024 sub-bins of ALL coarse bins folded onto one another)
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (sk[j] != GSR_DEPTH_KEY_CULLED) {
            atomicAdd(&s_eq[sk[j] >> GSR_EQ_SHIFT], 1u);
            atomicAdd(&s_eq2[(sk[j] >> GSR_EQ_SHIFT2) & ((uint32_t)GSR_EQ_BINS - 1u)], 1u);
        }
    __syncthreads();
    // Round 1 (one barrier): the frame's true key range and the sampled keys per coarse bin (thread t owns the coarse bins 4t .. 4t+3)
    uint32_t c[4], fold[4];
    {
        const uint4 c4 = reinterpret_cast<const uint4*>(s_eq)[tid], f4 = reinterpret_cast<const uint4*>(s_eq2)[tid];
        c[0] = c4.x; c[1] = c4.y; c[2] = c4.z; c[3] = c4.w;
        fold[0] = f4.x; fold[1] = f4.y; fold[2] = f4.z; fold[3] = f4.w;
        uint32_t nm = 0, mx = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            nm = max(nm, rg[j].x);
            mx = max(mx, rg[j].y);
        }
        // the fullest coarse bin of the sample (the lowest one on ties): count << 10 | (1023 - bin); counts <= 4096
        uint32_t hp = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) hp = max(hp, (c[i] << 10) | ((uint32_t)GSR_EQ_BINS - 1u - (4u * (uint32_t)tid + (uint32_t)i)));
        nm = wave_incl_max_u32(nm);
        mx = wave_incl_max_u32(mx);
        hp = wave_incl_max_u32(hp);
        const uint32_t cs = wave_incl_scan_u32(c[0] + c[1] + c[2] + c[3], lane);
        if (lane == 63) { s_nmin[w] = nm; s_max[w] = mx; s_w[w] = cs; s_hotp[w] = hp; }
    }
    __syncthreads();
    const uint32_t tmin = ~max(max(s_nmin[0], s_nmin[1]), max(s_nmin[2], s_nmin[3]));
    const uint32_t tmax = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    uint32_t hot = GSR_EQ_NO_HOT, hot_bg = 0u, hot_floor = 1u;
    {   // Round 2 (one barrier): coarse bin b in [b_lo, b_hi] gets 1 + floor(spare * c[b] / C) buckets; their exclusive prefix is the table.
        // (Every sampled key lies inside the true range, so C counts exactly the sampled keys of the bins in range.)
        const bool any = tmax >= tmin;      // (nothing listed: tmin = 0xFFFFFFFF, tmax = 0)
        const uint32_t b_lo = any ? tmin >> GSR_EQ_SHIFT : 0u, b_hi = any ? min(tmax >> GSR_EQ_SHIFT, (uint32_t)GSR_EQ_BINS - 1u) : 0u;
        const uint32_t C = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        const uint32_t nbins = b_hi - b_lo + 1u, spare = (uint32_t)DS_NB - 2u - nbins;      // >= 1022
        uint32_t nb[4], nsum = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t b = 4u * (uint32_t)tid + (uint32_t)i;
            const bool in = b >= b_lo && b <= b_hi;
            // (spare * c <= 2046 * 4096)
            nb[i] = in ? 1u + (C ? (spare * c[i]) / C : spare / nbins) : 0u;
            nsum += nb[i];
        }
        const uint32_t nincl = wave_incl_scan_u32(nsum, lane);
        if (lane == 63) s_w2[w] = nincl;
        __syncthreads();
        uint32_t run = nincl - nsum;
#pragma unroll
        for (int k2 = 0; k2 < WG_WAVES; ++k2)
            if (k2 < w) run += s_w2[k2];
        uint4 e;
        e.x = run | (nb[0] << 16); run += nb[0];
        e.y = run | (nb[1] << 16); run += nb[1];
        e.z = run | (nb[2] << 16); run += nb[2];
        e.w = run | (nb[3] << 16);
        reinterpret_cast<uint4*>(s_eq)[tid] = e;
        if (blockIdx.x == 0) {
            reinterpret_cast<uint4*>(eq_tab)[tid] = e;
            if (tid == 0) { frame[2] = tmin; frame[3] = tmax; frame[6] = tmin; frame[7] = tmax; }
        }
        // Second level (gsr_frame.h), workgroup-uniform and rare: one coarse bin holds an eighth of the sample or more.  Its buckets are spread
        // over its 1024 sub-bins in proportion to THEIR sampled mass: the histogram of (key >> 7) & 1023 over ALL sample keys (the sub-bins of all
        // coarse bins folded onto one another) is the hot bin's sub-bin histogram plus a flat background of the other bins' keys, (C - c_hot) / 1024 each:
        // sub-bin j takes the buckets [S + NB * F(<j) / F, S + NB * F(<=j) / F).
        const uint32_t hpm = max(max(s_hotp[0], s_hotp[1]), max(s_hotp[2], s_hotp[3]));
        const uint32_t cH = hpm >> 10, H = (uint32_t)GSR_EQ_BINS - 1u - (hpm & ((uint32_t)GSR_EQ_BINS - 1u));
        hot = (any && C >= 256u && cH * 8u >= C) ? H : GSR_EQ_NO_HOT;
        hot_bg = (C - cH + (uint32_t)GSR_EQ_BINS - 1u) / (uint32_t)GSR_EQ_BINS;
        hot_floor = max(1u, cH >> 10);
    }
    __syncthreads();
    if (hot != GSR_EQ_NO_HOT) {
        uint32_t f[4] = {fold[0], fold[1], fold[2], fold[3]};
        // the folded histogram = the hot bin's sub-bins + a flat background of the keys of all other coarse bins
        // ... and every sub-bin keeps a floor of hot_floor = max(1, c_hot / 1024) -- its share if the hot bin were filled evenly: where the bin really
        // is a wall or a run of ties the excess dominates and takes most of the buckets; where it is merely a well-filled bin the excess is noise
        // and the floor keeps the second level close to the uniform split of the first (without it a few lucky sub-bins took ALL the bin's buckets
        // and an eighth of the frame shared a handful of them: the grown scene of the training run fell back to the LSD passes on 15 % of its frames)
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = (f[i] > hot_bg ? f[i] - hot_bg : 0u) + hot_floor;
        const uint32_t fsum = f[0] + f[1] + f[2] + f[3];
        const uint32_t fincl = wave_incl_scan_u32(fsum, lane);
        if (lane == 63) s_w3[w] = fincl;
        __syncthreads();
        const uint32_t F = s_w3[0] + s_w3[1] + s_w3[2] + s_w3[3];
        if (F == 0u) {      // (cannot happen with the floor; kept as a guard; workgroup-uniform)
            hot = GSR_EQ_NO_HOT;
        } else {
            uint32_t run = fincl - fsum;
#pragma unroll
            for (int k2 = 0; k2 < WG_WAVES; ++k2)
                if (k2 < w) run += s_w3[k2];
            const uint32_t eH = s_eq[hot], S = eH & 0xFFFFu, NB = eH >> 16;
            uint32_t e2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t st2 = S + (uint32_t)(((uint64_t)NB * run) / F), en2 = S + (uint32_t)(((uint64_t)NB * (run + f[i])) / F);
                run += f[i];
                e2[i] = st2 | ((en2 - st2) << 16);
            }
            const uint4 q = make_uint4(e2[0], e2[1], e2[2], e2[3]);
            reinterpret_cast<uint4*>(s_eq2)[tid] = q;
            if (blockIdx.x == 0) reinterpret_cast<uint4*>(eq_tab + GSR_EQ_BINS)[tid] = q;
            __syncthreads();
        }
    }
    if (blockIdx.x == 0 && tid == 0) eq_tab[2 * GSR_EQ_BINS] = hot;
    const EqView eqv{s_eq, s_eq2, hot};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int64_t e0 = base + ((int64_t)v * DS_THREADS + tid) * 4;
        uint32_t dd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dd[j] = ds_bucket(k[v][j], eqv);      // (keys past P were loaded as "no tile": bucket 2047, never counted)
            if (e0 + j < P) {
                atomicAdd(&h_cnt[dd[j]], 1u);
                if (t[v][j]) atomicAdd(&h_tile[dd[j]], t[v][j]);
            }
        }
        // the bucket of every key, for ds_scatter (8 bytes per four keys; the array is padded past P)
        if (e0 < P) *reinterpret_cast<uint2*>(bucket_of + e0) = make_uint2(dd[0] | (dd[1] << 16), dd[2] | (dd[3] << 16));
    }
    __syncthreads();
    uint32_t* crow = cnt_tab + (int64_t)blockIdx.x * DS_NB;
    uint32_t* trow = tile_tab + (int64_t)blockIdx.x * DS_NB;
#pragma unroll
    for (int i = 0; i < DS_DPT; ++i) {
        crow[i * DS_THREADS + tid] = h_cnt[i * DS_THREADS + tid];
        trow[i * DS_THREADS + tid] = h_tile[i * DS_THREADS + tid];
    }
}

// ---- D2 ------------------------------------------------------------------------------------------------------------
// tables are [workgroup][bucket]; a workgroup of this kernel owns 16 buckets, thread = (bucket, 1/16 of the rows): the 16
// lanes of a row segment read 64 contiguous bytes.  cnt_tab[b][d] <- keys of bucket d in workgroups < b; totals of both tables.
__global__ void __launch_bounds__(DS_THREADS)
ds_scan(int nblocks, uint32_t* __restrict__ cnt_tab, const uint32_t* __restrict__ tile_tab, uint32_t* __restrict__ cnt_total,
        uint32_t* __restrict__ tile_total) {
    __shared__ uint32_t s_c[16][16], s_t[16][16];
    const int tid = threadIdx.x, dl = tid & 15, bg = tid >> 4;
    const int d = blockIdx.x * 16 + dl;
    const int chunk = (nblocks + 15) / 16;
    const int lo = min(bg * chunk, nblocks), hi = min(lo + chunk, nblocks);
    constexpr int KEEP = 16;            // rows per thread kept in registers between the two sweeps (<= 256 workgroups = 1 M keys)
    const bool keep = chunk <= KEEP;
    uint32_t v[KEEP];
    uint32_t csum = 0, tsum = 0;
    if (keep) {
        uint32_t tt[KEEP];
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {      // unconditional loads (clamped row): one round trip for all of them
            const int b = min(lo + k, nblocks - 1);
            v[k] = cnt_tab[(int64_t)b * DS_NB + d];
            tt[k] = tile_tab[(int64_t)b * DS_NB + d];
        }
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            if (lo + k >= hi) { v[k] = 0u; tt[k] = 0u; }
            csum += v[k];
            tsum += tt[k];
        }
    } else {
#pragma unroll 4
        for (int b = lo; b < hi; ++b) {
            csum += cnt_tab[(int64_t)b * DS_NB + d];
            tsum += tile_tab[(int64_t)b * DS_NB + d];
        }
    }
    s_c[bg][dl] = csum;
    s_t[bg][dl] = tsum;
    __syncthreads();
    uint32_t run = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g)
        if (g < bg) run += s_c[g][dl];
    if (bg == 15) cnt_total[d] = run + csum;
    if (bg == 0) {
        uint32_t tt = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g) tt += s_t[g][dl];
        tile_total[d] = tt;
    }
    if (keep) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            if (lo + k < hi) cnt_tab[(int64_t)(lo + k) * DS_NB + d] = run;
            run += v[k];
        }
    } else {
#pragma unroll 4
        for (int b = lo; b < hi; ++b) {
            const uint32_t c = cnt_tab[(int64_t)b * DS_NB + d];
            cnt_tab[(int64_t)b * DS_NB + d] = run;
            run += c;
        }
    }
}

// ---- D3 ------------------------------------------------------------------------------------------------------------
// 512 threads per 4096 keys: with 256 the launch put ONE wave on every SIMD, and a lone wave issues a VALU instruction every
// ~5 cycles (DESIGN: the measured issue ceiling needs >= 3 waves per SIMD) -- the kernel was bound by the 11 ballots per 64 keys
// at a fifth of the issue rate.  Eight waves per workgroup halve the rounds per wave and let two waves share every SIMD.
constexpr int S3_THREADS = 512;
constexpr int S3_WAVES = S3_THREADS / 64;
constexpr int S3_IPT = DS_ITEMS / S3_THREADS;          // 8 keys per thread
constexpr int S3_DPT = DS_NB / S3_THREADS;             // 4 buckets per thread
static_assert(S3_IPT == 8 && S3_DPT == 4, "layout");

// exclusive scan over the workgroup of the per-thread totals of DPT values (thread t owns entries t*DPT .. t*DPT+DPT-1)
template <int DPT, int NW>
__device__ __forceinline__ uint32_t ds_block_excl_scan(const uint32_t (&v)[DPT], uint32_t* wsum, int lane, int w) {
    uint32_t tsum = 0;
#pragma unroll
    for (int i = 0; i < DPT; ++i) tsum += v[i];
    const uint32_t incl = wave_incl_scan_u32(tsum, lane);
    __syncthreads();
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t wbase = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k)
        if (k < w) wbase += wsum[k];
    return wbase + incl - tsum;
}

// plan entry of window s (GSR_DS_PLAN_WORDS = 12 words): begin, end (elements of the bucket-ordered array), first bucket, end bucket (bit 31: the last
// bucket is more than a capacity of one key value, its chunks behind the first are written by helpers), tile instances in front of the segment, number of
// listed Gaussians, smallest key the segment's buckets can hold, one past the largest; helper job of the window's workgroup: first element of the chunk
// (GSR_DS_NO_HELP: none), its length | its bucket << 16, first element of that bucket, tile instances in front of that bucket
__global__ void __launch_bounds__(S3_THREADS)
ds_scatter(int P, int nblocks, const uint32_t* __restrict__ keys, const uint16_t* __restrict__ bucket_of, const uint32_t* __restrict__ frame, const uint32_t* __restrict__ eq_tab,
           const uint32_t* __restrict__ cnt_tab, const uint32_t* __restrict__ cnt_total, const uint32_t* __restrict__ tile_total,
           uint2* __restrict__ pairs, uint32_t* __restrict__ order, uint32_t* __restrict__ offsets, uint2* __restrict__ rect_sorted,
           uint32_t* __restrict__ plan, int nseg_cap) {
    __shared__ __attribute__((aligned(16))) uint16_t wave_cnt[S3_WAVES][DS_NB];      // 32 KB (a wave counts <= 512 keys)
    __shared__ __attribute__((aligned(16))) uint32_t digit_base[DS_NB];              //  8 KB
    __shared__ __attribute__((aligned(16))) uint32_t s_eq[GSR_EQ_BINS], s_eq2[GSR_EQ_BINS];      //  8 KB equalised bucket tables (ds_hist): the plan workgroup only
    __shared__ uint32_t wsum[S3_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    static_assert(GSR_EQ_BINS == 2 * S3_THREADS, "one 8-byte load per thread");

    if ((int)blockIdx.x == nblocks) {
        // ---- the segment plan (one workgroup, beside the scattering ones) ----
        // the equalised bucket tables of ds_hist (both levels; the second is garbage and never read when there is no hot bin); the barriers of the scans
        // below publish them
        const uint32_t hot = eq_tab[2 * GSR_EQ_BINS];
        const EqView eqv{s_eq, s_eq2, hot};
        reinterpret_cast<uint2*>(s_eq)[tid] = reinterpret_cast<const uint2*>(eq_tab)[tid];
        reinterpret_cast<uint2*>(s_eq2)[tid] = reinterpret_cast<const uint2*>(eq_tab + GSR_EQ_BINS)[tid];
        uint32_t* cnt_excl = reinterpret_cast<uint32_t*>(&wave_cnt[0][0]);      // [2048]; entry 2047 = number of listed Gaussians
        uint32_t* tile_excl = cnt_excl + DS_NB;
        uint32_t c[S3_DPT], t[S3_DPT];
        {
            const uint4 a = reinterpret_cast<const uint4*>(cnt_total)[tid], b = reinterpret_cast<const uint4*>(tile_total)[tid];
            c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w;
            t[0] = b.x; t[1] = b.y; t[2] = b.z; t[3] = b.w;
        }
        uint32_t crun = ds_block_excl_scan<S3_DPT, S3_WAVES>(c, wsum, lane, w);
        uint32_t trun = ds_block_excl_scan<S3_DPT, S3_WAVES>(t, wsum, lane, w);
        {
            uint32_t ce[S3_DPT], te[S3_DPT];
#pragma unroll
            for (int i = 0; i < S3_DPT; ++i) {
                ce[i] = crun; te[i] = trun;
                crun += c[i];
                trun += t[i];
            }
            reinterpret_cast<uint4*>(cnt_excl)[tid] = make_uint4(ce[0], ce[1], ce[2], ce[3]);
            reinterpret_cast<uint4*>(tile_excl)[tid] = make_uint4(te[0], te[1], te[2], te[3]);
        }
        __syncthreads();
        const uint32_t listed = cnt_excl[DS_CULL];
        const uint32_t tmin = frame[6], tmax = frame[7];
        const uint32_t used = (s_eq[GSR_EQ_BINS - 1] & 0xFFFFu) + (s_eq[GSR_EQ_BINS - 1] >> 16);      // buckets the table hands out
        auto lower_bound = [&](uint32_t x) {      // first bucket in [0, 2047] whose first element is >= x (2047 if none)
            uint32_t lo = 0, hi = DS_CULL;
#pragma unroll 1
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (cnt_excl[mid] >= x) hi = mid; else lo = mid + 1;
            }
            return lo;
        };
        // bucket d holds ONE key value (by the table: the keys that map to it are [first key of d, first key of d + 1) inside the frame's range)
        auto one_key_bucket = [&](uint32_t d) {
            const uint32_t lo = max(tmin, ds_bucket_first_key(d, eqv));
            const uint32_t hi = d + 1u >= used ? tmax + 1u : min(tmax + 1u, ds_bucket_first_key(d + 1u, eqv));
            return hi <= lo + 1u;
        };
        for (int s = tid; s < nseg_cap; s += S3_THREADS) {
            uint4 e = make_uint4(0u, 0u, 0u, 0u);
            uint4 h = make_uint4(GSR_DS_NO_HELP, 0u, 0u, 0u);
            uint32_t tb = 0, key_lo = 0, key_hi = 0;
            const uint64_t x0 = (uint64_t)s * DS_SEG;
            if (x0 < listed) {
                const uint32_t d0 = lower_bound((uint32_t)x0);
                const uint64_t x1 = x0 + DS_SEG;
                uint32_t d1 = x1 >= listed ? DS_CULL : lower_bound((uint32_t)x1);
                e = make_uint4(cnt_excl[d0], cnt_excl[d1], d0, d1);
                tb = tile_excl[d0];
                // the keys of buckets [d0, d1): [first key of d0, first key of d1), inside the frame's true range (keys < 2^27: no overflow)
                key_lo = max(tmin, ds_bucket_first_key(d0, eqv));
                key_hi = d1 >= used ? tmax + 1u : min(tmax + 1u, ds_bucket_first_key(d1, eqv));
                if (e.y > e.x) {
                    // the segment's last bucket (the one that holds its last element; a bucket beyond the LDS capacity is always the last: the next one
                    // starts outside the window).  More than a capacity of ONE key value: the owner writes its first chunk, the workgroups of the
                    // windows the bucket covers write the others (below) -- flagged in bit 31 of the end bucket
                    const uint32_t dl = lower_bound(e.y) - 1u;
                    if (e.y - cnt_excl[dl] > (uint32_t)DS_CAP && one_key_bucket(dl)) e.w = d1 | GSR_DS_TIE_HELPED;
                }
            }
            // HELPER JOB of window s (whether it owns a segment or not): chunk c >= 1 of a bucket of more than a capacity of one key value goes to the
            // first window that starts at or behind the chunk.  Such a bucket began more than a window in front of the chunk, so it is the bucket that holds the
            // element at the start of the PREVIOUS window.
            if (s >= 1 && x0 - DS_SEG < listed) {
                const uint32_t p = (uint32_t)(x0 - DS_SEG);
                const uint32_t dc = lower_bound(p + 1u) - 1u;      // (cnt_excl[0] = 0 <= p: at least 1)
                const uint32_t bb = cnt_excl[dc], be = cnt_excl[dc + 1u];
                if (be - bb > (uint32_t)DS_CAP && p < be) {
                    const uint32_t c = ((uint32_t)x0 - bb) / (uint32_t)DS_CAP, cs = bb + c * (uint32_t)DS_CAP;
                    if (c >= 1u && (uint64_t)cs + DS_SEG > x0 && cs < be && one_key_bucket(dc))
                        h = make_uint4(cs, min((uint32_t)DS_CAP, be - cs) | (dc << 16), bb, tile_excl[dc]);
                }
            }
            reinterpret_cast<uint4*>(plan)[3 * s] = e;
            reinterpret_cast<uint4*>(plan)[3 * s + 1] = make_uint4(tb, listed, key_lo, key_hi);
            reinterpret_cast<uint4*>(plan)[3 * s + 2] = h;
        }
        return;
    }

    const uint32_t R32 = frame[0];
    // the workgroup's keys are requested first: their trip overlaps the bucket-base prologue below.  Wave w owns the
    // contiguous run [w * 512, w * 512 + 512) of the workgroup's keys, item r of a lane is key r * 64 + lane of the run
    const int64_t wave_base = (int64_t)blockIdx.x * DS_ITEMS + (int64_t)w * (64 * S3_IPT);
    uint32_t key[S3_IPT], dig[S3_IPT];
#pragma unroll
    for (int r = 0; r < S3_IPT; ++r) {
        const int64_t idx = wave_base + r * 64 + lane;
        key[r] = keys[idx < P ? idx : (int64_t)P - 1];
        dig[r] = bucket_of[idx < P ? idx : (int64_t)P - 1];      // (ds_hist looked the bucket up in the equalised tables once)
    }
    {   // digit_base[d] = (exclusive scan of the bucket totals)[d] + keys of bucket d in earlier workgroups
        uint32_t v[S3_DPT];
        const uint4 a = reinterpret_cast<const uint4*>(cnt_total)[tid];
        const uint4 bh = reinterpret_cast<const uint4*>(cnt_tab + (int64_t)blockIdx.x * DS_NB)[tid];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        uint32_t run = ds_block_excl_scan<S3_DPT, S3_WAVES>(v, wsum, lane, w);
        // LDS accesses of this kernel never use the "thread t owns buckets 4t .. 4t+3" pattern with 32-bit accesses (a lane stride
        // of several words piles the lanes onto a few banks): the thread's four bases go out as one 16-byte write, the count tables
        // are cleared with 16-byte writes and scanned with bucket = round * 512 + thread.
        uint4 db;
        db.x = run + bh.x; run += v[0];
        db.y = run + bh.y; run += v[1];
        db.z = run + bh.z; run += v[2];
        db.w = run + bh.w;
        reinterpret_cast<uint4*>(digit_base)[tid] = db;
        uint4* wc4 = reinterpret_cast<uint4*>(&wave_cnt[0][0]);      // 32 KB = 2048 x 16 bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) wc4[i * S3_THREADS + tid] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t rank[S3_IPT];
#pragma unroll
    for (int r = 0; r < S3_IPT; ++r) {
        const int64_t idx = wave_base + r * 64 + lane;
        const bool valid = idx < P;
        const uint32_t d = dig[r];
        const uint64_t mask = match_digit(d, GSR_DS_BITS, __ballot(valid));
        const uint32_t prior = valid ? (uint32_t)wave_cnt[w][d] : 0u;
        rank[r] = prior + (uint32_t)__popcll(mask & lt_mask);
        if (valid && (mask & lt_mask) == 0ull) wave_cnt[w][d] = (uint16_t)(prior + (uint32_t)__popcll(mask));
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // keys of a bucket are few per workgroup (4096 keys over ~1000+ buckets): no LDS staging, every pair goes straight to
    // its slot; per bucket the waves' counts become exclusive offsets
#pragma unroll
    for (int i = 0; i < S3_DPT; ++i) {
        const int d = i * S3_THREADS + tid;
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < S3_WAVES; ++k) {
            const uint32_t t = wave_cnt[k][d];
            wave_cnt[k][d] = (uint16_t)run;
            run += t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < S3_IPT; ++r) {
        const int64_t idx = wave_base + r * 64 + lane;
        if (idx < P) {
            const uint32_t d = dig[r];
            const uint32_t pos = digit_base[d] + (uint32_t)wave_cnt[w][d] + rank[r];
            if (d != DS_CULL) {
                pairs[pos] = make_uint2(key[r], (uint32_t)idx);
            } else {      // no tile: behind every listed Gaussian, in index order; the inclusive scan stays at R
                order[pos] = (uint32_t)idx;
                offsets[pos] = R32;
                rect_sorted[pos] = make_uint2(0u, 0u);
            }
        }
    }
}

// ---- D4 ------------------------------------------------------------------------------------------------------------
// 512 threads: the sort is a chain of dependent LDS round trips per wave and pass (key -> ballots -> running count -> store),
// so the lever is the number of 64-item steps per wave: 8 waves halve it against 4, and two workgroups of 8 waves give every
// SIMD four waves to interleave.
constexpr int SG_THREADS = 512;
constexpr int SG_WAVES = SG_THREADS / 64;
constexpr int SG_OUT = DS_CAP / SG_THREADS;            // 8 elements per thread in the output phase

struct SegLds {
    uint32_t (*key)[DS_CAP];
    uint16_t (*idx)[DS_CAP];
    __device__ __forceinline__ void load(int buf, uint32_t i, uint32_t& k, uint32_t& v) const { k = key[buf][i]; v = idx[buf][i]; }
    __device__ __forceinline__ void store(int buf, uint32_t i, uint32_t k, uint32_t v) const { key[buf][i] = k; idx[buf][i] = (uint16_t)v; }
};
struct SegGlobal {      // the segment's slice of the two pair arrays; keys are rebased on the fly
    uint2* p0;
    uint2* p1;
    uint32_t base_key;
    __device__ __forceinline__ void load(int buf, uint32_t i, uint32_t& k, uint32_t& v) const {
        const uint2 t = buf ? p1[i] : p0[i];
        k = t.x - base_key; v = t.y;
    }
    __device__ __forceinline__ void store(int buf, uint32_t i, uint32_t k, uint32_t v) const {
        const uint2 t = make_uint2(k + base_key, v);
        if (buf) p1[i] = t; else p0[i] = t;
    }
};

// exclusive scan over the 512 threads of the per-thread value; wsum: SG_WAVES words of LDS; two barriers
__device__ __forceinline__ uint32_t seg_excl_scan(uint32_t v, uint32_t* wsum, int lane, int w) {
    const uint32_t incl = wave_incl_scan_u32(v, lane);
    __syncthreads();
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t wbase = 0;
#pragma unroll
    for (int k = 0; k < SG_WAVES; ++k)
        if (k < w) wbase += wsum[k];
    return wbase + incl - v;
}

// Stable LSD radix sort of n (key, value) items held in buffer 0 of `m` on the low `nbits` key bits, passes of <= 9 bits,
// by the whole workgroup.  Returns the buffer that holds the result.  Wave w owns a contiguous eighth of the items; per pass:
// count (LDS atomics on the wave's own table) -> scan over (bin, wave) -> rank with ballot matching against running counts.
// CntT: uint16_t for segments that fit the LDS buffers (counts and offsets < 65536; two bins share a 32-bit atomic),
// uint32_t for the oversized ones.
template <class Mem, typename CntT>
__device__ __forceinline__ int seg_sort(const Mem& m, uint32_t n, int nbits, CntT (*wave_cnt)[DS_PASS_BINS], uint32_t* wsum) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int cur = 0;
    if (nbits <= 0) return cur;
    const int npass = (nbits + DS_PASS_BITS - 1) / DS_PASS_BITS;
    const int pb = (nbits + npass - 1) / npass;
    const uint32_t q = (((n + (uint32_t)SG_WAVES - 1u) / (uint32_t)SG_WAVES) + 63u) & ~63u;
    const uint32_t wbeg = min((uint32_t)w * q, n), wend = min(wbeg + q, n);
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    int sh = 0;
#pragma unroll 1
    for (int p = 0; p < npass; ++p, sh += pb) {
        const int bits = min(pb, nbits - sh);
        const uint32_t dmask = (1u << bits) - 1u;
#pragma unroll
        for (int k = 0; k < SG_WAVES; ++k) wave_cnt[k][tid] = (CntT)0;
        __syncthreads();
#pragma unroll 2
        for (uint32_t i = wbeg + lane; i < wend; i += 64u) {
            uint32_t k, v;
            m.load(cur, i, k, v);
            const uint32_t d = (k >> sh) & dmask;
            if (sizeof(CntT) == 2) atomicAdd(reinterpret_cast<uint32_t*>(wave_cnt[w]) + (d >> 1), 1u << ((d & 1u) * 16u));
            else atomicAdd(reinterpret_cast<uint32_t*>(wave_cnt[w]) + d, 1u);
        }
        __syncthreads();
        {   // thread t owns bin t: offsets over (bin, wave)
            uint32_t c[SG_WAVES], tot = 0;
#pragma unroll
            for (int k = 0; k < SG_WAVES; ++k) { c[k] = wave_cnt[k][tid]; tot += c[k]; }
            uint32_t run = seg_excl_scan(tot, wsum, lane, w);
#pragma unroll
            for (int k = 0; k < SG_WAVES; ++k) { wave_cnt[k][tid] = (CntT)run; run += c[k]; }
        }
        __syncthreads();
#pragma unroll 1
        for (uint32_t i0 = wbeg; i0 < wend; i0 += 64u) {      // wave-uniform trips
            const uint32_t i = i0 + (uint32_t)lane;
            const bool valid = i < wend;
            uint32_t k = 0, v = 0;
            if (valid) m.load(cur, i, k, v);
            const uint32_t d = (k >> sh) & dmask;
            const uint64_t mask = match_digit(d, bits, __ballot(valid));
            const uint32_t prior = valid ? (uint32_t)wave_cnt[w][d] : 0u;
            if (valid) {
                m.store(cur ^ 1, prior + (uint32_t)__popcll(mask & lt_mask), k, v);
                if ((mask & lt_mask) == 0ull) wave_cnt[w][d] = (CntT)(prior + (uint32_t)__popcll(mask));
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        cur ^= 1;
    }
    return cur;
}

__device__ __forceinline__ uint32_t rect_area(const uint2 r) {
    return ((r.x >> 16) - (r.x & 0xFFFFu)) * ((r.y >> 16) - (r.y & 0xFFFFu));
}

// Output of up to 4096 consecutive sorted elements [c0, c0 + m) of the segment: element p of the chunk belongs to thread
// p % 512, round p / 512 (lane-contiguous stores).  id[j]: the Gaussian id of element j * 512 + tid.  Returns the number of
// tile instances of the chunk (every thread).
__device__ __forceinline__ uint32_t seg_output(const uint32_t (&id)[SG_OUT], uint32_t m, uint32_t gpos0 /*global position of the chunk*/,
                                              uint32_t tile_base, const uint2* __restrict__ rect, uint32_t* __restrict__ order,
                                              uint2* __restrict__ rect_sorted, uint32_t* __restrict__ offsets,
                                              uint2* __restrict__ block_first, uint32_t bf_cap, uint32_t last_listed, bool r_ok,
                                              uint32_t (*s_wt)[SG_WAVES]) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint2 rc[SG_OUT];
#pragma unroll
    for (int j = 0; j < SG_OUT; ++j) rc[j] = rect[id[j]];      // (lanes past m carry id 0: a valid address)
    uint32_t t[SG_OUT], incl[SG_OUT];
#pragma unroll
    for (int j = 0; j < SG_OUT; ++j) {
        const uint32_t p = (uint32_t)j * SG_THREADS + (uint32_t)tid;
        t[j] = p < m ? rect_area(rc[j]) : 0u;
        incl[j] = wave_incl_scan_u32(t[j], lane);
        if (lane == 63) s_wt[j][w] = incl[j];
    }
    __syncthreads();
    uint32_t run = tile_base;
    constexpr uint32_t IT = GSR_TS_ITEMS;
#pragma unroll
    for (int j = 0; j < SG_OUT; ++j) {
        const uint32_t p = (uint32_t)j * SG_THREADS + (uint32_t)tid;
        uint32_t wpre = 0, rtot = 0;
#pragma unroll
        for (int k = 0; k < SG_WAVES; ++k) {
            const uint32_t x = s_wt[j][k];
            if (k < w) wpre += x;
            rtot += x;
        }
        const uint32_t in = run + wpre + incl[j], ex = in - t[j];
        if (p < m) {
            const uint32_t g = gpos0 + p;
            order[g] = id[j];
            rect_sorted[g] = rc[j];
            offsets[g] = in;
            if (t[j] && r_ok) {
                // block_first[k] = (depth-order index, instances before it) of the Gaussian that owns instance k * IT; one
                // past the last block: the last Gaussian with tiles (tilesort.hip reads entry b and b + 1)
                for (uint32_t b = (ex + IT - 1u) / IT; b <= (in - 1u) / IT; ++b)
                    if (b < bf_cap) block_first[b] = make_uint2(g, ex);
                if (g == last_listed) {
                    const uint32_t b = (in + IT - 1u) / IT;
                    if (b < bf_cap) block_first[b] = make_uint2(g, ex);
                }
            }
        }
        run += rtot;
    }
    __syncthreads();      // s_wt is reused by the next chunk
    return run - tile_base;
}

// `n` <= DS_CAP consecutive elements [begin, begin + n) of the bucket-ordered array, keys in [base_key, base_key + 2^nbits): sorted in LDS and written
// out (order, rectangles, inclusive tile scan from `tile_base`, emission block table).  Returns the tile instances of the run (every thread).
struct SegLdsMem {
    uint32_t (*s_key)[DS_CAP];
    uint16_t (*s_idx)[DS_CAP];
    uint16_t (*wave_cnt)[DS_PASS_BINS];
    uint32_t* wsum;
    uint32_t (*s_wt)[SG_WAVES];
};
__device__ __forceinline__ uint32_t seg_sort_in_lds(const SegLdsMem& L, const uint2* __restrict__ pairs0, uint32_t begin, uint32_t n, uint32_t base_key, int nbits,
                                                    uint32_t tile_base, const uint2* __restrict__ rect, uint32_t* __restrict__ order, uint2* __restrict__ rect_sorted,
                                                    uint32_t* __restrict__ offsets, uint2* __restrict__ block_first, uint32_t bf_cap, uint32_t last_listed, bool r_ok) {
    const int tid = threadIdx.x;
    SegLds m{L.s_key, L.s_idx};
    {   // all loads, then all LDS writes (a load inside "if (p < n) lds[p] = ..." is waited for one by one)
        uint32_t kk[SG_OUT];
#pragma unroll
        for (int j = 0; j < SG_OUT; ++j) {
            const uint32_t p = (uint32_t)j * SG_THREADS + (uint32_t)tid;
            kk[j] = pairs0[begin + (p < n ? p : 0u)].x;
        }
#pragma unroll
        for (int j = 0; j < SG_OUT; ++j) {
            const uint32_t p = (uint32_t)j * SG_THREADS + (uint32_t)tid;
            if (p < n) {
                L.s_key[0][p] = kk[j] - base_key;
                L.s_idx[0][p] = (uint16_t)p;
            }
        }
    }
    __syncthreads();
    const int cur = seg_sort<SegLds, uint16_t>(m, n, nbits, L.wave_cnt, L.wsum);
    uint32_t ix[SG_OUT], id[SG_OUT];
#pragma unroll
    for (int j = 0; j < SG_OUT; ++j) {
        const uint32_t p = (uint32_t)j * SG_THREADS + (uint32_t)tid;
        ix[j] = p < n ? (uint32_t)L.s_idx[cur][p] : 0u;
    }
#pragma unroll
    for (int j = 0; j < SG_OUT; ++j) {
        const uint32_t p = (uint32_t)j * SG_THREADS + (uint32_t)tid;
        const uint32_t v = pairs0[begin + ix[j]].y;
        id[j] = p < n ? v : 0u;
    }
    return seg_output(id, n, begin, tile_base, rect, order, rect_sorted, offsets, block_first, bf_cap, last_listed, r_ok, L.s_wt);
}

__global__ void __launch_bounds__(SG_THREADS)
ds_segsort(const uint32_t* __restrict__ plan, const uint32_t* __restrict__ frame, const uint32_t* __restrict__ cnt_total, const uint32_t* __restrict__ cnt_tab,
           const uint32_t* __restrict__ tile_tab, int nblocks, uint2* pairs0, uint2* pairs1,
           const uint2* __restrict__ rect, uint32_t* __restrict__ order, uint2* __restrict__ rect_sorted, uint32_t* __restrict__ offsets,
           uint2* __restrict__ block_first, uint32_t bf_cap, uint32_t* slow_word /*mapped host word or NULL*/) {
    __shared__ uint32_t s_key[2][DS_CAP];                       // 32 KB (a bucket beyond the LDS capacity: the 32-bit count table instead)
    __shared__ uint16_t s_idx[2][DS_CAP];                       // 16 KB
    __shared__ uint16_t wave_cnt[SG_WAVES][DS_PASS_BINS];       //  8 KB
    __shared__ uint32_t wsum[SG_WAVES];
    __shared__ uint32_t s_wt[SG_OUT][SG_WAVES];
    __shared__ uint32_t s_bc[DS_NB];                            //  8 KB bucket sizes of an oversized segment
    __shared__ uint32_t s_mm[2][SG_WAVES];
    static_assert(sizeof(uint32_t) * SG_WAVES * DS_PASS_BINS <= sizeof(uint32_t) * 2 * DS_CAP, "count table of the oversized path");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint4 e = reinterpret_cast<const uint4*>(plan)[3 * blockIdx.x];
    const uint4 e2 = reinterpret_cast<const uint4*>(plan)[3 * blockIdx.x + 1];
    const uint4 e3 = reinterpret_cast<const uint4*>(plan)[3 * blockIdx.x + 2];
    const uint32_t begin = e.x, end = e.y, d0 = e.z, d1 = e.w & ~GSR_DS_TIE_HELPED;
    const bool tie_helped = (e.w & GSR_DS_TIE_HELPED) != 0u;
    const uint32_t tile_base = e2.x, listed = e2.y;
    const bool r_ok = frame[1] == 0u && (int32_t)frame[0] >= 0;      // R < 2^31 (else the host refuses the frame: no table writes)
    if (e3.x != GSR_DS_NO_HELP) {
        // ---- helper job (round 6): chunk [cs, cs + len) of a bucket of EQUAL keys that is longer than the LDS capacity.  Equal keys are already in their
        // final order (ds_scatter is stable): only the output is left, and its tile scan starts at the bucket's tile base + the tile counts of the bucket's
        // elements in front of the chunk.  Those are whole contributions of ds_hist's workgroups -- the [workgroup][bucket] tables hold their sizes (scanned by
        // ds_scan: cnt_tab[b][d] = elements of bucket d from workgroups < b) and tile sums -- plus a part of ONE workgroup's, at most 4096 elements, which are
        // added up from their rectangles.  Constant time whatever the bucket's length (first version: every helper summed ALL elements in front of its
        // chunk, ~5 us per 4096: a bucket of 60 000 took 73 us).
        const uint32_t cs = e3.x, len = e3.y & 0xFFFFu, dbk = e3.y >> 16, bb = e3.z, front = cs - bb;
        if (tid == 0) s_mm[1][0] = front;      // start (inside the bucket) of the workgroup contribution the chunk start cuts; `front`: it cuts none
        __syncthreads();
        uint32_t tsum = 0;
#pragma unroll 1
        for (int b0 = 0; b0 < nblocks; b0 += SG_THREADS) {
            const int b = min(b0 + tid, nblocks - 1);
            const uint32_t c0 = cnt_tab[(int64_t)b * DS_NB + dbk], tt = tile_tab[(int64_t)b * DS_NB + dbk];
            const uint32_t cn = cnt_tab[(int64_t)min(b + 1, nblocks - 1) * DS_NB + dbk], ct = cnt_total[dbk];
            const uint32_t c1 = b + 1 < nblocks ? cn : ct;
            const bool valid = b0 + tid < nblocks, whole = valid && c1 <= front;      // (selects, not branches: a load under a branch is waited for alone)
            tsum += whole ? tt : 0u;
            if (valid && !whole && c0 < front) s_mm[1][0] = c0;      // (exactly one workgroup's contribution straddles the chunk start)
        }
        __syncthreads();
        {
            const uint32_t p0 = s_mm[1][0], part = front - p0;      // <= 4096 elements [bb + p0, bb + front)
            uint32_t idv[SG_OUT];
#pragma unroll
            for (int j = 0; j < SG_OUT; ++j) idv[j] = pairs0[bb + p0 + min((uint32_t)j * SG_THREADS + (uint32_t)tid, part ? part - 1u : 0u)].y;
            uint2 rv[SG_OUT];
#pragma unroll
            for (int j = 0; j < SG_OUT; ++j) rv[j] = rect[idv[j]];
#pragma unroll
            for (int j = 0; j < SG_OUT; ++j) tsum += (uint32_t)j * SG_THREADS + (uint32_t)tid < part ? rect_area(rv[j]) : 0u;
        }
        tsum = wave_incl_scan_u32(tsum, lane);
        if (lane == 63) s_mm[0][w] = tsum;
        __syncthreads();
        uint32_t tfront = 0;
#pragma unroll
        for (int k2 = 0; k2 < SG_WAVES; ++k2) tfront += s_mm[0][k2];
        uint32_t id[SG_OUT];
#pragma unroll
        for (int j = 0; j < SG_OUT; ++j) {
            const uint32_t p = (uint32_t)j * SG_THREADS + (uint32_t)tid;
            const uint32_t v = pairs0[cs + (p < len ? p : 0u)].y;
            id[j] = p < len ? v : 0u;
        }
        seg_output(id, len, cs, e3.w + tfront, rect, order, rect_sorted, offsets, block_first, bf_cap, listed - 1u, r_ok, s_wt);
        __syncthreads();
    }
    if (end <= begin) return;
    const uint32_t n = end - begin;
    // keys of the segment lie in [base_key, top_key): the key span of its buckets under the equalised mapping (the plan workgroup of
    // ds_scatter inverted the table)
    const uint32_t base_key = e2.z, top_key = e2.w;
    const uint32_t span = top_key > base_key ? top_key - base_key : 0u;
    const int nbits = span <= 1u ? 0 : 32 - __clz((int)(span - 1u));
    // (the segments tile [0, listed): exactly one ends at `listed`.  Its end BUCKET need not be 2047 -- behind the last
    // non-empty bucket come empty ones that start at `listed` too -- so the element count decides, not the bucket)
    const uint32_t last_listed = end == listed ? end - 1u : 0xFFFFFFFFu;
    const SegLdsMem L{s_key, s_idx, wave_cnt, wsum, s_wt};
    if (n <= (uint32_t)DS_CAP) {
        seg_sort_in_lds(L, pairs0, begin, n, base_key, nbits, tile_base, rect, order, rect_sorted, offsets, block_first, bf_cap, last_listed, r_ok);
        return;
    }
    // ---- oversized segment (round 6: bounded).  It is a run of WHOLE buckets in key order, so it is cut into groups of consecutive buckets of
    // at most DS_CAP elements, each sorted in LDS like an ordinary segment (with the segment's key span), one after the other.  Only a single
    // bucket beyond the capacity is left: its own key span is measured (min / max of its keys) -- a run of EQUAL keys, which the second-level
    // table isolates, needs no pass at all -- otherwise the radix passes go through global memory for that bucket alone.  Reported to the host
    // (which then prefers the LSD passes for a while) only when it really is slower than they are: passes through global memory, or more than
    // DS_SLOW_CHUNKS chunks of output by this one workgroup -- measured (profiles/r06_depth_distribution_probe.json, 750 000 keys on 48 values =
    // 15 600 ties per bucket): ~25 us per chunk of 4096, ds_segsort 92 us where the whole frame's LSD passes + scan take 84 us against this path's
    // 31 us + the longest segment; break-even at ~2 chunks.
    const uint32_t nbk = d1 - d0;      // <= 2047
    {   // s_bc[i] = elements of the segment's buckets 0 .. i (inclusive prefix of the bucket sizes; thread t owns entries 4t .. 4t+3).  The groups are
        // found by binary search in it: walking the sizes one by one was a chain of one LDS round trip per bucket, and the segment in front of a
        // crowd holds hundreds of (empty) buckets -- the crowd frame of tools/gpu_depth_distribution_probe.py spent 40 of its 69 us there
        static_assert(DS_NB == 4 * SG_THREADS, "four bucket sizes per thread");
        uint32_t c4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t b = 4u * (uint32_t)tid + (uint32_t)i;
            c4[i] = cnt_total[min(d0 + b, (uint32_t)DS_NB - 1u)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4u * (uint32_t)tid + (uint32_t)i >= nbk) c4[i] = 0u;
        uint32_t run = seg_excl_scan(c4[0] + c4[1] + c4[2] + c4[3], wsum, lane, w);
#pragma unroll
        for (int i = 0; i < 4; ++i) { run += c4[i]; s_bc[4 * tid + i] = run; }
    }
    __syncthreads();
    uint32_t g_begin = begin, g_d = 0, tb = tile_base;
#pragma unroll 1
    while (g_d < nbk) {      // (every thread does the same searches: workgroup-uniform control flow)
        // the group: buckets [g_d, g_e) with g_e the largest end whose elements fit the capacity -- at least one bucket
        const uint32_t before = g_d ? s_bc[g_d - 1u] : 0u;
        uint32_t lo = g_d + 1u, hi = nbk;      // largest e in [g_d + 1, nbk] with s_bc[e - 1] - before <= DS_CAP (or g_d + 1)
#pragma unroll 1
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1u) >> 1;
            if (s_bc[mid - 1u] - before <= (uint32_t)DS_CAP) lo = mid; else hi = mid - 1u;
        }
        const uint32_t g_e = lo, n_g = s_bc[g_e - 1u] - before;
        if (n_g != 0u && n_g <= (uint32_t)DS_CAP) {
            tb += seg_sort_in_lds(L, pairs0, g_begin, n_g, base_key, nbits, tb, rect, order, rect_sorted, offsets, block_first, bf_cap, last_listed, r_ok);
            __syncthreads();
        } else if (n_g != 0u && tie_helped) {
            // one bucket of n_g > DS_CAP keys, ONE key value by the table: its first chunk here, the others by the helpers (above); nothing behind it
            uint32_t id[SG_OUT];
#pragma unroll
            for (int j = 0; j < SG_OUT; ++j) id[j] = pairs0[g_begin + (uint32_t)j * SG_THREADS + (uint32_t)tid].y;
            seg_output(id, (uint32_t)DS_CAP, g_begin, tb, rect, order, rect_sorted, offsets, block_first, bf_cap, last_listed, r_ok, s_wt);
        } else if (n_g != 0u) {
            // one bucket of n_g > DS_CAP keys: its true key span
            uint32_t kmn = 0xFFFFFFFFu, kmx = 0u;
            for (uint32_t i0 = (uint32_t)tid; i0 < n_g; i0 += (uint32_t)SG_OUT * SG_THREADS) {      // eight keys per trip (clamped index: duplicates change no min / max)
                uint32_t kk[SG_OUT];
#pragma unroll
                for (int j = 0; j < SG_OUT; ++j) kk[j] = pairs0[g_begin + min(i0 + (uint32_t)j * SG_THREADS, n_g - 1u)].x;
#pragma unroll
                for (int j = 0; j < SG_OUT; ++j) { kmn = min(kmn, kk[j]); kmx = max(kmx, kk[j]); }
            }
            kmn = ~wave_incl_max_u32(~kmn);
            kmx = wave_incl_max_u32(kmx);
            if (lane == 63) { s_mm[0][w] = kmn; s_mm[1][w] = kmx; }
            __syncthreads();
#pragma unroll
            for (int k2 = 0; k2 < SG_WAVES; ++k2) { kmn = min(kmn, s_mm[0][k2]); kmx = max(kmx, s_mm[1][k2]); }
            const uint32_t sp = kmx - kmn + 1u;
            const int nb_g = sp <= 1u ? 0 : 32 - __clz((int)(sp - 1u));
            if (slow_word && tid == 0 && (nb_g > 0 || n_g > (uint32_t)DS_SLOW_CHUNKS * (uint32_t)DS_CAP))
                __hip_atomic_store(slow_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            SegGlobal m{pairs0 + g_begin, pairs1 + g_begin, kmn};
            uint32_t (*cnt32)[DS_PASS_BINS] = reinterpret_cast<uint32_t (*)[DS_PASS_BINS]>(&s_key[0][0]);
            __syncthreads();
            const int cur = seg_sort<SegGlobal, uint32_t>(m, n_g, nb_g, cnt32, wsum);
            const uint2* sorted = cur ? m.p1 : m.p0;
#pragma unroll 1
            for (uint32_t c0 = 0; c0 < n_g; c0 += (uint32_t)DS_CAP) {
                const uint32_t mm = min((uint32_t)DS_CAP, n_g - c0);
                uint32_t id[SG_OUT];
#pragma unroll
                for (int j = 0; j < SG_OUT; ++j) {
                    const uint32_t p = (uint32_t)j * SG_THREADS + (uint32_t)tid;
                    const uint32_t v = sorted[c0 + (p < mm ? p : 0u)].y;
                    id[j] = p < mm ? v : 0u;
                }
                tb += seg_output(id, mm, g_begin + c0, tb, rect, order, rect_sorted, offsets, block_first, bf_cap, last_listed, r_ok, s_wt);
            }
            __syncthreads();
        }
        g_begin += n_g;
        g_d = g_e;
    }
}

}  // namespace

size_t gsr_depth_bucket_blocks(int P) { return ((size_t)(P > 0 ? P : 1) + DS_ITEMS - 1) / DS_ITEMS; }
size_t gsr_depth_bucket_segments(int P) { return ((size_t)(P > 0 ? P : 1) + DS_SEG - 1) / DS_SEG + 1; }

// keys[P] (27-bit depth keys, GSR_DEPTH_KEY_CULLED for Gaussians without a tile), tiles[P], rect[P], frame = the words the
// key-producing kernel's last workgroup wrote, wg_range = the key ranges its workgroups left (gsr_frame.h)  ->  order[P], rect_sorted[P], offsets[P] (inclusive scan of the
// tile counts in depth order), block_first[bf_cap]
void gsr_launch_depth_bucket_sort(int P, const uint32_t* keys, const uint32_t* tiles, const uint2* rect, uint32_t* frame,
                                  const uint2* wg_range, int n_range, const GsrDepthSortBufs& b, uint32_t* order, uint2* rect_sorted,
                                  uint32_t* offsets, uint2* block_first, uint32_t block_first_cap, uint32_t* slow_word, hipStream_t st) {
    const int nblocks = (int)gsr_depth_bucket_blocks(P);
    const int nseg_cap = (int)gsr_depth_bucket_segments(P);
    hipLaunchKernelGGL(ds_hist, dim3(nblocks), dim3(DS_THREADS), 0, st, P, keys, tiles, frame, wg_range, n_range, b.eq_tab, b.bucket_of, b.cnt_tab, b.tile_tab);
    hipLaunchKernelGGL(ds_scan, dim3(DS_NB / 16), dim3(DS_THREADS), 0, st, nblocks, b.cnt_tab, b.tile_tab, b.cnt_total, b.tile_total);
    hipLaunchKernelGGL(ds_scatter, dim3(nblocks + 1), dim3(S3_THREADS), 0, st, P, nblocks, keys, b.bucket_of, frame, b.eq_tab, b.cnt_tab, b.cnt_total,
                       b.tile_total, b.pairs[0], order, offsets, rect_sorted, b.plan, nseg_cap);
    hipLaunchKernelGGL(ds_segsort, dim3(nseg_cap), dim3(SG_THREADS), 0, st, b.plan, frame, b.cnt_total, b.cnt_tab, b.tile_tab, nblocks, b.pairs[0], b.pairs[1], rect, order,
                       rect_sorted, offsets, block_first, block_first_cap, slow_word);
}
