// Binning kernels: prefix sum of tiles_touched in depth order, instance emission
// (K3 duplicateWithKeys) and per-tile ranges (K5 identifyTileRanges) of the reference rasterizer,
// redesigned for wave64 (algorithm: SURVEY.md Appendix A.3).
#include "gsr_internal.h"
#include "gsr_wave.h"
#include "gsr_frame.h"

namespace {

constexpr int SC_THREADS = 256;
constexpr int SC_IPT = GSR_SCAN_ITEMS / SC_THREADS;   // 4 consecutive items per thread

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) { return gsrw::wave_incl_scan_u32(v, lane); }

__device__ __forceinline__ uint32_t rect_tiles(const uint2 r) {
    return ((r.x >> 16) - (r.x & 0xFFFFu)) * ((r.y >> 16) - (r.y & 0xFFFFu));      // <= 2^24 (make_cam limits the grid)
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {      // total in every lane: DPP scan, then lane 63's value
    const uint64_t t = gsrw::wave_incl_scan_u64(v, 0);
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(t >> 32), 63) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)t, 63);
}

// pass 1: per-workgroup totals of the tile counts in depth order.  The count of a Gaussian is the area of its
// (band-clamped) tile rectangle; the gathered rectangles are written out in depth order (rect_sorted), so that pass 2
// and the emission stream them instead of repeating the 8-byte random gather.  Sums are 64-bit: a total beyond
// 2^31 - 1 must be DETECTED on the host, not wrapped.
// GATHER = false: rect_sorted was already filled by the last pass of the depth sort (sort.hip, os_pass<LAST>)
template <bool GATHER>
__global__ void __launch_bounds__(SC_THREADS)
scan_block_sums(int P, const uint32_t* __restrict__ order, const uint2* __restrict__ rect,
                uint2* __restrict__ rect_sorted, uint64_t* __restrict__ block_sums) {
    __shared__ uint64_t wsum[SC_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * GSR_SCAN_ITEMS + tid * SC_IPT;
    uint64_t s = 0;
    // Loads are unconditional (index clamped to the last Gaussian) and all issued before the first use: with the load inside
    // "if (base + k < P)" the compiler emits load -> s_waitcnt vmcnt(0) per element, SC_IPT serial round trips in a kernel
    // that is nothing but one round trip.
    uint2 r[SC_IPT];
    if (GATHER) {
        uint32_t id[SC_IPT];
#pragma unroll
        for (int k = 0; k < SC_IPT; ++k) id[k] = order[min(base + k, (int64_t)P - 1)];
#pragma unroll
        for (int k = 0; k < SC_IPT; ++k) r[k] = rect[id[k]];
    } else {
#pragma unroll
        for (int k = 0; k < SC_IPT; ++k) r[k] = rect_sorted[min(base + k, (int64_t)P - 1)];
    }
#pragma unroll
    for (int k = 0; k < SC_IPT; ++k)
        if (base + k < P) {
            if (GATHER) rect_sorted[base + k] = r[k];
            s += rect_tiles(r[k]);
        }
    s = wave_sum_u64(s);
    if (lane == 0) wsum[w] = s;
    __syncthreads();
    if (tid == 0) block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// pass 2: every workgroup sums the totals of the workgroups before it (a few KB from L2), then scans its
// own 1024 items.  offsets[j] = inclusive prefix in depth order (low 32 bits); the last workgroup publishes R (64-bit).
// Also fills the per-block table of the fused emission (tilesort.hip): block_first[k] = depth-order index of the
// Gaussian that owns instance k * GSR_TS_ITEMS, and, one past the last block, the last Gaussian with tiles.
__global__ void __launch_bounds__(SC_THREADS)
scan_finish(int P, const uint2* __restrict__ rect_sorted, const uint64_t* __restrict__ block_sums,
            uint32_t* __restrict__ offsets, uint2* __restrict__ block_first, uint32_t block_first_cap,
            uint32_t* __restrict__ num_rendered, uint32_t* host_word, uint32_t seq) {
    __shared__ uint64_t wsum[SC_THREADS / 64];
    __shared__ uint64_t wtot[SC_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * GSR_SCAN_ITEMS + tid * SC_IPT;
    // every load of the kernel is issued up front, unconditionally (clamped indices): the block sums of the workgroups
    // before this one (four per thread cover 1024 workgroups = 1 M Gaussians; beyond that a loop) and the thread's items
    uint2 rr[SC_IPT + 1];
#pragma unroll
    for (int k = 0; k <= SC_IPT; ++k) rr[k] = rect_sorted[min(base + k, (int64_t)P - 1)];
    uint64_t bs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[j] = block_sums[min(tid + j * SC_THREADS, (int)blockIdx.x)];      // (entry blockIdx.x exists)
    uint64_t pre = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (tid + j * SC_THREADS < (int)blockIdx.x) pre += bs[j];
    for (int b = tid + 4 * SC_THREADS; b < (int)blockIdx.x; b += SC_THREADS) pre += block_sums[b];
    pre = wave_sum_u64(pre);
    if (lane == 0) wsum[w] = pre;
    uint32_t v[SC_IPT + 1];
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k <= SC_IPT; ++k) {
        v[k] = (base + k < P) ? rect_tiles(rr[k]) : 0u;     // v[SC_IPT]: the next thread's first count
        if (k < SC_IPT) s += v[k];
    }
    const uint64_t incl = gsrw::wave_incl_scan_u64(s, lane);
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    uint64_t run = wsum[0] + wsum[1] + wsum[2] + wsum[3];
#pragma unroll
    for (int k = 0; k < SC_THREADS / 64; ++k)
        if (k < w) run += wtot[k];
    run += incl - s;
    constexpr uint64_t IT = GSR_TS_ITEMS;
#pragma unroll
    for (int k = 0; k < SC_IPT; ++k) {
        const uint64_t excl = run;
        run += v[k];
        if (base + k < P) {
            offsets[base + k] = (uint32_t)run;
            if (v[k] && run <= 0x7FFFFFFFull) {
                for (uint64_t b = (excl + IT - 1) / IT; b <= (run - 1) / IT; ++b)
                    if (b < block_first_cap) block_first[b] = make_uint2((uint32_t)(base + k), (uint32_t)excl);
                if (v[k + 1] == 0u) {     // counts are non-zero exactly on a prefix of the depth order: this is the last one
                    const uint64_t b = (run + IT - 1) / IT;
                    if (b < block_first_cap) block_first[b] = make_uint2((uint32_t)(base + k), (uint32_t)excl);
                }
            }
        }
        if (base + k == (int64_t)P - 1) {
            num_rendered[0] = (uint32_t)run;
            num_rendered[1] = (uint32_t)(run >> 32);
            if (host_word) {   // publish R to the spinning host: value first, then the sequence number (system-scope release)
                __hip_atomic_store(&host_word[0], (uint32_t)run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&host_word[2], (uint32_t)(run >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&host_word[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Instance emission in depth order.  One WORKGROUP per 64 consecutive Gaussians of the depth order; their instances
// form one contiguous run that the four waves write 64 slots at a time, round-robin (fully coalesced): output slot k
// (lane = k mod 64) finds its source Gaussian with a 6-step binary search over the group's inclusive prefix (held one
// per lane, fetched with ds_bpermute shuffles), then derives its tile from the Gaussian's rectangle (y-major, x
// fastest).  (One wave per group measured 48 us on the 1 M / 1080p frame: depth order packs the largest splats --
// hundreds of tiles each -- into the same few groups, and a lone wave walks their ~200 chunks as one chain of
// dependent cross-lane searches.)
constexpr int EMIT_WAVES = 4;        // waves per 64-Gaussian group (1: 48 us, 4: 39 us, 8: 49 us on the bench frame)

template <typename KeyT>
__global__ void __launch_bounds__(EMIT_WAVES * 64)
emit_instances(int P, int gx, const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
               const uint2* __restrict__ rect_sorted, KeyT* __restrict__ inst_keys, uint32_t* __restrict__ inst_vals,
               float4* __restrict__ splats) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * 64;
    if (j0 >= P) return;
    const int64_t j = j0 + lane;
    uint32_t id = 0, incl = 0, minx = 0, w = 1, miny = 0;
    const uint32_t base = j0 > 0 ? offsets[j0 - 1] : 0u;   // wave-uniform
    const int last = (int)((P - j0) < 64 ? (P - j0 - 1) : 63);
    const uint32_t total = offsets[j0 + last] - base;      // wave-uniform: instances of the whole group
    // most groups have fewer than 64 instances: waves 1..3 leave before the dependent order -> rect loads
    if (wv != 0 && (uint32_t)wv * 64u >= total) return;
    if (j < P) {
        id = order[j];
        incl = offsets[j] - base;
        const uint2 r = rect_sorted[j];
        minx = r.x & 0xFFFFu;
        w = (r.x >> 16) - minx;
        miny = r.y & 0xFFFFu;
    } else {
        incl = 0xFFFFFFFFu;   // never selected: search looks for the first prefix > k
    }
    const uint32_t excl_self = incl - ((j < P) ? (offsets[j] - (j > 0 ? offsets[j - 1] : 0u)) : 0u);
    // first emission index of this Gaussian -> 4th quad of its splat record (the blend backward writes its
    // per-instance gradient records at emission indices, see render_bwd.hip)
    if (splats && wv == 0 && j < P) reinterpret_cast<uint32_t*>(splats + (int64_t)id * 4 + 3)[2] = base + excl_self;
    for (uint32_t k0 = (uint32_t)wv * 64u; k0 < total; k0 += EMIT_WAVES * 64u) {
        const uint32_t k = k0 + lane;
        int lo = 0, hi = last;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi) >> 1;
            const uint32_t v = __shfl(incl, mid, 64);
            if (v > k) hi = mid; else lo = mid + 1;
        }
        if (lo > last) lo = last;
        const uint32_t src_excl = __shfl(excl_self, lo, 64);
        const uint32_t src_id = __shfl(id, lo, 64);
        const uint32_t src_minx = __shfl(minx, lo, 64);
        const uint32_t src_w = __shfl(w, lo, 64);
        const uint32_t src_miny = __shfl(miny, lo, 64);
        if (k < total) {
            const uint32_t local = k - src_excl;
            const uint32_t ry = local / src_w;
            const uint32_t rx = local - ry * src_w;
            const uint32_t tile = (src_miny + ry) * (uint32_t)gx + src_minx + rx;
            inst_keys[(int64_t)base + k] = (KeyT)tile;
            inst_vals[(int64_t)base + k] = src_id;
        }
    }
}

// ranges[t] = [first, last+1) of tile t's run in the sorted instance list; untouched tiles stay (0,0).
// Every lane takes 16 bytes of keys (8 x u16 / 4 x u32) in one load; the key before its first one comes from the
// neighbouring lane (ds_bpermute) -- one wave-wide load moves 1 KB instead of the 128 B of a 16-bit scalar load.
template <typename KeyT>
__global__ void __launch_bounds__(256)
tile_ranges(int64_t R, const KeyT* __restrict__ keys, uint2* __restrict__ ranges) {
    constexpr int VEC = 16 / (int)sizeof(KeyT);
    const int lane = threadIdx.x & 63;
    const int64_t nvec = (R + VEC - 1) / VEC;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x - lane; v0 < nvec; v0 += stride) {   // wave-uniform trips
        const int64_t v = v0 + lane, base = v * VEC;
        uint32_t k[VEC];
        if (base + VEC <= R) {
            const uint4 t = *reinterpret_cast<const uint4*>(keys + base);
            const uint32_t wd[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                k[j] = sizeof(KeyT) == 2 ? ((wd[j / 2] >> (16 * (j & 1))) & 0xFFFFu) : wd[j % 4];
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) k[j] = base + j < R ? (uint32_t)keys[base + j] : 0u;
        }
        uint32_t prev = (uint32_t)__shfl_up((int)k[VEC - 1], 1, 64);
        if (lane == 0 && base > 0 && base < R) prev = (uint32_t)keys[base - 1];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int64_t i = base + j;
            if (i < R) {
                const uint32_t t = k[j], p = j ? k[j - 1] : prev;
                if (i == 0) {
                    ranges[t].x = 0;
                } else if (p != t) {
                    ranges[p].y = (uint32_t)i;
                    ranges[t].x = (uint32_t)i;
                }
                if (i == R - 1) ranges[t].y = (uint32_t)R;
            }
        }
    }
}

// Two-axis sharding: the splat records of ALL Gaussians arrive from an all-gather, each produced by the
// rank that owns the Gaussian with the FULL-frame tile rectangle in its 4th quad.  This kernel copies them into the
// geometry state and derives what the binning stages need for THIS rank's band of tile rows: band-clamped rectangle,
// tile count, depth-sort key (culled / out-of-band Gaussians sort last) -- the same values preprocess_fwd_kernel writes
// when it runs with the band itself.
__global__ void __launch_bounds__(256)
splat_ingest(int P, const float4* __restrict__ records, int y0, int y1, float4* __restrict__ splats, uint2* __restrict__ rect,
             uint32_t* __restrict__ tiles, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, GsrFrameStatsDev fs) {
    GsrFrameAcc acc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 q0 = records[i * 4 + 0], q1 = records[i * 4 + 1], q2 = records[i * 4 + 2], q3 = records[i * 4 + 3];
        const uint32_t rx = __float_as_uint(q3.x), ry = __float_as_uint(q3.y), full = __float_as_uint(q3.w);
        const int minx = (int)(rx & 0xFFFFu), maxx = (int)(rx >> 16), miny = (int)(ry & 0xFFFFu), maxy = (int)(ry >> 16);
        const int bminy = miny < y0 ? y0 : (miny > y1 ? y1 : miny);
        const int bmaxy = maxy < y0 ? y0 : (maxy > y1 ? y1 : maxy);
        const uint32_t t = full ? (uint32_t)((maxx - minx) * (bmaxy - bminy)) : 0u;
        const uint2 rc = make_uint2((uint32_t)minx | ((uint32_t)maxx << 16), (uint32_t)bminy | ((uint32_t)bmaxy << 16));
        splats[i * 4 + 0] = q0;
        splats[i * 4 + 1] = q1;
        splats[i * 4 + 2] = q2;
        splats[i * 4 + 3] = make_float4(__uint_as_float(rc.x), __uint_as_float(rc.y), 0.f, __uint_as_float(t));
        rect[i] = rc;
        tiles[i] = t;
        const uint32_t key = gsr_depth_key(q2.y, t != 0u, acc.ovf);      // q2.y = view-space depth
        keys[i] = key;
        vals[i] = (uint32_t)i;
        acc.add(key, t);
    }
    gsr_frame_stats_commit(fs, acc.tiles, acc.kmin, acc.kmax, acc.ovf);
}

// fallback of the 27-bit depth sort (a listed Gaussian deeper than 13 107): the full 32-bit keys of round 2
__global__ void __launch_bounds__(256)
rekey_full(int P, const float4* __restrict__ splats, const uint32_t* __restrict__ tiles, uint32_t* __restrict__ keys,
           uint32_t* __restrict__ vals) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        keys[i] = tiles[i] ? __float_as_uint(splats[i * 4 + 2].y) : 0xFFFFFFFFu;
        vals[i] = (uint32_t)i;
    }
}

}  // namespace

int gsr_launch_splat_ingest(int P, const float* records, int y0, int y1, float4* splats, uint2* rect, uint32_t* tiles,
                            uint32_t* keys, uint32_t* vals, const GsrFrameStatsDev& fs, hipStream_t st) {
    int64_t nb = ((int64_t)P + 255) / 256;
    if (nb > GSR_FRAME_MAX_GROUPS) nb = GSR_FRAME_MAX_GROUPS;      // (gsr_frame.h: tickets)
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(splat_ingest, dim3((int)nb), dim3(256), 0, st, P, reinterpret_cast<const float4*>(records), y0, y1, splats,
                       rect, tiles, keys, vals, fs);
    return (int)nb;
}

void gsr_launch_rekey_full(int P, const float4* splats, const uint32_t* tiles, uint32_t* keys, uint32_t* vals, hipStream_t st) {
    int64_t nb = ((int64_t)P + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(rekey_full, dim3((int)nb), dim3(256), 0, st, P, splats, tiles, keys, vals);
}

void gsr_launch_scan_tiles(int P, const uint32_t* order, const uint2* rect, uint2* rect_sorted, uint32_t* offsets,
                           uint64_t* block_sums, uint2* block_first, uint32_t block_first_cap, uint32_t* num_rendered,
                           uint32_t* host_word, uint32_t seq, bool rect_already_sorted, hipStream_t st) {
    const int nb = (P + GSR_SCAN_ITEMS - 1) / GSR_SCAN_ITEMS;
    if (rect_already_sorted)
        hipLaunchKernelGGL(scan_block_sums<false>, dim3(nb), dim3(SC_THREADS), 0, st, P, order, rect, rect_sorted, block_sums);
    else
        hipLaunchKernelGGL(scan_block_sums<true>, dim3(nb), dim3(SC_THREADS), 0, st, P, order, rect, rect_sorted, block_sums);
    hipLaunchKernelGGL(scan_finish, dim3(nb), dim3(SC_THREADS), 0, st, P, rect_sorted, block_sums, offsets, block_first,
                       block_first_cap, num_rendered, host_word, seq);
}

void gsr_launch_emit(int P, int gx, const uint32_t* order, const uint32_t* offsets, const uint2* rect,
                     void* inst_keys, bool key16, uint32_t* inst_vals, float4* splats, hipStream_t st) {
    const int nb = (int)(((int64_t)P + 63) / 64);        // one workgroup per 64 Gaussians of the depth order
    if (key16)
        hipLaunchKernelGGL(emit_instances<uint16_t>, dim3(nb), dim3(EMIT_WAVES * 64), 0, st, P, gx, order, offsets, rect,
                           (uint16_t*)inst_keys, inst_vals, splats);
    else
        hipLaunchKernelGGL(emit_instances<uint32_t>, dim3(nb), dim3(EMIT_WAVES * 64), 0, st, P, gx, order, offsets, rect,
                           (uint32_t*)inst_keys, inst_vals, splats);
}

void gsr_launch_ranges(int64_t R, int n_tiles, const void* sorted_keys, bool key16, uint2* ranges, bool already_zeroed,
                       hipStream_t st) {
    if (!already_zeroed) (void)hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)n_tiles, st);
    if (R <= 0) return;
    int64_t nb = (R / (key16 ? 8 : 4) + 255) / 256 + 1;
    if (nb > 2048) nb = 2048;
    if (key16)
        hipLaunchKernelGGL(tile_ranges<uint16_t>, dim3((int)nb), dim3(256), 0, st, R, (const uint16_t*)sorted_keys, ranges);
    else
        hipLaunchKernelGGL(tile_ranges<uint32_t>, dim3((int)nb), dim3(256), 0, st, R, (const uint32_t*)sorted_keys, ranges);
}
