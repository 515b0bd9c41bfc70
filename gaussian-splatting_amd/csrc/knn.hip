// Mean squared distance to the 3 nearest neighbours of every point -- replaces `simple_knn._C.distCUDA2` of the
// un-vendored submodule submodules/simple-knn (.gitmodules:1-3), called once per scene at
// scene/gaussian_model.py:159 to initialise the Gaussian scales.
//
// The result is the EXACT 3-NN (self excluded by index, coincident points count with distance 0), so any
// space-partitioning order works; this one is laid out for 64-wide waves:
//   knn_bbox         min / max of the cloud (ordered-uint atomics, one set per workgroup)
//   knn_morton       30-bit Morton code of every point inside that box  -> keys; vals = point index
//   radix sort       the rasterizer's own pair sort (sort.hip)
//   knn_boxes        one workgroup per BOX = 256 consecutive points of the sorted order: gathers them into a
//                    float4 stream (x, y, z, original index) and writes the box's AABB
//   knn_query        workgroup b answers the 256 points of box b (lane = query).  It first scans its own box (seeds
//                    the three best distances), then visits the other boxes outward in sort order b-1, b+1, b-2, ...
//                    A box is scanned only if some lane's distance to its AABB is below that lane's current third-best
//                    (exact pruning).  Box bounds and candidate points are wave-uniform, so they arrive through scalar
//                    loads and every lane spends 11 VALU ops per candidate (3 sub, 3 fma, 5 min/max insertion).
#include "gsr_internal.h"

namespace {

constexpr int KNN_BOX = 256;
constexpr float KNN_INF = 3.402823466e+38f;     // FLT_MAX: "no neighbour yet"

__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

__global__ void knn_bbox_init(uint32_t* bbox) {
    if (threadIdx.x < 3) bbox[threadIdx.x] = 0xFFFFFFFFu;       // min x,y,z
    else if (threadIdx.x < 6) bbox[threadIdx.x] = 0u;           // max x,y,z
}

__global__ void __launch_bounds__(256)
knn_bbox(int N, const float* __restrict__ pts, uint32_t* __restrict__ bbox) {
    __shared__ uint32_t s_min[3], s_max[3];
    if (threadIdx.x < 3) { s_min[threadIdx.x] = 0xFFFFFFFFu; s_max[threadIdx.x] = 0u; }
    __syncthreads();
    uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t o = f2ord(pts[i * 3 + k]);
            mn[k] = min(mn[k], o);
            mx[k] = max(mx[k], o);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = min(mn[k], (uint32_t)__shfl_xor((int)mn[k], off, 64));
            mx[k] = max(mx[k], (uint32_t)__shfl_xor((int)mx[k], off, 64));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_min[k], mn[k]); atomicMax(&s_max[k], mx[k]); }
    }
    __syncthreads();
    if (threadIdx.x < 3) { atomicMin(&bbox[threadIdx.x], s_min[threadIdx.x]); atomicMax(&bbox[3 + threadIdx.x], s_max[threadIdx.x]); }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {      // 10 bits -> every third bit
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ void __launch_bounds__(256)
knn_morton(int N, const float* __restrict__ pts, const uint32_t* __restrict__ bbox, uint32_t* __restrict__ keys,
           uint32_t* __restrict__ vals) {
    const float lo[3] = {ord2f(bbox[0]), ord2f(bbox[1]), ord2f(bbox[2])};
    const float hi[3] = {ord2f(bbox[3]), ord2f(bbox[4]), ord2f(bbox[5])};
    float inv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) inv[k] = hi[k] > lo[k] ? 1023.0f / (hi[k] - lo[k]) : 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t code = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float t = (pts[i * 3 + k] - lo[k]) * inv[k];
            t = fminf(1023.0f, fmaxf(0.0f, t));                 // NaN coordinates land in cell 0
            code |= spread10((uint32_t)t) << (2 - k);
        }
        keys[i] = code;
        vals[i] = (uint32_t)i;
    }
}

__global__ void __launch_bounds__(KNN_BOX)
knn_boxes(int N, const float* __restrict__ pts, const uint32_t* __restrict__ order, float4* __restrict__ sorted,
          float4* __restrict__ box_lo, float4* __restrict__ box_hi) {
    __shared__ float s_lo[4][3], s_hi[4][3];
    const int64_t j = (int64_t)blockIdx.x * KNN_BOX + threadIdx.x;
    float p[3] = {KNN_INF, KNN_INF, KNN_INF}, q[3] = {-KNN_INF, -KNN_INF, -KNN_INF};
    if (j < N) {
        const uint32_t i = order[j];
        const float x = pts[(int64_t)i * 3 + 0], y = pts[(int64_t)i * 3 + 1], z = pts[(int64_t)i * 3 + 2];
        sorted[j] = make_float4(x, y, z, __uint_as_float(i));
        p[0] = q[0] = x; p[1] = q[1] = y; p[2] = q[2] = z;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            p[k] = fminf(p[k], __shfl_xor(p[k], off, 64));
            q[k] = fmaxf(q[k], __shfl_xor(q[k], off, 64));
        }
        if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6][k] = p[k]; s_hi[threadIdx.x >> 6][k] = q[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float l[3], h[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            l[k] = fminf(fminf(s_lo[0][k], s_lo[1][k]), fminf(s_lo[2][k], s_lo[3][k]));
            h[k] = fmaxf(fmaxf(s_hi[0][k], s_hi[1][k]), fmaxf(s_hi[2][k], s_hi[3][k]));
        }
        box_lo[blockIdx.x] = make_float4(l[0], l[1], l[2], 0.f);
        box_hi[blockIdx.x] = make_float4(h[0], h[1], h[2], 0.f);
    }
}

struct Best3 {
    float b0, b1, b2;
    __device__ __forceinline__ void insert(float d) {
        const float c0 = fmaxf(b0, d);
        b0 = fminf(b0, d);
        const float c1 = fmaxf(b1, c0);
        b1 = fminf(b1, c0);
        b2 = fminf(b2, c1);
    }
};

__device__ __forceinline__ float dist2(float qx, float qy, float qz, const float4 p) {
    const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

__global__ void __launch_bounds__(KNN_BOX)
knn_query(int N, int n_boxes, const float4* __restrict__ sorted, const float4* __restrict__ box_lo,
          const float4* __restrict__ box_hi, float* __restrict__ out) {
    const int own = blockIdx.x;
    const int64_t j = (int64_t)own * KNN_BOX + threadIdx.x;
    const bool live = j < N;
    const float4 me = live ? sorted[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    Best3 best = {KNN_INF, KNN_INF, KNN_INF};
    // dead lanes (tail of the last box) must never ask for a scan: their third-best is pinned at -1
    if (!live) best.b0 = best.b1 = best.b2 = -1.0f;

    // own box: seeds the three best; a point is not its own neighbour
    {
        const int64_t base = (int64_t)own * KNN_BOX;
        const int cnt = (int)min((int64_t)KNN_BOX, (int64_t)N - base);
        for (int c = 0; c < cnt; ++c) {
            const float4 p = sorted[base + c];
            float d = dist2(me.x, me.y, me.z, p);
            d = (c == (int)threadIdx.x) ? KNN_INF : d;
            if (live) best.insert(d);
        }
    }
    // other boxes, outward in sort order
    for (int k = 1; k < n_boxes; ++k) {
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int b = side == 0 ? own - k : own + k;
            if (b < 0 || b >= n_boxes) continue;
            const float4 lo = box_lo[b], hi = box_hi[b];
            const float ex = fmaxf(0.0f, fmaxf(lo.x - me.x, me.x - hi.x));
            const float ey = fmaxf(0.0f, fmaxf(lo.y - me.y, me.y - hi.y));
            const float ez = fmaxf(0.0f, fmaxf(lo.z - me.z, me.z - hi.z));
            const float dbox = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
            // dbox is a lower bound of the distance to every point of the box computed with the same fma chain on
            // component offsets that are no larger in magnitude, so it can not exceed a true candidate's distance
            if (__builtin_amdgcn_ballot_w64(dbox < best.b2) == 0ull) continue;
            const int64_t base = (int64_t)b * KNN_BOX;
            const int cnt = (int)min((int64_t)KNN_BOX, (int64_t)N - base);
            for (int c = 0; c < cnt; ++c) {
                const float4 p = sorted[base + c];
                const float d = dist2(me.x, me.y, me.z, p);
                if (live) best.insert(d);
            }
        }
    }
    if (live) out[__float_as_uint(me.w)] = (best.b0 + best.b1 + best.b2) / 3.0f;
}

inline int grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

struct GsrKnnScratch {
    uint32_t* bbox;           // [8]
    uint32_t* keys[2];        // [N] x2
    uint32_t* vals[2];        // [N] x2
    uint32_t* sort_hist;
    uint32_t* digit_total;    // [2048]
    float4* sorted;           // [N]
    float4* box_lo;           // [n_boxes]
    float4* box_hi;
    size_t bytes;
};

static GsrKnnScratch knn_carve(char* base, int N) {
    GsrKnnScratch s;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off += (bytes + 127) & ~(size_t)127;
        return p;
    };
    const int n_boxes = (N + KNN_BOX - 1) / KNN_BOX;
    s.bbox = (uint32_t*)take(32);
    for (int i = 0; i < 2; ++i) s.keys[i] = (uint32_t*)take((size_t)N * 4);
    for (int i = 0; i < 2; ++i) s.vals[i] = (uint32_t*)take((size_t)N * 4);
    s.sort_hist = (uint32_t*)take((size_t)256 * (size_t)gsr_sort_blocks(N, true) * 4);
    s.digit_total = (uint32_t*)take(2048 * 4);
    s.sorted = (float4*)take((size_t)N * 16);
    s.box_lo = (float4*)take((size_t)n_boxes * 16);
    s.box_hi = (float4*)take((size_t)n_boxes * 16);
    s.bytes = off;
    return s;
}

size_t gsr_knn_scratch_bytes_impl(int N) { return knn_carve(nullptr, N < 0 ? 0 : N).bytes; }

void gsr_launch_knn(int N, const float* points, float* out, void* scratch, hipStream_t st) {
    GsrKnnScratch s = knn_carve((char*)scratch, N);
    const int n_boxes = (N + KNN_BOX - 1) / KNN_BOX;
    hipLaunchKernelGGL(knn_bbox_init, dim3(1), dim3(64), 0, st, s.bbox);
    hipLaunchKernelGGL(knn_bbox, dim3(grid_for(N)), dim3(256), 0, st, N, points, s.bbox);
    hipLaunchKernelGGL(knn_morton, dim3(grid_for(N)), dim3(256), 0, st, N, points, s.bbox, s.keys[0], s.vals[0]);
    const int cur = gsr_radix_sort_pairs(s.keys, s.vals, N, 30, 8, s.sort_hist, s.digit_total,
                                         N < (1 << 21) ? GSR_SORT_ITEMS_SMALL : GSR_SORT_ITEMS, st);
    hipLaunchKernelGGL(knn_boxes, dim3(n_boxes), dim3(KNN_BOX), 0, st, N, points, s.vals[cur], s.sorted, s.box_lo, s.box_hi);
    hipLaunchKernelGGL(knn_query, dim3(n_boxes), dim3(KNN_BOX), 0, st, N, n_boxes, s.sorted, s.box_lo, s.box_hi, out);
}
