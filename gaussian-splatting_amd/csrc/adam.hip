// Fused dense Adam step -- first "next row" beyond the rasterizer: the optimizer step that follows
// loss.backward() in every training iteration (train.py:177-186, torch.optim.Adam groups built at
// scene/gaussian_model.py:178-211, eps 1e-15).  torch's foreach Adam issues several multi-tensor kernels and measured
// 0.94 ms per step on the 59 floats x 1 M Gaussians of the bench scene; one fused pass reads p, g, m, v and writes p, m, v
// once (28 B per parameter = 1.65 GB -> HBM-bound).  Arithmetic follows torch.optim.Adam (no weight decay, no amsgrad):
//   m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "gsr_internal.h"
#include "gsr_adam_math.h"

namespace {

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float om_b1, float b2, float om_b2, float step_size,
                                      float inv_bc2_sqrt, float eps) {
    gsr_adam1(p, g, m, v, om_b1, b2, om_b2, step_size, inv_bc2_sqrt, eps);
}

__device__ __forceinline__ void adam4(float4& p, const float4& g, float4& m, float4& v, float om_b1, float b2, float om_b2,
                                      float step_size, float inv_bc2_sqrt, float eps) {
    adam1(p.x, g.x, m.x, v.x, om_b1, b2, om_b2, step_size, inv_bc2_sqrt, eps);
    adam1(p.y, g.y, m.y, v.y, om_b1, b2, om_b2, step_size, inv_bc2_sqrt, eps);
    adam1(p.z, g.z, m.z, v.z, om_b1, b2, om_b2, step_size, inv_bc2_sqrt, eps);
    adam1(p.w, g.w, m.w, v.w, om_b1, b2, om_b2, step_size, inv_bc2_sqrt, eps);
}
#ifdef GSR_SIMT_SHIM      // (tests/simt/: the kernel source compiled for the host, where a cache hint has no meaning)
__device__ __forceinline__ float4 ntload4(const float4* p) { return *p; }
__device__ __forceinline__ void ntstore4(float4* p, const float4& v) { *p = v; }
#else
typedef float float4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntload4(const float4* p) {
    const float4v t = __builtin_nontemporal_load(reinterpret_cast<const float4v*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void ntstore4(float4* p, const float4& v) {
    float4v t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<float4v*>(p));
}
#endif

// one tensor, walked by `nthreads` threads of which this one is number `tid0`
__device__ __forceinline__ void adam_walk(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                          int64_t n, float om_b1, float b2, float om_b2, float step_size, float inv_bc2_sqrt, float eps,
                                          int vec, int64_t tid0, int64_t stride) {
    if (vec) {
        const int64_t n4 = n >> 2;
        float4* p4 = reinterpret_cast<float4*>(p);
        const float4* g4 = reinterpret_cast<const float4*>(g);
        float4* m4 = reinterpret_cast<float4*>(m);
        float4* v4 = reinterpret_cast<float4*>(v);
        // two independent 64-byte groups per trip (8 x 16-byte loads in flight per lane), streamed past the caches:
        // every element is touched exactly once per step
        int64_t i = tid0;
        for (; i + stride < n4; i += 2 * stride) {
            const int64_t k = i + stride;
            float4 pa = ntload4(p4 + i), ma = ntload4(m4 + i), va = ntload4(v4 + i);
            const float4 ga = ntload4(g4 + i);
            float4 pb = ntload4(p4 + k), mb = ntload4(m4 + k), vb = ntload4(v4 + k);
            const float4 gb = ntload4(g4 + k);
            adam4(pa, ga, ma, va, om_b1, b2, om_b2, step_size, inv_bc2_sqrt, eps);
            adam4(pb, gb, mb, vb, om_b1, b2, om_b2, step_size, inv_bc2_sqrt, eps);
            ntstore4(p4 + i, pa); ntstore4(m4 + i, ma); ntstore4(v4 + i, va);
            ntstore4(p4 + k, pb); ntstore4(m4 + k, mb); ntstore4(v4 + k, vb);
        }
        for (; i < n4; i += stride) {
            float4 pp = ntload4(p4 + i), mm = ntload4(m4 + i), vv = ntload4(v4 + i);
            const float4 gg = ntload4(g4 + i);
            adam4(pp, gg, mm, vv, om_b1, b2, om_b2, step_size, inv_bc2_sqrt, eps);
            ntstore4(p4 + i, pp); ntstore4(m4 + i, mm); ntstore4(v4 + i, vv);
        }
        for (int64_t j = (n4 << 2) + tid0; j < n; j += stride) {
            float pp = p[j], mm = m[j], vv = v[j];
            adam1(pp, g[j], mm, vv, om_b1, b2, om_b2, step_size, inv_bc2_sqrt, eps);
            p[j] = pp; m[j] = mm; v[j] = vv;
        }
    } else {
        for (int64_t j = tid0; j < n; j += stride) {
            float pp = p[j], mm = m[j], vv = v[j];
            adam1(pp, g[j], mm, vv, om_b1, b2, om_b2, step_size, inv_bc2_sqrt, eps);
            p[j] = pp; m[j] = mm; v[j] = vv;
        }
    }
}

__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
            float om_b1, float b2, float om_b2, float step_size, float inv_bc2_sqrt, float eps, int vec) {
    adam_walk(p, g, m, v, n, om_b1, b2, om_b2, step_size, inv_bc2_sqrt, eps, vec, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
              (int64_t)gridDim.x * blockDim.x);
}

// Several tensors in ONE launch (gsr_adam_step_multi): a 3DGS model is five or six parameter tensors, four of them small --
// 6-20 us of kernel each, separated by launch boundaries.  Every tensor gets the block range [block_begin, next block_begin) and
// is walked exactly as by its own launch.
struct AdamTensorDev {
    float* p; const float* g; float* m; float* v;
    int64_t n;
    float om_b1, b2, om_b2, step_size, inv_bc2_sqrt, eps;
    int vec, block_begin;
};
struct AdamBatchDev {
    AdamTensorDev t[GSR_ADAM_MAX_TENSORS];
    int count, total_blocks;
};
__global__ void __launch_bounds__(256)
adam_multi_kernel(AdamBatchDev b) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < GSR_ADAM_MAX_TENSORS; ++j)
        if (j < b.count && (int)blockIdx.x >= b.t[j].block_begin) k = j;
    const AdamTensorDev& t = b.t[k];
    const int end = k + 1 < b.count ? b.t[k + 1].block_begin : b.total_blocks;
    const int64_t nthreads = (int64_t)(end - t.block_begin) * blockDim.x;
    adam_walk(t.p, t.g, t.m, t.v, t.n, t.om_b1, t.b2, t.om_b2, t.step_size, t.inv_bc2_sqrt, t.eps, t.vec,
              (int64_t)((int)blockIdx.x - t.block_begin) * blockDim.x + threadIdx.x, nthreads);
}

// Sparse variant (SparseGaussianAdam.step(visibility, N), train.py:180-183): the tensor is N rows of M elements; rows of
// Gaussians that were not visible in this iteration are skipped entirely -- parameter AND both moments stay untouched.
// [RECALLED, un-vendored source] the reference's kernel applies no bias correction:
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;  p += -lr m / (sqrt(v) + eps)
// Vectorised like the dense kernel: the tensor is walked as float4 groups (16-byte accesses, 1 KB per wave instruction);
// a group's first element is mapped to its row with ONE 32-bit division, the other three follow by comparison (a group
// spans at most four rows); groups whose rows are all invisible issue no load at all, so the traffic is that of the
// visible rows only.  Partially visible groups write the untouched components back unchanged.
__device__ __forceinline__ void sparse_adam1(float& p, float g, float& m, float& v, float lr, float b1, float om_b1, float b2,
                                             float om_b2, float eps) {
    gsr_sparse_adam1(p, g, m, v, lr, b1, om_b1, b2, om_b2, eps);
}

// one tensor, walked by the threads tid0, tid0 + stride, ... (its own launch, or its block range of the multi-tensor launch)
__device__ __forceinline__ void sparse_adam_walk(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                 const uint8_t* __restrict__ visible, int64_t N, uint32_t M, float lr, float b1, float om_b1, float b2,
                                                 float om_b2, float eps, int vec, int64_t tid0, int64_t stride) {
    const int64_t n = N * (int64_t)M;
    const int64_t n4 = vec ? (n >> 2) : 0;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (int64_t i = tid0; i < n4; i += stride) {
        const uint64_t e0 = (uint64_t)i << 2;
        // n < 2^32 for every tensor of a 3DGS model below 89 M Gaussians: one 32-bit division per four elements
        const uint64_t row0 = (e0 >> 32) ? e0 / M : (uint64_t)((uint32_t)e0 / M);
        uint32_t r = (uint32_t)(e0 - row0 * M);
        uint64_t row = row0;
        bool vis[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            vis[k] = visible[row] != 0;
            if (++r >= M) { r = 0; ++row; }
            if (row >= (uint64_t)N) row = (uint64_t)N - 1;       // (only past the last element of the tensor)
        }
        if (!(vis[0] | vis[1] | vis[2] | vis[3])) continue;
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        if (vis[0]) sparse_adam1(pp.x, gg.x, mm.x, vv.x, lr, b1, om_b1, b2, om_b2, eps);
        if (vis[1]) sparse_adam1(pp.y, gg.y, mm.y, vv.y, lr, b1, om_b1, b2, om_b2, eps);
        if (vis[2]) sparse_adam1(pp.z, gg.z, mm.z, vv.z, lr, b1, om_b1, b2, om_b2, eps);
        if (vis[3]) sparse_adam1(pp.w, gg.w, mm.w, vv.w, lr, b1, om_b1, b2, om_b2, eps);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    for (int64_t i = (n4 << 2) + tid0; i < n; i += stride) {
        if (!visible[i / M]) continue;
        float pp = p[i], mm = m[i], vv = v[i];
        sparse_adam1(pp, g[i], mm, vv, lr, b1, om_b1, b2, om_b2, eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

__global__ void __launch_bounds__(256)
sparse_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                   const uint8_t* __restrict__ visible, int64_t N, uint32_t M, float lr, float b1, float om_b1, float b2,
                   float om_b2, float eps, int vec) {
    sparse_adam_walk(p, g, m, v, visible, N, M, lr, b1, om_b1, b2, om_b2, eps, vec, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
                     (int64_t)gridDim.x * blockDim.x);
}

// Round 5: the six parameter tensors of a 3DGS model in ONE launch (gsr_sparse_adam_step_multi) -- SparseGaussianAdam.step used to issue one
// launch (and one ctypes call) per parameter group; at P ~ 100 K the whole training iteration is bound by the host's launch rate, and at 1 M
// five launch gaps of ~4 us separate kernels of 8-30 us.  Every tensor gets the block range [block_begin, next block_begin) and is walked exactly
// as by its own launch: results are the bits of gsr_sparse_adam_step on every tensor.
struct SparseAdamTensorDev {
    float* p; const float* g; float* m; float* v;
    uint32_t M;
    float lr, eps;
    int vec, block_begin;
};
struct SparseAdamBatchDev {
    SparseAdamTensorDev t[GSR_ADAM_MAX_TENSORS];
    const uint8_t* visible;
    int64_t N;
    float b1, om_b1, b2, om_b2;
    int count, total_blocks;
};
__global__ void __launch_bounds__(256)
sparse_adam_multi_kernel(SparseAdamBatchDev b) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < GSR_ADAM_MAX_TENSORS; ++j)
        if (j < b.count && (int)blockIdx.x >= b.t[j].block_begin) k = j;
    const SparseAdamTensorDev& t = b.t[k];
    const int end = k + 1 < b.count ? b.t[k + 1].block_begin : b.total_blocks;
    sparse_adam_walk(t.p, t.g, t.m, t.v, b.visible, b.N, t.M, t.lr, b.b1, b.om_b1, b.b2, b.om_b2, t.eps, t.vec,
                     (int64_t)((int)blockIdx.x - t.block_begin) * blockDim.x + threadIdx.x, (int64_t)(end - t.block_begin) * blockDim.x);
}

}  // namespace

void gsr_launch_sparse_adam(float* p, const float* g, float* m, float* v, const uint8_t* visible, int64_t N, int64_t M,
                            double lr, double beta1, double beta2, double eps, hipStream_t st) {
    const int64_t n = N * M;
    if (n <= 0) return;
    const int vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && M < (1ll << 31)) ? 1 : 0;
    const int64_t work = vec ? (n + 3) / 4 : n;
    int64_t nb = (work + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(sparse_adam_kernel, dim3((int)nb), dim3(256), 0, st, p, g, m, v, visible, N, (uint32_t)M, (float)lr,
                       (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, vec);
}

void gsr_launch_sparse_adam_multi(const GsrSparseAdamTensor* tensors, int count, const uint8_t* visible, int64_t N, double beta1, double beta2,
                                  hipStream_t st) {
    SparseAdamBatchDev b;
    b.count = 0;
    int blocks = 0;
    for (int i = 0; i < count && b.count < GSR_ADAM_MAX_TENSORS; ++i) {
        const GsrSparseAdamTensor& a = tensors[i];
        const int64_t n = N * a.M;
        if (n <= 0) continue;
        SparseAdamTensorDev& t = b.t[b.count++];
        t.p = a.param; t.g = a.grad; t.m = a.exp_avg; t.v = a.exp_avg_sq; t.M = (uint32_t)a.M;
        t.lr = (float)a.lr; t.eps = (float)a.eps;
        t.vec = ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0 && a.M < (1ll << 31)) ? 1 : 0;
        const int64_t work = t.vec ? (n + 3) / 4 : n;
        int64_t nb = (work + 255) / 256;
        if (nb > 8192) nb = 8192;      // (the single-tensor launcher's cap: the same walk per tensor)
        t.block_begin = blocks;
        blocks += (int)nb;
    }
    if (b.count == 0) return;
    b.visible = visible; b.N = N;
    b.b1 = (float)beta1; b.om_b1 = (float)(1.0 - beta1); b.b2 = (float)beta2; b.om_b2 = (float)(1.0 - beta2);
    b.total_blocks = blocks;
    hipLaunchKernelGGL(sparse_adam_multi_kernel, dim3(blocks), dim3(256), 0, st, b);
}

void gsr_launch_adam(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                     int step, hipStream_t st) {
    if (n <= 0) return;
    // hyper-parameters stay in double until the last moment, like torch: 1 - 0.999f would already be off by 1.3e-5 relative
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    const int vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) ? 1 : 0;
    int64_t work = vec ? (n + 3) / 4 : n;
    int64_t nb = (work + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((int)nb), dim3(256), 0, st, p, g, m, v, n, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), step_size, inv_bc2_sqrt, (float)eps, vec);
}

void gsr_launch_adam_multi(const GsrAdamTensor* tensors, int count, hipStream_t st) {
    AdamBatchDev b;
    b.count = 0;
    int blocks = 0;
    for (int i = 0; i < count && b.count < GSR_ADAM_MAX_TENSORS; ++i) {
        const GsrAdamTensor& a = tensors[i];
        if (a.n <= 0) continue;
        AdamTensorDev& t = b.t[b.count++];
        const double bc1 = 1.0 - pow(a.beta1, (double)a.step);
        const double bc2 = 1.0 - pow(a.beta2, (double)a.step);
        t.p = a.param; t.g = a.grad; t.m = a.exp_avg; t.v = a.exp_avg_sq; t.n = a.n;
        t.om_b1 = (float)(1.0 - a.beta1); t.b2 = (float)a.beta2; t.om_b2 = (float)(1.0 - a.beta2);
        t.step_size = (float)(a.lr / bc1); t.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2)); t.eps = (float)a.eps;
        t.vec = ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0) ? 1 : 0;
        const int64_t work = t.vec ? (t.n + 3) / 4 : t.n;
        int64_t nb = (work + 255) / 256;
        if (nb > 4096) nb = 4096;
        t.block_begin = blocks;
        blocks += (int)nb;
    }
    if (b.count == 0) return;
    b.total_blocks = blocks;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, st, b);
}
