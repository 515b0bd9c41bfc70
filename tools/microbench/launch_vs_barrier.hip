// Micro-benchmark (measurement tool, not product): what does a dependency between two phases of a latency-bound chain cost
//   (a) as a kernel boundary in one HIP stream,            (b) the same chain replayed from a hipGraph,
//   (c) as a grid-wide barrier inside one persistent kernel (flat counter / per-XCD counters + one global counter)?
// Each phase touches `bytes` of memory per workgroup so that the boundary really has to make data visible device-wide.
//   hipcc --offload-arch=gfx950 -O3 -o launch_vs_barrier launch_vs_barrier.hip && ./launch_vs_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) phase_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[(i * 97) % n] = in[i] + 1u;
}

__device__ __forceinline__ void grid_barrier_flat(uint32_t* ctr, uint32_t target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE);      // agent scope by default for global atomics
        while (__atomic_load_n(ctr, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
// two-level: workgroups of one XCD (blockIdx % 8 on gfx942 / gfx950 round-robin dispatch) meet on their own counter; the last
// arrival of each XCD bumps the global one; everybody spins on the global word
__device__ __forceinline__ void grid_barrier_xcd(uint32_t* ctrs, uint32_t epoch, uint32_t per_xcd, uint32_t n_xcd) {
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t* mine = ctrs + 32 * (1 + (blockIdx.x & 7));
        const uint32_t old = __atomic_fetch_add(mine, 1u, __ATOMIC_ACQ_REL);
        if (old + 1 == epoch * per_xcd) __atomic_fetch_add(ctrs, 1u, __ATOMIC_RELEASE);
        while (__atomic_load_n(ctrs, __ATOMIC_ACQUIRE) < epoch * n_xcd) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

template <int MODE>
__global__ void __launch_bounds__(256) persistent_kernel(uint32_t* a, uint32_t* b, int n, int phases, uint32_t* ctrs, uint32_t base_epoch) {
    for (int p = 0; p < phases; ++p) {
        const uint32_t* in = (p & 1) ? b : a;
        uint32_t* out = (p & 1) ? a : b;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[(i * 97) % n] = in[i] + 1u;
        if (p + 1 < phases) {
            __threadfence();
            if (MODE == 0) grid_barrier_flat(ctrs, (base_epoch + p + 1) * gridDim.x);
            else grid_barrier_xcd(ctrs, base_epoch + p + 1, gridDim.x / 8, 8);
        }
    }
}

int main() {
    const int phases = 12, reps = 50;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t *a, *b, *ctrs;
    const int nmax = 1 << 22;
    CK(hipMalloc(&a, nmax * 4)); CK(hipMalloc(&b, nmax * 4)); CK(hipMalloc(&ctrs, 4096));
    CK(hipMemset(a, 0, nmax * 4)); CK(hipMemset(b, 0, nmax * 4));
    for (int n : {1 << 12, 1 << 20, 1 << 22}) {
        for (int grid : {256, 512, 1024}) {
            float ms;
            // (a) stream of dependent launches
            for (int w = 0; w < 3; ++w) for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(phase_kernel, dim3(grid), dim3(256), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, n);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(phase_kernel, dim3(grid), dim3(256), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, n);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms, e0, e1));
            const float t_launch = ms * 1e3f / (reps * phases);
            // (b) graph
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(phase_kernel, dim3(grid), dim3(256), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, n);
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms, e0, e1));
            const float t_graph = ms * 1e3f / (reps * phases);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            // (c) persistent kernel, flat and two-level barriers (grid <= resident capacity: 256 CUs x 8 workgroups of 256)
            float t_pers[2];
            for (int mode = 0; mode < 2; ++mode) {
                CK(hipMemsetAsync(ctrs, 0, 4096, st));
                uint32_t epoch = 0;
                auto launch = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(persistent_kernel<0>, dim3(grid), dim3(256), 0, st, a, b, n, phases, ctrs, epoch);
                    else hipLaunchKernelGGL(persistent_kernel<1>, dim3(grid), dim3(256), 0, st, a, b, n, phases, ctrs, epoch);
                    epoch += phases - 1;
                };
                for (int w = 0; w < 3; ++w) launch();
                CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < reps; ++r) launch();
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms, e0, e1));
                t_pers[mode] = ms * 1e3f / (reps * phases);
            }
            printf("n %8d grid %5d: per phase  stream launches %6.2f us   graph %6.2f us   persistent flat barrier %6.2f us   per-XCD barrier %6.2f us\n",
                   n, grid, t_launch, t_graph, t_pers[0], t_pers[1]);
            fflush(stdout);
        }
    }
    return 0;
}
