// SIMT-on-CPU shim (TEST INFRASTRUCTURE, build container only): just enough of the HIP device language to run the INTEGER kernels of csrc/ --
// 256-thread workgroups, wave64 ballots, DPP moves, LDS, barriers -- lane by lane on a CPU, so that kernel source can be brought up and
// regression-tested without a GPU (tests/test_simt_tilesort_cpu.py).  Found ahead of the real <hip/hip_runtime.h> through -Itests/simt.
//
// Execution model (tests/simt/simt_runtime.h): one workgroup at a time, every lane a fiber (ucontext) on ONE OS thread.  A lane runs until it
// reaches a workgroup barrier or a wave-level operation (ballot, DPP move, wave barrier), where it waits for the other lanes.  Between two such
// points the lanes of a wave run one after the other FROM LANE 63 DOWN TO LANE 0 -- not in lock step: a kernel that lets one lane read what
// another lane of the same wave writes must have a wave-level operation in between (the csrc/ kernels use __builtin_amdgcn_wave_barrier there).
// The descending order makes the one lock-step idiom of the ranking loops come out as on the hardware: every lane reads a counter, then the
// LOWEST matching lane writes it back (`prior = cnt[d]; ...; if (leader) cnt[d] = prior + n;`) -- the writer runs after its readers.
// Not modelled: timing, bank conflicts, exec-masked wave operations (every live lane of a wave must take part in every wave-level operation).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
// LDS: one copy per process is enough (one workgroup runs at a time).  Every __shared__ array lands in the ELF section "simt_lds", which run_block()
// fills with a poison pattern before each workgroup: LDS holds garbage when a workgroup starts on the hardware, not zeros or the previous block's values.
// (clang only: g++ refuses to put the statics of inline / template functions and of plain functions into one named section -- there LDS keeps what the
// previous workgroup left, zeros at first.)
#ifdef __clang__
#define __shared__ static __attribute__((section("simt_lds")))
#else
#define __shared__ static
#endif
#define __launch_bounds__(...)
#define GSR_SIMT_SHIM 1

struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return {x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return {x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }      // (launches run synchronously: so does this)

namespace simt {
struct Idx { unsigned x, y, z; };
void syncthreads();
uint64_t ballot(bool pred);
int dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl);
void wave_barrier();
}  // namespace simt
extern simt::Idx threadIdx, blockIdx, blockDim, gridDim;

// hipLaunchKernelGGL runs the grid HERE, one workgroup after the other (tests/simt/simt_runtime.h), so the launchers of csrc/ -- grids, argument
// lists, launch order -- are exercised as they stand.  A workgroup that dead-locks or uses an unmodelled operation sets simt::launch_error.
#include <functional>
#include <vector>
namespace simt {
bool run_block(unsigned block, unsigned grid, int nthreads, const std::function<void()>& body);
bool schedule_shuffled();
uint32_t schedule_random();
extern const char* launch_error;
}  // namespace simt
template <class K, class... A>
static inline void simt_launch(K kernel, dim3 grid, dim3 block, A... args) {
    if (simt::schedule_shuffled()) {      // SIMT_SCHEDULE: the workgroups of the grid in a random order (simt_runtime.h)
        std::vector<unsigned> order(grid.x);
        for (unsigned b = 0; b < grid.x; ++b) order[b] = b;
        for (unsigned i = grid.x; i > 1; --i) std::swap(order[i - 1], order[simt::schedule_random() % i]);
        for (unsigned k = 0; k < grid.x && !simt::launch_error; ++k) (void)simt::run_block(order[k], grid.x, (int)block.x, [&] { kernel(args...); });
        return;
    }
    for (unsigned b = 0; b < grid.x && !simt::launch_error; ++b) (void)simt::run_block(b, grid.x, (int)block.x, [&] { kernel(args...); });
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) simt_launch(kernel, dim3(grid), dim3(block), __VA_ARGS__)

#define __syncthreads() simt::syncthreads()
#define __ballot(p) simt::ballot((bool)(p))
#define __builtin_amdgcn_ballot_w64(p) simt::ballot((bool)(p))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) simt::dpp((int)(old), (int)(src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_wave_barrier() simt::wave_barrier()
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __popcll(x) __builtin_popcountll((unsigned long long)(x))
// v_bitop3_b32: bit i of the result = truth table entry (a_i << 2 | b_i << 1 | c_i); v_bfe_i32: sign-extended bit field
static inline uint32_t __builtin_amdgcn_bitop3_b32(uint32_t a, uint32_t b, uint32_t c, unsigned tt) {
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i) r |= ((tt >> ((((a >> i) & 1u) << 2) | (((b >> i) & 1u) << 1) | ((c >> i) & 1u))) & 1u) << i;
    return r;
}
static inline int __builtin_amdgcn_sbfe(int v, unsigned off, unsigned width) {
    const uint32_t f = ((uint32_t)v >> off) & ((1u << width) - 1u);
    return (int)(f ^ (1u << (width - 1))) - (int)(1u << (width - 1));
}
#define __popc(x) __builtin_popcount((unsigned)(x))

using std::max;
using std::min;
static inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }      // one OS thread: atomic by construction
static inline int atomicAdd(int* p, int v) { const int o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }

// further device-language pieces of the integer kernels (depthsort.hip)
#define __ATOMIC_RELAXED_SHIM 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __HIP_MEMORY_SCOPE_AGENT 1
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_load(p, order, scope) (*(p))
#define __clz(x) ((x) ? __builtin_clz((unsigned)(x)) : 32)
#define __clzll(x) ((x) ? __builtin_clzll((unsigned long long)(x)) : 64)
#define __ffsll(x) __builtin_ffsll((long long)(x))
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// ---- pieces needed by the per-Gaussian and blend kernels (harnesses that define __HIPCC__ to get the device-side helpers of the csrc/ headers) ----
#define __hip_atomic_fetch_add(p, v, order, scope) ([&] { auto* p_ = (p); const auto o_ = *p_; *p_ = o_ + (v); return o_; }())
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __builtin_amdgcn_s_getreg(x) 0u
#define __builtin_amdgcn_sched_barrier(x) do { } while (0)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_s_sleep(x) do { } while (0)
#define __fdiv_rn(a, b) ((a) / (b))
#define __expf(x) expf(x)
static inline unsigned long long wall_clock64() { return 0ull; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
namespace simt { uint32_t readlane(uint32_t v, int lane); }
#define __builtin_amdgcn_readlane(v, lane) ((int)simt::readlane((uint32_t)(v), (lane)))
static inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; if (v > o) *p = v; return o; }

// v_permlane32_swap / v_permlane16_swap: (new first operand, new second operand) after swapping the first operand's upper half (odd rows) with the
// second operand's lower half (even rows); __shfl_xor
namespace simt { uint2 permlane32_swap(uint32_t a, uint32_t b); uint2 permlane16_swap(uint32_t a, uint32_t b); uint32_t shfl_xor(uint32_t v, int m); }
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) simt::permlane32_swap((a), (b))
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) simt::permlane16_swap((a), (b))
static inline int __shfl_xor(int v, int m, int = 64) { return (int)simt::shfl_xor((uint32_t)v, m); }
static inline uint32_t __shfl_xor(uint32_t v, int m, int = 64) { return simt::shfl_xor(v, m); }
static inline float __shfl_xor(float v, int m, int = 64) { uint32_t u; memcpy(&u, &v, 4); u = simt::shfl_xor(u, m); memcpy(&v, &u, 4); return v; }

// ---- shuffles / further atomics / explicit-rounding intrinsics (binning.hip, knn.hip, density.hip) ----
namespace simt { uint32_t shfl(uint32_t v, int src); uint32_t shfl_up(uint32_t v, int delta); }
static inline uint32_t __shfl(uint32_t v, int src, int = 64) { return simt::shfl(v, src); }
static inline int __shfl(int v, int src, int = 64) { return (int)simt::shfl((uint32_t)v, src); }
static inline float __shfl(float v, int src, int = 64) { return __uint_as_float(simt::shfl(__float_as_uint(v), src)); }
static inline uint32_t __shfl_up(uint32_t v, int d, int = 64) { return simt::shfl_up(v, d); }
static inline int __shfl_up(int v, int d, int = 64) { return (int)simt::shfl_up((uint32_t)v, d); }
static inline float __shfl_up(float v, int d, int = 64) { return __uint_as_float(simt::shfl_up(__float_as_uint(v), d)); }
static inline uint32_t atomicMin(uint32_t* p, uint32_t v) { const uint32_t o = *p; if (v < o) *p = v; return o; }
static inline uint32_t atomicMax(uint32_t* p, uint32_t v) { const uint32_t o = *p; if (v > o) *p = v; return o; }
static inline int atomicMin(int* p, int v) { const int o = *p; if (v < o) *p = v; return o; }
static inline int atomicMax(int* p, int v) { const int o = *p; if (v > o) *p = v; return o; }
static inline uint32_t atomicOr(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o | v; return o; }
static inline float atomicAdd(float* p, float v) { const float o = *p; *p = o + v; return o; }
#define __fmul_rn(a, b) ((float)(a) * (float)(b))
#define __fadd_rn(a, b) ((float)(a) + (float)(b))
#define __fsub_rn(a, b) ((float)(a) - (float)(b))
#define __fsqrt_rn(a) sqrtf(a)
#define __frcp_rn(a) (1.0f / (a))

// ---- raw buffer access (ssim.hip): a buffer resource = base + size in bytes; out-of-range loads return 0, out-of-range stores are dropped ----
struct __amdgpu_buffer_rsrc_t { char* base; uint32_t bytes; };
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, int /*stride*/, int num, int /*flags*/) { return {(char*)p, (uint32_t)num}; }
static inline int __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
    const uint32_t o = (uint32_t)voff + (uint32_t)soff;
    int v = 0;
    if ((uint64_t)o + 4 <= r.bytes) memcpy(&v, r.base + o, 4);
    return v;
}
static inline void __builtin_amdgcn_raw_buffer_store_b32(int v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
    const uint32_t o = (uint32_t)voff + (uint32_t)soff;
    if ((uint64_t)o + 4 <= r.bytes) memcpy(r.base + o, &v, 4);
}

// ---- the slice of the HIP RUNTIME API that csrc/gsr_api.cpp uses: one "device" whose memory is host memory, launches that have completed when
// hipLaunchKernelGGL returns (so every stream / event synchronisation is a no-op and a "mapped" host word is simply the same address) ----
#include <stdlib.h>
#include <time.h>
typedef struct simt_event* hipEvent_t;
struct simt_event { double ms; };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
#define hipHostMallocMapped 1u
#define hipHostMallocPortable 2u
#define hipHostMallocCoherent 4u
static inline const char* hipGetErrorString(hipError_t) { return simt::launch_error ? simt::launch_error : "SIMT shim"; }
static inline hipError_t hipGetLastError() { return simt::launch_error ? 719 : hipSuccess; }      // a workgroup that dead-locked / used an unmodelled operation
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : 2; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : 2; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline double simt_now_ms() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)calloc(1, sizeof(simt_event)); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->ms = simt_now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
#define hipStreamNonBlocking 1u
#define hipEventDisableTiming 2u
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)calloc(1, 8); return hipSuccess; }      // (launches complete inline: a second stream is a name)
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->ms - a->ms); return hipSuccess; }
