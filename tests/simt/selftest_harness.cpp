// Self-test of the SIMT-on-CPU shim (tests/test_simt_shim_cpu.py): the wave-level primitives the csrc/ kernels are written in, run on the
// shim and compared with plain loops -- DPP scans (row_shr / row_bcast / wave_shr / wave_shl), ballots and digit matching, workgroup scans,
// the cross-lane reductions of the blend backward (DPP + permlane swaps), readlane, __shfl_xor, and the case that once went wrong: a lane
// that LEAVES the kernel right after a wave-level operation must not take its value away from lanes that have not read it yet.
#define __HIPCC__ 1
#include "hip/hip_runtime.h"
#include <stdio.h>
#include "gsr_wave.h"
namespace tu_bwd {
#include "render_bwd.hip"
}
#include "simt_runtime.h"
using namespace gsrw;

static uint32_t o_scan[256], o_max[256], o_excl[256], o_shr[256], o_shl[256], o_xor[256];
static uint64_t o_bal[256], o_match[256], o_scan64[256];
static float o_r2[256], o_r4[256], o_s63[256], o_rl[256][10];

static void kern() {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    __shared__ uint32_t wsum[4];
    o_scan[tid] = wave_incl_scan_u32((uint32_t)(lane + 1), lane);
    o_scan64[tid] = wave_incl_scan_u64(0x100000000ull + (uint64_t)lane, lane);
    o_max[tid] = wave_incl_max_u32((uint32_t)((lane * 37) % 64));
    o_shr[tid] = dpp_src_u32<0x138, 0xf>((uint32_t)(lane + 100));
    o_shl[tid] = (uint32_t)__builtin_amdgcn_update_dpp(-2, lane + 200, 0x130, 0xf, 0xf, false);
    o_xor[tid] = (uint32_t)__shfl_xor(lane * 3, 5, 64);
    o_bal[tid] = __ballot((lane % 3) == 0);
    o_match[tid] = match_digit((uint32_t)(lane % 5), 3, __ballot(lane < 50));
    uint32_t v[1] = {(uint32_t)(tid % 7)};
    o_excl[tid] = block_excl_scan<1>(v, wsum, lane, w);
    const float a = (float)(lane + 1), b = (float)(1000 + 2 * lane), c = (float)(lane * lane), d = 0.5f * lane;
    o_r2[tid] = tu_bwd::reduce2(a, b);
    o_r4[tid] = tu_bwd::reduce4(a, b, c, d);
    o_s63[tid] = tu_bwd::wave_sum_to_lane63(a);
    float x[10];
    for (int i = 0; i < 10; ++i) x[i] = (float)(lane * 100 + i) * 0.5f;
    for (int i = 0; i < 10; ++i) o_rl[tid][i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[i]), 63));      // ... and the kernel ends here
}

extern "C" int simt_selftest(void) {
    if (!simt::run_block(0, 1, 256, [] { kern(); })) { printf("run_block failed: %s\n", simt::g.error ? simt::g.error : "?"); return 1000; }
    int bad = 0;
    auto fail = [&](const char* what, int t) { if (bad < 10) printf("%s: thread %d\n", what, t); ++bad; };
    uint32_t run = 0;
    float sa = 0, sb = 0, sc = 0, sd = 0;
    for (int l = 0; l < 64; ++l) { sa += l + 1; sb += 1000 + 2 * l; sc += l * l; sd += 0.5f * l; }
    for (int t = 0; t < 256; ++t) {
        const int lane = t & 63;
        if (o_scan[t] != (uint32_t)((lane + 1) * (lane + 2) / 2)) fail("wave_incl_scan_u32", t);
        if (o_scan64[t] != (uint64_t)(lane + 1) * 0x100000000ull + (uint64_t)(lane * (lane + 1) / 2)) fail("wave_incl_scan_u64", t);
        uint32_t m = 0; for (int l = 0; l <= lane; ++l) m = std::max(m, (uint32_t)((l * 37) % 64));
        if (o_max[t] != m) fail("wave_incl_max_u32", t);
        if (o_shr[t] != (lane ? (uint32_t)(lane - 1 + 100) : 0u)) fail("wave_shr:1", t);
        if (o_shl[t] != (lane < 63 ? (uint32_t)(lane + 1 + 200) : (uint32_t)-2)) fail("wave_shl:1", t);
        if (o_xor[t] != (uint32_t)((lane ^ 5) * 3)) fail("__shfl_xor", t);
        uint64_t bb = 0; for (int l = 0; l < 64; ++l) if (l % 3 == 0) bb |= 1ull << l;
        if (o_bal[t] != bb) fail("__ballot", t);
        uint64_t mm = 0; for (int l = 0; l < 50; ++l) if (l % 5 == lane % 5) mm |= 1ull << l;
        if (o_match[t] != mm) fail("match_digit", t);
        if (o_excl[t] != run) fail("block_excl_scan", t);
        run += (uint32_t)(t % 7);
        if (lane == 31 && o_r2[t] != sa) fail("reduce2 lane 31", t);
        if (lane == 63 && o_r2[t] != sb) fail("reduce2 lane 63", t);
        if (lane == 15 && o_r4[t] != sa) fail("reduce4 lane 15", t);
        if (lane == 31 && o_r4[t] != sc) fail("reduce4 lane 31", t);
        if (lane == 47 && o_r4[t] != sb) fail("reduce4 lane 47", t);
        if (lane == 63 && o_r4[t] != sd) fail("reduce4 lane 63", t);
        if (lane == 63 && o_s63[t] != sa) fail("wave_sum_to_lane63", t);
        for (int i = 0; i < 10; ++i) if (o_rl[t][i] != (float)(63 * 100 + i) * 0.5f) fail("readlane before the kernel's end", t);
    }
    return bad;
}

// ---- what SIMT_SCHEDULE is for: two kernels that are WRONG on a GPU and right in the shim's default order ----
static uint32_t o_racy[256], o_chain[64];
static void racy_handover() {      // wave 0 hands a value to the other waves through LDS -- without the barrier
    __shared__ uint32_t box;
    const int tid = threadIdx.x;
    if (tid == 0) box = 0;
    __syncthreads();
    if (tid == 63) box = 42;       // (lane 63 of wave 0: the first lane the default sweep runs)
    o_racy[tid] = box;             // missing __syncthreads()
}
static void block_chain(uint32_t* cell) {      // every workgroup assumes its predecessor has already run
    if (threadIdx.x == 0) { o_chain[blockIdx.x] = *cell; *cell = blockIdx.x + 1; }
}
extern "C" int simt_selftest_detects_order_dependence(void) {
    if (!simt::run_block(0, 1, 256, [] { racy_handover(); })) return -1;
    int wrong = 0;
    for (int t = 0; t < 256; ++t) wrong += o_racy[t] != 42;
    uint32_t cell = 0;
    hipLaunchKernelGGL(block_chain, 64, 64, 0, nullptr, &cell);
    int out_of_order = 0;
    for (int b = 0; b < 64; ++b) out_of_order += o_chain[b] != (uint32_t)b;
    return (wrong ? 1 : 0) | (out_of_order ? 2 : 0);
}

// ---- LDS holds garbage when a workgroup starts (the "simt_lds" section is poisoned before every workgroup), not what the previous workgroup left ----
static uint32_t o_lds[2][64];
static int lds_round = 0;
static void lds_fresh() {
    __shared__ uint32_t a[64];
    o_lds[lds_round][threadIdx.x] = a[threadIdx.x];
    a[threadIdx.x] = 0;
}
extern "C" int simt_selftest_lds_is_garbage(void) {
    for (lds_round = 0; lds_round < 2; ++lds_round)
        if (!simt::run_block((unsigned)lds_round, 2, 64, [] { lds_fresh(); })) return -1;
    int zeros = 0, same = 0;
    for (int t = 0; t < 64; ++t) { zeros += (o_lds[0][t] == 0) + (o_lds[1][t] == 0); same += o_lds[0][t] == o_lds[1][t]; }
    return zeros + same;      // 0: garbage both times, and different garbage
}
