// wave64 / workgroup primitives shared by the binning and sorting kernels (device only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsrw {

constexpr int WG_THREADS = 256;
constexpr int WG_WAVES = WG_THREADS / 64;

// Wave-wide inclusive scans with DPP moves (gfx9 family: row_shr inside the 16-lane rows, row_bcast:15 / row_bcast:31
// across them) instead of __shfl_up: a __shfl_up is a ds_bpermute_b32 -- an LDS-crossbar round trip of ~100 cycles that
// the next step depends on, six in a row per scan -- while the DPP forms are plain VALU operand modifiers (the compiler
// fuses most steps into v_add_u32_dpp).  Lanes without a source (row start, masked row) read 0.  All lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_src_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
#define GSR_DPP_SCAN_STEPS(STEP) STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int /*lane*/) {
#define GSR_STEP(C, M) v += dpp_src_u32<C, M>(v);
    GSR_DPP_SCAN_STEPS(GSR_STEP)
#undef GSR_STEP
    return v;
}

__device__ __forceinline__ uint32_t wave_incl_max_u32(uint32_t v) {
#define GSR_STEP(C, M) v = max(v, dpp_src_u32<C, M>(v));
    GSR_DPP_SCAN_STEPS(GSR_STEP)
#undef GSR_STEP
    return v;
}

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, int /*lane*/) {
#define GSR_STEP(C, M) v += ((uint64_t)dpp_src_u32<C, M>((uint32_t)(v >> 32)) << 32) | dpp_src_u32<C, M>((uint32_t)v);
    GSR_DPP_SCAN_STEPS(GSR_STEP)
#undef GSR_STEP
    return v;
}

// Exclusive scan over the 256 threads of a workgroup of the per-thread totals of DPT values (thread t owns entries
// t*DPT .. t*DPT+DPT-1); returns the exclusive prefix of the thread's first entry.  wsum: 4-entry LDS scratch; two barriers.
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
    for (int k = 0; k < WG_WAVES; ++k)
        if (k < w) wbase += wsum[k];
    return wbase + incl - tsum;
}

// a / b and the remainder for a < 2^24, 0 < b < 2^16 with a quotient < 2^16: one v_rcp_f32 and a +-1 correction instead
// of the ~30-instruction integer division sequence
__device__ __forceinline__ uint32_t div_small(uint32_t a, uint32_t b, uint32_t& rem) {
    uint32_t q = (uint32_t)((float)a * __builtin_amdgcn_rcpf((float)b));
    int r = (int)a - (int)(q * b);
    if (r < 0) { --q; r += (int)b; }
    else if (r >= (int)b) { ++q; r -= (int)b; }
    rem = (uint32_t)r;
    return q;
}

// 64-bit mask of the lanes whose `digit` (low `bits` bits significant) equals this lane's, among the lanes in `valid_mask`.
// FOUR VALU instructions per bit: the bit sign-extended by one v_bfe_i32, one v_cmp for the ballot, and "mask & ~(ballot ^ bit)" as one
// v_bitop3_b32 per half (truth table 0x90 = a & ~(b ^ c)).  The plain form ("mask &= bit ? bal : ~bal") compiles to EIGHT (v_and, two v_cmp,
// v_cndmask 0 / -1, two v_xor, two v_and).  Every ranking kernel of the forward's binning chain spends most of its VALU here (emit_scatter: 16 items x 7
// bits per thread and block; bucket_scatter 16 x 6; ds_scatter 16 x 11; ds_segsort 2-3 passes x 9).  Measured, same box, interleaved
// (profiles/r05_ab_candidates.json): depth sort + emission + tile sort 0.182 -> 0.174 ms; bins bit-exact on the GPU suites.
__device__ __forceinline__ uint64_t match_digit(uint32_t digit, int bits, uint64_t valid_mask) {
    uint32_t lo = (uint32_t)valid_mask, hi = (uint32_t)(valid_mask >> 32);
    for (int b = 0; b < bits; ++b) {            // wave-uniform trip count
        const int sx = __builtin_amdgcn_sbfe((int)digit, (unsigned)b, 1u);      // 0 / -1
        const uint64_t bal = __ballot(sx != 0);
        lo = __builtin_amdgcn_bitop3_b32(lo, (uint32_t)bal, (uint32_t)sx, 0x90);
        hi = __builtin_amdgcn_bitop3_b32(hi, (uint32_t)(bal >> 32), (uint32_t)sx, 0x90);
    }
    return ((uint64_t)hi << 32) | lo;
}

}  // namespace gsrw
