// Per-Gaussian streaming kernels: forward preprocess, fused backward (K8+K9), markVisible (K10).
// Replaces preprocessCUDA / computeCov2DCUDA / checkFrustum of the un-vendored reference rasterizer
// (reference call site: gaussian_renderer/__init__.py:91-110; algorithm: SURVEY.md Appendix A.2 / A.6).
//
// COMPILE WITH -ffp-contract=off : the integer-deciding arithmetic (radius, tile rectangle) must match
// oracle/torch_oracle.py bit for bit (see gsr_math.h).
//
// gfx950 notes: HBM-streaming, one Gaussian per lane, 256-thread workgroups, grid-stride capped at 2048 workgroups.
// The 192-byte SH record per Gaussian (and its 192-byte gradient) dominates the streams.  A lane reading "its" record
// directly touches 64 different cache lines per load instruction (192-byte lane stride), which measured ~55 % of the
// achievable bandwidth; instead each wave moves the 12 KB SH block of its 64 consecutive Gaussians with fully
// coalesced 16-byte accesses through a wave-private LDS tile (row stride 52 floats: conflict-free ds_read_b128 /
// ds_write_b128 per lane), skipping rows of culled Gaussians at 16-byte granularity.  Camera matrices are
// wave-uniform -> scalar loads.
#include "gsr_internal.h"
#include "gsr_adam_math.h"
#include "gsr_frame.h"

namespace {

// occupancy attributes of the two per-Gaussian kernels (tuning builds: GSR_EXTRA_FLAGS=-DGSR_PRE_OCC_FWD=__attribute__((amdgpu_waves_per_eu(3,3)))).
// Default: none -- the kernels take 177-212 VGPRs = two waves per SIMD.  Three waves per SIMD (168 VGPRs; three 53 KB workgroups also fill
// the CU's LDS exactly) cost 25-49 spilled registers in round 2 (measured slower) and 8-36 with the round-4 kernels.
#ifndef GSR_PRE_OCC_FWD
#define GSR_PRE_OCC_FWD
#endif
// Round 6: the backward's covariance chain runs in fp64 (gsr_math.h gsr_project_backward_r); left alone the split-SH instantiation
// allocates 259 registers = ONE wave per SIMD.  Pinned at two waves per SIMD every instantiation fits 256 without a spill.
#ifndef GSR_PRE_OCC_BWD
#ifdef GSR_SIMT_SHIM      // (tests/simt: the kernel source compiled for the host)
#define GSR_PRE_OCC_BWD
#else
#define GSR_PRE_OCC_BWD __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
#endif

constexpr int SH_ROW = 52;        // LDS row stride in floats for a 48-float SH record (52*l mod 64 hits 16 distinct bank quads)

__device__ __forceinline__ void load_cam(const GsrCamDev& c, GsrCam& cam) {
    cam.W = c.W; cam.H = c.H; cam.gx = c.gx; cam.gy = c.gy;
    cam.focal_x = c.focal_x; cam.focal_y = c.focal_y; cam.limx = c.limx; cam.limy = c.limy;
    cam.scale_modifier = c.scale_modifier; cam.sh_degree = c.sh_degree; cam.M = c.M;
    cam.antialiasing = c.antialiasing; cam.tile_y0 = c.tile_y0; cam.tile_y1 = c.tile_y1; cam.snug = c.snug;
#pragma unroll
    for (int i = 0; i < 16; ++i) { cam.view[i] = c.view[i]; cam.proj[i] = c.proj[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) cam.campos[i] = c.campos[i];
}

// "All twelve loads are issued before the first LDS write."  A scheduling barrier alone does not guarantee that: it is not a
// memory operation, so the IR optimiser may sink every load across it, next to its LDS write -- which it did in the split-SH
// forward the moment the surrounding kernel changed (13 x load -> s_waitcnt vmcnt(0) -> ds_write_b128, tools/isa_audit.py).
// One empty asm statement that names ALL the loaded registers as operands can only be placed after the last load was issued.
#ifdef GSR_SIMT_SHIM      // (tests/simt/: the kernel source compiled for the host; a compiler-scheduling fence has no meaning there)
__device__ __forceinline__ void fence_loaded12(float4 (&)[12]) {}
#else
typedef float gsr_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fence_loaded12(float4 (&v)[12]) {
    gsr_v4f r[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) r[k] = (gsr_v4f){v[k].x, v[k].y, v[k].z, v[k].w};
    asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                      "+v"(r[9]), "+v"(r[10]), "+v"(r[11]));
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = make_float4(r[k].x, r[k].y, r[k].z, r[k].w);
    __builtin_amdgcn_sched_barrier(0);
}
#endif

// Cooperative, coalesced copy of the SH rows of the wave's 64 Gaussians [i0, i0+64) from global memory into the wave's
// LDS tile.  Only rows whose bit is set in `rows` are fetched (16 bytes at a time).  M == 16 only.
__device__ __forceinline__ void wave_load_sh16(const float* __restrict__ shs, int64_t i0, int P, uint64_t rows, int lane,
                                               float* tile) {
    const float4* src = reinterpret_cast<const float4*>(shs + i0 * 48);
    // all twelve loads are issued before the first LDS write: written as "tile[..] = src[idx]" per iteration the compiler
    // waited for every load before issuing the next one (twelve serial memory round trips per 64 Gaussians -- the fused
    // backward ran 141 us against 101 us for the split form, whose loader already had this shape)
    float4 v[12];
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int idx = it * 64 + lane;          // float4 index inside the 64 x 12 block
        const int g = idx / 12;
        v[it] = src[(((rows >> g) & 1ull) && i0 + g < P) ? idx : 0];      // (piece 0 is always valid)
    }
    fence_loaded12(v);      // keep the loads together: the scheduler otherwise sinks each one to its LDS write
    // ... and unconditional LDS writes (a conditional write makes the compiler sink the load into the branch, which brings
    // the serial round trips back); rows that were not fetched receive piece 0 and are never read
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int idx = it * 64 + lane;
        const int g = idx / 12, part = idx - g * 12;
        *reinterpret_cast<float4*>(tile + g * SH_ROW + part * 4) = v[it];
    }
    __builtin_amdgcn_wave_barrier();
}
// Two-step form of the same copy: the 12 coalesced 16-byte loads of the wave's block are ISSUED first (into registers) and
// written to the LDS tile later, so that their latency overlaps the projection arithmetic instead of following it.
__device__ __forceinline__ void wave_issue_sh16(const float* __restrict__ shs, int64_t i0, int P, int lane, float4 (&reg)[12]) {
    const float4* src = reinterpret_cast<const float4*>(shs + i0 * 48);
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int idx = it * 64 + lane;
        const int g = idx / 12;
        reg[it] = (i0 + g < P) ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ void wave_commit_sh16(const float4 (&reg)[12], uint64_t rows, int lane, float* tile) {
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int idx = it * 64 + lane;
        const int g = idx / 12, part = idx - g * 12;
        if ((rows >> g) & 1ull) *reinterpret_cast<float4*>(tile + g * SH_ROW + part * 4) = reg[it];
    }
    __builtin_amdgcn_wave_barrier();
}
// ... and back: every row (all 64, or up to P) is written -- zero rows included.
__device__ __forceinline__ void wave_store_sh16(float* __restrict__ dst_all, int64_t i0, int P, int lane, const float* tile) {
    float4* dst = reinterpret_cast<float4*>(dst_all + i0 * 48);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int idx = it * 64 + lane;
        const int g = idx / 12, part = idx - g * 12;
        if (i0 + g < P) dst[idx] = *reinterpret_cast<const float4*>(tile + g * SH_ROW + part * 4);
    }
    __builtin_amdgcn_wave_barrier();
}

// "separate_sh" call form (gaussian_renderer/__init__.py:82-100): coefficient 0 lives in dc[P,1,3], coefficients 1..15
// in rest[P,15,3].  Both blocks of the wave's 64 Gaussians are contiguous (768 B and 11520 B).
// Both blocks are copied with 16-byte accesses EXACTLY as they lie in memory -- rest rows at tile[g * 45], dc rows at
// tile[SPLIT_DC + g * 3] -- so the copy needs no per-float address arithmetic (the first version re-laid them out as padded
// 48-float rows: ~20 VALU instructions per 16 bytes on divisions by 45 plus four 4-byte LDS writes made the split-form
// forward 117 us against 74 us for the fused form).  Readers go through GsrShRowSplit (gsr_math.h): scalar LDS reads at
// an odd row stride, bank-conflict-free.
constexpr int SPLIT_DC = 64 * 45;      // float offset of the dc rows inside the tile (2880 + 192 = 3072 floats <= 64 * SH_ROW)
__device__ __forceinline__ void wave_load_sh_split_dense(const float* __restrict__ dc, const float* __restrict__ rest, int64_t i0,
                                                         int P, uint64_t rows, int lane, float* tile) {
    const int nrow = (int)((P - i0) < 64 ? (P - i0) : 64);
    const float* src = rest + i0 * 45;
    const float* sdc = dc + i0 * 3;
    const int nrest = nrow * 45, ndc = nrow * 3;
    // Branch-free issue of all thirteen 16-byte loads (a piece that is not wanted, or not completely inside the block, reads
    // the block's first 16 bytes instead -- always valid, always cached): with a branch per piece the compiler waits for
    // every load before it issues the next one, twelve serial round trips per 64 Gaussians.
    // rows != all: screen-sharded ranks fetch only the pieces that hold a wanted row.
    float4 v[12];
    bool ok[12];      // (kept for the address select only)
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int f = (it * 64 + lane) * 4;
        const int g0 = f / 45, g1 = (f + 3) / 45;
        const bool want = rows == ~0ull || ((((rows >> (g0 & 63)) | (rows >> (g1 & 63))) & 1ull) != 0);
        ok[it] = want && f + 3 < nrest;
        v[it] = *reinterpret_cast<const float4*>(src + (ok[it] ? f : 0));
    }
    const int fd = lane * 4;
    const bool okd = rows != 0ull && fd + 3 < ndc;
    const float4 vd = *reinterpret_cast<const float4*>(sdc + (okd ? fd : 0));
    fence_loaded12(v);      // keep the loads together: the scheduler otherwise sinks each one to its LDS write
    // ... and unconditional LDS writes (a conditional write makes the compiler sink the load into the branch, which brings
    // the serial round trips back): pieces that were not fetched land in rows nobody reads, or -- past the end of the two
    // blocks -- in the 256 spare floats at the end of the tile
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int f = (it * 64 + lane) * 4;
        *reinterpret_cast<float4*>(tile + (f < 64 * 45 ? f : 64 * 48 + lane * 4)) = v[it];
    }
    *reinterpret_cast<float4*>(tile + (fd < 64 * 3 ? SPLIT_DC + fd : 64 * 48 + lane * 4)) = vd;
    (void)ok; (void)okd;
    // ragged ends (only when the block is the last one and its float count is not a multiple of four)
    if (lane < (nrest & 3)) tile[(nrest & ~3) + lane] = src[(nrest & ~3) + lane];
    if (lane < (ndc & 3)) tile[SPLIT_DC + (ndc & ~3) + lane] = sdc[(ndc & ~3) + lane];
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void wave_store_sh_split_dense(float* __restrict__ d_dc, float* __restrict__ d_rest, int64_t i0, int P,
                                                          int lane, const float* tile) {
    const int nrow = (int)((P - i0) < 64 ? (P - i0) : 64);
    float* dst = d_rest + i0 * 45;
    const int nrest = nrow * 45;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 12; ++it) {
        const int f = (it * 64 + lane) * 4;
        if (f + 3 < nrest) {
            *reinterpret_cast<float4*>(dst + f) = *reinterpret_cast<const float4*>(tile + f);
        } else if (f < nrest) {
            for (int c = 0; c < 3; ++c)
                if (f + c < nrest) dst[f + c] = tile[f + c];
        }
    }
    float* ddc = d_dc + i0 * 3;
    const int ndc = nrow * 3, fd = lane * 4;
    if (fd + 3 < ndc) {
        *reinterpret_cast<float4*>(ddc + fd) = *reinterpret_cast<const float4*>(tile + SPLIT_DC + fd);
    } else if (fd < ndc) {
        for (int c = 0; c < 3; ++c)
            if (fd + c < ndc) ddc[fd + c] = tile[SPLIT_DC + fd + c];
    }
    __builtin_amdgcn_wave_barrier();
}
// SPLIT = the separate dc / rest form (always staged: the C ABI accepts it only for M == 16 and 16-byte aligned pointers)
// fs: frame statistics (gsr_frame.h) -- the kernel's last workgroup publishes R and the depth-key range
template <bool SPLIT>
__global__ void __launch_bounds__(256) GSR_PRE_OCC_FWD
preprocess_fwd_kernel(GsrCamDev camd, int P, const float* __restrict__ means3D, const float* __restrict__ shs,
                      const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                      const float* __restrict__ scales, const float* __restrict__ rotations,
                      const float* __restrict__ cov3D_precomp, float4* __restrict__ splats,
                      uint2* __restrict__ rect, uint32_t* __restrict__ tiles, uint32_t* __restrict__ clamped_out,
                      uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, int32_t* __restrict__ radii,
                      GsrFrameStatsDev fs) {
    __shared__ __attribute__((aligned(16))) float s_sh[4][64 * SH_ROW];
    GsrCam cam;
    load_cam(camd, cam);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* tile = s_sh[wv];
    const float* dc = SPLIT ? camd.sh_dc : nullptr;          // split form: `shs` holds coefficients 1..15
    const bool staged_sh = shs != nullptr && cam.M == 16;
    // bands of a quarter of the frame or more: still cheaper than a second latency phase (measured: 0.098 ms two-phase vs
    // 0.074 ms speculative for a quarter-frame band at 1 M Gaussians)
    const bool speculative = (cam.tile_y1 - cam.tile_y0) * 4 >= cam.gy;
    // wave-uniform trip count: every lane of a wave runs the same iterations (lanes past P idle inside)
    GsrFrameAcc acc;
    for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; i0 < P; i0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = i0 + lane;
        const bool in_range = i < P;
        GsrSplat sp;
        sp.radius = 0; sp.tiles = 0; sp.minx = sp.miny = sp.maxx = sp.maxy = 0; sp.depth = 0.f;
        float mean[3] = {0.f, 0.f, 0.f};
        bool vis = false;
        // full frame on one GPU: nearly every Gaussian needs its colour, so the SH block is requested up front, together
        // with the geometry, and one memory latency is exposed per iteration instead of two (projection -> visibility ->
        // SH loads).  The block goes to the LDS tile as soon as it lands -- for all rows, the visibility is not known yet --
        // so its 48 registers are free again before the projection arithmetic starts.  Under screen sharding only ~1/N of
        // the rows are needed: there the loads wait for the visibility mask.
        const bool spec_sh = staged_sh && speculative;
        float4 g_rot = make_float4(0.f, 0.f, 0.f, 0.f);
        float g_s[3] = {0.f, 0.f, 0.f}, g_op = 0.f;
        if (spec_sh) {
            float4 shreg[12];
            if (!SPLIT) wave_issue_sh16(shs, i0, P, lane, shreg);
            if (in_range) {     // the geometry loads ride in the same latency window
                mean[0] = means3D[i * 3 + 0]; mean[1] = means3D[i * 3 + 1]; mean[2] = means3D[i * 3 + 2];
                g_op = opacities[i];
                if (!cov3D_precomp) {
                    g_s[0] = scales[i * 3 + 0]; g_s[1] = scales[i * 3 + 1]; g_s[2] = scales[i * 3 + 2];
                    g_rot = reinterpret_cast<const float4*>(rotations)[i];
                }
            }
            if (SPLIT) wave_load_sh_split_dense(dc, shs, i0, P, ~0ull, lane, tile);
            else wave_commit_sh16(shreg, ~0ull, lane, tile);
        }
        if (in_range) {
            if (!spec_sh) {
                mean[0] = means3D[i * 3 + 0]; mean[1] = means3D[i * 3 + 1]; mean[2] = means3D[i * 3 + 2];
                g_op = opacities[i];
            }
            float cov[6];
            if (cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; ++k) cov[k] = cov3D_precomp[i * 6 + k];
            } else {
                if (!spec_sh) {
                    g_s[0] = scales[i * 3 + 0]; g_s[1] = scales[i * 3 + 1]; g_s[2] = scales[i * 3 + 2];
                    g_rot = reinterpret_cast<const float4*>(rotations)[i];
                }
                const float q[4] = {g_rot.x, g_rot.y, g_rot.z, g_rot.w};
                gsr_cov3d(g_s, cam.scale_modifier, q, cov);
            }
            vis = gsr_project(cam, mean, cov, g_op, sp);
        }
        // colours are needed only by Gaussians that touch this rank's band of tile rows (all visible ones on one GPU):
        // with the screen sharded over N GPUs each rank streams ~1/N of the SH records
        const bool need_color = vis && sp.tiles > 0;
        if (staged_sh) {
            const uint64_t rows = __ballot(need_color);
            if (rows) {
                if (spec_sh) { /* already staged */ }
                else if (SPLIT) wave_load_sh_split_dense(dc, shs, i0, P, rows, lane, tile);
                else wave_load_sh16(shs, i0, P, rows, lane, tile);
            }
        }
        if (!in_range) continue;
        uint32_t clampbits = 0;
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
        if (vis) {
            float rgb[3] = {0.f, 0.f, 0.f};
            if (!need_color) {
                // visible but outside the band: never blended here, record keeps geometry only
            } else if (colors_precomp) {
                rgb[0] = colors_precomp[i * 3 + 0]; rgb[1] = colors_precomp[i * 3 + 1]; rgb[2] = colors_precomp[i * 3 + 2];
            } else if (staged_sh) {
                if (SPLIT) gsr_sh_to_rgb_row(cam.sh_degree, 16, GsrShRowSplit{tile + SPLIT_DC + lane * 3, tile + lane * 45}, mean, cam.campos, rgb, clampbits);
                else gsr_sh_to_rgb_row(cam.sh_degree, 16, GsrShRowAligned{tile + lane * SH_ROW}, mean, cam.campos, rgb, clampbits);
            } else if (!SPLIT) {
                gsr_sh_to_rgb(cam.sh_degree, cam.M, shs + i * (int64_t)cam.M * 3, mean, cam.campos, rgb, clampbits);
            }
            q0 = make_float4(sp.px, sp.py, sp.conA, sp.conB);
            q1 = make_float4(sp.conC, sp.opacity, rgb[0], rgb[1]);
            // tau = 2 ln(255 opacity) + slack: a splat reaches alpha >= 1/255 only where its quadratic form is <= tau
            // (box-cull threshold of the blend kernels); 1/depth feeds the inverse-depth image.
            q2 = make_float4(rgb[2], sp.depth, sp.tau, gsr_inv_depth(sp.depth));
        }
        const uint2 rc = make_uint2(sp.minx | (sp.maxx << 16), sp.miny | (sp.maxy << 16));
        splats[i * 4 + 0] = q0;
        splats[i * 4 + 1] = q1;
        splats[i * 4 + 2] = q2;
        // 4th quad: tile rectangle, first emission index (filled in by emit_instances), tiles_touched -- as raw bits
        splats[i * 4 + 3] = make_float4(__uint_as_float(rc.x), __uint_as_float(rc.y), 0.f, __uint_as_float(sp.tiles));
        rect[i] = rc;
        tiles[i] = sp.tiles;
        if (clamped_out) clamped_out[i] = clampbits;
        radii[i] = sp.radius;
        // depth-sort key (gsr_internal.h): 27 bits of bits(depth) - bits(0.2f); Gaussians with no tile in the band sort last.
        const uint32_t key = gsr_depth_key(sp.depth, sp.tiles != 0u, acc.ovf);
        keys[i] = key;
        vals[i] = (uint32_t)i;
        acc.add(key, sp.tiles);
    }
    gsr_frame_stats_commit(fs, acc.tiles, acc.kmin, acc.kmax, acc.ovf);
}


// The Adam step of the two SH tensors straight from the gradient tile (gsr_backward_preprocess_sh_adam): the gradient never
// travels to HBM and back, and the parameter rows the kernel loaded a moment ago are re-read from L2.  Same pieces, same 16-byte
// accesses as wave_store_sh_split_dense; four pieces per trip (p, m, v of four pieces = 48 registers in flight).
// rows: bit r set = row r takes part (dense Adam: all rows; SparseGaussianAdam: the visible ones).  A piece spans at most two
// rows; components of rows that do not take part are written back unchanged.
__device__ __forceinline__ void adam_piece(float4& p, const float4& g, float4& m, float4& v, const float (&a)[6], int sparse, bool k0, bool k1,
                                           bool k2, bool k3) {
    if (sparse) {
        if (k0) gsr_sparse_adam1(p.x, g.x, m.x, v.x, a[0], a[1], a[2], a[3], a[4], a[5]);
        if (k1) gsr_sparse_adam1(p.y, g.y, m.y, v.y, a[0], a[1], a[2], a[3], a[4], a[5]);
        if (k2) gsr_sparse_adam1(p.z, g.z, m.z, v.z, a[0], a[1], a[2], a[3], a[4], a[5]);
        if (k3) gsr_sparse_adam1(p.w, g.w, m.w, v.w, a[0], a[1], a[2], a[3], a[4], a[5]);
    } else {
        gsr_adam1(p.x, g.x, m.x, v.x, a[0], a[1], a[2], a[3], a[4], a[5]);
        gsr_adam1(p.y, g.y, m.y, v.y, a[0], a[1], a[2], a[3], a[4], a[5]);
        gsr_adam1(p.z, g.z, m.z, v.z, a[0], a[1], a[2], a[3], a[4], a[5]);
        gsr_adam1(p.w, g.w, m.w, v.w, a[0], a[1], a[2], a[3], a[4], a[5]);
    }
}
__device__ __forceinline__ void wave_adam_sh_split_dense(const GsrShAdamDev& ad, int64_t i0, int P, uint64_t rows, int lane, const float* tile) {
    const int nrow = (int)((P - i0) < 64 ? (P - i0) : 64);
    const int nrest = nrow * 45, ndc = nrow * 3;
    float* pr = ad.rest + i0 * 45; float* mr = ad.rest_m + i0 * 45; float* vr = ad.rest_v + i0 * 45;
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int it0 = 0; it0 < 12; it0 += 4) {     // (a real loop: unrolled, the three trips' loads were hoisted together -- 360 registers)
        float4 pp[4], mm[4], vv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {           // (pieces beyond the block read its first 16 bytes: always valid)
            const int f = ((it0 + k) * 64 + lane) * 4;
            const int fo = f + 3 < nrest ? f : 0;
            pp[k] = *reinterpret_cast<const float4*>(pr + fo);
            mm[k] = *reinterpret_cast<const float4*>(mr + fo);
            vv[k] = *reinterpret_cast<const float4*>(vr + fo);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int f = ((it0 + k) * 64 + lane) * 4;
            if (f + 3 < nrest) {
                const float4 g = *reinterpret_cast<const float4*>(tile + f);
                const int r0 = f / 45, r1 = (f + 1) / 45, r2 = (f + 2) / 45, r3 = (f + 3) / 45;
                adam_piece(pp[k], g, mm[k], vv[k], ad.rest_a, ad.sparse, (rows >> r0) & 1ull, (rows >> r1) & 1ull, (rows >> r2) & 1ull,
                           (rows >> r3) & 1ull);
                *reinterpret_cast<float4*>(pr + f) = pp[k];
                *reinterpret_cast<float4*>(mr + f) = mm[k];
                *reinterpret_cast<float4*>(vr + f) = vv[k];
            } else if (f < nrest) {             // ragged end of the last block
                for (int c = 0; c < 3; ++c)
                    if (f + c < nrest && ((rows >> ((f + c) / 45)) & 1ull)) {
                        float p1 = pr[f + c], m1 = mr[f + c], v1 = vr[f + c];
                        if (ad.sparse) gsr_sparse_adam1(p1, tile[f + c], m1, v1, ad.rest_a[0], ad.rest_a[1], ad.rest_a[2], ad.rest_a[3], ad.rest_a[4], ad.rest_a[5]);
                        else gsr_adam1(p1, tile[f + c], m1, v1, ad.rest_a[0], ad.rest_a[1], ad.rest_a[2], ad.rest_a[3], ad.rest_a[4], ad.rest_a[5]);
                        pr[f + c] = p1; mr[f + c] = m1; vr[f + c] = v1;
                    }
            }
        }
    }
    float* pd = ad.dc + i0 * 3; float* md = ad.dc_m + i0 * 3; float* vd = ad.dc_v + i0 * 3;
    const int fd = lane * 4;
    if (fd + 3 < ndc) {
        float4 p4 = *reinterpret_cast<const float4*>(pd + fd), m4 = *reinterpret_cast<const float4*>(md + fd), v4 = *reinterpret_cast<const float4*>(vd + fd);
        const float4 g = *reinterpret_cast<const float4*>(tile + SPLIT_DC + fd);
        adam_piece(p4, g, m4, v4, ad.dc_a, ad.sparse, (rows >> (fd / 3)) & 1ull, (rows >> ((fd + 1) / 3)) & 1ull, (rows >> ((fd + 2) / 3)) & 1ull,
                   (rows >> ((fd + 3) / 3)) & 1ull);
        *reinterpret_cast<float4*>(pd + fd) = p4;
        *reinterpret_cast<float4*>(md + fd) = m4;
        *reinterpret_cast<float4*>(vd + fd) = v4;
    } else if (fd < ndc) {
        for (int c = 0; c < 3; ++c)
            if (fd + c < ndc && ((rows >> ((fd + c) / 3)) & 1ull)) {
                float p1 = pd[fd + c], m1 = md[fd + c], v1 = vd[fd + c];
                if (ad.sparse) gsr_sparse_adam1(p1, tile[SPLIT_DC + fd + c], m1, v1, ad.dc_a[0], ad.dc_a[1], ad.dc_a[2], ad.dc_a[3], ad.dc_a[4], ad.dc_a[5]);
                else gsr_adam1(p1, tile[SPLIT_DC + fd + c], m1, v1, ad.dc_a[0], ad.dc_a[1], ad.dc_a[2], ad.dc_a[3], ad.dc_a[4], ad.dc_a[5]);
                pd[fd + c] = p1; md[fd + c] = m1; vd[fd + c] = v1;
            }
    }
    __builtin_amdgcn_wave_barrier();
}

// `shs` carries no __restrict__: in the ADAM instantiation it is adam.rest, which the kernel also writes (the update in place)
template <bool SPLIT, bool ADAM = false>
__global__ void __launch_bounds__(256) GSR_PRE_OCC_BWD
preprocess_bwd_kernel(GsrCamDev camd, int P, const float* __restrict__ means3D, const float* shs,
                      const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                      const float* __restrict__ scales, const float* __restrict__ rotations,
                      const float* __restrict__ cov3D_precomp, const int32_t* __restrict__ radii,
                      const uint32_t* __restrict__ clamped, const float4* __restrict__ grads,
                      float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacity,
                      float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
                      float* __restrict__ dL_dscales, float* __restrict__ dL_drotations, GsrShAdamDev adam) {
    __shared__ __attribute__((aligned(16))) float s_sh[4][64 * SH_ROW];
    GsrCam cam;
    load_cam(camd, cam);
    const int M = cam.M;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* tile = s_sh[wv];
    const float* dc = SPLIT ? camd.sh_dc : nullptr;     // split form: `shs` / dL_dsh hold coefficients 1..15, dc / dL_ddc coefficient 0
    float* dL_ddc = SPLIT ? camd.dL_dsh_dc : nullptr;
    const bool staged_sh = shs != nullptr && M == 16;
    constexpr int sh_row = SH_ROW;
    for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; i0 < P; i0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = i0 + lane;
        const bool in_range = i < P;
        float dmean[3] = {0.f, 0.f, 0.f};
        float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // (what is stored; the chain itself runs in GsrBwdReal, gsr_math.h)
        float dscale[3] = {0.f, 0.f, 0.f};
        float drot[4] = {0.f, 0.f, 0.f, 0.f};
        float dop = 0.f;
        float dm2x = 0.f, dm2y = 0.f;
        float drgb[3] = {0.f, 0.f, 0.f};
        // One latency phase per iteration: the per-Gaussian inputs (radius, 48-byte gradient record, position, scale,
        // rotation, opacity) are requested first and the SH block right behind them -- for all rows: 88 % of them are
        // visible on the bench frame, and waiting for the visibility mask first costs a second full memory latency.
        int rad = 0;
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0, q4 = g0;
        float mean[3] = {0.f, 0.f, 0.f}, s[3] = {0.f, 0.f, 0.f}, q[4], cov[6], op_in = 0.f;
        if (in_range) {
            rad = radii[i];
            g0 = grads[i * 3 + 0]; g1 = grads[i * 3 + 1]; g2 = grads[i * 3 + 2];
            mean[0] = means3D[i * 3 + 0]; mean[1] = means3D[i * 3 + 1]; mean[2] = means3D[i * 3 + 2];
            op_in = opacities[i];
            if (cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; ++k) cov[k] = cov3D_precomp[i * 6 + k];
            } else {
                s[0] = scales[i * 3 + 0]; s[1] = scales[i * 3 + 1]; s[2] = scales[i * 3 + 2];
                q4 = reinterpret_cast<const float4*>(rotations)[i];
            }
        }
        if (staged_sh) {
            if (SPLIT) wave_load_sh_split_dense(dc, shs, i0, P, ~0ull, lane, tile);
            else wave_load_sh16(shs, i0, P, ~0ull, lane, tile);
        }
        const bool vis = in_range && rad > 0;
        if (vis) {
            GsrSplatGrad g;
            g.dpx = g0.x; g.dpy = g0.y; g.dconA = g0.z; g.dconB = g0.w;
            g.dconC = g1.x; g.dopacity = g1.y; g.dr = g1.z; g.dg = g1.w;
            g.db = g2.x; g.dinvdepth = g2.y;
            drgb[0] = g.dr; drgb[1] = g.dg; drgb[2] = g.db;
            GsrBwdReal covr[6], dcovr[6] = {0, 0, 0, 0, 0, 0};
            if (!cov3D_precomp) {
                q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
                gsr_cov3d_r<GsrBwdReal>(s, cam.scale_modifier, q, covr);
            } else {
#pragma unroll
                for (int k = 0; k < 6; ++k) covr[k] = cov[k];
            }
            gsr_project_backward_r<GsrBwdReal>(cam, mean, covr, op_in, g, dmean, dcovr, dop);
            // the returned means2D gradient is in NDC-scaled units (SURVEY A.5 units trap)
            dm2x = g.dpx * (0.5f * (float)cam.W);
            dm2y = g.dpy * (0.5f * (float)cam.H);
            if (!cov3D_precomp) gsr_cov3d_backward_r<GsrBwdReal>(s, cam.scale_modifier, q, dcovr, dscale, drot);
#pragma unroll
            for (int k = 0; k < 6; ++k) dcov[k] = (float)dcovr[k];
        }
        if (vis) {
            if (shs) {
                // The colour clamp mask is recomputed from the SH record (a few hundred FLOPs on data that is loaded anyway)
                // instead of being read from the forward's array: under screen sharding the forward evaluates colours
                // only for Gaussians in the rank's band, while this kernel runs for every visible Gaussian on every rank.
                float rgb_unused[3];
                uint32_t clampbits = 0;
                if (staged_sh) {   // in place in the LDS row: each 48-byte group is read before it is overwritten with its gradient
                    if (SPLIT) {
                        const GsrShRowSplit row{tile + SPLIT_DC + lane * 3, tile + lane * 45};
                        gsr_sh_to_rgb_row(cam.sh_degree, 16, row, mean, cam.campos, rgb_unused, clampbits);
                        gsr_sh_backward_row(cam.sh_degree, 16, row, mean, cam.campos, clampbits, drgb,
                                            GsrShRowSplitOut{tile + SPLIT_DC + lane * 3, tile + lane * 45}, dmean);
                    } else {
                        const GsrShRowAligned row{tile + lane * sh_row};
                        gsr_sh_to_rgb_row(cam.sh_degree, 16, row, mean, cam.campos, rgb_unused, clampbits);
                        gsr_sh_backward_row(cam.sh_degree, 16, row, mean, cam.campos, clampbits, drgb,
                                            GsrShRowAlignedOut{tile + lane * sh_row}, dmean);
                    }
                } else if (!SPLIT) {
                    gsr_sh_to_rgb(cam.sh_degree, M, shs + i * (int64_t)M * 3, mean, cam.campos, rgb_unused, clampbits);
                    gsr_sh_backward(cam.sh_degree, M, shs + i * (int64_t)M * 3, mean, cam.campos, clampbits, drgb,
                                    dL_dsh + i * (int64_t)M * 3, dmean);
                }
            }
        } else if (shs) {
            if (staged_sh) {
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    if (SPLIT) {      // the lane's own 45 + 3 floats, nothing else (rows are not padded)
                        if (k < 11) {
                            tile[lane * 45 + k * 4 + 0] = 0.f; tile[lane * 45 + k * 4 + 1] = 0.f;
                            tile[lane * 45 + k * 4 + 2] = 0.f; tile[lane * 45 + k * 4 + 3] = 0.f;
                        } else {
                            tile[lane * 45 + 44] = 0.f;
                            tile[SPLIT_DC + lane * 3 + 0] = 0.f; tile[SPLIT_DC + lane * 3 + 1] = 0.f; tile[SPLIT_DC + lane * 3 + 2] = 0.f;
                        }
                    } else {
                        *reinterpret_cast<float4*>(tile + lane * sh_row + k * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            } else if (in_range && !SPLIT) {
                float* o = dL_dsh + i * (int64_t)M * 3;
                for (int k = 0; k < M * 3; ++k) o[k] = 0.f;
            }
        }
        if (staged_sh) {
            if (SPLIT && ADAM) wave_adam_sh_split_dense(adam, i0, P, adam.sparse ? __ballot(vis) : ~0ull, lane, tile);
            else if (SPLIT) wave_store_sh_split_dense(dL_ddc, dL_dsh, i0, P, lane, tile);
            else wave_store_sh16(dL_dsh, i0, P, lane, tile);
        }
        if (!in_range) continue;
        dL_dmeans2D[i * 3 + 0] = dm2x; dL_dmeans2D[i * 3 + 1] = dm2y; dL_dmeans2D[i * 3 + 2] = 0.f;
        if (dL_dcolors) { dL_dcolors[i * 3 + 0] = drgb[0]; dL_dcolors[i * 3 + 1] = drgb[1]; dL_dcolors[i * 3 + 2] = drgb[2]; }
        dL_dopacity[i] = dop;
        dL_dmeans3D[i * 3 + 0] = dmean[0]; dL_dmeans3D[i * 3 + 1] = dmean[1]; dL_dmeans3D[i * 3 + 2] = dmean[2];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (dL_dcov3D) dL_dcov3D[i * 6 + k] = dcov[k];
        if (dL_dscales) {
            dL_dscales[i * 3 + 0] = dscale[0]; dL_dscales[i * 3 + 1] = dscale[1]; dL_dscales[i * 3 + 2] = dscale[2];
            reinterpret_cast<float4*>(dL_drotations)[i] = make_float4(drot[0], drot[1], drot[2], drot[3]);
        }
    }
}

__global__ void __launch_bounds__(256)
mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ vm, uint8_t* __restrict__ present) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = means3D[i * 3 + 0], y = means3D[i * 3 + 1], z = means3D[i * 3 + 2];
        const float pvz = vm[2] * x + vm[6] * y + vm[10] * z + vm[14];
        present[i] = pvz > GSR_NEAR_Z ? 1 : 0;
    }
}

int g_stream_grid_cap = 1024;      // option preprocess_grid_cap (tuning): workgroups of the grid-stride per-Gaussian kernels
                                   // (two interleaved A/B sessions: 1024 -> 0.068 / 0.072 ms, 2048 -> 0.070 / 0.076, 512 -> 0.076, 4096 -> 0.074)

inline int stream_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > g_stream_grid_cap) b = g_stream_grid_cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

int gsr_launch_preprocess(const GsrCamDev& cam, int P, const float* means3D, const float* shs,
                          const float* colors_precomp, const float* opacities, const float* scales,
                          const float* rotations, const float* cov3D_precomp, GsrGeom g, int32_t* radii,
                          const GsrFrameStatsDev& fs, hipStream_t st) {
    const int grid = stream_grid(P) < GSR_FRAME_MAX_GROUPS ? stream_grid(P) : GSR_FRAME_MAX_GROUPS;      // (gsr_frame.h: tickets)
#define GSR_PRE_FWD(SPLIT_)                                                                                                            \
    hipLaunchKernelGGL((preprocess_fwd_kernel<SPLIT_>), dim3(grid), dim3(256), 0, st, cam, P, means3D, shs, colors_precomp,              \
                       opacities, scales, rotations, cov3D_precomp, g.splats, g.rect, g.tiles,                                         \
                       /*clamped (recomputed by the backward)*/ nullptr, g.keys[0], g.vals[0], radii, fs)
    if (cam.sh_dc) GSR_PRE_FWD(true);
    else GSR_PRE_FWD(false);
#undef GSR_PRE_FWD
    return grid;
}

void gsr_launch_preprocess_backward(const GsrCamDev& cam, int P, const float* means3D, const float* shs,
                                    const float* colors_precomp, const float* opacities, const float* scales,
                                    const float* rotations, const float* cov3D_precomp, const int32_t* radii,
                                    GsrGeom g, const float* splat_grads, float* dL_dmeans2D, float* dL_dcolors,
                                    float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                                    float* dL_dscales, float* dL_drotations, hipStream_t st) {
#define GSR_PRE_BWD(SPLIT_)                                                                                                     \
    hipLaunchKernelGGL((preprocess_bwd_kernel<SPLIT_>), dim3(stream_grid(P)), dim3(256), 0, st, cam, P, means3D, shs,             \
                       colors_precomp, opacities, scales, rotations, cov3D_precomp, radii, g.clamped,                                 \
                       reinterpret_cast<const float4*>(splat_grads), dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D,     \
                       dL_dsh, dL_dscales, dL_drotations, GsrShAdamDev{})
    if (cam.sh_dc) GSR_PRE_BWD(true);
    else GSR_PRE_BWD(false);
#undef GSR_PRE_BWD
}

void gsr_launch_preprocess_backward_sh_adam(const GsrCamDev& cam, int P, const float* means3D, const float* opacities,
                                            const float* scales, const float* rotations, const float* cov3D_precomp,
                                            const int32_t* radii, GsrGeom g, const float* splat_grads, float* dL_dmeans2D,
                                            float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dscales,
                                            float* dL_drotations, const GsrShAdamDev& adam, hipStream_t st) {
    // split-SH form, M == 16: cam.sh_dc / `shs` are adam.dc / adam.rest (read for the colour clamp, then updated in place)
    hipLaunchKernelGGL((preprocess_bwd_kernel<true, true>), dim3(stream_grid(P)), dim3(256), 0, st, cam, P, means3D,
                       (const float*)adam.rest, (const float*)nullptr, opacities, scales, rotations, cov3D_precomp, radii, g.clamped,
                       reinterpret_cast<const float4*>(splat_grads), dL_dmeans2D, (float*)nullptr, dL_dopacity, dL_dmeans3D, dL_dcov3D,
                       (float*)nullptr, dL_dscales, dL_drotations, adam);
}

void gsr_launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t st) {
    hipLaunchKernelGGL(mark_visible_kernel, dim3(stream_grid(P)), dim3(256), 0, st, P, means3D, view, present);
}


void gsr_set_preprocess_grid_cap(int cap) { g_stream_grid_cap = cap < 64 ? 64 : cap; }
