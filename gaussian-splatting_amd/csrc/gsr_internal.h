// Internal declarations shared by the translation units of libgsr_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "gsr.h"
#include "gsr_math.h"
#include "gsr_frame.h"

// ---- per-call camera block handed to kernels by value (matrices are read from device memory) ----
struct GsrCamDev {
    int W, H, gx, gy;
    float focal_x, focal_y, limx, limy, scale_modifier;
    int sh_degree, M, antialiasing, tile_y0, tile_y1;
    int snug;                // gsr_math.h GsrCam::snug
    const float* view;
    const float* proj;
    const float* campos;
    const float* bg;
    const float* sh_dc;      // split SH form (GsrRasterSettings.sh_dc), else NULL
    float* dL_dsh_dc;
};

// ---- scratch layouts (GeometryState / BinningState / ImageState), 128-byte aligned carve ----
static inline size_t gsr_align128(size_t x) { return (x + 127) & ~(size_t)127; }

#define GSR_SORT_ITEMS 4096      // items per workgroup in a radix pass (256 threads x 16)
#define GSR_SORT_ITEMS_SMALL 1024
#define GSR_SCAN_ITEMS 1024      // items per workgroup in the tiles_touched scan
#ifndef GSR_TS_ITEMS
#define GSR_TS_ITEMS 4096        // instances per workgroup of the fused emission / tile sort (tilesort.hip).  (Round 3 tuned it with -DGSR_TS_ITEMS=2048 builds; the
                                 // block tables of depthsort.hip / gsr_api.cpp have since been sized for 4096 and a 2048 build aborts: not a supported switch any more.)
#endif
// capacity of the per-block "first Gaussian" table the scan kernel fills: enough for 64 tiles per Gaussian on average;
// frames beyond that get the table from a fallback kernel once R is known (gsr_launch_fill_block_first)
static inline size_t gsr_block_first_cap(int P) { return (size_t)(P > 0 ? P : 1) / 64 + 66; }

// ---- bucket depth sort (depthsort.hip): one bucket pass over the keys + an LDS sort per segment of buckets ----
#define GSR_DS_ITEMS 4096        // keys per workgroup of the bucket histogram / scatter
#define GSR_DS_BITS 11
#define GSR_DS_BUCKETS 2048      // bucket 2047: Gaussians without a tile (they sort last)
#define GSR_DS_SEG 2048          // a segment = the buckets that start inside one window of this many elements
#define GSR_DS_CAP 4096          // largest segment sorted in LDS; beyond: the same passes through global memory (slow, reported)
#define GSR_DS_MAX_P (3 << 20)   // above: the LSD radix sort (the [workgroup][bucket] tables grow with P)
// Histogram-equalised bucket mapping (round 6, depthsort.hip ds_hist).  The 27-bit key space is cut into GSR_EQ_BINS coarse bins of 2^GSR_EQ_SHIFT keys
// (64 per octave of depth); every ds_hist workgroup reads the same GSR_EQ_SAMPLE keys -- GSR_EQ_SAMPLE / 16 windows of 16 consecutive keys spread
// evenly over the array -- and builds the same tables from them: (first bucket, buckets) per coarse bin, and for ONE "hot" coarse bin (an eighth of
// the sample or more) the same per sub-bin of 2^GSR_EQ_SHIFT2 keys.
#define GSR_EQ_SHIFT 17
#define GSR_EQ_BINS 1024         // 2^(GSR_DEPTH_KEY_BITS - GSR_EQ_SHIFT)
#define GSR_EQ_SHIFT2 7          // GSR_EQ_SHIFT - log2(GSR_EQ_BINS)
#define GSR_EQ_SAMPLE 4096
#define GSR_EQ_NO_HOT 0xFFFFFFFFu
// table buffer (ds_hist -> ds_scatter): GSR_EQ_BINS words level 1, GSR_EQ_BINS words level 2, then the hot bin (GSR_EQ_NO_HOT: no second level)
#define GSR_EQ_TAB_WORDS (2 * GSR_EQ_BINS + 16)
#define GSR_DS_PLAN_WORDS 12           // words per window of the segment plan (depthsort.hip ds_scatter)
#define GSR_DS_NO_HELP 0xFFFFFFFFu     // plan word 8: the window's workgroup has no helper job
#define GSR_DS_TIE_HELPED 0x80000000u  // plan word 3, bit 31: the segment's last bucket is more than a capacity of one key value; helpers write its chunks behind the first
struct GsrDepthSortBufs {
    uint2* pairs[2];             // [P] (key, id) in bucket order / scratch of an oversized segment
    uint32_t* cnt_tab;           // [workgroups][2048] keys per bucket, then their exclusive prefix over the workgroups
    uint32_t* tile_tab;          // [workgroups][2048] tile instances per bucket
    uint32_t* cnt_total;         // [2048]
    uint32_t* tile_total;        // [2048]
    uint32_t* plan;              // [segments][GSR_DS_PLAN_WORDS]
    uint16_t* bucket_of;         // [P] the depth bucket of every key (written by ds_hist, read by ds_scatter: the table lookup is done once)
    uint32_t* eq_tab;            // [GSR_EQ_TAB_WORDS] first bucket | buckets << 16 per coarse bin, the same per sub-bin of the hot coarse bin, the hot bin (ds_hist's workgroup 0)
};
size_t gsr_depth_bucket_blocks(int P);
size_t gsr_depth_bucket_segments(int P);

struct GsrGeom {                 // P-sized
    float4* splats;              // [4P]  (x,y,conA,conB) (conC,opacity,r,g) (b,depth,tau,1/depth) (rect.x,rect.y,goffset,tiles as bits)
    uint2* rect;                 // [P]   x = minx | maxx<<16 ; y = miny | maxy<<16 (band-clamped)
    uint32_t* tiles;             // [P]   tiles_touched
    uint32_t* clamped;           // [P]   colour clamp bits
    uint32_t* keys[2];           // [P]x2 depth-sort keys (ping-pong)
    uint32_t* vals[2];           // [P]x2 Gaussian ids   (ping-pong); vals[0] = depth order (4 passes: even)
    uint2* rect_sorted;          // [P]   tile rectangles in depth order (written by the scan)
    uint32_t* offsets;           // [P]   inclusive scan of tiles_touched in depth order
    uint64_t* block_sums;        // [ceil(P/GSR_SCAN_ITEMS)]
    uint2* block_first;          // [gsr_block_first_cap(P)] per 4096-instance block: (depth-order index of its first Gaussian, instances before it)
    uint32_t* sort_hist;         // [512 * nblocks_small(P)] radix block histograms (LSD depth sort)
    uint32_t* digit_total;       // [512]
    GsrDepthSortBufs ds;         // bucket depth sort (depthsort.hip); carved for P <= GSR_DS_MAX_P
    uint32_t* num_rendered;      // frame words (gsr_frame.h): [0..1] R as 64 bits, [2] smallest, [3] largest depth key of a listed Gaussian
    uint2* wg_range;             // [GSR_FRAME_MAX_GROUPS] per-workgroup depth-key ranges of the key-producing kernel
    size_t bytes;
};
GsrGeom gsr_carve_geom(char* base, int P);

// ---- depth-sort keys (round 3): 27 bits instead of 32 -> 3 radix passes of 9 bits instead of 4 of 8 ----
// Every listed Gaussian passed the near cull view.z > 0.2 (Appendix A.2 step 1), and positive fp32 bit patterns order like the
// floats, so key = bits(depth) - bits(0.2f) >= 1 orders the listed Gaussians exactly like bits(depth).  27 bits cover
// 16 octaves: depths below 0.2 * 2^16 = 13 107.  A deeper Gaussian sets *key_overflow (a mapped host word); the host then
// re-keys with the full 32 bits and repeats the depth sort with 4 passes (gsr_api.cpp) -- bit-identical order either way.
// Gaussians without a tile in the band sort last (key = 2^27 - 1; stable sort: among themselves in index order).
#define GSR_DEPTH_KEY_BITS 27
#define GSR_DEPTH_KEY_BASE 0x3E4CCCCDu
#define GSR_DEPTH_KEY_CULLED ((1u << GSR_DEPTH_KEY_BITS) - 1u)
static_assert((1 << (GSR_DEPTH_KEY_BITS - GSR_EQ_SHIFT)) == GSR_EQ_BINS && (1 << (GSR_EQ_SHIFT - GSR_EQ_SHIFT2)) == GSR_EQ_BINS, "equalised bucket tables");
#ifdef __HIPCC__
// `overflow` is a per-thread flag the caller reports ONCE, after its loop, through gsr_frame_stats_commit: a store through an
// (unrestricted) host-word pointer inside the streaming loop made hipcc serialise the loop's batched loads (ISA audit: the
// split-SH loader of the preprocess became a 14-long load -> wait chain).
__device__ __forceinline__ uint32_t gsr_depth_key(float depth, bool listed, bool& overflow) {
    if (!listed) return GSR_DEPTH_KEY_CULLED;
    uint32_t k = __float_as_uint(depth) - GSR_DEPTH_KEY_BASE;
    if (k >= GSR_DEPTH_KEY_CULLED) {
        overflow = true;
        k = GSR_DEPTH_KEY_CULLED - 1u;
    }
    return k;
}
// the two derived fields of a splat record, written by the preprocess and recomputed bit-identically from (opacity, depth) by the
// receiver of a packed record (route.hip): tau = gsr_tau(opacity) (gsr_math.h) and 1 / depth, an explicit single rounding
__device__ __forceinline__ float gsr_inv_depth(float depth) { return __fdiv_rn(1.0f, depth); }
#endif

struct GsrBinning {              // R-sized
    // > 65536 tiles (LSD sort, sort.hip): keys[2] tile ids / vals[2] Gaussian ids, ping-pong; list = vals[passes & 1]
    // <= 65536 tiles (fused emission + two-level sort, tilesort.hip): words = keys[0] (8R bytes, contiguous with
    //   keys[1]) holds the packed level-1 output, vals[0] the final point_list
    uint32_t* keys[2];
    uint32_t* vals[2];
    uint32_t* sort_hist;         // LSD: [256 * nblocks_small(R)]; fused: level-1 table [nb1 * nblk]
    uint32_t* hist2;             // fused: level-2 table [(nblk + 256) * 256]
    uint32_t* digit_total;       // [256]
    uint32_t* bucket_base;       // [257]
    uint32_t* blk2_start;        // [257]
    uint32_t* tile_base;         // [65536]
    uint2* block_first;          // [ceil(R / GSR_TS_ITEMS) + 2]  (fallback table, see gsr_block_first_cap)
    uint32_t* meta;
    size_t bytes;
};
GsrBinning gsr_carve_binning(char* base, int64_t R);

struct GsrImage {
    float* final_T;              // [H*W]
    uint32_t* n_contrib;         // [H*W]
    uint2* ranges;               // [n_tiles]
    uint32_t* block_steps;       // [n_tiles*4] list entries the forward blended per 8x8 block (tracking build): the backward's work estimate
    uint32_t* tile_order;        // [n_tiles]   band-local tile indices, heaviest first (written by the backward's plan kernel)
    size_t bytes;
};
GsrImage gsr_carve_image(char* base, int W, int H);

// ---- kernel launchers (one per translation unit) ----
// preprocess.hip  (compiled with -ffp-contract=off)
// (the key-producing launchers return their grid size: the number of entries of fs.wg_range they fill)
int gsr_launch_preprocess(const GsrCamDev& cam, int P, const float* means3D, const float* shs,
                          const float* colors_precomp, const float* opacities, const float* scales,
                          const float* rotations, const float* cov3D_precomp, GsrGeom g, int32_t* radii,
                          const GsrFrameStatsDev& fs, hipStream_t st);
void gsr_launch_preprocess_backward(const GsrCamDev& cam, int P, const float* means3D, const float* shs,
                                    const float* colors_precomp, const float* opacities, const float* scales,
                                    const float* rotations, const float* cov3D_precomp, const int32_t* radii,
                                    GsrGeom g, const float* splat_grads /*[P,12]*/, float* dL_dmeans2D,
                                    float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D,
                                    float* dL_dsh, float* dL_dscales, float* dL_drotations, hipStream_t st);
// the same kernel, but the SH gradient of the split form is not written: the Adam update of the two SH tensors is applied in
// place from the gradient tile (gsr_backward_preprocess_sh_adam); dense: every row, sparse: rows with radii > 0
struct GsrShAdamDev {
    float* dc;  float* dc_m;  float* dc_v;            // [P,1,3]
    float* rest; float* rest_m; float* rest_v;        // [P,15,3]
    // dense (torch.optim.Adam): om_b1, b2, om_b2, step_size, inv_bc2_sqrt, eps;  sparse: lr, b1, om_b1, b2, om_b2, eps
    float dc_a[6], rest_a[6];
    int sparse;
};
void gsr_launch_preprocess_backward_sh_adam(const GsrCamDev& cam, int P, const float* means3D, const float* opacities,
                                            const float* scales, const float* rotations, const float* cov3D_precomp,
                                            const int32_t* radii, GsrGeom g, const float* splat_grads, float* dL_dmeans2D,
                                            float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dscales,
                                            float* dL_drotations, const GsrShAdamDev& adam, hipStream_t st);
void gsr_set_preprocess_grid_cap(int cap);      // tuning knob (option preprocess_grid_cap)
void gsr_launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t st);

// sort.hip: LSD radix sort of (u32 key, u32 value) pairs on bits [0, nbits); returns the index (0/1) of the
// ping-pong buffer that holds the result.  n is known on the host.  items = keys per workgroup: 1024, 2048 or 4096
// (hist must hold 2^digit_bits * ceil(n / items) counters).
// rect / rect_sorted (optional): the last pass also writes rect_sorted[pos] = rect[value] (depth sort: the tile rectangles
// in depth order, which the scan and the emission stream afterwards)
int gsr_radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], int64_t n, int nbits, int max_digit_bits, uint32_t* hist,
                         uint32_t* digit_total, int items, hipStream_t st, const uint2* rect = nullptr, uint2* rect_sorted = nullptr);
int gsr_radix_sort_pairs_k16(uint16_t* keys[2], uint32_t* vals[2], int64_t n, int nbits, int max_digit_bits, uint32_t* hist,
                             uint32_t* digit_total, int items, hipStream_t st);
// depthsort.hip: depth order + rectangles in depth order + inclusive scan of the tile counts + the emission's block table
void gsr_launch_depth_bucket_sort(int P, const uint32_t* keys, const uint32_t* tiles, const uint2* rect, uint32_t* frame,
                                  const uint2* wg_range, int n_range, const GsrDepthSortBufs& b, uint32_t* order, uint2* rect_sorted,
                                  uint32_t* offsets, uint2* block_first, uint32_t block_first_cap, uint32_t* slow_word, hipStream_t st);
// pass plan shared by the sorter and by code that must know which ping-pong buffer holds the result
int gsr_sort_plan(int nbits, int max_digit_bits, int* pass_bits /*[8]*/);
#define GSR_DEPTH_DIGIT_BITS 9      // 27-bit depth keys: 3 passes of 9 bits (round 2: 4 x 8 on 32 bits; 3 x 11 measured slower: 113 vs 91 us)
#define GSR_SORT_MAX_DIGITS 512     // table rows (digit values) the geometry buffer provides for the depth sort
#define GSR_TILE_DIGIT_BITS 8       // tile ids: ceil(bits/8) passes of equal width
static inline int64_t gsr_sort_blocks(int64_t n, bool small_blocks) {
    const int64_t items = small_blocks ? GSR_SORT_ITEMS_SMALL : GSR_SORT_ITEMS;
    return (n + items - 1) / items;
}

// binning.hip: scan of tiles_touched in depth order, instance emission, tile ranges
// host_word (mapped pinned, may be NULL): [0] = R low word, [2] = R high word, [1] = seq (stored last)
void gsr_launch_scan_tiles(int P, const uint32_t* order, const uint2* rect, uint2* rect_sorted /*[P]*/, uint32_t* offsets,
                           uint64_t* block_sums, uint2* block_first, uint32_t block_first_cap, uint32_t* num_rendered,
                           uint32_t* host_word, uint32_t seq, bool rect_already_sorted, hipStream_t st);
// legacy emission (frames with more than 65536 tiles): 32-bit tile ids
void gsr_launch_emit(int P, int gx, const uint32_t* order, const uint32_t* offsets, const uint2* rect_sorted,
                     void* inst_keys, bool key16, uint32_t* inst_vals, float4* splats /*NULL: skip the goffset write*/,
                     hipStream_t st);

// tilesort.hip: fused emission + two-level stable tile sort + ranges (frames with <= 65536 tiles)
struct GsrTileSortPlan { bool fused; int lb, hb; bool word64; };
void gsr_tile_sort_plan(int n_tiles, int P, GsrTileSortPlan* plan);
void gsr_launch_fill_block_first(int P, const uint32_t* offsets, uint2* block_first, uint32_t cap, hipStream_t st);
void gsr_launch_tile_sort_level1(const GsrTileSortPlan& plan, int64_t R, int gx, const uint2* block_first,
                                 const uint32_t* offsets, const uint2* rect_sorted, const uint32_t* order, void* words,
                                 uint32_t* hist1, uint32_t* digit_total, uint32_t* bucket_base, uint32_t* blk2_start,
                                 float4* splats /*NULL: inference*/, hipStream_t st);
void gsr_launch_tile_sort_level2(const GsrTileSortPlan& plan, int64_t R, int n_tiles, const void* words, uint32_t* point_list,
                                 const uint32_t* bucket_base, const uint32_t* blk2_start, uint32_t* hist2, uint32_t* tile_base,
                                 uint2* ranges, hipStream_t st, int h0 = 0, int h1 = 0 /*level-1 buckets [h0, h1) of a band; (0, 0) = all*/);
void gsr_set_level2_scan_mode(int v);      // tilesort.hip (option level2_scan_mode)
// sort.hip: in-place exclusive scan of every digit row of a [ndigits][nblocks] block-histogram table + row totals
void gsr_launch_rs_scan(uint32_t* block_hist, int nblocks, int ndigits, uint32_t* digit_total, hipStream_t st);
void gsr_launch_ranges(int64_t R, int n_tiles, const void* sorted_keys, bool key16, uint2* ranges, bool already_zeroed,
                       hipStream_t st);

// render_fwd.hip / render_bwd.hip
// ---- measurement block of the blend kernels (device memory, allocated by gsr_profile_enable only) -------------------------------------
// [0..5] work counters (include/gsr.h), [6] mode: 0 = counters (atomics at the end of every wave), 1 = per-wave trace, [8 + 4 w ..] trace
// entry of wave w: start, end (s_memrealtime), placement | kernel << 40, steps.
#define GSR_TRACE_MODE_WORD 6
#define GSR_TRACE_BASE 8
#define GSR_MEASURE_WORDS (GSR_TRACE_BASE + 4 * (size_t)GSR_TRACE_WAVES)
#ifdef __HIPCC__
__device__ __forceinline__ bool gsr_trace_mode(const unsigned long long* counters) {
    return __builtin_nontemporal_load(counters + GSR_TRACE_MODE_WORD) != 0ull;
}
// steps: bits 0-15 blend steps; in trace mode the kernels also pack the time (10 ns ticks, saturating at 65535) they spent walking survivors
// (bits 16-31), getting batches ready -- waiting for the gathered records, box tests, staging -- (bits 32-47) and storing (bits 48-63)
__device__ __forceinline__ unsigned long long gsr_trace_pack(uint32_t steps, unsigned long long walk, unsigned long long prep, unsigned long long store) {
    auto sat = [](unsigned long long v) { return v > 65535ull ? 65535ull : v; };
    return (unsigned long long)(steps > 65535u ? 65535u : steps) | (sat(walk) << 16) | (sat(prep) << 32) | (sat(store) << 48);
}
__device__ __forceinline__ void gsr_trace_wave(unsigned long long* counters, unsigned long long t0, uint32_t wave, uint32_t kernel, unsigned long long steps) {
    if (wave >= (uint32_t)GSR_TRACE_WAVES) return;
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);         // HW_REG_HW_ID: wave [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);        // HW_REG_XCC_ID [3:0]
    unsigned long long* e = counters + GSR_TRACE_BASE + (size_t)wave * 4;
    e[0] = t0;
    e[1] = wall_clock64();
    e[2] = (unsigned long long)hw | ((unsigned long long)(xcc & 15u) << 32) | ((unsigned long long)kernel << 40);
    e[3] = steps;
}
#endif

void gsr_launch_render_forward(const GsrCamDev& cam, const uint2* ranges, const uint32_t* point_list,
                               const float4* splats, float* final_T, uint32_t* n_contrib, uint32_t* block_steps /*NULL unless tracking*/,
                               float* out_color, float* out_invdepth, int variant,
                               unsigned long long* counters /*NULL or [4] work counters*/, hipStream_t st,
                               int tile_off = 0, int tile_cnt = -1 /*the launch blends tiles [tile_off, tile_off + tile_cnt) of the band (multiple of 8; -1 = all)*/);
void gsr_set_render_fwd_lds_pad(int bytes);      // render_fwd.hip (tuning option render_fwd_lds_pad)
// variant: 0 = default (independent quadrant waves); 1 (global atomics) and 4 (round 1's workgroup-per-tile kernel) exist
// only in builds with -DGSR_AB_VARIANTS
int gsr_render_backward_variant_available(int variant);
int gsr_render_forward_variant_available(int variant);
void gsr_launch_render_backward(const GsrCamDev& cam, const uint2* ranges, const uint32_t* point_list,
                                const float4* splats, const float* final_T, const uint32_t* n_contrib,
                                const uint32_t* block_steps, uint32_t* tile_order /*NULL: tiles in index order*/,
                                const float* dL_dpix, const float* dL_dinvdepth, float* splat_grads /*[P,12] variant 1*/,
                                float* inst_grads /*[4][R,12]*/, uint32_t* inst_flag /*[R]*/, int64_t R, int variant,
                                int order_mode /*plan kernel: 1 tiles by the sum of their blocks, 2 tiles by their heaviest half, 3 half tiles (waves)*/,
                                unsigned long long* counters, hipStream_t st);
size_t gsr_reduce_units(int64_t R);      // units of 1024 instance records the reduce works in
void gsr_launch_reduce_instances(int P, int64_t R, const uint32_t* order, const uint32_t* offsets, const float4* splats,
                                 const float* inst_grads, const uint32_t* inst_flag, float* splat_grads, uint2* unit_first,
                                 float* unit_piece, hipStream_t st);

// backward scratch (caller-owned, gsr_backward_scratch_bytes): per-Gaussian record, per-instance records, maps
// record slots per instance: one per 16x8 half tile (the measurement build's per-quadrant kernel needs four)
#ifdef GSR_AB_VARIANTS
#define GSR_BWD_SLOTS 4
#else
#define GSR_BWD_SLOTS 2
#endif
struct GsrBwdScratch {
    float* splat_grads;     // [P,12]
    float* inst_grads;      // [GSR_BWD_SLOTS][R,12]  one record slot per (half tile, instance), slot-major
    uint32_t* inst_flag;    // [R]  byte q != 0: slot q of the instance has a record
    uint2* unit_first;      // [units]  reduce: (owner of the unit's first record, that owner's first record)
    float* unit_piece;      // [units][2][12]  reduce: head / tail partial rows of every unit
    size_t bytes;
};
GsrBwdScratch gsr_carve_bwd(char* base, int P, int64_t R);

// adam.hip
void gsr_launch_adam(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                     int step, hipStream_t st);

// ssim.hip
void gsr_launch_ssim_forward(int planes, int H, int W, const float* img1, const float* img2, float* ssim_map, float* dm_dmu1,
                             float* dm_dex2, float* dm_dexy, hipStream_t st);
void gsr_launch_ssim_backward(int planes, int H, int W, const float* img1, const float* img2, const float* dL_dmap,
                              const float* dm_dmu1, const float* dm_dex2, const float* dm_dexy, float* dL_dimg1, hipStream_t st);

// adam.hip: sparse step (rows of M elements, skipped entirely when visible[row] == 0)
void gsr_launch_adam_multi(const GsrAdamTensor* tensors, int count /*<= GSR_ADAM_MAX_TENSORS*/, hipStream_t st);
void gsr_launch_sparse_adam_multi(const GsrSparseAdamTensor* tensors, int count, const uint8_t* visible, int64_t N, double beta1, double beta2,
                                  hipStream_t st);
void gsr_launch_sparse_adam(float* p, const float* g, float* m, float* v, const uint8_t* visible, int64_t N, int64_t M,
                            double lr, double beta1, double beta2, double eps, hipStream_t st);
// density.hip: per-iteration density-control statistics
void gsr_launch_density_stats(int P, const float* grad, const uint8_t* visible, const int32_t* radii, float* accum, float* denom,
                              float* max_radii, hipStream_t st);
// knn.hip
size_t gsr_knn_scratch_bytes_impl(int N);
void gsr_launch_knn(int N, const float* points, float* out, void* scratch, hipStream_t st);
// ssim.hip: mean-SSIM form (no map round trip)
void gsr_set_ssim_variant(int v);      // 0 = marching waves (default), 1 = LDS tiles (A/B)
void gsr_set_ssim_target_waves(int v); // tuning: waves per launch the marching form aims for
int64_t gsr_ssim_partial_count_impl(int planes, int H, int W);
void gsr_launch_ssim_mean_forward(int planes, int H, int W, const float* img1, const float* img2, float* partials,
                                  float* mean_out, float* dm_dmu1, float* dm_dex2, float* dm_dexy, hipStream_t st);
void gsr_launch_ssim_mean_backward(int planes, int H, int W, const float* img1, const float* img2, const float* dL_dmean,
                                   const float* dm_dmu1, const float* dm_dex2, const float* dm_dexy, float* dL_dimg1,
                                   hipStream_t st);
// ssim.hip: the fused training loss (1 - lambda) L1 + lambda (1 - SSIM) in the SSIM kernels' single pass
void gsr_launch_train_loss_forward(int planes, int H, int W, const float* img1, const float* img2, float lambda, float* partials,
                                   float* loss_out, float* dm_dmu1, float* dm_dex2, float* dm_dexy, hipStream_t st);
void gsr_launch_train_loss_backward(int planes, int H, int W, const float* img1, const float* img2, const float* dL_dloss, float lambda,
                                    const float* dm_dmu1, const float* dm_dex2, const float* dm_dexy, float* dL_dimg1, hipStream_t st);
// binning.hip: gathered splat records -> geometry state of this rank's band (two-axis sharding)
int gsr_launch_splat_ingest(int P, const float* records, int y0, int y1, float4* splats, uint2* rect, uint32_t* tiles,
                            uint32_t* keys, uint32_t* vals, const GsrFrameStatsDev& fs, hipStream_t st);
// binning.hip: full 32-bit depth keys from the splat records (fallback of the 27-bit depth sort)
void gsr_launch_rekey_full(int P, const float4* splats, const uint32_t* tiles, uint32_t* keys, uint32_t* vals, hipStream_t st);

// route.hip: Gaussian-sharded rendering -- destination-targeted exchange of packed splat records
#define GSR_MAX_BANDS 64
size_t gsr_route_scratch_bytes_impl(int P, int n_bands);
void gsr_launch_route_count(int P, const float* records, int n_bands, const int32_t* bounds, uint32_t* block_counts,
                            uint32_t* band_counts, hipStream_t st);
void gsr_launch_route_pack(int P, const float* records, int n_bands, const int32_t* bounds, const int64_t* band_offsets,
                           const uint32_t* block_offsets, float* packed, int32_t* send_ids, uint32_t cap /*0xFFFFFFFF: no limit*/,
                           const uint32_t* band_counts /*fixed-capacity form: header rows; else NULL*/, hipStream_t st);
int gsr_launch_ingest_packed(int P, const float* packed, int y0, int y1, float4* splats, uint2* rect, uint32_t* tiles,
                             uint32_t* keys, uint32_t* vals, const GsrFrameStatsDev& fs, int seg_rows /*0 or capacity + 1*/, hipStream_t st);
void gsr_launch_route_add_rows(int64_t n, const int32_t* ids, const float* rows, float* out, hipStream_t st);
