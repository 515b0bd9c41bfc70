/*
 * gsr.h -- C ABI of libgsr_hip.so, the MI355X (gfx950) differentiable Gaussian-splat rasterizer.
 *
 * This is the drop-in boundary for ONE path of graphdeco-inria/gaussian-splatting: the native
 * operator behind `GaussianRasterizer` / `GaussianRasterizationSettings`
 * (reference call sites: gaussian_renderer/__init__.py:14, :36-52, :91-110; backward via
 * loss.backward() at train.py:142).  In the reference that operator lives in the un-vendored
 * submodule `submodules/diff-gaussian-rasterization` (.gitmodules:4-7) whose torch extension `_C`
 * exports `rasterize_gaussians`, `rasterize_gaussians_backward` and `mark_visible`; each entry point
 * below names the `_C` function it replaces.  A maintainer binds these with ctypes (see
 * INTEGRATION.md; gaussian-splatting_amd/diff_gaussian_rasterization/_lib.py is that binding).
 *
 * Conventions
 *   - plain C: pointers + sizes only, no torch / C++ types; every data pointer is a DEVICE pointer
 *     (tensor.data_ptr()) to contiguous fp32 / int32 data unless stated otherwise.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream).
 *     gsr_rasterize_forward learns num_rendered (it sizes the binning buffer) mid-pipeline like the reference, but
 *     without hipStreamSynchronize and -- since ABI 4 -- off the critical path: the count does not depend on the depth
 *     order, so the per-Gaussian preprocess kernel sums it and its last workgroup stores the 64-bit count + a sequence
 *     number into a mapped, pinned host word; the calling thread queues the depth sort, then polls the word (200 us of
 *     spinning, then sched_yield() between polls).
 *   - return value: GSR_OK (0) or a negative GsrStatus; gsr_last_error() gives the message for the
 *     calling thread.  No exception crosses the ABI.
 *   - no per-frame device allocation inside: scratch memory is caller-owned and obtained through the three
 *     resize callbacks (geometry / binning / image state), exactly like the reference's
 *     resizeFunctional lambdas; forward returns with the three buffers populated and the caller keeps
 *     them alive for gsr_rasterize_backward.  The library itself owns only a few 64-byte CONTROL BLOCKS per device,
 *     created the first time a device (or one more concurrent caller on it) is seen and kept for the life of the
 *     process: the mapped host word of the read-back above, a 64-byte device counter beside it, one mapped
 *     "oversized depth bucket" flag, and -- only when option fwd_bands > 1 -- one auxiliary HIP stream with five
 *     events.  Their creation (hipHostMalloc / hipMalloc / hipStreamCreate) synchronises the device once; do the
 *     first call outside a stream capture.
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_ABI_VERSION 4

typedef enum GsrStatus {
    GSR_OK = 0,
    GSR_ERR_INVALID_ARG = -1,   /* bad size / NULL pointer / inconsistent optional inputs */
    GSR_ERR_HIP = -2,           /* a HIP runtime call or kernel launch failed */
    GSR_ERR_ALLOC = -3,         /* a resize callback returned NULL */
    GSR_ERR_UNSUPPORTED = -4    /* e.g. more than 2^24 tiles, SH degree > 3 */
} GsrStatus;

/* Mirrors GaussianRasterizationSettings (gaussian_renderer/__init__.py:36-50), same field meaning.
 * bg / viewmatrix / projmatrix / campos are device pointers (they are CUDA tensors in the
 * reference).  viewmatrix / projmatrix are the row-major tensors the reference passes, i.e. the
 * TRANSPOSED math matrices (scene/cameras.py:86-88): flat index i + 4*j = math element (row i, col j). */
typedef struct GsrRasterSettings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    const float* bg;          /* [3] */
    float scale_modifier;
    const float* viewmatrix;  /* [16] */
    const float* projmatrix;  /* [16] */
    int32_t sh_degree;        /* active degree 0..3 */
    const float* campos;      /* [3] */
    int32_t prefiltered;      /* always 0 in the reference (gaussian_renderer/__init__.py:47) */
    int32_t debug;            /* 1: synchronise + check after every kernel */
    int32_t antialiasing;
    /* Extension for screen-tile sharding across GPUs (not in the reference):
     * only tile rows [tile_y0, tile_y1) are binned and rendered; pixels outside are left untouched.
     * tile_y1 <= 0 means "all rows". radii are unaffected by the band. */
    int32_t tile_y0;
    int32_t tile_y1;
    /* Extension: 1 = the caller will not run the backward for this forward (inference / torch.no_grad()): state that only
     * the backward reads (final_T, n_contrib, first-emission indices) is not written.  0 = reference behaviour. */
    int32_t no_backward;
    /* Extension -- the "separate_sh" call form the reference uses when its accelerated rasterizer is installed
     * (gaussian_renderer/__init__.py:82-100: rasterizer(dc = features_dc, shs = features_rest, ...)): when sh_dc is
     * non-NULL, SH coefficient 0 is read from sh_dc[P,1,3] and `shs` holds coefficients 1..M-1 as [P,M-1,3] (M still
     * counts ALL coefficients; supported for degree-3 storage, M == 16, with 16-byte aligned pointers -- else
     * GSR_ERR_UNSUPPORTED and the caller concatenates).  In the backward dL_dsh then receives [P,M-1,3] and dL_dsh_dc[P,1,3] the
     * gradient of the DC term.  This avoids the torch.cat of the two parameter tensors (scene/gaussian_model.py:121-125)
     * and the split of its gradient in every iteration.  Both NULL = fused [P,M,3] form. */
    const float* sh_dc;
    float* dL_dsh_dc;
} GsrRasterSettings;

/* Resize callback: make the buffer at least `bytes` long and return its (device) base address,
 * 128-byte aligned or better.  Mirrors the reference's std::function<char*(size_t)> lambdas that
 * call tensor.resize_(). */
typedef void* (*GsrResizeFn)(void* user, size_t bytes);

int gsr_abi_version(void);
const char* gsr_last_error(void);

/* Scratch sizes (bytes) -- mirrors the reference's required<GeometryState/BinningState/ImageState>(). */
size_t gsr_geometry_bytes(int P);
size_t gsr_binning_bytes(int64_t R, int n_tiles);
size_t gsr_image_bytes(int width, int height);
size_t gsr_backward_scratch_bytes(int P, int64_t R);

/*
 * Replaces _C.rasterize_gaussians.
 *   P            number of Gaussians; M = SH coefficients per channel in `shs` ((max_degree+1)^2).
 *   means3D[P,3], opacities[P], shs[P,M,3] XOR colors_precomp[P,3],
 *   (scales[P,3] AND rotations[P,4]) XOR cov3D_precomp[P,6]; absent inputs are NULL.
 *   out_color[3,H,W], out_invdepth[1,H,W] (may be NULL), radii[P] int32.
 *   *num_rendered receives R, the number of (Gaussian, tile) instances that were binned: the tiles of the snug rectangle of
 *   every Gaussian's alpha >= 1/255 ellipse (option snug_tiles), a subset of the reference's tile square -- the reference would
 *   report ~1.4x as many for the same frame; the rendered outputs are the same bits.
 * P == 0: out_color / out_invdepth are zero-filled, *num_rendered = 0, no callback is invoked.
 */
int gsr_rasterize_forward(const GsrRasterSettings* settings, int P, int M,
                          const float* means3D, const float* shs, const float* colors_precomp,
                          const float* opacities, const float* scales, const float* rotations,
                          const float* cov3D_precomp,
                          GsrResizeFn geom_resize, void* geom_user,
                          GsrResizeFn binning_resize, void* binning_user,
                          GsrResizeFn image_resize, void* image_user,
                          float* out_color, float* out_invdepth, int32_t* radii,
                          int32_t* num_rendered, void* stream);

/*
 * Replaces _C.rasterize_gaussians_backward.
 *   geom/binning/image buffers: the three buffers forward filled (same P, R, settings, inputs).
 *   dL_dout_color[3,H,W]; dL_dout_invdepth[1,H,W] or NULL (treated as zero).
 * Outputs (every element is overwritten -- zeros for Gaussians with radii == 0 -- so the caller may pass
 * uninitialised memory; the reference's glue allocates them with torch::zeros):
 *   dL_dmeans2D[P,3]  (x,y in NDC-scaled units = pixel gradient * (0.5 W, 0.5 H); z = 0)
 *   dL_dcolors[P,3] (may be NULL unless colors_precomp was given: it is an intermediate otherwise), dL_dopacity[P],
 *   dL_dmeans3D[P,3], dL_dcov3D[P,6] (may be NULL unless cov3D_precomp was given), dL_dsh[P,M,3] (NULL if no shs),
 *   dL_dscales[P,3], dL_drotations[P,4] (NULL if cov3D_precomp was given).
 *   bwd_scratch: caller-owned scratch of gsr_backward_scratch_bytes(P, num_rendered) bytes (per-instance
 *   gradient records, the emission-order inverse map and the per-Gaussian 2-D gradient record).
 *   If splat_grads_out is non-NULL it receives the device address (inside bwd_scratch) of the per-Gaussian
 *   record [P,12] = [dpx,dpy,dA,dB,dC,dopacity,dr,dg,db,dinvdepth,0,0] -- the 48-byte payload that is
 *   reduce-scattered between GPUs when the screen is sharded.
 */
int gsr_rasterize_backward(const GsrRasterSettings* settings, int P, int M, int32_t num_rendered,
                           const float* means3D, const float* shs, const float* colors_precomp,
                           const float* opacities, const float* scales, const float* rotations,
                           const float* cov3D_precomp, const int32_t* radii,
                           const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                           const float* dL_dout_color, const float* dL_dout_invdepth,
                           float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity,
                           float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                           float* dL_dscales, float* dL_drotations, void* bwd_scratch,
                           float** splat_grads_out, void* stream);

/*
 * The two halves of gsr_rasterize_backward, exposed separately for screen-sharded multi-GPU training (SURVEY.md 8(e),
 * no reference counterpart): every rank runs gsr_backward_blend on its own band of tiles, the 48-byte per-Gaussian
 * records ([P,12], pointer returned in *splat_grads_out) are summed across ranks (RCCL all-reduce / reduce-scatter),
 * and gsr_backward_preprocess turns the summed records into the parameter gradients.
 * gsr_rasterize_backward == gsr_backward_blend followed by gsr_backward_preprocess on the same record array.
 */
int gsr_backward_blend(const GsrRasterSettings* settings, int P, int32_t num_rendered,
                       const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                       const float* dL_dout_color, const float* dL_dout_invdepth,
                       void* bwd_scratch, float** splat_grads_out, void* stream);
int gsr_backward_preprocess(const GsrRasterSettings* settings, int P, int M,
                            const float* means3D, const float* shs, const float* colors_precomp,
                            const float* opacities, const float* scales, const float* rotations,
                            const float* cov3D_precomp, const int32_t* radii, const void* geom_buffer,
                            const float* splat_grads,
                            float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity,
                            float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                            float* dL_dscales, float* dL_drotations, void* stream);

/*
 * Two-axis sharding (no reference counterpart): the per-Gaussian stages are sharded over the GAUSSIAN
 * axis (every rank owns P/G Gaussians, their parameters and optimizer state), binning + blending over the PIXEL axis
 * (bands of tile rows).  Forward: gsr_preprocess_forward on the own shard -> all-gather of the 64-byte splat records ->
 * gsr_rasterize_from_splats on the own band.  Backward: gsr_backward_blend on the own band (records of ALL Gaussians)
 * -> reduce-scatter (sum) of the 48-byte gradient records -> gsr_backward_preprocess on the own shard (geom_buffer may
 * be NULL there).  No parameter or parameter-gradient ever crosses ranks.
 *
 * gsr_preprocess_forward: projects the P Gaussians of the shard; writes radii[P] and splat_records[P,16] floats
 *   (x,y,conA,conB | conC,opacity,r,g | b,depth,tau,1/depth | rect.x,rect.y,0,tiles as bits) with the FULL-frame tile
 *   rectangle (tile_y0/tile_y1 of the settings are ignored).  geom_scratch: gsr_geometry_bytes(P) bytes.
 * gsr_rasterize_from_splats: bins and blends P gathered records inside the band the settings name; the records are
 *   copied into the geometry buffer (obtained through the callback), so the caller's array is not modified.  Buffers
 *   and outputs as gsr_rasterize_forward; a record whose tile count is 0 (culled, or a zero padding row) is ignored.
 */
int gsr_preprocess_forward(const GsrRasterSettings* settings, int P, int M,
                           const float* means3D, const float* shs, const float* colors_precomp,
                           const float* opacities, const float* scales, const float* rotations,
                           const float* cov3D_precomp, void* geom_scratch, int32_t* radii, float* splat_records,
                           void* stream);
int gsr_rasterize_from_splats(const GsrRasterSettings* settings, int P, const float* splat_records,
                              GsrResizeFn geom_resize, void* geom_user,
                              GsrResizeFn binning_resize, void* binning_user,
                              GsrResizeFn image_resize, void* image_user,
                              float* out_color, float* out_invdepth, int32_t* num_rendered, void* stream);

/*
 * Gaussian-sharded rendering (round 3; SURVEY.md 8(e), no reference counterpart): like the two-axis form, but the projected
 * splats are sent only where they are needed.  Rank g projects its P/G Gaussians (gsr_preprocess_forward), finds for every
 * band of tile rows [band_bounds[b], band_bounds[b+1]) the records whose tile rectangle touches it (gsr_route_count ->
 * band_counts, which the ranks exchange to size the buffers), packs them STABLY per band as 48-byte records
 * (x, y, conA, conB | conC, opacity, r, g | b, depth, rect.x bits, rect.y bits) together with their shard-local indices
 * (gsr_route_pack), and a variable-size all-to-all delivers segment b to rank b.  The receiver concatenates the segments in
 * rank order (= global Gaussian order when shards are contiguous index ranges, so depth ties resolve as on one GPU) and
 * calls gsr_rasterize_from_packed on its band: depth sort, scan, emission and tile sort then run on the band's Gaussians
 * only.  Backward: gsr_backward_blend on the received set -> reverse all-to-all of the [n,12] gradient rows ->
 * gsr_route_return adds the returned rows into splat_grads[P,12] of the shard (band order, no atomics, deterministic) ->
 * gsr_backward_preprocess.
 *   band_bounds : HOST array of n_bands + 1 tile-row indices (n_bands <= 64);  band_counts : DEVICE uint32[n_bands]
 *   band_offsets: HOST array of n_bands + 1 row offsets into packed / send_ids (exclusive scan of the counts)
 *   scratch     : gsr_route_scratch_bytes(P, n_bands) bytes, filled by gsr_route_count and read by gsr_route_pack
 */
size_t gsr_route_scratch_bytes(int P, int n_bands);
int gsr_route_count(int P, const float* splat_records, int n_bands, const int32_t* band_bounds, void* scratch,
                    uint32_t* band_counts, void* stream);
int gsr_route_pack(int P, const float* splat_records, int n_bands, const int32_t* band_bounds, const int64_t* band_offsets,
                   const void* scratch, float* packed, int32_t* send_ids, void* stream);
int gsr_rasterize_from_packed(const GsrRasterSettings* settings, int P, const float* packed_records,
                              GsrResizeFn geom_resize, void* geom_user,
                              GsrResizeFn binning_resize, void* binning_user,
                              GsrResizeFn image_resize, void* image_user,
                              float* out_color, float* out_invdepth, int32_t* num_rendered, void* stream);
int gsr_route_return(int P, int n_bands, const int64_t* band_offsets, const int32_t* send_ids, const float* returned,
                     float* splat_grads, void* stream);

/*
 * FIXED-CAPACITY form of the same exchange (round 4): no count matrix travels to the host before the records do.
 * gsr_route_pack_fixed lays the records of band b into a segment of capacity + 1 rows of `packed` (and of `send_ids`):
 *   row b*(capacity+1)            header: word 0 = band_counts[b] (may exceed capacity), word 1 = capacity, rest 0; send id -1
 *   rows .. + 1 .. + capacity     the first min(band_counts[b], capacity) records of the band, stable; unused rows: send id -1
 * Segment sizes do not depend on the counts, so the all-to-all has equal splits.  The receiver passes the n_segments segments it
 * got (rank order) to gsr_rasterize_from_segments: header rows and rows past a segment's count enter the frame as Gaussians
 * without tiles (P = n_segments * (capacity + 1) for gsr_backward_blend / gsr_backward_scratch_bytes), so depth ties still resolve
 * in (source rank, index) order.  The gradient rows go back in the same layout; gsr_route_return with band_offsets[b] =
 * b*(capacity+1) skips the rows whose send id is negative.  A segment whose count exceeds the capacity has lost records: the
 * caller compares band_counts with the capacity ON THE DEVICE, agrees on the outcome with the other ranks and repeats the frame
 * with the exact form (parallel.py does this without an extra host synchronisation).
 *   band_counts : DEVICE uint32[n_bands] as written by gsr_route_count;  scratch as for gsr_route_pack
 */
int gsr_route_pack_fixed(int P, const float* splat_records, int n_bands, const int32_t* band_bounds, int capacity,
                         const void* scratch, const uint32_t* band_counts, float* packed, int32_t* send_ids, void* stream);
int gsr_rasterize_from_segments(const GsrRasterSettings* settings, int n_segments, int capacity, const float* segments,
                                GsrResizeFn geom_resize, void* geom_user,
                                GsrResizeFn binning_resize, void* binning_user,
                                GsrResizeFn image_resize, void* image_user,
                                float* out_color, float* out_invdepth, int32_t* num_rendered, void* stream);

/*
 * OPT-IN fusion of the optimizer into the backward (no reference counterpart): gsr_backward_preprocess for the split-SH form
 * (settings->sh_dc = coefficient 0 [P,1,3], shs_rest = coefficients 1..15 [P,15,3], M = 16) that does NOT write the two SH
 * gradients but applies their Adam step in place, from the gradient tile in LDS: the gradient (204 B per Gaussian) never travels
 * to HBM and back.  settings->sh_dc and shs_rest are read (colour clamp) and then UPDATED; the moments likewise.
 *   sparse = 0: torch.optim.Adam on every row (rows without gradient take the g = 0 step), bias correction with step_dc / step_rest
 *   sparse = 1: SparseGaussianAdam on the rows with radii > 0, no bias correction
 * Every parameter comes out bit-identical to gsr_backward_preprocess followed by gsr_adam_step / gsr_sparse_adam_step on the two
 * tensors.  The caller owns the consequences: the step happens inside backward, once per call (no gradient accumulation).
 * All six SH arrays must be 16-byte aligned; colors_precomp is not supported in this form.
 */
typedef struct GsrShAdam {
    float* dc_exp_avg;
    float* dc_exp_avg_sq;
    float* rest_exp_avg;
    float* rest_exp_avg_sq;
    double lr_dc, lr_rest, beta1, beta2, eps;
    int32_t step_dc, step_rest;      /* 1-based, after incrementing; ignored when sparse */
    int32_t sparse;
    int32_t reserved;
} GsrShAdam;
int gsr_backward_preprocess_sh_adam(const GsrRasterSettings* settings, int P, int M, const float* means3D, float* shs_rest,
                                    const float* opacities, const float* scales, const float* rotations,
                                    const float* cov3D_precomp, const int32_t* radii, const void* geom_buffer,
                                    const float* splat_grads, float* dL_dmeans2D, float* dL_dopacity, float* dL_dmeans3D,
                                    float* dL_dcov3D, float* dL_dscales, float* dL_drotations, const GsrShAdam* adam,
                                    void* stream);

/*
 * Fused dense Adam step on one fp32 tensor of n elements (the optimizer step of train.py:177-186).
 * Same arithmetic as torch.optim.Adam(betas, eps) without weight decay / amsgrad; `step` is the 1-based step count
 * AFTER incrementing; state tensors exp_avg / exp_avg_sq are updated in place.
 */
int gsr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  double lr, double beta1, double beta2, double eps, int32_t step, void* stream);
/*
 * The same step for several tensors (a 3DGS model: positions, SH, opacities, scales, rotations -- one param group each,
 * scene/gaussian_model.py:178-211) in as few launches as possible: up to GSR_ADAM_MAX_TENSORS tensors share one kernel launch,
 * longer lists are split.  Results are bit-identical to gsr_adam_step on every tensor.
 */
#define GSR_ADAM_MAX_TENSORS 8
typedef struct GsrAdamTensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    double lr, beta1, beta2, eps;
    int32_t step;            /* 1-based, after incrementing */
    int32_t reserved;
} GsrAdamTensor;
int gsr_adam_step_multi(const GsrAdamTensor* tensors, int32_t count, void* stream);

/*
 * Sparse Adam step: replaces `_C.adamUpdate` behind `SparseGaussianAdam.step(visibility, N)` of the
 * reference's accelerated rasterizer (imported at train.py:37-41 and scene/gaussian_model.py:24-27, stepped at
 * train.py:180-183).  The tensor is N rows of M elements; rows with visible[row] == 0 are skipped entirely (parameter and
 * both moments untouched).  [RECALLED -- the source is an un-vendored submodule] no bias correction:
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr m / (sqrt(v) + eps).   visible is uint8 / bool [N].
 */
int gsr_sparse_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* visible,
                         int64_t N, int64_t M, double lr, double beta1, double beta2, double eps, void* stream);
/* The same step for all parameter groups of one `SparseGaussianAdam.step(visibility, N)` call (the six tensors of
 * scene/gaussian_model.py:183-190, one group each) in ONE launch: tensor i is N rows of tensors[i].M elements with its own lr / eps;
 * visibility, N and the betas are shared.  Up to GSR_ADAM_MAX_TENSORS tensors per launch, longer lists are split.  Bit-identical to
 * gsr_sparse_adam_step on every tensor.  (Added in round 5; additive, the ABI version stays 4.) */
typedef struct GsrSparseAdamTensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t M;               /* elements per row (Gaussian) */
    double lr, eps;
} GsrSparseAdamTensor;
int gsr_sparse_adam_step_multi(const GsrSparseAdamTensor* tensors, int32_t count, const uint8_t* visible, int64_t N, double beta1,
                               double beta2, void* stream);

/*
 * Per-iteration statistics of adaptive density control: GaussianModel.add_densification_stats
 * (scene/gaussian_model.py:471-473, called at train.py:167) and the max_radii2D update of train.py:166, as ONE pass instead of
 * boolean-mask torch ops (three host synchronisations per training iteration):
 *   for visible i:  grad_accum[i] += sqrt(g[i,0]^2 + g[i,1]^2);  denom[i] += 1;  max_radii2D[i] = max(max_radii2D[i], radii[i])
 * viewspace_grad[P,3] = the operator's dL/dmeans2D; visible uint8/bool [P] or NULL (= radii > 0, what
 * gaussian_renderer/__init__.py:123 passes); radii int32 [P] or NULL (then max_radii2D is not touched).
 */
int gsr_density_stats(int P, const float* viewspace_grad, const uint8_t* visible, const int32_t* radii, float* grad_accum,
                      float* denom, float* max_radii2D, void* stream);

/*
 * Replaces `simple_knn._C.distCUDA2` (un-vendored submodule submodules/simple-knn, .gitmodules:1-3; called once per
 * scene at scene/gaussian_model.py:159): mean_dist2[i] = mean of the squared Euclidean distances
 * from points[i] to its 3 nearest OTHER points (exact; coincident points count with distance 0; fewer than 4 points
 * leaves FLT_MAX terms in the mean, as in the reference).  points[N,3] fp32, scratch of gsr_knn_scratch_bytes(N) bytes.
 */
size_t gsr_knn_scratch_bytes(int N);
int gsr_knn_mean_dist2(int N, const float* points, float* mean_dist2, void* scratch, void* stream);

/*
 * Fused SSIM map: replaces the un-vendored `fused_ssim` extension (train.py:31-35,122; its
 * `fusedssim` / `fusedssim_backward`), same numerics as utils/loss_utils.py:56-87 (11x11 Gaussian window, sigma 1.5,
 * zero padding 5, C1 = 0.01^2, C2 = 0.03^2).  Images are [planes, H, W] fp32 (planes = batch x channels).
 * forward writes the SSIM map and -- when the three derivative maps are non-NULL (training) -- dm/dmu1, dm/dE[x^2],
 * dm/dE[xy]; backward turns dL/dmap into dL/dimg1.
 */
int gsr_ssim_forward(int planes, int H, int W, const float* img1, const float* img2, float* ssim_map,
                     float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream);
int gsr_ssim_backward(int planes, int H, int W, const float* img1, const float* img2, const float* dL_dmap,
                      const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                      float* dL_dimg1, void* stream);

/*
 * Mean-SSIM form of the same kernels -- what `fused_ssim(img1, img2)` returns (train.py:122): the forward writes one
 * partial sum per wave (per tile in the LDS-tiled A/B form) into `partials` -- gsr_ssim_partial_count floats is room for
 * either form -- and their mean, added in fixed order, into mean_out[1]; the backward takes dL/dmean as ONE device scalar.  Saves the SSIM-map round trip and the framework's
 * reduction / broadcast kernels; results equal gsr_ssim_forward + mean up to fp32 summation order.
 */
int64_t gsr_ssim_partial_count(int planes, int H, int W);
int gsr_ssim_mean_forward(int planes, int H, int W, const float* img1, const float* img2, float* partials, float* mean_out,
                          float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream);
int gsr_ssim_mean_backward(int planes, int H, int W, const float* img1, const float* img2, const float* dL_dmean,
                           const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1,
                           void* stream);

/*
 * The whole training loss of train.py:119-126 in the same two kernels (round 3, SURVEY.md 8(f) N1 "fused SSIM + L1"):
 *     loss = (1 - lambda_dssim) * mean|img1 - img2| + lambda_dssim * (1 - mean SSIM(img1, img2))
 * (utils/loss_utils.py:40-41 for the L1 term).  The forward reads both images once (the L1 sum comes from the pixels the
 * SSIM window already staged) and writes loss_out[3] = (loss, L1, SSIM) -- the two parts for the progress bar of
 * train.py:147-151; the backward takes dL/dloss as ONE device scalar and writes dL/dimg1 = -lambda dSSIM/dimg1 +
 * (1 - lambda) sign(img1 - img2) / count once.  partials: 2 * gsr_ssim_partial_count(planes, H, W) floats.
 */
int gsr_train_loss_forward(int planes, int H, int W, const float* img1, const float* img2, float lambda_dssim, float* partials,
                           float* loss_out, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream);
int gsr_train_loss_backward(int planes, int H, int W, const float* img1, const float* img2, const float* dL_dloss,
                            float lambda_dssim, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                            float* dL_dimg1, void* stream);

/* Replaces _C.mark_visible: present[i] = 1 iff Gaussian i is in front of the 0.2 near plane. */
int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/*
 * Introspection for tests and bench (no reference counterpart).  Views into the buffers forward filled.
 * Pointers are device pointers valid while the buffers live.
 */
typedef struct GsrForwardViews {
    const float* splats;          /* [P,16] x,y,conA,conB | conC,opacity,r,g | b,depth,tau,1/depth | rect,goffset,tiles (bits) */
    const uint32_t* tiles_touched;/* [P] */
    const uint32_t* depth_order;  /* [P] Gaussian ids in (depth, id) order; culled ones last */
    const uint32_t* point_list;   /* [R] Gaussian ids sorted by (tile, depth, id) */
    const uint32_t* ranges;       /* [n_tiles,2] start,end */
    const float* final_T;         /* [H*W] */
    const uint32_t* n_contrib;    /* [H*W] */
    const uint32_t* tile_scan;    /* [P] inclusive scan of tiles_touched in depth order (ABI 4) */
} GsrForwardViews;
int gsr_forward_views(int P, int64_t R, int width, int height,
                      const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                      GsrForwardViews* out);

/* Per-stage GPU timing (HIP events on `stream`, recorded around each kernel group when enabled).
 * Stage ids index GSR_STAGE_*; times accumulate until reset.  Used by bench.py for roofline.achieved. */
enum {
    GSR_STAGE_PREPROCESS = 0,
    GSR_STAGE_DEPTH_SORT = 1,
    GSR_STAGE_SCAN = 2,
    GSR_STAGE_EMIT = 3,
    GSR_STAGE_TILE_SORT = 4,
    GSR_STAGE_RANGES = 5,
    GSR_STAGE_RENDER = 6,
    GSR_STAGE_RENDER_BWD = 7,
    GSR_STAGE_PREPROCESS_BWD = 8,
    GSR_STAGE_GATHER_BWD = 9,
    GSR_STAGE_COLOR = 10,            /* (unused since ABI 4: the split preprocess was removed) */
    GSR_STAGE_R_WAIT = 11,           /* GPU idle between the depth sort and the first kernel the host launches after reading R */
    GSR_STAGE_COUNT = 12
};
int gsr_profile_enable(int on);      /* bit 0: per-stage events; bit 1: work counters (slow the blend kernels: count in a separate pass);
                                      * bit 2: per-wave trace of the blend kernels instead of the counters (gsr_profile_trace) */
int gsr_profile_reset(void);
/* Resolves pending events (synchronises them) and returns accumulated ms and launch counts per stage. */
int gsr_profile_read(float* ms_out, int32_t* count_out, int n);
/* Work counters of the blend kernels, accumulated over the launches made while profiling is enabled (bench.py turns them
 * into achieved FLOP/s): [0] forward (8x8 pixel block, list entry) pairs blended by a whole wave, [1] forward batches of
 * 64 entries box-tested, [2] / [3] the same for the blend backward.  reset != 0 clears them after reading.
 * [4] / [5]: steps of the HEAVIEST wave of the forward / backward launches (max, not sum): with the wave counts this gives the
 * tail of a launch, slowest wave / mean wave, which is what clustered scenes stress. */
#define GSR_COUNTER_COUNT 6
int gsr_profile_counters(uint64_t* out, int n, int reset);
/* Per-wave trace of the two blend kernels (measurement only; gsr_profile_enable(4)): entry w of the most recent launch's wave w is four
 * 64-bit words -- start and end time (100 MHz constant clock, s_memrealtime), placement (HW_ID bits 0-31, XCC id bits 32-35,
 * kernel bits 40-41: 1 forward, 2 backward) and the wave's blend steps.  Waves that left early (nothing to blend) keep a zero entry.
 * Copies min(max_waves, GSR_TRACE_WAVES) entries to `out`, clears the device copy, returns the number copied (negative: error).
 * This is how the tail of a blend launch was measured (tools/gpu_wave_trace.py). */
#define GSR_TRACE_WAVES 65536
int gsr_profile_trace(uint64_t* out, int max_waves);

/* Switches.  The product library accepts the tuning knobs sort_small_block_threshold, sort_mid_block_threshold,
 * sort_items_large, tile_sort_mode (0 fused two-level sort, 1 legacy LSD passes) and
 *   depth_sort_mode   0 = automatic: the bucket depth sort (4 launches, csrc/depthsort.hip) up to 3 M Gaussians, the LSD radix
 *                     passes beyond and for 64 frames after a frame whose depths crowded one bucket; 1 = always LSD; 2 = always
 *                     the bucket sort (all three give bit-identical bins)
 *   level2_scan_mode  0 = automatic, 1 = the tile sort's level-2 scan as its own launch, 2 = folded into the scatter
 * The measurement build (GSR_AB=1 python build.py -> lib_ab/, sources in tools/ab_variants/) also compiles render_fwd_variant 1 / 3,
 * render_bwd_variant 1 / 4 / 5 and the round-5 options fwd_bands (level-2 sort + blend band by band on two HIP streams) and
 * render_fwd_lds_pad (resident waves of the forward blend capped through LDS) -- all measured and rejected
 * (profiles/r05_ab_fwd_bands_occupancy.json); the product accepts only the defaults.  (ABI 4 removed the options of experiments
 * whose code left the sources: onesweep depth sort, color_overlap, first_hist_in_preprocess, sh_dma.)  Unknown names /
 * unavailable values return an error.
 * Switches of the product library (every setting gives the same results; 0 is the form they replaced):
 *   snug_tiles       1 = a Gaussian is binned into the tiles its alpha >= 1/255 ellipse can reach, 0 = into the reference's square
 *                    of radius 3 sqrt(lambda_max) (same outputs, ~1.4x the instances)
 *   bwd_heavy_first  launch order of the blend backward, planned from the forward's per-block step counts: 2 (default) = tiles by
 *                    their heaviest half, 1 = tiles by the sum of their four blocks (round 3), 3 = every half tile on its own,
 *                    0 = index order; scheduling only -- the gradients are the same bits in every order
 * and ssim_variant (0 = marching-wave SSIM / training-loss kernels, 1 = the LDS-tiled form), ssim_target_waves (launch shape of
 * the marching form), preprocess_grid_cap.
 * Test hook: debug_dirty_control_block = t preloads t tickets into the frame counter of the NEXT forward call, once (a counter that is
 * not zero when a frame begins, csrc/gsr_frame.h): that frame is wrong or refused, the ones after it must be right again. */
int gsr_set_option(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
