// C ABI of libgsr_hip.so (include/gsr.h): argument checking, scratch carving, kernel orchestration.
// Host-side counterpart of the reference's rasterize_points.cu / CudaRasterizer::Rasterizer glue
// (un-vendored; call sites gaussian_renderer/__init__.py:91-110, train.py:142) -- re-designed around a
// depth-sort + stable tile-sort binning pipeline instead of one 64-bit key sort.
#include <hip/hip_runtime.h>

#include <sched.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "gsr_internal.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_OK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(GSR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

// ---- options / profiling ----
int g_render_fwd_variant = 0;
int g_render_bwd_variant = 0;
int g_depth_sort_mode = 0;      // 0 = automatic (bucket sort, depthsort.hip, up to GSR_DS_MAX_P Gaussians; LSD radix sort beyond, and
                                // for a while after a frame whose depths crowded one bucket), 1 = always LSD, 2 = always bucket sort
int g_snug_tiles = 1;           // 1 = bin every Gaussian into its snug tile rectangle (gsr_math.h); 0 = the reference's square (A/B)
int g_bwd_heavy_first = 2;      // launch order of the blend backward (plan kernel, render_bwd.hip): 0 = tiles in index order, 1 = heaviest tiles first (sum of
                                // the four blocks: round 3), 2 = tiles by their heaviest half (default: blend backward 0.343 -> 0.324 ms on the bench frame),
                                // 3 = every half tile (= wave) on its own, heaviest first (0.325; slower than 2 on the clustered scene: the two halves of
                                // a tile no longer share their gathered records in one XCD's L2)
int g_tile_sort_mode = 0;       // 0 = fused emission + two-level sort (tilesort.hip), 1 = legacy emit + LSD passes (A/B)

struct PendingEvent { int stage; hipEvent_t a, b; };
std::mutex g_prof_mu;
bool g_prof_on = false;
bool g_count_on = false;      // work counters of the blend kernels (bit 1 of gsr_profile_enable): they serialise on two global words
                              // and slow the kernels several times over, so they are never on while stages are being timed
std::vector<PendingEvent> g_pending;
std::vector<hipEvent_t> g_pool;
double g_stage_ms[GSR_STAGE_COUNT] = {0};
int g_stage_n[GSR_STAGE_COUNT] = {0};
constexpr int GSR_MAX_DEVICES = 64;
unsigned long long* g_counters_dev[GSR_MAX_DEVICES] = {nullptr};     // per device: work counters of the blend kernels (measurement only)

hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
struct StageTimer {
    int stage; hipStream_t st; bool on; hipEvent_t a, b;
    StageTimer(int stage_, hipStream_t st_) : stage(stage_), st(st_), on(g_prof_on) {
        if (on) { std::lock_guard<std::mutex> l(g_prof_mu); a = get_event(); b = get_event(); (void)hipEventRecord(a, st); }
    }
    ~StageTimer() {
        if (on) { (void)hipEventRecord(b, st); std::lock_guard<std::mutex> l(g_prof_mu); g_pending.push_back({stage, a, b}); }
    }
};

int check_stage(const GsrRasterSettings* s, hipStream_t st, const char* what) {
    if (s->debug) {
        hipError_t e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) return fail(GSR_ERR_HIP, std::string("after ") + what + ": " + hipGetErrorString(e));
    }
    return GSR_OK;
}
#define STAGE_CHECK(what) do { int rc_ = check_stage(settings, st, what); if (rc_ != GSR_OK) return rc_; } while (0)

int bits_for(uint32_t n_values) {   // number of bits needed to represent 0 .. n_values-1
    int b = 0;
    while (b < 32 && (1ull << b) < (unsigned long long)n_values) ++b;
    return b < 1 ? 1 : b;
}

int make_cam(const GsrRasterSettings* s, int M, GsrCamDev& c) {
    if (!s) return fail(GSR_ERR_INVALID_ARG, "settings is NULL");
    if (s->image_width <= 0 || s->image_height <= 0) return fail(GSR_ERR_INVALID_ARG, "image size must be positive");
    if (!s->bg || !s->viewmatrix || !s->projmatrix || !s->campos)
        return fail(GSR_ERR_INVALID_ARG, "bg / viewmatrix / projmatrix / campos must be device pointers");
    if (s->sh_degree < 0 || s->sh_degree > 3) return fail(GSR_ERR_UNSUPPORTED, "sh_degree must be 0..3");
    c.W = s->image_width;
    c.H = s->image_height;
    c.gx = (c.W + GSR_TILE - 1) / GSR_TILE;
    c.gy = (c.H + GSR_TILE - 1) / GSR_TILE;
    if (c.gx > 65535 || c.gy > 65535) return fail(GSR_ERR_UNSUPPORTED, "more than 65535 tiles per axis");
    // a Gaussian's tile count (rectangle area) and the tile ids must stay well inside 32 bits
    if ((int64_t)c.gx * c.gy > (1ll << 24)) return fail(GSR_ERR_UNSUPPORTED, "more than 2^24 tiles");
    // fp32 host arithmetic, same expression as oracle/torch_oracle.py:preprocess
    c.focal_x = (float)c.W / (2.0f * s->tanfovx);
    c.focal_y = (float)c.H / (2.0f * s->tanfovy);
    c.limx = 1.3f * s->tanfovx;
    c.limy = 1.3f * s->tanfovy;
    c.scale_modifier = s->scale_modifier;
    c.sh_degree = s->sh_degree;
    c.M = M;
    c.antialiasing = s->antialiasing ? 1 : 0;
    c.snug = g_snug_tiles;
    int y0 = s->tile_y0, y1 = s->tile_y1;
    if (y1 <= 0) { y0 = 0; y1 = c.gy; }
    if (y0 < 0) y0 = 0;
    if (y1 > c.gy) y1 = c.gy;
    if (y0 > y1) y0 = y1;
    c.tile_y0 = y0;
    c.tile_y1 = y1;
    c.view = s->viewmatrix;
    c.proj = s->projmatrix;
    c.campos = s->campos;
    c.bg = s->bg;
    c.sh_dc = s->sh_dc;
    c.dL_dsh_dc = s->dL_dsh_dc;
    return GSR_OK;
}

int check_inputs(int P, int M, const float* means3D, const float* shs, const float* colors_precomp,
                 const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                 int sh_degree) {
    if (P < 0) return fail(GSR_ERR_INVALID_ARG, "P < 0");
    if (P == 0) return GSR_OK;
    if (!means3D || !opacities) return fail(GSR_ERR_INVALID_ARG, "means3D / opacities are NULL");
    if ((shs == nullptr) == (colors_precomp == nullptr))
        return fail(GSR_ERR_INVALID_ARG, "provide exactly one of shs / colors_precomp");
    const bool sr = scales != nullptr && rotations != nullptr;
    if ((scales == nullptr) != (rotations == nullptr) || sr == (cov3D_precomp != nullptr))
        return fail(GSR_ERR_INVALID_ARG, "provide exactly one of (scales, rotations) / cov3D_precomp");
    if (shs) {
        if (M <= 0 || M > 16) return fail(GSR_ERR_UNSUPPORTED, "SH coefficient count M must be 1..16");
        if ((sh_degree + 1) * (sh_degree + 1) > M) return fail(GSR_ERR_INVALID_ARG, "sh_degree needs more coefficients than M");
    }
    return GSR_OK;
}

// split SH form (settings->sh_dc): coefficient 0 in sh_dc[P,1,3], coefficients 1..M-1 in shs[P,M-1,3]
int check_split_sh(const GsrRasterSettings* s, int P, int M, const float* shs, bool backward) {
    if (!s->sh_dc || P == 0) return GSR_OK;
    if (!shs) return fail(GSR_ERR_INVALID_ARG, "sh_dc given without shs (the coefficients 1..M-1)");
    if (M != 16) return fail(GSR_ERR_UNSUPPORTED, "split SH form needs M == 16 (degree-3 storage); concatenate smaller records");
    if ((((uintptr_t)s->sh_dc) | ((uintptr_t)shs)) & 15) return fail(GSR_ERR_UNSUPPORTED, "split SH form needs 16-byte aligned sh_dc / shs");
    if (backward && !s->dL_dsh_dc) return fail(GSR_ERR_INVALID_ARG, "sh_dc given but dL_dsh_dc is NULL");
    return GSR_OK;
}

// R read-back word: mapped + portable + coherent pinned host memory: [0] = R low word, [1] = sequence number, [2] = R high word,
// [3] = "a depth key needed more than 27 bits".  ("The bucket depth sort met an oversized segment" is NOT part of the leased slot
// (the segment sort may still be running when the slot goes back to the pool): it is a per-device mapped word, g_slow_word.)
// Written by the last workgroup of the key-producing kernel (gsr_frame.h), which folds the frame's statistics in `state`, a
// 64-byte block of device memory that is zero between frames.  Words are LEASED per call from a per-device pool (ADVICE r02:
// keyed by device, not by thread -- a host thread that comes and goes leaks nothing, concurrent callers never share a word);
// never freed (the runtime releases them with the context).
struct HostWord {
    uint32_t* host = nullptr; uint32_t* dev = nullptr; uint32_t* state = nullptr; uint32_t seq = 0;
};
std::mutex g_hw_mu;
std::vector<HostWord> g_hw_pool[GSR_MAX_DEVICES];
std::atomic<int64_t> g_last_R[GSR_MAX_DEVICES];      // per device: sizes the speculative binning buffer of the next frame
std::atomic<int> g_lsd_frames[GSR_MAX_DEVICES];      // per device: frames for which the automatic depth sort stays with the LSD passes
std::atomic<int> g_lsd_backoff[GSR_MAX_DEVICES];     // per device: length of the next such stay (doubles per failed retry; heuristic only)
std::atomic<int> g_lsd_probation[GSR_MAX_DEVICES];   // per device: clean bucket-sort frames still needed after a stay before the back-off is forgotten
uint32_t* g_slow_word_dev[GSR_MAX_DEVICES] = {nullptr};      // its device-side address
std::atomic<int> g_debug_dirty_tickets{0};           // TEST HOOK (gsr_set_option debug_dirty_control_block): tickets preloaded into the next frame's counter, once
uint32_t* g_slow_word[GSR_MAX_DEVICES] = {nullptr};  // per device, mapped host memory: set (1) by ds_segsort when a segment overflowed the LDS capacity;
                                                     // read and cleared by the next lease on that device.  A heuristic flag: a store that races the
                                                     // clear is at worst seen one frame later or lost once, never attributed to another device
struct HostWordLease {
    int dev = -1;
    HostWord hw;
    hipStream_t st = nullptr;
    bool settled = true;      // false while a kernel that will still write the word may be pending
    ~HostWordLease() {
        if (dev < 0) return;
        if (!settled) {      // error paths only: never hand a word with a pending writer (or a half-filled state block) to the next call
            (void)hipMemsetAsync(hw.state, 0, 64, st);      // the ticket counter of gsr_frame.h (behind whatever kernel may still add to it)
            (void)hipStreamSynchronize(st);
        }
        std::lock_guard<std::mutex> l(g_hw_mu);
        g_hw_pool[dev].push_back(hw);
    }
};
int lease_host_word(HostWordLease& lease, hipStream_t st) {
    int dev_id = 0;
    HIP_OK(hipGetDevice(&dev_id));
    if (dev_id < 0 || dev_id >= GSR_MAX_DEVICES) return fail(GSR_ERR_UNSUPPORTED, "device ordinal out of range");
    {
        std::lock_guard<std::mutex> l(g_hw_mu);
        if (!g_hw_pool[dev_id].empty()) { lease.hw = g_hw_pool[dev_id].back(); g_hw_pool[dev_id].pop_back(); }
    }
    if (!lease.hw.host) {      // a new slot (first call on this device, or one more concurrent caller): 64 B mapped host + 64 B device, kept for good
        HostWord nw;
        hipError_t e = hipHostMalloc((void**)&nw.host, 64, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&nw.dev, nw.host, 0);
        if (e == hipSuccess) e = hipMalloc((void**)&nw.state, 64);
        // cleared ON THE CALLER'S STREAM, in front of the kernel that will count in it: hipMemset() of device memory is queued on the null stream and
        // returns before it has run, and a non-blocking stream (every torch side stream) is not ordered behind the null stream -- the clear could land
        // in the middle of the first frame, lose its tickets, and leave a counter that is never zero again (round 6: a GPU suite run lost
        // the second caller's first frame of tests/test_gpu_parity.py::test_concurrent_forward_calls_from_two_host_threads and every frame after it)
        if (e == hipSuccess) e = hipMemsetAsync(nw.state, 0, 64, st);
        if (e != hipSuccess) {      // (nothing half-made is leaked or pooled)
            if (nw.state) (void)hipFree(nw.state);
            if (nw.host) (void)hipHostFree(nw.host);
            return fail(GSR_ERR_HIP, std::string("control block allocation: ") + hipGetErrorString(e));
        }
        for (int i = 0; i < 16; ++i) nw.host[i] = 0;
        lease.hw = nw;
    }
    {
        std::lock_guard<std::mutex> l(g_hw_mu);
        if (!g_slow_word[dev_id]) {
            uint32_t* w = nullptr;
            uint32_t* wd = nullptr;
            hipError_t e = hipHostMalloc((void**)&w, 64, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent);
            if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&wd, w, 0);
            if (e != hipSuccess) {      // (the slot leased above goes back to the pool -- lease.dev is set first -- nothing leaks)
                if (w) (void)hipHostFree(w);
                lease.dev = dev_id;
                lease.st = st;
                return fail(GSR_ERR_HIP, std::string("slow-word allocation: ") + hipGetErrorString(e));
            }
            for (int i = 0; i < 16; ++i) w[i] = 0;
            g_slow_word_dev[dev_id] = wd;
            g_slow_word[dev_id] = w;
        }
    }
    lease.hw.host[3] = 0;
    uint32_t* slow = g_slow_word[dev_id];
    // (read and cleared in ONE step: two concurrent callers never both see the same report and double the stay twice -- ADVICE r05)
    if (__atomic_load_n(slow, __ATOMIC_RELAXED) && __atomic_exchange_n(slow, 0u, __ATOMIC_RELAXED)) {
                                 // a recent frame on this device: thousands of Gaussians in one depth bucket -- its segment went
                                 // through global memory.  Stay with the LSD passes for a while, then try again: 64 frames the first
        int back = g_lsd_backoff[dev_id].load();      // time, twice as long after every retry that met an oversized segment again
        if (back < 64) back = 64;                     // (a scene that always crowds a bucket pays one slow frame in 65, 129, ... 8193)
        g_lsd_frames[dev_id].store(back);
        g_lsd_backoff[dev_id].store(back >= 8192 ? 8192 : back * 2);
        g_lsd_probation[dev_id].store(0);
    } else if (g_lsd_probation[dev_id].load() > 0 && g_lsd_frames[dev_id].load() == 0) {
        //  the stay is over and bucket-sort frames come back clean: after three of them (the flag is written by a kernel that
        // may still be running when the next lease looks) the doubling is forgotten -- isolated crowded frames of a long run no longer add up
        if (g_lsd_probation[dev_id].fetch_sub(1) == 1) g_lsd_backoff[dev_id].store(64);
    }
    lease.dev = dev_id;
    lease.st = st;
    return GSR_OK;
}
uint32_t* slow_word_dev(int dev_id) {      // device-side address of the device's "oversized segment" flag (mapped host memory; cached with it)
    return g_slow_word[dev_id] ? g_slow_word_dev[dev_id] : nullptr;
}
unsigned long long* counters_for_current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= GSR_MAX_DEVICES) return nullptr;
    return g_counters_dev[d];
}

#if defined(__x86_64__) || defined(__i386__)
#define GSR_CPU_RELAX() __builtin_ia32_pause()
#elif defined(__aarch64__)
#define GSR_CPU_RELAX() asm volatile("yield" ::: "memory")
#else
#define GSR_CPU_RELAX() do { } while (0)
#endif

// keys per workgroup of a radix pass: 1024 below 512 K keys, 2048 below 3 M, 4096 above.  Measured on the depth sort of
// 1 M keys with the round-2 kernels (one box, interleaved): 1024 -> 0.0892 ms, 2048 -> 0.0855 ms, 4096 -> 0.0919 ms.
int64_t g_small_block_threshold = (int64_t)512 * 1024;
int64_t g_mid_block_threshold = (int64_t)3 * 1024 * 1024;
int g_sort_items_large = GSR_SORT_ITEMS;                      // keys per workgroup above the thresholds (1024 / 2048 / 4096 / 8192)
int sort_items(int64_t n) {
    if (n < g_small_block_threshold) return GSR_SORT_ITEMS_SMALL;
    if (n < g_mid_block_threshold) return 2048 < g_sort_items_large ? 2048 : g_sort_items_large;
    return g_sort_items_large;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
GsrGeom gsr_carve_geom(char* base, int P) {
    GsrGeom g;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off = gsr_align128(off + bytes); return p; };
    const size_t n = (size_t)(P > 0 ? P : 1);
    g.splats = (float4*)take(n * 64);
    g.rect = (uint2*)take(n * 8);
    g.tiles = (uint32_t*)take(n * 4);
    g.clamped = (uint32_t*)take(n * 4);
    g.keys[0] = (uint32_t*)take(n * 4);
    g.keys[1] = (uint32_t*)take(n * 4);
    g.vals[0] = (uint32_t*)take(n * 4);
    g.vals[1] = (uint32_t*)take(n * 4);
    g.rect_sorted = (uint2*)take(n * 8);
    g.offsets = (uint32_t*)take(n * 4);
    g.block_sums = (uint64_t*)take(((n + GSR_SCAN_ITEMS - 1) / GSR_SCAN_ITEMS) * 8);
    g.block_first = (uint2*)take(gsr_block_first_cap(P) * 8);
    g.sort_hist = (uint32_t*)take((size_t)GSR_SORT_MAX_DIGITS * (size_t)gsr_sort_blocks((int64_t)n, true) * 4);   // small workgroups: worst case
    g.digit_total = (uint32_t*)take(GSR_SORT_MAX_DIGITS * 4);
    {
        const bool ds = P <= GSR_DS_MAX_P;
        const size_t np = ds ? n : 1, nblk = ds ? gsr_depth_bucket_blocks(P) : 1, nseg = ds ? gsr_depth_bucket_segments(P) : 1;
        g.ds.pairs[0] = (uint2*)take(np * 8);
        g.ds.pairs[1] = (uint2*)take(np * 8);
        g.ds.cnt_tab = (uint32_t*)take(nblk * GSR_DS_BUCKETS * 4);
        g.ds.tile_tab = (uint32_t*)take(nblk * GSR_DS_BUCKETS * 4);
        g.ds.cnt_total = (uint32_t*)take(GSR_DS_BUCKETS * 4);
        g.ds.tile_total = (uint32_t*)take(GSR_DS_BUCKETS * 4);
        g.ds.plan = (uint32_t*)take(nseg * GSR_DS_PLAN_WORDS * 4);
        g.ds.eq_tab = (uint32_t*)take((size_t)GSR_EQ_TAB_WORDS * 4);
        g.ds.bucket_of = (uint16_t*)take((np + 64) * 2);
    }
    g.num_rendered = (uint32_t*)take(128);
    g.wg_range = (uint2*)take((size_t)GSR_FRAME_MAX_GROUPS * 8);
    g.bytes = off;
    return g;
}

GsrBinning gsr_carve_binning(char* base, int64_t R) {
    GsrBinning b;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off = gsr_align128(off + bytes); return p; };
    const size_t n = (size_t)(R > 0 ? R : 1);
    {   // one contiguous 8n-byte region: the two key buffers of the LSD sort, or the packed words of the fused sort
        char* wk = take(gsr_align128(n * 4) * 2);
        b.keys[0] = (uint32_t*)wk;
        b.keys[1] = (uint32_t*)(wk ? wk + gsr_align128(n * 4) : nullptr);
    }
    b.vals[0] = (uint32_t*)take(n * 4);
    b.vals[1] = (uint32_t*)take(n * 4);
    b.sort_hist = (uint32_t*)take((size_t)256 * (size_t)gsr_sort_blocks((int64_t)n, true) * 4);   // worst case
    const size_t nblk = (n + GSR_TS_ITEMS - 1) / GSR_TS_ITEMS;
    b.hist2 = (uint32_t*)take((nblk + 256) * 256 * 4);
    b.digit_total = (uint32_t*)take(256 * 4);
    b.bucket_base = (uint32_t*)take(257 * 4);
    b.blk2_start = (uint32_t*)take(257 * 4);
    b.tile_base = (uint32_t*)take(65536 * 4);
    b.block_first = (uint2*)take((nblk + 2) * 8);
    b.meta = (uint32_t*)take(128);
    b.bytes = off;
    return b;
}

GsrBwdScratch gsr_carve_bwd(char* base, int P, int64_t R) {
    GsrBwdScratch b;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off = gsr_align128(off + bytes); return p; };
    const size_t np = (size_t)(P > 0 ? P : 1), nr = (size_t)(R > 0 ? R : 1);
    b.splat_grads = (float*)take(np * 48);
    b.inst_grads = (float*)take(nr * 48 * GSR_BWD_SLOTS);
    b.inst_flag = (uint32_t*)take(nr * 4);
    const size_t nu = gsr_reduce_units((int64_t)nr);
    b.unit_first = (uint2*)take(nu * 8);
    b.unit_piece = (float*)take(nu * 96);
    b.bytes = off;
    return b;
}

GsrImage gsr_carve_image(char* base, int W, int H) {
    GsrImage im;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off = gsr_align128(off + bytes); return p; };
    const size_t npix = (size_t)W * (size_t)H;
    const size_t nt = (size_t)((W + GSR_TILE - 1) / GSR_TILE) * (size_t)((H + GSR_TILE - 1) / GSR_TILE);
    im.final_T = (float*)take(npix * 4);
    im.n_contrib = (uint32_t*)take(npix * 4);
    im.ranges = (uint2*)take(nt * 8);
    im.block_steps = (uint32_t*)take(nt * 16);
    im.tile_order = (uint32_t*)take(nt * 8);      // one entry per half tile (the backward's launch order)
    im.bytes = off;
    return im;
}

extern "C" {

int gsr_abi_version(void) { return GSR_ABI_VERSION; }
const char* gsr_last_error(void) { return g_err.c_str(); }

size_t gsr_geometry_bytes(int P) { return gsr_carve_geom(nullptr, P).bytes; }
size_t gsr_binning_bytes(int64_t R, int n_tiles) { (void)n_tiles; return gsr_carve_binning(nullptr, R).bytes; }
size_t gsr_image_bytes(int width, int height) { return gsr_carve_image(nullptr, width, height).bytes; }
size_t gsr_backward_scratch_bytes(int P, int64_t R) { return gsr_carve_bwd(nullptr, P, R).bytes; }

int gsr_set_option(const char* name, int value) {
    if (!name) return fail(GSR_ERR_INVALID_ARG, "option name is NULL");
    if (!strcmp(name, "render_fwd_variant")) {
        if (!gsr_render_forward_variant_available(value)) return fail(GSR_ERR_UNSUPPORTED, "render_fwd_variant not in this build (A/B variants need -DGSR_AB_VARIANTS)");
        g_render_fwd_variant = value;
        return GSR_OK;
    }
    if (!strcmp(name, "render_bwd_variant")) {
        if (!gsr_render_backward_variant_available(value)) return fail(GSR_ERR_UNSUPPORTED, "render_bwd_variant not in this build (A/B variants need -DGSR_AB_VARIANTS)");
        g_render_bwd_variant = value;
        return GSR_OK;
    }
    if (!strcmp(name, "sort_small_block_threshold")) { g_small_block_threshold = value; return GSR_OK; }
    if (!strcmp(name, "sort_mid_block_threshold")) { g_mid_block_threshold = value; return GSR_OK; }
    if (!strcmp(name, "sort_items_large")) {
        if (value != 1024 && value != 2048 && value != 4096 && value != 8192)
            return fail(GSR_ERR_INVALID_ARG, "sort_items_large must be 1024, 2048, 4096 or 8192");
        g_sort_items_large = value;
        return GSR_OK;
    }
    if (!strcmp(name, "depth_sort_mode")) {
        if (value < 0 || value > 2) return fail(GSR_ERR_INVALID_ARG, "depth_sort_mode must be 0 (automatic), 1 (LSD radix passes) or 2 (bucket sort)");
        g_depth_sort_mode = value;
        return GSR_OK;
    }
    if (!strcmp(name, "preprocess_grid_cap")) { gsr_set_preprocess_grid_cap(value); return GSR_OK; }
    if (!strcmp(name, "ssim_variant")) {
        if (value != 0 && value != 1) return fail(GSR_ERR_INVALID_ARG, "ssim_variant must be 0 (marching waves) or 1 (LDS tiles)");
        gsr_set_ssim_variant(value);
        return GSR_OK;
    }
    if (!strcmp(name, "ssim_target_waves")) { gsr_set_ssim_target_waves(value); return GSR_OK; }
    if (!strcmp(name, "snug_tiles")) {
        if (value != 0 && value != 1) return fail(GSR_ERR_INVALID_ARG, "snug_tiles must be 1 (snug tile rectangles) or 0 (the reference's tile square)");
        g_snug_tiles = value;
        return GSR_OK;
    }
    if (!strcmp(name, "bwd_heavy_first")) {
        if (value < 0 || value > 3) return fail(GSR_ERR_INVALID_ARG, "bwd_heavy_first must be 0 (tiles in index order), 1 (heaviest tiles first), 2 (tiles by their heaviest half) or 3 (half tiles, heaviest first)");
        g_bwd_heavy_first = value;
        return GSR_OK;
    }
    if (!strcmp(name, "level2_scan_mode")) {
        if (value < 0 || value > 2) return fail(GSR_ERR_INVALID_ARG, "level2_scan_mode must be 0 (automatic), 1 (separate scan launch) or 2 (scan folded into the scatter)");
        gsr_set_level2_scan_mode(value);
        return GSR_OK;
    }
#ifdef GSR_AB_VARIANTS
    if (!strcmp(name, "render_fwd_lds_pad")) {
        if (value < 0 || value > 60000) return fail(GSR_ERR_INVALID_ARG, "render_fwd_lds_pad must be 0..60000 bytes");
        gsr_set_render_fwd_lds_pad(value);
        return GSR_OK;
    }
#endif
    if (!strcmp(name, "debug_dirty_control_block")) {      // test hook, one shot: see frame_stats_for
        if (value < 0 || value > 2047) return fail(GSR_ERR_INVALID_ARG, "debug_dirty_control_block: 0..2047 tickets");
        g_debug_dirty_tickets.store(value);
        return GSR_OK;
    }
    if (!strcmp(name, "tile_sort_mode")) {
        if (value != 0 && value != 1) return fail(GSR_ERR_INVALID_ARG, "tile_sort_mode must be 0 (fused) or 1 (legacy LSD)");
        g_tile_sort_mode = value;
        return GSR_OK;
    }
    return fail(GSR_ERR_INVALID_ARG, std::string("unknown option ") + name);
}

int gsr_profile_enable(int on) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    const bool trace = (on & 4) != 0;
    g_count_on = (on & 2) != 0 || trace;
    int d = 0;
    if (g_count_on && hipGetDevice(&d) == hipSuccess && d >= 0 && d < GSR_MAX_DEVICES) {
        if (!g_counters_dev[d]) {
            // measurement only (the product path never allocates): one block per device ordinal, on the CURRENT device -- the six work
            // counters, the mode word and the per-wave trace area (gsr_internal.h)
            if (hipMalloc((void**)&g_counters_dev[d], GSR_MEASURE_WORDS * sizeof(unsigned long long)) != hipSuccess) g_counters_dev[d] = nullptr;
            else (void)hipMemset(g_counters_dev[d], 0, GSR_MEASURE_WORDS * sizeof(unsigned long long));
        }
        if (g_counters_dev[d]) {
            const unsigned long long mode = trace ? 1ull : 0ull;
            (void)hipMemcpy(g_counters_dev[d] + GSR_TRACE_MODE_WORD, &mode, sizeof(mode), hipMemcpyHostToDevice);
            (void)hipStreamSynchronize(nullptr);      // (null-stream work: finished before a kernel of a non-blocking stream reads the word)
        }
    }
    g_prof_on = (on & 1) != 0;
    return GSR_OK;
}
int gsr_profile_trace(uint64_t* out, int max_waves) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    if (!out || max_waves <= 0) return 0;
    unsigned long long* ctr = counters_for_current_device();
    if (!ctr) return 0;
    const int n = max_waves < GSR_TRACE_WAVES ? max_waves : GSR_TRACE_WAVES;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(out, ctr + GSR_TRACE_BASE, (size_t)n * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    (void)hipMemset(ctr + GSR_TRACE_BASE, 0, (size_t)GSR_TRACE_WAVES * 4 * sizeof(unsigned long long));
    (void)hipStreamSynchronize(nullptr);      // (the clear is queued on the null stream: done before a kernel of a non-blocking stream traces again)
    return n;
}
int gsr_profile_counters(uint64_t* out, int n, int reset) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    unsigned long long host[GSR_COUNTER_COUNT] = {0};
    unsigned long long* ctr = counters_for_current_device();
    if (ctr) {
        if (hipDeviceSynchronize() != hipSuccess) return fail(GSR_ERR_HIP, "counter read-back failed");      // (kernels of every stream have counted)
        if (hipMemcpy(host, ctr, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return fail(GSR_ERR_HIP, "counter read-back failed");
        if (reset) { (void)hipMemset(ctr, 0, sizeof(host)); (void)hipStreamSynchronize(nullptr); }
    }
    for (int i = 0; i < n && i < GSR_COUNTER_COUNT; ++i) out[i] = host[i];
    return GSR_OK;
}
int gsr_profile_reset(void) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    for (auto& p : g_pending) { (void)hipEventSynchronize(p.b); g_pool.push_back(p.a); g_pool.push_back(p.b); }
    g_pending.clear();
    for (int i = 0; i < GSR_STAGE_COUNT; ++i) { g_stage_ms[i] = 0; g_stage_n[i] = 0; }
    return GSR_OK;
}
int gsr_profile_read(float* ms_out, int32_t* count_out, int n) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    for (auto& p : g_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            g_stage_ms[p.stage] += ms;
            g_stage_n[p.stage] += 1;
        }
        g_pool.push_back(p.a);
        g_pool.push_back(p.b);
    }
    g_pending.clear();
    for (int i = 0; i < n && i < GSR_STAGE_COUNT; ++i) {
        if (ms_out) ms_out[i] = (float)g_stage_ms[i];
        if (count_out) count_out[i] = g_stage_n[i];
    }
    return GSR_OK;
}

// (profiling only) an event pair that goes back to the pool unless it was handed to g_pending -- ADVICE r02: the early
// returns of bin_and_render leaked two events per failed frame
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    bool armed = false;
    void arm(hipStream_t st) {
        if (!g_prof_on) return;
        std::lock_guard<std::mutex> l(g_prof_mu);
        a = get_event(); b = get_event(); armed = true;
        (void)hipEventRecord(a, st);
    }
    void commit(int stage, hipStream_t st) {
        if (!armed) return;
        (void)hipEventRecord(b, st);
        std::lock_guard<std::mutex> l(g_prof_mu);
        g_pending.push_back({stage, a, b});
        armed = false;
    }
    ~EventPair() {
        if (!armed) return;
        std::lock_guard<std::mutex> l(g_prof_mu);
        g_pool.push_back(a); g_pool.push_back(b);
    }
};

// Which ping-pong buffer holds the depth order is a pure function of the build (3 passes of 9 bits -> vals[1]); the 32-bit
// fallback sort (4 passes -> vals[0]) copies its result there, so the backward finds it without any state.
static int depth_order_buffer_index() {
    int pb[8];
    return gsr_sort_plan(GSR_DEPTH_KEY_BITS, GSR_DEPTH_DIGIT_BITS, pb) & 1;
}

// Host side of the R read-back: spin on the mapped word.  ADVICE r02: in training the host runs far ahead of the GPU, so the
// wait can be the whole rest of the previous iteration -- a hard spin would pin a core per rank.  200 us of PAUSE spinning
// (covers the forward-only case, where the wait is the scan bubble), then polls separated by sched_yield(), after 2 s the
// blocking path (which also surfaces kernel faults).
static int wait_for_R(HostWord& hw, uint32_t seq, hipStream_t st, const uint32_t* num_rendered_dev) {
    volatile uint32_t* w = hw.host;
    bool got = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 0;; ++spin) {
        if (w[1] == seq) { got = true; break; }
        if ((spin & 63) == 63) {
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us > 2.0e6) break;
            if (us > 200.0) sched_yield();
        } else {
            GSR_CPU_RELAX();
        }
    }
    if (!got) {
        HIP_OK(hipStreamSynchronize(st));
        if (w[1] != seq) {
            uint32_t r[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // frame words: [0..1] R, [8] "a depth key overflowed", [9] sequence number (gsr_frame.h)
            HIP_OK(hipMemcpy(r, num_rendered_dev, sizeof(r), hipMemcpyDeviceToHost));
            if (r[9] != seq) {
                // the stream is idle and NO workgroup drew the last ticket: the slot's counter was not zero when the frame began (or lost
                // tickets on the way).  It is never handed on in that state, and the frame is refused instead of rendered with a made-up R
                HIP_OK(hipMemsetAsync(hw.state, 0, 64, st));
                HIP_OK(hipStreamSynchronize(st));
                return fail(GSR_ERR_HIP, "the frame statistics of this call were never published (control block reset; the call can be repeated)");
            }
            hw.host[0] = r[0];
            hw.host[2] = r[1];
            hw.host[3] = r[8] ? 1u : 0u;      // (the fallback recovers the key-overflow flag too, not only R)
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return GSR_OK;
}

// The statistics block / frame words / host word a key-producing kernel needs (gsr_frame.h)
static GsrFrameStatsDev frame_stats_for(HostWordLease& lease, GsrGeom& g, uint32_t& seq_out) {
    seq_out = ++lease.hw.seq;
    lease.settled = false;
    GsrFrameStatsDev fs;
    fs.state = lease.hw.state + 2u * (seq_out & 1u);               // two 64-bit counters per slot, used alternately (gsr_frame.h)
    fs.state_other = lease.hw.state + 2u * ((seq_out & 1u) ^ 1u);
    if (const int t = g_debug_dirty_tickets.exchange(0)) {       // test hook: this frame meets a counter that is not zero (tests/test_simt_abi_cpu.py)
        const unsigned long long dirt = ((unsigned long long)t << 53) | 5ull;
        (void)hipMemcpyAsync(fs.state, &dirt, sizeof(dirt), hipMemcpyHostToDevice, lease.st);
        (void)hipStreamSynchronize(lease.st);
    }
    fs.frame = g.num_rendered;
    fs.wg_range = g.wg_range;
    fs.host_word = lease.hw.dev;
    fs.seq = seq_out;
    return fs;
}

static bool use_bucket_sort(int P, int dev_id) {
    if (P > GSR_DS_MAX_P || g_depth_sort_mode == 1) return false;
    if (g_depth_sort_mode == 2) return true;
    if (g_lsd_frames[dev_id].load() > 0) {
        const int left = g_lsd_frames[dev_id].fetch_sub(1);
        if (left > 0) {                                               // (concurrent callers: one decrement each, never a lost update)
            if (left == 1) g_lsd_probation[dev_id].store(3);          // the stay ends with this frame: the next ones are on probation
            return false;
        }
        g_lsd_frames[dev_id].store(0);
    }
    return true;
}

// Everything after the key-producing kernel (per-Gaussian preprocess or splat ingest): depth sort + scan, emission, tile sort,
// ranges, blend.  `g` holds the splat records, band-clamped rectangles / tile counts and the depth keys of all P Gaussians;
// `seq` is the sequence number that kernel's last workgroup publishes with R.
static int bin_and_render(const GsrRasterSettings* settings, const GsrCamDev& cam, int P, GsrGeom& g, HostWordLease& lease, uint32_t seq,
                          int n_range /*workgroups of the key-producing kernel*/, GsrResizeFn binning_resize, void* binning_user, GsrResizeFn image_resize, void* image_user,
                          float* out_color, float* out_invdepth, int32_t* num_rendered, hipStream_t st) {
    const int dev_id = lease.dev;
    HostWord& hw_slot = lease.hw;
    const int order_buf = depth_order_buffer_index();
    const uint32_t bf_cap = (uint32_t)gsr_block_first_cap(P);
    const bool bucket = use_bucket_sort(P, dev_id);
    // R = number of (Gaussian, tile) instances sizes the binning buffer and the emission grids, so the host must learn it
    // mid-pipeline (the reference has the same read-back).  It does not depend on the depth order: the key-producing kernel
    // has already summed it, and its last workgroup stores it with a sequence number straight into mapped pinned host memory
    // (gsr_frame.h).  The host queues the whole depth sort FIRST and only then looks at the word, so the read-back, the
    // callbacks and the emission launches all happen while the GPU sorts -- no idle bubble (rounds 1-3: R came from the
    // scan, after the sort, and the GPU idled 5-7 us per frame).
    {   StageTimer t(GSR_STAGE_DEPTH_SORT, st);
        if (bucket) {
            gsr_launch_depth_bucket_sort(P, g.keys[0], g.tiles, g.rect, g.num_rendered, g.wg_range, n_range, g.ds, g.vals[order_buf], g.rect_sorted, g.offsets,
                                         g.block_first, bf_cap, slow_word_dev(dev_id), st);
        } else {
            const int ob = gsr_radix_sort_pairs(g.keys, g.vals, P, GSR_DEPTH_KEY_BITS, GSR_DEPTH_DIGIT_BITS, g.sort_hist, g.digit_total,
                                                sort_items(P), st, g.rect, g.rect_sorted);
            if (ob != order_buf) return fail(GSR_ERR_HIP, "depth sort: unexpected result buffer");
        }
    }
    STAGE_CHECK("depth sort");
    if (!bucket) {
        StageTimer t(GSR_STAGE_SCAN, st);
        gsr_launch_scan_tiles(P, g.vals[order_buf], g.rect, g.rect_sorted, g.offsets, g.block_sums, g.block_first, bf_cap,
                              g.num_rendered + 4, nullptr, 0u, /*rect_already_sorted=*/true, st);
    }
    // (profiling only) what is left of the read-back on the GPU's time line: from the end of the depth sort / scan to the
    // first launch after the host has R
    EventPair wait_ev;
    wait_ev.arm(st);
    const int n_tiles = cam.gx * cam.gy;
    GsrTileSortPlan plan;
    gsr_tile_sort_plan(n_tiles, P, &plan);
    if (g_tile_sort_mode == 1) plan.fused = false;
    char* ibase = (char*)image_resize(image_user, gsr_image_bytes(cam.W, cam.H));
    // legacy path: the tile-range table is cleared here; the fused path writes every entry of the table itself
    if (ibase && !plan.fused)
        HIP_OK(hipMemsetAsync(gsr_carve_image(ibase, cam.W, cam.H).ranges, 0, sizeof(uint2) * (size_t)n_tiles, st));
    char* bbase = nullptr;
    size_t spec_bytes = 0;
    const int64_t last_R = g_last_R[dev_id].load();
    if (last_R > 0) {
        spec_bytes = gsr_binning_bytes(last_R + last_R / 4 + 4096, n_tiles);
        bbase = (char*)binning_resize(binning_user, spec_bytes);
    }
    int rc = wait_for_R(hw_slot, seq, st, g.num_rendered);
    if (rc != GSR_OK) return rc;
    lease.settled = true;
    if (hw_slot.host[3] != 0u) {
        // a listed Gaussian lies deeper than 0.2 * 2^16: its 27-bit key was clamped and the depth order above is not
        // trustworthy.  Rare path: full 32-bit keys from the splat records, 4 passes of 8 bits (round 2's sort), result
        // copied into the buffer the backward reads, scan again (R itself does not depend on the order).
        gsr_launch_rekey_full(P, g.splats, g.tiles, g.keys[0], g.vals[0], st);
        const int ob = gsr_radix_sort_pairs(g.keys, g.vals, P, 32, 8, g.sort_hist, g.digit_total, sort_items(P), st, g.rect, g.rect_sorted);
        if (ob != order_buf) HIP_OK(hipMemcpyAsync(g.vals[order_buf], g.vals[ob], (size_t)P * 4, hipMemcpyDeviceToDevice, st));
        gsr_launch_scan_tiles(P, g.vals[order_buf], g.rect, g.rect_sorted, g.offsets, g.block_sums, g.block_first, bf_cap,
                              g.num_rendered + 4, nullptr, 0u, true, st);
    }
    const uint64_t R64 = ((uint64_t)hw_slot.host[2] << 32) | (uint64_t)hw_slot.host[0];
    if (R64 > 0x7FFFFFFFull) return fail(GSR_ERR_UNSUPPORTED, "more than 2^31-1 tile instances");
    const int64_t R = (int64_t)R64;
    *num_rendered = (int32_t)R;
    g_last_R[dev_id].store(R);
    const size_t need_bytes = gsr_binning_bytes(R, n_tiles);
    if (!bbase || need_bytes > spec_bytes) bbase = (char*)binning_resize(binning_user, need_bytes);
    if (!bbase || !ibase) return fail(GSR_ERR_ALLOC, "binning / image buffer resize returned NULL");
    GsrBinning b = gsr_carve_binning(bbase, R);
    GsrImage im = gsr_carve_image(ibase, cam.W, cam.H);
    wait_ev.commit(GSR_STAGE_R_WAIT, st);
    float4* goffset_splats = settings->no_backward ? nullptr : g.splats;

    int list_buf = 0;
    if (R > 0 && plan.fused) {
        const uint2* block_first = g.block_first;
        const int64_t nblk = (R + GSR_TS_ITEMS - 1) / GSR_TS_ITEMS;
        {   StageTimer t(GSR_STAGE_EMIT, st);
            if ((uint64_t)nblk + 1 > (uint64_t)bf_cap) {      // more than ~64 tiles per Gaussian: table sized by R instead
                gsr_launch_fill_block_first(P, g.offsets, b.block_first, (uint32_t)(nblk + 2), st);
                block_first = b.block_first;
            }
            gsr_launch_tile_sort_level1(plan, R, cam.gx, block_first, g.offsets, g.rect_sorted, g.vals[order_buf], b.keys[0],
                                        b.sort_hist, b.digit_total, b.bucket_base, b.blk2_start, goffset_splats, st);
        }
        STAGE_CHECK("emit + level-1 sort");
        {   StageTimer t(GSR_STAGE_TILE_SORT, st);
            gsr_launch_tile_sort_level2(plan, R, n_tiles, b.keys[0], b.vals[0], b.bucket_base, b.blk2_start, b.hist2, b.tile_base,
                                        im.ranges, st);
        }
        STAGE_CHECK("level-2 sort + ranges");
    } else if (R > 0) {
        const bool key16 = n_tiles <= 65536;     // (only reachable through the A/B option: <= 65536 tiles normally take the fused path)
        {   StageTimer t(GSR_STAGE_EMIT, st);
            gsr_launch_emit(P, cam.gx, g.vals[order_buf], g.offsets, g.rect_sorted, b.keys[0], key16, b.vals[0], goffset_splats, st);
        }
        STAGE_CHECK("emit");
        {   StageTimer t(GSR_STAGE_TILE_SORT, st);
            if (key16) {
                uint16_t* k16[2] = {(uint16_t*)b.keys[0], (uint16_t*)b.keys[1]};
                // always two passes (zero high bits sort harmlessly), so that the list lands in vals[0] like the fused path's
                const int kb = bits_for((uint32_t)n_tiles) < 9 ? 9 : bits_for((uint32_t)n_tiles);
                list_buf = gsr_radix_sort_pairs_k16(k16, b.vals, R, kb, GSR_TILE_DIGIT_BITS, b.sort_hist,
                                                    b.digit_total, sort_items(R), st);
            } else {
                list_buf = gsr_radix_sort_pairs(b.keys, b.vals, R, bits_for((uint32_t)n_tiles), GSR_TILE_DIGIT_BITS, b.sort_hist,
                                                b.digit_total, sort_items(R), st);
            }
        }
        STAGE_CHECK("tile sort");
        {   StageTimer t(GSR_STAGE_RANGES, st);
            gsr_launch_ranges(R, n_tiles, b.keys[list_buf], key16, im.ranges, /*already_zeroed=*/true, st);
        }
        STAGE_CHECK("ranges");
    } else if (plan.fused) {
        HIP_OK(hipMemsetAsync(im.ranges, 0, sizeof(uint2) * (size_t)n_tiles, st));
    }
    {   StageTimer t(GSR_STAGE_RENDER, st);
        gsr_launch_render_forward(cam, im.ranges, b.vals[list_buf], g.splats, settings->no_backward ? nullptr : im.final_T,
                                  settings->no_backward ? nullptr : im.n_contrib, settings->no_backward ? nullptr : im.block_steps,
                                  out_color, out_invdepth, g_render_fwd_variant, g_count_on ? counters_for_current_device() : nullptr, st);
    }
    STAGE_CHECK("render");
    HIP_OK(hipGetLastError());
    return GSR_OK;
}


int gsr_rasterize_forward(const GsrRasterSettings* settings, int P, int M, const float* means3D, const float* shs,
                          const float* colors_precomp, const float* opacities, const float* scales,
                          const float* rotations, const float* cov3D_precomp, GsrResizeFn geom_resize, void* geom_user,
                          GsrResizeFn binning_resize, void* binning_user, GsrResizeFn image_resize, void* image_user,
                          float* out_color, float* out_invdepth, int32_t* radii, int32_t* num_rendered, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    GsrCamDev cam;
    int rc = make_cam(settings, M, cam);
    if (rc != GSR_OK) return rc;
    rc = check_inputs(P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, cam.sh_degree);
    if (rc != GSR_OK) return rc;
    rc = check_split_sh(settings, P, M, shs, false);
    if (rc != GSR_OK) return rc;
    if (!out_color || !num_rendered) return fail(GSR_ERR_INVALID_ARG, "out_color / num_rendered are NULL");
    const size_t npix = (size_t)cam.W * cam.H;
    *num_rendered = 0;
    if (P == 0) {   // reference behaviour: zero image (not background), nothing else touched
        HIP_OK(hipMemsetAsync(out_color, 0, npix * 3 * sizeof(float), st));
        if (out_invdepth) HIP_OK(hipMemsetAsync(out_invdepth, 0, npix * sizeof(float), st));
        return GSR_OK;
    }
    if (!radii) return fail(GSR_ERR_INVALID_ARG, "radii is NULL");
    if (!geom_resize || !binning_resize || !image_resize) return fail(GSR_ERR_INVALID_ARG, "resize callbacks are NULL");

    char* gbase = (char*)geom_resize(geom_user, gsr_geometry_bytes(P));
    if (!gbase) return fail(GSR_ERR_ALLOC, "geometry buffer resize returned NULL");
    GsrGeom g = gsr_carve_geom(gbase, P);
    HostWordLease lease;
    rc = lease_host_word(lease, st);
    if (rc != GSR_OK) return rc;
    uint32_t seq;
    const GsrFrameStatsDev fs = frame_stats_for(lease, g, seq);
    int n_range;
    {   StageTimer t(GSR_STAGE_PREPROCESS, st);
        n_range = gsr_launch_preprocess(cam, P, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, g, radii, fs, st);
    }
    STAGE_CHECK("preprocess");
    return bin_and_render(settings, cam, P, g, lease, seq, n_range, binning_resize, binning_user, image_resize, image_user, out_color, out_invdepth,
                          num_rendered, st);
}

int gsr_preprocess_forward(const GsrRasterSettings* settings, int P, int M, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                           const float* cov3D_precomp, void* geom_scratch, int32_t* radii, float* splat_records,
                           void* stream) {
    hipStream_t st = (hipStream_t)stream;
    GsrCamDev cam;
    int rc = make_cam(settings, M, cam);
    if (rc != GSR_OK) return rc;
    rc = check_inputs(P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, cam.sh_degree);
    if (rc != GSR_OK) return rc;
    rc = check_split_sh(settings, P, M, shs, false);
    if (rc != GSR_OK) return rc;
    if (P == 0) return GSR_OK;
    if (!geom_scratch || !radii || !splat_records) return fail(GSR_ERR_INVALID_ARG, "geom_scratch / radii / splat_records are NULL");
    if ((uintptr_t)splat_records & 15) return fail(GSR_ERR_INVALID_ARG, "splat_records must be 16-byte aligned");
    // the records leave this rank: rectangles and tile counts are those of the FULL frame, whatever band the settings name
    cam.tile_y0 = 0;
    cam.tile_y1 = cam.gy;
    GsrGeom g = gsr_carve_geom((char*)geom_scratch, P);
    g.splats = reinterpret_cast<float4*>(splat_records);
    {   StageTimer t(GSR_STAGE_PREPROCESS, st);
        GsrFrameStatsDev none;      // nothing is binned here: no frame statistics
        none.state = nullptr; none.frame = nullptr; none.wg_range = nullptr; none.host_word = nullptr; none.seq = 0;
        gsr_launch_preprocess(cam, P, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, g, radii, none, st);
    }
    STAGE_CHECK("preprocess (shard)");
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

// shared body of gsr_rasterize_from_splats (64-byte records) / gsr_rasterize_from_packed (48-byte records)
static int rasterize_from_records(const GsrRasterSettings* settings, int P, const float* records, bool packed, int seg_rows, GsrResizeFn geom_resize,
                                  void* geom_user, GsrResizeFn binning_resize, void* binning_user, GsrResizeFn image_resize,
                                  void* image_user, float* out_color, float* out_invdepth, int32_t* num_rendered, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    GsrCamDev cam;
    int rc = make_cam(settings, 0, cam);
    if (rc != GSR_OK) return rc;
    if (P < 0) return fail(GSR_ERR_INVALID_ARG, "P < 0");
    if (!out_color || !num_rendered) return fail(GSR_ERR_INVALID_ARG, "out_color / num_rendered are NULL");
    const size_t npix = (size_t)cam.W * cam.H;
    *num_rendered = 0;
    if (P == 0) {
        // No record reached this rank's band while the scene itself is not empty (the caller renders a band of a sharded frame):
        // the band's pixels are what the blend leaves where no splat lands -- the BACKGROUND (T = 1), not the zero image of the
        // reference's P == 0 early-out (with a white or random background the gathered frame, and the loss, differed
        // from one GPU's).  The blend runs over empty tile ranges.
        HIP_OK(hipMemsetAsync(out_color, 0, npix * 3 * sizeof(float), st));
        if (out_invdepth) HIP_OK(hipMemsetAsync(out_invdepth, 0, npix * sizeof(float), st));
        if (!image_resize) return fail(GSR_ERR_INVALID_ARG, "resize callbacks are NULL");
        char* ibase = (char*)image_resize(image_user, gsr_image_bytes(cam.W, cam.H));
        if (!ibase) return fail(GSR_ERR_ALLOC, "image buffer resize returned NULL");
        GsrImage im = gsr_carve_image(ibase, cam.W, cam.H);
        HIP_OK(hipMemsetAsync(im.ranges, 0, sizeof(uint2) * (size_t)cam.gx * cam.gy, st));
        const bool nb = settings->no_backward != 0;
        gsr_launch_render_forward(cam, im.ranges, nullptr, nullptr, nb ? nullptr : im.final_T, nb ? nullptr : im.n_contrib,
                                  nb ? nullptr : im.block_steps, out_color, out_invdepth, 0, nullptr, st);
        HIP_OK(hipGetLastError());
        return GSR_OK;
    }
    if (!records) return fail(GSR_ERR_INVALID_ARG, "records pointer is NULL");
    if ((uintptr_t)records & 15) return fail(GSR_ERR_INVALID_ARG, "records must be 16-byte aligned");
    if (!geom_resize || !binning_resize || !image_resize) return fail(GSR_ERR_INVALID_ARG, "resize callbacks are NULL");
    char* gbase = (char*)geom_resize(geom_user, gsr_geometry_bytes(P));
    if (!gbase) return fail(GSR_ERR_ALLOC, "geometry buffer resize returned NULL");
    GsrGeom g = gsr_carve_geom(gbase, P);
    HostWordLease lease;
    rc = lease_host_word(lease, st);
    if (rc != GSR_OK) return rc;
    uint32_t seq;
    const GsrFrameStatsDev fs = frame_stats_for(lease, g, seq);
    int n_range;
    {   StageTimer t(GSR_STAGE_PREPROCESS, st);
        if (packed) n_range = gsr_launch_ingest_packed(P, records, cam.tile_y0, cam.tile_y1, g.splats, g.rect, g.tiles, g.keys[0], g.vals[0], fs, seg_rows, st);
        else n_range = gsr_launch_splat_ingest(P, records, cam.tile_y0, cam.tile_y1, g.splats, g.rect, g.tiles, g.keys[0], g.vals[0], fs, st);
    }
    STAGE_CHECK("splat ingest");
    return bin_and_render(settings, cam, P, g, lease, seq, n_range, binning_resize, binning_user, image_resize, image_user, out_color, out_invdepth,
                          num_rendered, st);
}

int gsr_rasterize_from_splats(const GsrRasterSettings* settings, int P, const float* splat_records, GsrResizeFn geom_resize,
                              void* geom_user, GsrResizeFn binning_resize, void* binning_user, GsrResizeFn image_resize,
                              void* image_user, float* out_color, float* out_invdepth, int32_t* num_rendered, void* stream) {
    return rasterize_from_records(settings, P, splat_records, false, 0, geom_resize, geom_user, binning_resize, binning_user, image_resize,
                                  image_user, out_color, out_invdepth, num_rendered, stream);
}

int gsr_rasterize_from_packed(const GsrRasterSettings* settings, int P, const float* packed_records, GsrResizeFn geom_resize,
                              void* geom_user, GsrResizeFn binning_resize, void* binning_user, GsrResizeFn image_resize,
                              void* image_user, float* out_color, float* out_invdepth, int32_t* num_rendered, void* stream) {
    return rasterize_from_records(settings, P, packed_records, true, 0, geom_resize, geom_user, binning_resize, binning_user, image_resize,
                                  image_user, out_color, out_invdepth, num_rendered, stream);
}

int gsr_rasterize_from_segments(const GsrRasterSettings* settings, int n_segments, int capacity, const float* segments,
                                GsrResizeFn geom_resize, void* geom_user, GsrResizeFn binning_resize, void* binning_user,
                                GsrResizeFn image_resize, void* image_user, float* out_color, float* out_invdepth,
                                int32_t* num_rendered, void* stream) {
    if (n_segments < 1 || n_segments > GSR_MAX_BANDS) return fail(GSR_ERR_INVALID_ARG, "n_segments must be 1..64");
    if (capacity < 1) return fail(GSR_ERR_INVALID_ARG, "capacity < 1");
    const int64_t rows = (int64_t)n_segments * ((int64_t)capacity + 1);
    if (rows > 0x7FFFFFFFll) return fail(GSR_ERR_UNSUPPORTED, "n_segments * (capacity + 1) exceeds 2^31-1 rows");
    return rasterize_from_records(settings, (int)rows, segments, true, capacity + 1, geom_resize, geom_user, binning_resize, binning_user,
                                  image_resize, image_user, out_color, out_invdepth, num_rendered, stream);
}

// ---- Gaussian-sharded exchange (route.hip) ----
static int check_bands(int n_bands, const int32_t* band_bounds) {
    if (n_bands < 1 || n_bands > GSR_MAX_BANDS) return fail(GSR_ERR_UNSUPPORTED, "n_bands must be 1..64");
    if (!band_bounds) return fail(GSR_ERR_INVALID_ARG, "band_bounds is NULL");
    for (int b = 0; b < n_bands; ++b)
        if (band_bounds[b] > band_bounds[b + 1] || band_bounds[b] < 0) return fail(GSR_ERR_INVALID_ARG, "band_bounds must be non-negative and non-decreasing");
    return GSR_OK;
}

size_t gsr_route_scratch_bytes(int P, int n_bands) { return gsr_route_scratch_bytes_impl(P, n_bands < 1 ? 1 : n_bands); }

int gsr_route_count(int P, const float* splat_records, int n_bands, const int32_t* band_bounds, void* scratch, uint32_t* band_counts,
                    void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int rc = check_bands(n_bands, band_bounds);
    if (rc != GSR_OK) return rc;
    if (P < 0) return fail(GSR_ERR_INVALID_ARG, "P < 0");
    if (!band_counts) return fail(GSR_ERR_INVALID_ARG, "band_counts is NULL");
    if (P == 0) { HIP_OK(hipMemsetAsync(band_counts, 0, sizeof(uint32_t) * (size_t)n_bands, st)); return GSR_OK; }
    if (!splat_records || !scratch) return fail(GSR_ERR_INVALID_ARG, "splat_records / scratch are NULL");
    if ((uintptr_t)splat_records & 15) return fail(GSR_ERR_INVALID_ARG, "splat_records must be 16-byte aligned");
    gsr_launch_route_count(P, splat_records, n_bands, band_bounds, (uint32_t*)scratch, band_counts, st);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_route_pack(int P, const float* splat_records, int n_bands, const int32_t* band_bounds, const int64_t* band_offsets,
                   const void* scratch, float* packed, int32_t* send_ids, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int rc = check_bands(n_bands, band_bounds);
    if (rc != GSR_OK) return rc;
    if (P < 0) return fail(GSR_ERR_INVALID_ARG, "P < 0");
    if (P == 0) return GSR_OK;
    if (!band_offsets) return fail(GSR_ERR_INVALID_ARG, "band_offsets is NULL");
    if (band_offsets[n_bands] == band_offsets[0]) return GSR_OK;      // nothing to send
    if (!splat_records || !scratch || !packed || !send_ids) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    if (((uintptr_t)splat_records | (uintptr_t)packed) & 15) return fail(GSR_ERR_INVALID_ARG, "splat_records / packed must be 16-byte aligned");
    gsr_launch_route_pack(P, splat_records, n_bands, band_bounds, band_offsets, (const uint32_t*)scratch, packed, send_ids, 0xFFFFFFFFu,
                          nullptr, st);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_route_pack_fixed(int P, const float* splat_records, int n_bands, const int32_t* band_bounds, int capacity, const void* scratch,
                         const uint32_t* band_counts, float* packed, int32_t* send_ids, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int rc = check_bands(n_bands, band_bounds);
    if (rc != GSR_OK) return rc;
    if (P < 0 || capacity < 1) return fail(GSR_ERR_INVALID_ARG, "P < 0 or capacity < 1");
    if (!packed || !send_ids) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    if ((uintptr_t)packed & 15) return fail(GSR_ERR_INVALID_ARG, "packed must be 16-byte aligned");
    const size_t seg_rows = (size_t)capacity + 1;
    // unused rows and header rows carry the id -1 (gsr_route_return skips them)
    HIP_OK(hipMemsetAsync(send_ids, 0xFF, sizeof(int32_t) * seg_rows * (size_t)n_bands, st));
    if (P == 0) {      // an empty shard still sends its (zero) headers
        for (int b = 0; b < n_bands; ++b) HIP_OK(hipMemsetAsync(packed + (size_t)b * seg_rows * 12, 0, 48, st));
        return GSR_OK;
    }
    if (!splat_records || !scratch || !band_counts) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    if ((uintptr_t)splat_records & 15) return fail(GSR_ERR_INVALID_ARG, "splat_records must be 16-byte aligned");
    int64_t off[GSR_MAX_BANDS + 1];
    for (int b = 0; b <= n_bands; ++b) off[b] = (int64_t)b * (int64_t)seg_rows + 1;      // first record row of segment b
    gsr_launch_route_pack(P, splat_records, n_bands, band_bounds, off, (const uint32_t*)scratch, packed, send_ids, (uint32_t)capacity,
                          band_counts, st);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_route_return(int P, int n_bands, const int64_t* band_offsets, const int32_t* send_ids, const float* returned,
                     float* splat_grads, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (P < 0 || n_bands < 1 || n_bands > GSR_MAX_BANDS) return fail(GSR_ERR_INVALID_ARG, "bad P / n_bands");
    if (P == 0) return GSR_OK;
    if (!band_offsets || !splat_grads) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    HIP_OK(hipMemsetAsync(splat_grads, 0, (size_t)P * 48, st));
    for (int b = 0; b < n_bands; ++b) {      // band order: fixed association order of the <= n_bands terms per Gaussian
        const int64_t n = band_offsets[b + 1] - band_offsets[b];
        if (n < 0) return fail(GSR_ERR_INVALID_ARG, "band_offsets must be non-decreasing");
        if (n == 0) continue;
        if (!send_ids || !returned) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
        gsr_launch_route_add_rows(n, send_ids + band_offsets[b], returned + band_offsets[b] * 12, splat_grads, st);
    }
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

// Which ping-pong buffer holds the sorted list / the depth order is a pure function of the frame geometry (never of a
// run-time option), so the backward finds the forward's results without any state passed between the two calls:
// <= 65536 tiles: vals[0] (fused sort writes it; the legacy LSD sort takes 2 passes); more: 3 LSD passes -> vals[1].
static int list_buffer_index(int n_tiles) {
    if (n_tiles <= 65536) return 0;
    int pb[8];
    return gsr_sort_plan(bits_for((uint32_t)n_tiles), GSR_TILE_DIGIT_BITS, pb) & 1;
}

int gsr_backward_blend(const GsrRasterSettings* settings, int P, int32_t num_rendered, const void* geom_buffer,
                       const void* binning_buffer, const void* image_buffer, const float* dL_dout_color,
                       const float* dL_dout_invdepth, void* bwd_scratch, float** splat_grads_out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    GsrCamDev cam;
    int rc = make_cam(settings, 0, cam);
    if (rc != GSR_OK) return rc;
    if (P < 0) return fail(GSR_ERR_INVALID_ARG, "P < 0");
    if (P == 0) return GSR_OK;
    if (!geom_buffer || !binning_buffer || !image_buffer || !dL_dout_color || !bwd_scratch)
        return fail(GSR_ERR_INVALID_ARG, "state buffers / dL_dout_color / bwd_scratch are NULL");
    GsrGeom g = gsr_carve_geom((char*)geom_buffer, P);
    GsrBinning b = gsr_carve_binning((char*)binning_buffer, num_rendered);
    GsrImage im = gsr_carve_image((char*)image_buffer, cam.W, cam.H);
    const int list_buf = num_rendered > 0 ? list_buffer_index(cam.gx * cam.gy) : 0;
    GsrBwdScratch w = gsr_carve_bwd((char*)bwd_scratch, P, num_rendered);
    float* sg = w.splat_grads;
    if (splat_grads_out) *splat_grads_out = sg;
    {   StageTimer t(GSR_STAGE_RENDER_BWD, st);
        if (g_render_bwd_variant == 1 || num_rendered <= 0) {
            HIP_OK(hipMemsetAsync(sg, 0, (size_t)P * 12 * sizeof(float), st));
            if (num_rendered > 0)
                gsr_launch_render_backward(cam, im.ranges, b.vals[list_buf], g.splats, im.final_T, im.n_contrib, im.block_steps, nullptr,
                                           dL_dout_color, dL_dout_invdepth, sg, nullptr, nullptr, num_rendered, 1, 0, nullptr, st);
        } else {
            // (the flag words of the instances are cleared by the launcher: in the plan kernel's launch, or with a fill)
            gsr_launch_render_backward(cam, im.ranges, b.vals[list_buf], g.splats, im.final_T, im.n_contrib, im.block_steps,
                                       g_bwd_heavy_first ? im.tile_order : nullptr,
                                       dL_dout_color, dL_dout_invdepth, nullptr, w.inst_grads, w.inst_flag, num_rendered,
                                       g_render_bwd_variant, g_bwd_heavy_first, g_count_on ? counters_for_current_device() : nullptr, st);
        }
    }
    STAGE_CHECK("render backward blend");
    if (g_render_bwd_variant != 1 && num_rendered > 0) {
        StageTimer t(GSR_STAGE_GATHER_BWD, st);
        gsr_launch_reduce_instances(P, num_rendered, g.vals[depth_order_buffer_index()], g.offsets, g.splats, w.inst_grads,
                                    w.inst_flag, sg, w.unit_first, w.unit_piece, st);
    }
    STAGE_CHECK("render backward reduce");
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_backward_preprocess(const GsrRasterSettings* settings, int P, int M, const float* means3D, const float* shs,
                            const float* colors_precomp, const float* opacities, const float* scales,
                            const float* rotations, const float* cov3D_precomp, const int32_t* radii,
                            const void* geom_buffer, const float* splat_grads, float* dL_dmeans2D, float* dL_dcolors,
                            float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales,
                            float* dL_drotations, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    GsrCamDev cam;
    int rc = make_cam(settings, M, cam);
    if (rc != GSR_OK) return rc;
    rc = check_inputs(P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, cam.sh_degree);
    if (rc != GSR_OK) return rc;
    rc = check_split_sh(settings, P, M, shs, true);
    if (rc != GSR_OK) return rc;
    if (P == 0) return GSR_OK;
    (void)geom_buffer;      // the colour clamp bits are recomputed; a NULL geometry buffer is accepted (Gaussian-sharded backward)
    if (!radii || !splat_grads) return fail(GSR_ERR_INVALID_ARG, "radii / splat_grads are NULL");
    if (!dL_dmeans2D || !dL_dopacity || !dL_dmeans3D) return fail(GSR_ERR_INVALID_ARG, "gradient outputs are NULL");
    if (colors_precomp && !dL_dcolors) return fail(GSR_ERR_INVALID_ARG, "dL_dcolors is NULL");
    if (cov3D_precomp && !dL_dcov3D) return fail(GSR_ERR_INVALID_ARG, "dL_dcov3D is NULL");
    if (shs && !dL_dsh) return fail(GSR_ERR_INVALID_ARG, "dL_dsh is NULL");
    if (settings->sh_dc && ((((uintptr_t)dL_dsh) | ((uintptr_t)settings->dL_dsh_dc)) & 15))
        return fail(GSR_ERR_UNSUPPORTED, "split SH form needs 16-byte aligned dL_dsh / dL_dsh_dc");
    if (scales && (!dL_dscales || !dL_drotations)) return fail(GSR_ERR_INVALID_ARG, "dL_dscales / dL_drotations are NULL");
    GsrGeom g = gsr_carve_geom(geom_buffer ? (char*)geom_buffer : nullptr, P);
    {   StageTimer t(GSR_STAGE_PREPROCESS_BWD, st);
        gsr_launch_preprocess_backward(cam, P, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                       radii, g, splat_grads, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D,
                                       dL_dsh, scales ? dL_dscales : nullptr, scales ? dL_drotations : nullptr, st);
    }
    STAGE_CHECK("preprocess backward");
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_backward_preprocess_sh_adam(const GsrRasterSettings* settings, int P, int M, const float* means3D, float* shs_rest,
                                    const float* opacities, const float* scales, const float* rotations,
                                    const float* cov3D_precomp, const int32_t* radii, const void* geom_buffer,
                                    const float* splat_grads, float* dL_dmeans2D, float* dL_dopacity, float* dL_dmeans3D,
                                    float* dL_dcov3D, float* dL_dscales, float* dL_drotations, const GsrShAdam* adam,
                                    void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!settings || !adam) return fail(GSR_ERR_INVALID_ARG, "settings / adam are NULL");
    GsrCamDev cam;
    int rc = make_cam(settings, M, cam);
    if (rc != GSR_OK) return rc;
    rc = check_inputs(P, M, means3D, shs_rest, nullptr, opacities, scales, rotations, cov3D_precomp, cam.sh_degree);
    if (rc != GSR_OK) return rc;
    rc = check_split_sh(settings, P, M, shs_rest, false);
    if (rc != GSR_OK) return rc;
    if (M != 16 || !settings->sh_dc) return fail(GSR_ERR_UNSUPPORTED, "the fused SH Adam step needs the split SH form with M = 16 (sh_dc + 15 coefficients)");
    if (P == 0) return GSR_OK;
    if (!radii || !splat_grads) return fail(GSR_ERR_INVALID_ARG, "radii / splat_grads are NULL");
    if (!dL_dmeans2D || !dL_dopacity || !dL_dmeans3D) return fail(GSR_ERR_INVALID_ARG, "gradient outputs are NULL");
    if (cov3D_precomp && !dL_dcov3D) return fail(GSR_ERR_INVALID_ARG, "dL_dcov3D is NULL");
    if (scales && (!dL_dscales || !dL_drotations)) return fail(GSR_ERR_INVALID_ARG, "dL_dscales / dL_drotations are NULL");
    if (!adam->dc_exp_avg || !adam->dc_exp_avg_sq || !adam->rest_exp_avg || !adam->rest_exp_avg_sq)
        return fail(GSR_ERR_INVALID_ARG, "Adam moments are NULL");
    if (((uintptr_t)settings->sh_dc | (uintptr_t)shs_rest | (uintptr_t)adam->dc_exp_avg | (uintptr_t)adam->dc_exp_avg_sq |
         (uintptr_t)adam->rest_exp_avg | (uintptr_t)adam->rest_exp_avg_sq) & 15)
        return fail(GSR_ERR_UNSUPPORTED, "the fused SH Adam step needs 16-byte aligned SH tensors and moments");
    if (!adam->sparse && (adam->step_dc < 1 || adam->step_rest < 1)) return fail(GSR_ERR_INVALID_ARG, "step < 1");
    GsrShAdamDev d;
    d.dc = const_cast<float*>(settings->sh_dc); d.dc_m = adam->dc_exp_avg; d.dc_v = adam->dc_exp_avg_sq;
    d.rest = shs_rest; d.rest_m = adam->rest_exp_avg; d.rest_v = adam->rest_exp_avg_sq;
    d.sparse = adam->sparse ? 1 : 0;
    auto fill = [&](float (&a)[6], double lr, int step) {
        if (d.sparse) {      // lr, b1, 1 - b1, b2, 1 - b2, eps  (gsr_launch_sparse_adam)
            a[0] = (float)lr; a[1] = (float)adam->beta1; a[2] = (float)(1.0 - adam->beta1); a[3] = (float)adam->beta2;
            a[4] = (float)(1.0 - adam->beta2); a[5] = (float)adam->eps;
        } else {             // 1 - b1, b2, 1 - b2, step size, 1 / sqrt(1 - b2^t), eps  (gsr_launch_adam: doubles until the last moment)
            const double bc1 = 1.0 - pow(adam->beta1, (double)step), bc2 = 1.0 - pow(adam->beta2, (double)step);
            a[0] = (float)(1.0 - adam->beta1); a[1] = (float)adam->beta2; a[2] = (float)(1.0 - adam->beta2);
            a[3] = (float)(lr / bc1); a[4] = (float)(1.0 / sqrt(bc2)); a[5] = (float)adam->eps;
        }
    };
    fill(d.dc_a, adam->lr_dc, adam->step_dc);
    fill(d.rest_a, adam->lr_rest, adam->step_rest);
    GsrGeom g = gsr_carve_geom(geom_buffer ? (char*)geom_buffer : nullptr, P);
    {   StageTimer t(GSR_STAGE_PREPROCESS_BWD, st);
        gsr_launch_preprocess_backward_sh_adam(cam, P, means3D, opacities, scales, rotations, cov3D_precomp, radii, g, splat_grads,
                                               dL_dmeans2D, dL_dopacity, dL_dmeans3D, dL_dcov3D, scales ? dL_dscales : nullptr,
                                               scales ? dL_drotations : nullptr, d, st);
    }
    STAGE_CHECK("preprocess backward (fused SH Adam)");
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_rasterize_backward(const GsrRasterSettings* settings, int P, int M, int32_t num_rendered, const float* means3D,
                           const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, const float* cov3D_precomp, const int32_t* radii,
                           const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                           const float* dL_dout_color, const float* dL_dout_invdepth, float* dL_dmeans2D,
                           float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                           float* dL_dscales, float* dL_drotations, void* bwd_scratch, float** splat_grads_out, void* stream) {
    if (P == 0) return GSR_OK;
    float* sg = nullptr;
    int rc = gsr_backward_blend(settings, P, num_rendered, geom_buffer, binning_buffer, image_buffer, dL_dout_color,
                                dL_dout_invdepth, bwd_scratch, &sg, stream);
    if (rc != GSR_OK) return rc;
    if (splat_grads_out) *splat_grads_out = sg;
    return gsr_backward_preprocess(settings, P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   radii, geom_buffer, sg, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D,
                                   dL_dsh, dL_dscales, dL_drotations, stream);
}

int gsr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                  double beta2, double eps, int32_t step, void* stream) {
    if (n < 0 || step < 1) return fail(GSR_ERR_INVALID_ARG, "n < 0 or step < 1");
    if (n == 0) return GSR_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    gsr_launch_adam(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_adam_step_multi(const GsrAdamTensor* tensors, int32_t count, void* stream) {
    if (count < 0) return fail(GSR_ERR_INVALID_ARG, "count < 0");
    if (count == 0) return GSR_OK;
    if (!tensors) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    for (int i = 0; i < count; ++i) {
        const GsrAdamTensor& a = tensors[i];
        if (a.n < 0 || a.step < 1) return fail(GSR_ERR_INVALID_ARG, "n < 0 or step < 1");
        if (a.n > 0 && (!a.param || !a.grad || !a.exp_avg || !a.exp_avg_sq)) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    }
    for (int i = 0; i < count; i += GSR_ADAM_MAX_TENSORS)
        gsr_launch_adam_multi(tensors + i, count - i < GSR_ADAM_MAX_TENSORS ? count - i : GSR_ADAM_MAX_TENSORS, (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_sparse_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* visible,
                         int64_t N, int64_t M, double lr, double beta1, double beta2, double eps, void* stream) {
    if (N < 0 || M < 0) return fail(GSR_ERR_INVALID_ARG, "N < 0 or M < 0");
    if (N == 0 || M == 0) return GSR_OK;
    if (M >= (1ll << 31) || N >= (1ll << 31)) return fail(GSR_ERR_UNSUPPORTED, "sparse Adam: N and M must be below 2^31 (the kernel indexes rows and columns with 32 bits)");
    if (!param || !grad || !exp_avg || !exp_avg_sq || !visible) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    gsr_launch_sparse_adam(param, grad, exp_avg, exp_avg_sq, visible, N, M, lr, beta1, beta2, eps, (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_sparse_adam_step_multi(const GsrSparseAdamTensor* tensors, int32_t count, const uint8_t* visible, int64_t N, double beta1, double beta2,
                               void* stream) {
    if (count < 0 || N < 0) return fail(GSR_ERR_INVALID_ARG, "count < 0 or N < 0");
    if (count == 0 || N == 0) return GSR_OK;
    if (!tensors || !visible) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    if (N >= (1ll << 31)) return fail(GSR_ERR_UNSUPPORTED, "sparse Adam: N and M must be below 2^31 (the kernel indexes rows and columns with 32 bits)");
    for (int i = 0; i < count; ++i) {
        const GsrSparseAdamTensor& a = tensors[i];
        if (a.M < 0) return fail(GSR_ERR_INVALID_ARG, "M < 0");
        if (a.M >= (1ll << 31)) return fail(GSR_ERR_UNSUPPORTED, "sparse Adam: N and M must be below 2^31 (the kernel indexes rows and columns with 32 bits)");
        if (a.M > 0 && (!a.param || !a.grad || !a.exp_avg || !a.exp_avg_sq)) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    }
    for (int i = 0; i < count; i += GSR_ADAM_MAX_TENSORS)
        gsr_launch_sparse_adam_multi(tensors + i, count - i < GSR_ADAM_MAX_TENSORS ? count - i : GSR_ADAM_MAX_TENSORS, visible, N, beta1, beta2,
                                     (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_density_stats(int P, const float* viewspace_grad, const uint8_t* visible, const int32_t* radii, float* grad_accum, float* denom,
                      float* max_radii2D, void* stream) {
    if (P < 0) return fail(GSR_ERR_INVALID_ARG, "P < 0");
    if (P == 0) return GSR_OK;
    if (!viewspace_grad || !grad_accum || !denom) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    if (!visible && !radii) return fail(GSR_ERR_INVALID_ARG, "give the visibility mask or the radii (visible = radii > 0)");
    if (radii && !max_radii2D) return fail(GSR_ERR_INVALID_ARG, "radii given but max_radii2D is NULL");
    gsr_launch_density_stats(P, viewspace_grad, visible, radii, grad_accum, denom, max_radii2D, (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

size_t gsr_knn_scratch_bytes(int N) { return gsr_knn_scratch_bytes_impl(N); }

int gsr_knn_mean_dist2(int N, const float* points, float* mean_dist2, void* scratch, void* stream) {
    if (N < 0) return fail(GSR_ERR_INVALID_ARG, "N < 0");
    if (N == 0) return GSR_OK;
    if (!points || !mean_dist2 || !scratch) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    gsr_launch_knn(N, points, mean_dist2, scratch, (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_ssim_forward(int planes, int H, int W, const float* img1, const float* img2, float* ssim_map, float* dm_dmu1,
                     float* dm_dsigma1_sq, float* dm_dsigma12, void* stream) {
    if (planes < 0 || H <= 0 || W <= 0) return fail(GSR_ERR_INVALID_ARG, "bad image size");
    if (planes == 0) return GSR_OK;
    if (planes > 65535) return fail(GSR_ERR_UNSUPPORTED, "more than 65535 image planes");
    if (!img1 || !img2 || !ssim_map) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    if ((dm_dmu1 == nullptr) != (dm_dsigma1_sq == nullptr) || (dm_dmu1 == nullptr) != (dm_dsigma12 == nullptr))
        return fail(GSR_ERR_INVALID_ARG, "give all three derivative maps or none");
    gsr_launch_ssim_forward(planes, H, W, img1, img2, ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int64_t gsr_ssim_partial_count(int planes, int H, int W) {
    if (planes <= 0 || H <= 0 || W <= 0) return 0;
    return gsr_ssim_partial_count_impl(planes, H, W);
}

int gsr_ssim_mean_forward(int planes, int H, int W, const float* img1, const float* img2, float* partials, float* mean_out,
                          float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream) {
    if (planes <= 0 || H <= 0 || W <= 0) return fail(GSR_ERR_INVALID_ARG, "bad image size");
    if (planes > 65535) return fail(GSR_ERR_UNSUPPORTED, "more than 65535 image planes");
    if (gsr_ssim_partial_count_impl(planes, H, W) > 0x7FFFFFFFll) return fail(GSR_ERR_UNSUPPORTED, "too many tiles");
    if (!img1 || !img2 || !partials || !mean_out) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    if ((dm_dmu1 == nullptr) != (dm_dsigma1_sq == nullptr) || (dm_dmu1 == nullptr) != (dm_dsigma12 == nullptr))
        return fail(GSR_ERR_INVALID_ARG, "give all three derivative maps or none");
    gsr_launch_ssim_mean_forward(planes, H, W, img1, img2, partials, mean_out, dm_dmu1, dm_dsigma1_sq, dm_dsigma12,
                                 (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_ssim_mean_backward(int planes, int H, int W, const float* img1, const float* img2, const float* dL_dmean,
                           const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1,
                           void* stream) {
    if (planes <= 0 || H <= 0 || W <= 0) return fail(GSR_ERR_INVALID_ARG, "bad image size");
    if (planes > 65535) return fail(GSR_ERR_UNSUPPORTED, "more than 65535 image planes");
    if (!img1 || !img2 || !dL_dmean || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1)
        return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    gsr_launch_ssim_mean_backward(planes, H, W, img1, img2, dL_dmean, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1,
                                  (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_train_loss_forward(int planes, int H, int W, const float* img1, const float* img2, float lambda_dssim, float* partials,
                           float* loss_out, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream) {
    if (planes <= 0 || H <= 0 || W <= 0) return fail(GSR_ERR_INVALID_ARG, "bad image size");
    if (planes > 65535) return fail(GSR_ERR_UNSUPPORTED, "more than 65535 image planes");
    if (gsr_ssim_partial_count_impl(planes, H, W) > 0x3FFFFFFFll) return fail(GSR_ERR_UNSUPPORTED, "too many tiles");
    if (!img1 || !img2 || !partials || !loss_out) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    if ((dm_dmu1 == nullptr) != (dm_dsigma1_sq == nullptr) || (dm_dmu1 == nullptr) != (dm_dsigma12 == nullptr))
        return fail(GSR_ERR_INVALID_ARG, "give all three derivative maps or none");
    gsr_launch_train_loss_forward(planes, H, W, img1, img2, lambda_dssim, partials, loss_out, dm_dmu1, dm_dsigma1_sq, dm_dsigma12,
                                  (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_train_loss_backward(int planes, int H, int W, const float* img1, const float* img2, const float* dL_dloss, float lambda_dssim,
                            const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1, void* stream) {
    if (planes <= 0 || H <= 0 || W <= 0) return fail(GSR_ERR_INVALID_ARG, "bad image size");
    if (planes > 65535) return fail(GSR_ERR_UNSUPPORTED, "more than 65535 image planes");
    if (!img1 || !img2 || !dL_dloss || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1)
        return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    gsr_launch_train_loss_backward(planes, H, W, img1, img2, dL_dloss, lambda_dssim, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1,
                                   (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_ssim_backward(int planes, int H, int W, const float* img1, const float* img2, const float* dL_dmap,
                      const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1, void* stream) {
    if (planes < 0 || H <= 0 || W <= 0) return fail(GSR_ERR_INVALID_ARG, "bad image size");
    if (planes == 0) return GSR_OK;
    if (planes > 65535) return fail(GSR_ERR_UNSUPPORTED, "more than 65535 image planes");
    if (!img1 || !img2 || !dL_dmap || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1)
        return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    gsr_launch_ssim_backward(planes, H, W, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1, (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream) {
    (void)projmatrix;
    if (P < 0) return fail(GSR_ERR_INVALID_ARG, "P < 0");
    if (P == 0) return GSR_OK;
    if (!means3D || !viewmatrix || !present) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    gsr_launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return GSR_OK;
}

int gsr_forward_views(int P, int64_t R, int width, int height, const void* geom_buffer, const void* binning_buffer,
                      const void* image_buffer, GsrForwardViews* out) {
    if (!out || !geom_buffer || !image_buffer) return fail(GSR_ERR_INVALID_ARG, "NULL pointer");
    GsrGeom g = gsr_carve_geom((char*)geom_buffer, P);
    GsrImage im = gsr_carve_image((char*)image_buffer, width, height);
    const int n_tiles = ((width + GSR_TILE - 1) / GSR_TILE) * ((height + GSR_TILE - 1) / GSR_TILE);
    out->splats = (const float*)g.splats;
    out->tiles_touched = g.tiles;
    out->depth_order = g.vals[depth_order_buffer_index()];
    out->point_list = nullptr;
    if (binning_buffer && R > 0) {
        GsrBinning b = gsr_carve_binning((char*)binning_buffer, R);
        out->point_list = b.vals[list_buffer_index(n_tiles)];
    }
    out->ranges = (const uint32_t*)im.ranges;
    out->final_T = im.final_T;
    out->n_contrib = im.n_contrib;
    out->tile_scan = g.offsets;
    return GSR_OK;
}

}  // extern "C"
