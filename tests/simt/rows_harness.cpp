// The kernels around the default path, compiled for the HOST through the SIMT-on-CPU shim and run through their own launchers
// (tests/test_simt_rows_cpu.py): the LSD radix sort (csrc/sort.hip: the depth sort's fall-back, the tile sort of frames beyond 65536 tiles, the
// Morton sort of the k-NN), the legacy binning path (csrc/binning.hip: tile scan, instance emission, tile ranges), distCUDA2 (csrc/knn.hip) and
// the density-control statistics (csrc/density.hip), the fused / sparse Adam steps (csrc/adam.hip).  TEST INFRASTRUCTURE, never part of libgsr_hip.so.
#define __HIPCC__ 1
#include "hip/hip_runtime.h"
#include "sort.hip"
#include "binning.hip"
#include "knn.hip"
#include "density.hip"
#include "adam.hip"
#include "simt_runtime.h"
#include <vector>

static char g_err[256];
static int finish(const char* what) {
    if (!simt::launch_error) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, simt::launch_error);
    simt::launch_error = nullptr;
    return -1;
}

extern "C" {

const char* simt_rows_last_error(void) { return g_err; }

// stable LSD sort of (key, value) pairs on the low nbits; key16 != 0: 16-bit keys.  keys / vals are sorted in place; rect (optional) is gathered
// by the last pass into rect_sorted.  items: keys per workgroup (1024 / 2048 / 4096).
int simt_radix_sort(int key16, int64_t n, int nbits, int max_digit_bits, int items, void* keys, uint32_t* vals, const uint2* rect, uint2* rect_sorted) {
    const size_t nb = (size_t)((n + items - 1) / items);
    std::vector<uint32_t> hist((size_t)(1u << 11) * (nb + 1)), total(1u << 11), v1((size_t)n + 16);
    uint32_t* vv[2] = {vals, v1.data()};
    int cur;
    if (key16) {
        std::vector<uint16_t> k1((size_t)n + 16);
        uint16_t* kk[2] = {(uint16_t*)keys, k1.data()};
        cur = gsr_radix_sort_pairs_k16(kk, vv, n, nbits, max_digit_bits, hist.data(), total.data(), items, nullptr);
        if (cur) { memcpy(keys, k1.data(), (size_t)n * 2); memcpy(vals, v1.data(), (size_t)n * 4); }
    } else {
        std::vector<uint32_t> k1((size_t)n + 16);
        uint32_t* kk[2] = {(uint32_t*)keys, k1.data()};
        cur = gsr_radix_sort_pairs(kk, vv, n, nbits, max_digit_bits, hist.data(), total.data(), items, nullptr, rect, rect_sorted);
        if (cur) { memcpy(keys, k1.data(), (size_t)n * 4); memcpy(vals, v1.data(), (size_t)n * 4); }
    }
    return finish("radix sort");
}

// the legacy binning path after the depth order is known: tile scan (offsets, block table, R), 32-bit or 16-bit instance emission, LSD tile sort,
// tile ranges.  Returns R or -1.
int64_t simt_legacy_bins(int P, int gx, int gy, const uint32_t* order, const uint2* rect, uint32_t* offsets, uint2* rect_sorted, uint32_t* point_list,
                         int64_t r_cap, uint2* ranges, int key16) {
    const int n_tiles = gx * gy;
    std::vector<uint64_t> block_sums((size_t)P / GSR_SCAN_ITEMS + 16);
    std::vector<uint2> block_first(gsr_block_first_cap(P) + 16);
    std::vector<uint32_t> frame(64, 0u);
    gsr_launch_scan_tiles(P, order, rect, rect_sorted, offsets, block_sums.data(), block_first.data(), (uint32_t)gsr_block_first_cap(P), frame.data() + 4, nullptr, 0u, false, nullptr);
    if (finish("scan")) return -1;
    const int64_t R = P > 0 ? (int64_t)offsets[P - 1] : 0;
    if (R > r_cap) { snprintf(g_err, sizeof(g_err), "R exceeds the capacity"); return -1; }
    for (int t = 0; t < n_tiles; ++t) ranges[t] = make_uint2(0u, 0u);
    if (R == 0) return 0;
    std::vector<uint32_t> keys((size_t)R + 16), vals((size_t)R + 16);
    gsr_launch_emit(P, gx, order, offsets, rect_sorted, keys.data(), key16 != 0, vals.data(), nullptr, nullptr);
    if (finish("emit")) return -1;
    int nbits = 1;
    while ((1ll << nbits) < n_tiles) ++nbits;
    if (simt_radix_sort(key16, R, key16 && nbits < 9 ? 9 : nbits, GSR_TILE_DIGIT_BITS, 4096, keys.data(), vals.data(), nullptr, nullptr)) return -1;
    gsr_launch_ranges(R, n_tiles, keys.data(), key16 != 0, ranges, true, nullptr);
    if (finish("ranges")) return -1;
    memcpy(point_list, vals.data(), (size_t)R * 4);
    return R;
}

int simt_knn(int N, const float* points, float* out) {
    std::vector<char> scratch(gsr_knn_scratch_bytes_impl(N) + 256);
    gsr_launch_knn(N, points, out, scratch.data(), nullptr);
    return finish("knn");
}

int simt_density_stats(int P, const float* grad, const uint8_t* visible, const int32_t* radii, float* accum, float* denom, float* max_radii) {
    gsr_launch_density_stats(P, grad, visible, radii, accum, denom, max_radii, nullptr);
    return finish("density stats");
}

// fused dense Adam over several tensors in one launch (gsr_adam_step_multi) / one tensor (gsr_adam_step); SparseGaussianAdam's row-sparse step
int simt_adam_multi(const GsrAdamTensor* tensors, int count) {
    gsr_launch_adam_multi(tensors, count, nullptr);
    return finish("adam multi");
}
int simt_adam(float* p, const float* g, float* m, float* v, int64_t n, double lr, double b1, double b2, double eps, int step) {
    gsr_launch_adam(p, g, m, v, n, lr, b1, b2, eps, step, nullptr);
    return finish("adam");
}
int simt_sparse_adam_multi(const GsrSparseAdamTensor* tensors, int count, const uint8_t* visible, int64_t N, double b1, double b2) {
    gsr_launch_sparse_adam_multi(tensors, count, visible, N, b1, b2, nullptr);
    return finish("sparse adam multi");
}
int simt_sparse_adam(float* p, const float* g, float* m, float* v, const uint8_t* visible, int64_t N, int64_t M, double lr, double b1, double b2, double eps) {
    gsr_launch_sparse_adam(p, g, m, v, visible, N, M, lr, b1, b2, eps, nullptr);
    return finish("sparse adam");
}

}  // extern "C"
