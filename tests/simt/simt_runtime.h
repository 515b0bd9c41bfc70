// Runtime of the SIMT-on-CPU shim (tests/simt/hip/hip_runtime.h): fibers, workgroup barrier, wave-level rendezvous.  Include ONCE, in the
// harness translation unit, after the kernels.  TEST INFRASTRUCTURE.
#pragma once
#include <ucontext.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <vector>

simt::Idx threadIdx, blockIdx, blockDim, gridDim;

namespace simt {
const char* launch_error = nullptr;
constexpr int WAVE = 64;
constexpr size_t STACK_BYTES = 256 * 1024;
struct Wave {
    int alive = 0, count = 0;
    unsigned phase = 0;
    uint64_t val[2][WAVE];
};
struct State {
    int nthreads = 0, alive = 0, cur = -1;
    std::vector<ucontext_t> ctx;
    std::vector<char> done;
    std::vector<char> stacks;
    ucontext_t sched;
    int bar_count = 0;
    unsigned bar_phase = 0;
    std::vector<Wave> waves;
    std::function<void()> body;
    const char* error = nullptr;
} g;

static void yield() { swapcontext(&g.ctx[g.cur], &g.sched); }

// SIMT_SCHEDULE=<seed> (non-zero): a poor man's race detector.  The waves of a workgroup are swept in a random order and a random half of them sits out each
// sweep (so a wave runs many wave-level steps ahead of another unless a barrier holds it), and the workgroups of a grid run in a random order.  A kernel
// whose result depends on either -- an LDS hand-over without its barrier, a block that assumes its predecessor has run -- then fails the same parity tests
// that pass in the default order (waves 0..n-1 in step, blocks 0..grid-1).  Lanes inside a wave keep their order: a wave is lock-step on the hardware.
static uint64_t sched_state = 0;
static bool sched_on = false, sched_read = false;
bool schedule_shuffled() {
    if (!sched_read) {
        sched_read = true;
        const char* e = getenv("SIMT_SCHEDULE");
        const unsigned long long seed = e ? strtoull(e, nullptr, 10) : 0;
        sched_on = seed != 0;
        sched_state = seed * 0x9E3779B97F4A7C15ull + 1;
    }
    return sched_on;
}
uint32_t schedule_random() {
    sched_state = sched_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(sched_state >> 33);
}

void syncthreads() {
    const unsigned ph = g.bar_phase;
    if (++g.bar_count == g.alive) { g.bar_count = 0; ++g.bar_phase; yield(); return; }      // (the last thread waits for its turn too)
    while (g.bar_phase == ph) yield();
}

// deposit v, wait for the live lanes of the wave; returns the buffer holding every lane's value (dead lanes: 0)
static const uint64_t* exchange(uint64_t v) {
    const int t = g.cur, w = t / WAVE, lane = t % WAVE;
    Wave& wv = g.waves[w];
    const unsigned ph = wv.phase;
    uint64_t* buf = wv.val[ph & 1u];
    buf[lane] = v;
    if (++wv.count == wv.alive) {
        if (wv.alive < WAVE)      // lanes that have left the kernel contribute 0 (their slots may hold values of earlier operations)
            for (int l = 0; l < WAVE; ++l)
                if (g.done[w * WAVE + l]) buf[l] = 0;
        wv.count = 0;
        ++wv.phase;
        yield();      // the last lane to arrive (lane 0 in a descending sweep) must not run ahead of the others: it continues in its turn of the next sweep
    } else {
        while (wv.phase == ph) yield();
    }
    return buf;
}

uint64_t ballot(bool pred) {
    const uint64_t* b = exchange(pred ? 1u : 0u);
    uint64_t m = 0;
    for (int l = 0; l < WAVE; ++l) m |= (uint64_t)(b[l] & 1u) << l;
    return m;
}

void wave_barrier() { (void)exchange(0); }

uint32_t readlane(uint32_t v, int lane) { return (uint32_t)exchange(v)[lane & 63]; }

uint32_t shfl(uint32_t v, int src) { return (uint32_t)exchange(v)[src & 63]; }
uint32_t shfl_up(uint32_t v, int delta) { const uint64_t* b = exchange(v); const int lane = g.cur % WAVE; return lane >= delta ? (uint32_t)b[lane - delta] : v; }
uint32_t shfl_xor(uint32_t v, int m) { return (uint32_t)exchange(v)[(g.cur % WAVE) ^ (m & 63)]; }

// both operands travel in one exchange (a in the low, b in the high word)
uint2 permlane32_swap(uint32_t a, uint32_t b) {
    const uint64_t* buf = exchange((uint64_t)a | ((uint64_t)b << 32));
    const int lane = g.cur % WAVE;
    auto A = [&](int l) { return (uint32_t)buf[l]; };
    auto B = [&](int l) { return (uint32_t)(buf[l] >> 32); };
    return lane < 32 ? make_uint2(A(lane), A(lane + 32)) : make_uint2(B(lane - 32), B(lane));
}
uint2 permlane16_swap(uint32_t a, uint32_t b) {
    const uint64_t* buf = exchange((uint64_t)a | ((uint64_t)b << 32));
    const int lane = g.cur % WAVE, row = lane >> 4;
    auto A = [&](int l) { return (uint32_t)buf[l]; };
    auto B = [&](int l) { return (uint32_t)(buf[l] >> 32); };
    // odd rows of the first operand <-> even rows of the second
    return (row & 1) ? make_uint2(B(lane - 16), B(lane)) : make_uint2(A(lane), A(lane + 16));
}

int dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const uint64_t* b = exchange((uint32_t)src);
    const int lane = g.cur % WAVE, row = lane >> 4, bank = (lane >> 2) & 3;
    if (!((row_mask >> row) & 1) || !((bank_mask >> bank) & 1)) return old;
    int from = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11F) {                    // row_shr:n
        const int n = ctrl & 0xF;
        if ((lane & 15) >= n) from = lane - n;
    } else if (ctrl == 0x138) {                              // wave_shr:1
        if (lane >= 1) from = lane - 1;
    } else if (ctrl == 0x130) {                              // wave_shl:1 -- the next lane's value
        if (lane < WAVE - 1) from = lane + 1;
    } else if (ctrl == 0x142) {                              // row_bcast:15 -- lane 15 of a row to every lane of the next row
        if (row >= 1) from = row * 16 - 1;
    } else if (ctrl == 0x143) {                              // row_bcast:31 -- lane 31 to rows 2 and 3
        if (row >= 2) from = 31;
    } else {
        static char msg[64];
        snprintf(msg, sizeof(msg), "simt::dpp: control code 0x%x not modelled", ctrl);
        g.error = msg;
        return old;
    }
    if (from < 0) return bound_ctrl ? 0 : old;
    return (int)(uint32_t)b[from];
}

static void trampoline() {
    g.body();
    const int t = g.cur, w = t / WAVE;
    g.done[t] = 1;
    // a lane that has left no longer takes part in barriers / wave operations (the csrc/ kernels only leave workgroup-uniformly)
    --g.alive;
    --g.waves[w].alive;      // (its slots in the exchange buffers stay as they are: other lanes may not have read the last operation's result yet)
    swapcontext(&g.ctx[t], &g.sched);
}

extern "C" char __start_simt_lds[] __attribute__((weak)), __stop_simt_lds[] __attribute__((weak));      // bounds of the "simt_lds" section (GNU ld)
static void poison_lds(unsigned block) {
    if (!__start_simt_lds) return;
    memset(__start_simt_lds, (int)(((0x5Au + block * 37u) | 1u) & 0xFFu), (size_t)(__stop_simt_lds - __start_simt_lds));      // (a whole library has megabytes of LDS arrays: memset speed matters)
}

// run `body` as ONE workgroup of `nthreads` threads (multiple of 64) with blockIdx.x = block; false: deadlock or unmodelled operation
bool run_block(unsigned block, unsigned grid, int nthreads, const std::function<void()>& body) {
    g.nthreads = g.alive = nthreads;
    g.ctx.assign(nthreads, ucontext_t());
    g.done.assign(nthreads, 0);
    if (g.stacks.size() != (size_t)nthreads * STACK_BYTES) g.stacks.assign((size_t)nthreads * STACK_BYTES, 0);
    g.waves.assign(nthreads / WAVE, Wave());
    for (auto& w : g.waves) { w.alive = WAVE; memset(w.val, 0, sizeof(w.val)); }
    g.bar_count = 0;
    g.bar_phase = 0;
    g.body = body;
    g.error = nullptr;
    poison_lds(block);
    blockIdx = {block, 0, 0};
    gridDim = {grid, 1, 1};
    blockDim = {(unsigned)nthreads, 1, 1};
    for (int t = 0; t < nthreads; ++t) {
        getcontext(&g.ctx[t]);
        g.ctx[t].uc_stack.ss_sp = g.stacks.data() + (size_t)t * STACK_BYTES;
        g.ctx[t].uc_stack.ss_size = STACK_BYTES;
        g.ctx[t].uc_link = &g.sched;
        makecontext(&g.ctx[t], trampoline, 0);
    }
    long idle_sweeps = 0;
    const bool shuffled = schedule_shuffled();
    const int nw = nthreads / WAVE;
    std::vector<int> order(nw);
    for (int w = 0; w < nw; ++w) order[w] = w;
    while (g.alive > 0) {
        const int alive_before = g.alive;
        const unsigned bar_before = g.bar_phase;
        unsigned long phases_before = 0;
        for (auto& w : g.waves) phases_before += w.phase;
        if (shuffled)
            for (int i = nw - 1; i > 0; --i) std::swap(order[i], order[schedule_random() % (unsigned)(i + 1)]);
        for (int k = 0; k < nw; ++k) {
            const int w = order[k];
            if (shuffled && k > 0 && (schedule_random() & 1u)) continue;      // (the first wave of the sweep always runs)
            for (int lane = WAVE - 1; lane >= 0; --lane) {          // lanes of a wave: 63 down to 0 (see hip_runtime.h)
                const int t = w * WAVE + lane;
                if (g.done[t]) continue;
                g.cur = t;
                threadIdx = {(unsigned)t, 0, 0};
                swapcontext(&g.sched, &g.ctx[t]);
            }
        }
        unsigned long phases_after = 0;
        for (auto& w : g.waves) phases_after += w.phase;
        if (g.alive == alive_before && g.bar_phase == bar_before && phases_after == phases_before) {
            if (++idle_sweeps > (shuffled ? 4096 : 4)) { launch_error = g.error = "simt: deadlock (a barrier or wave-level operation that not every live lane reaches)"; return false; }
        } else idle_sweeps = 0;
    }
    if (g.error) launch_error = g.error;
    return g.error == nullptr;
}
}  // namespace simt
