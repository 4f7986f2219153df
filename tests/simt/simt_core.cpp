// SIMT emulator core (test infrastructure, see simt_emu.h): fibers, scheduler, barriers, wave exchange buffers.
#include <sys/mman.h>
#include <string.h>

#include <deque>
#include <vector>

#include "simt_emu.h"

namespace simt {

Lane* g_cur = nullptr;
dim3 g_blockIdx, g_blockDim, g_gridDim;
int g_last_error = 0;
bool g_dma_deferred = getenv("CBX_EMU_DMA") && !strcmp(getenv("CBX_EMU_DMA"), "deferred");

namespace {

constexpr size_t STACK_BYTES = 512 * 1024;
constexpr int MAX_THREADS = 1024;

struct PendingOp {
    char* dst;
    int size;  // 0: a buffer load / store (already executed; only occupies its slot of the in-order counter)
    unsigned char data[16];
};

struct Fiber {
    std::deque<PendingOp> vm;  // this lane's outstanding vector-memory operations, oldest first (deferred-DMA mode)
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    Lane lane;
    unsigned shfl_seq[6];   // pairwise xor shuffles executed per mask
    int nbar = 0;           // workgroup barriers this thread has reached
    const char* waiting = "";
};

struct WaveState {
    int lanes = 0, alive = 0, arrived = 0;
    unsigned gen = 0;
    alignas(16) unsigned char xbuf[2][WAVE][XSLOT];
    alignas(16) float result[2][32 * 32];  // D tile of the MFMA in flight (written by the last arriver), by exchange parity
    unsigned shfl_post[WAVE][6];
    unsigned long long shfl_box[WAVE][6][2];
};

struct BlockState {
    int alive = 0, arrived = 0;
    unsigned gen = 0;
};

const void* g_main_bottom = nullptr;  // the scheduler's (OS thread's) stack, as AddressSanitizer wants it named on a switch back
size_t g_main_size = 0;
Fiber g_fibers[MAX_THREADS];
std::vector<WaveState> g_waves;
BlockState g_block;
void* g_sched_sp = nullptr;
int g_curf = 0;
unsigned long g_progress = 0;
const std::function<void()>* g_body = nullptr;
std::vector<unsigned char> g_dyn;
bool g_abandon = false;
// CBX_EMU_SCHED=reverse | random[:seed]: the order in which the scheduler resumes the lanes of a workgroup.  Results must not depend on it;
// a kernel with a missing barrier (a lane reading LDS another lane has not written yet) does.
int g_sched_mode = 0;
unsigned long long g_sched_rng = 0x9E3779B97F4A7C15ull;
struct SchedInit {
    SchedInit() {
        const char* e = getenv("CBX_EMU_SCHED");
        if (!e) return;
        if (!strncmp(e, "reverse", 7)) g_sched_mode = 1;
        else if (!strncmp(e, "random", 6)) {
            g_sched_mode = 2;
            if (e[6] == ':') g_sched_rng ^= strtoull(e + 7, nullptr, 10) * 0x2545F4914F6CDD1Dull;
        }
    }
} g_sched_init;

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define SIMT_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
#endif
#endif
extern "C" void simt_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl simt_switch
    .type simt_switch,@function
simt_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size simt_switch, .-simt_switch
)");

Fiber& cur() { return g_fibers[g_curf]; }
WaveState& wave() { return g_waves[cur().lane.wave]; }

void release_wave(WaveState& w) {
    w.arrived = 0;
    ++w.gen;
    ++g_progress;
}
void release_block() {
    g_block.arrived = 0;
    ++g_block.gen;
    ++g_progress;
}

void vm_retire(Fiber& f, size_t keep) {
    while (f.vm.size() > keep) {
        const PendingOp& o = f.vm.front();
        if (o.size) memcpy(o.dst, o.data, o.size);
        f.vm.pop_front();
    }
}

void fiber_main() {
#ifdef SIMT_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &g_main_bottom, &g_main_size);  // first entry: learn the scheduler's stack
#endif
    (*g_body)();
    Fiber& f = cur();
    vm_retire(f, 0);
    f.done = true;
    ++g_progress;
    WaveState& w = g_waves[f.lane.wave];
    // a finished lane no longer takes part in barriers / exchanges (s_barrier counts the waves still running)
    if (--w.alive > 0 && w.arrived == w.alive) release_wave(w);
    if (--g_block.alive > 0 && g_block.arrived == g_block.alive) release_block();
#ifdef SIMT_ASAN
    __sanitizer_start_switch_fiber(nullptr, g_main_bottom, g_main_size);  // nullptr: this fiber's fake stack is released
#endif
    simt_switch(&f.sp, g_sched_sp);
    abort();  // never resumed
}

void init_fiber(Fiber& f) {
    if (!f.stack) {
        f.stack = (char*)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (f.stack == MAP_FAILED) {
            perror("simt: mmap of a fiber stack");
            abort();
        }
    }
    uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                // the "return address" of fiber_main (never used)
    *--sp = (void*)&fiber_main;     // popped by simt_switch's ret; the slot is 16-byte aligned, so fiber_main starts ABI-aligned
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    f.sp = sp;
    f.done = false;
    f.vm.clear();
    f.nbar = 0;
    f.lane.xpar = 0;
    memset(f.shfl_seq, 0, sizeof(f.shfl_seq));
    f.waiting = "";
}

}  // namespace

void vm_push_dma(void* dst, const void* src, int size) {
    PendingOp o;
    o.dst = (char*)dst;
    o.size = size;
    if (size > 16) {
        fprintf(stderr, "simt: LDS-DMA of %d bytes per lane\n", size);
        abort();
    }
    if (src) memcpy(o.data, src, size);
    else memset(o.data, 0, size);
    cur().vm.push_back(o);
}
void vm_push_other() {
    PendingOp o;
    o.dst = nullptr;
    o.size = 0;
    cur().vm.push_back(o);
}
// CBX_EMU_DMA_SLACK=k weakens every explicit wait by k operations: the self-check that the deferred mode can fail (tests/test_simt_kernels.py)
static int g_vm_slack = getenv("CBX_EMU_DMA_SLACK") ? atoi(getenv("CBX_EMU_DMA_SLACK")) : 0;
void vm_wait(int n) { vm_retire(cur(), (n < 0 ? 0 : (size_t)n) + (size_t)g_vm_slack); }

void yield() {
    Fiber& f = cur();
#ifdef SIMT_ASAN
    void* fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, g_main_bottom, g_main_size);
    simt_switch(&f.sp, g_sched_sp);
    __sanitizer_finish_switch_fiber(fake, &g_main_bottom, &g_main_size);
#else
    simt_switch(&f.sp, g_sched_sp);
#endif
}
void note_progress() { ++g_progress; }

void wave_sync() {
    WaveState& w = wave();
    if (++w.arrived == w.alive) {
        release_wave(w);
        return;
    }
    const unsigned g = w.gen;
    cur().waiting = "wave exchange / MFMA";
    do yield();
    while (w.gen == g);
}

// CBX_EMU_DROP_BARRIER=k: every thread skips its k-th workgroup barrier -- the self-check that the scheduling modes can see a missing barrier
static int g_drop_barrier = getenv("CBX_EMU_DROP_BARRIER") ? atoi(getenv("CBX_EMU_DROP_BARRIER")) : -1;

// Split form of wave_sync for collectives whose result is computed ONCE: the last lane to arrive computes, then releases the others.
bool wave_arrive_last() {
    WaveState& w = wave();
    return ++w.arrived == w.alive;
}
void wave_release() { release_wave(wave()); }
void wave_wait_released() {
    WaveState& w = wave();
    const unsigned g = w.gen;
    cur().waiting = "MFMA";
    do yield();
    while (w.gen == g);
}
float* wave_result() { return wave().result[cur().lane.xpar]; }

void block_sync() {
    if (g_drop_barrier >= 0 && cur().nbar++ == g_drop_barrier) return;
    if (++g_block.arrived == g_block.alive) {
        release_block();
        return;
    }
    const unsigned g = g_block.gen;
    cur().waiting = "workgroup barrier";
    do yield();
    while (g_block.gen == g);
}


unsigned long long shfl_xor_pair(unsigned long long bits, int mi) {
    Fiber& f = cur();
    WaveState& w = wave();
    const int l = f.lane.lane, p = l ^ (1 << mi);
    const unsigned k = f.shfl_seq[mi]++;
    w.shfl_box[l][mi][k & 1] = bits;
    w.shfl_post[l][mi] = k + 1;
    ++g_progress;
    if (p >= w.lanes) return bits;
    f.waiting = "xor shuffle partner";
    while (w.shfl_post[p][mi] < k + 1) yield();
    ++g_progress;
    return w.shfl_box[p][mi][k & 1];
}

void* dyn_lds() { return g_dyn.data(); }

int launch(dim3 grid, dim3 block, size_t dyn_bytes, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > MAX_THREADS || (long)grid.x * grid.y * grid.z <= 0) {
        fprintf(stderr, "simt: bad launch geometry (%u x %u x %u threads)\n", block.x, block.y, block.z);
        g_last_error = hipErrorLaunchFailure;
        return -1;
    }
    if (g_sched_sp != nullptr && g_cur != nullptr) {
        fprintf(stderr, "simt: nested launch\n");
        abort();
    }
    g_blockDim = block;
    g_gridDim = grid;
    g_body = &body;
    const int nwaves = (nthreads + WAVE - 1) / WAVE;
    if ((int)g_waves.size() < nwaves) g_waves.resize(nwaves);
    g_dyn.assign(dyn_bytes + 16, 0xCD);  // LDS is not zero-initialised
    // workgroups run one after the other, in dispatch order -- or, under CBX_EMU_SCHED, reversed / shuffled: nothing may depend on it
    // (the split-context decode attention must merge in split order whoever arrives last)
    const unsigned long nblocks = (unsigned long)grid.x * grid.y * grid.z;
    static std::vector<unsigned long> border;
    border.resize(nblocks);
    // CBX_EMU_WG_ORDER=dispatch keeps the workgroups in index order under any lane schedule: the hardware dispatches in index order, and the
    // producer / consumer launches (gemv_pair.hip) are the one place that relies on it
    static const bool wg_in_order = getenv("CBX_EMU_WG_ORDER") && !strcmp(getenv("CBX_EMU_WG_ORDER"), "dispatch");
    for (unsigned long b = 0; b < nblocks; ++b) border[b] = (g_sched_mode == 1 && !wg_in_order) ? nblocks - 1 - b : b;
    if (g_sched_mode == 2 && !wg_in_order)
        for (unsigned long b = nblocks - 1; b > 0; --b) {
            g_sched_rng = g_sched_rng * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(border[b], border[(unsigned long)((g_sched_rng >> 33) % (b + 1))]);
        }
    for (unsigned long bi = 0; bi < nblocks; ++bi) {
            {
                const unsigned bx = (unsigned)(border[bi] % grid.x), by = (unsigned)((border[bi] / grid.x) % grid.y), bz = (unsigned)(border[bi] / ((unsigned long)grid.x * grid.y));
                g_blockIdx = dim3(bx, by, bz);
                g_block = BlockState();
                g_block.alive = nthreads;
                for (int w = 0; w < nwaves; ++w) {
                    WaveState& ws = g_waves[w];
                    ws.lanes = ws.alive = std::min(WAVE, nthreads - w * WAVE);
                    ws.arrived = 0;
                    ws.gen = 0;
                    memset(ws.shfl_post, 0, sizeof(ws.shfl_post));
                }
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = g_fibers[t];
                    init_fiber(f);
                    f.lane.flat = t;
                    f.lane.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.lane.lane = t % WAVE;
                    f.lane.wave = t / WAVE;
                    f.lane.wave_lanes = g_waves[t / WAVE].lanes;
                    f.lane.wave_x = &g_waves[t / WAVE].xbuf[0][0][0];
                }
                int remaining = nthreads;
                static std::vector<int> order;
                order.resize(nthreads);
                for (int t = 0; t < nthreads; ++t) order[t] = g_sched_mode == 1 ? nthreads - 1 - t : t;
                while (remaining > 0) {
                    const unsigned long before = g_progress;
                    if (g_sched_mode == 2)  // a fresh permutation per sweep: lanes (and waves) overtake each other wherever no barrier forbids it
                        for (int t = nthreads - 1; t > 0; --t) {
                            g_sched_rng = g_sched_rng * 6364136223846793005ull + 1442695040888963407ull;
                            std::swap(order[t], order[(int)((g_sched_rng >> 33) % (unsigned)(t + 1))]);
                        }
                    for (int oi = 0; oi < nthreads; ++oi) {
                        const int t = order[oi];
                        Fiber& f = g_fibers[t];
                        if (f.done) continue;
                        g_curf = t;
                        g_cur = &f.lane;
#ifdef SIMT_ASAN
                        void* fake = nullptr;
                        __sanitizer_start_switch_fiber(&fake, f.stack, STACK_BYTES);
                        simt_switch(&g_sched_sp, f.sp);
                        __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#else
                        simt_switch(&g_sched_sp, f.sp);
#endif
                        if (f.done) --remaining;
                    }
                    if (remaining > 0 && g_progress == before) {
                        fprintf(stderr, "simt: DEADLOCK in block (%u,%u,%u): %d of %d threads stuck\n", bx, by, bz, remaining, nthreads);
                        int shown = 0;
                        for (int t = 0; t < nthreads && shown < 8; ++t)
                            if (!g_fibers[t].done) fprintf(stderr, "  thread %d waits at: %s\n", t, g_fibers[t].waiting), ++shown;
                        g_cur = nullptr;
                        g_last_error = hipErrorLaunchFailure;
                        return -1;  // the stuck fibers are abandoned; their stacks are re-initialised by the next launch
                    }
                }
            }
    }
    g_cur = nullptr;
    return 0;
}

}  // namespace simt
