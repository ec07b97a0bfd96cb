// TEST INFRASTRUCTURE -- wavefront emulator (see include/hip/hip_runtime.h in this directory).
//
// Execution: emu::launch runs the blocks of a grid on a pool of host threads; inside a block the waves run one
// after the other (none of the kernels uses a block-level barrier); a wave is 64 fibers on private stacks
// switched by a dozen instructions of x86-64 assembly.  A lane runs until it needs another lane (xlane()),
// records what it asks for and hands the processor to the next runnable lane; the lane that finds nobody left to
// run resolves the pending operations and releases the lanes that took part.
//
// Divergence: lanes may wait at different call sites.  The lanes at the lowest code address are released first
// (structured code laid out in source order -- the build uses -fno-reorder-blocks -- puts the inside of an `if`
// or a loop before what follows it), the others stay blocked until the released ones join them or finish.
// Every operation resolved with fewer lanes than the wave has live is counted (emu_partial_ops()).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <sys/mman.h>
#include <dlfcn.h>
#include <signal.h>
#include <ucontext.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#include <mutex>
#include <map>

extern "C" void emu_switch(void **save_sp, void *new_sp);
asm(".text\n"
    ".globl emu_switch\n"
    ".type emu_switch,@function\n"
    "emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size emu_switch, .-emu_switch\n");

// the kernels' dynamic LDS: `extern __shared__ uint8_t lds[]` resolves to this (one per host thread = one per running block)
__thread __attribute__((aligned(64))) uint8_t lds[160 * 1024];

namespace emu {

static const size_t STACK_BYTES = 8u << 20;     // per lane; untouched pages cost nothing (MAP_NORESERVE)

struct Pending {
    void *site;
    int op;
    uint64_t a;
    int64_t b, c;
    uint32_t ctrl;
    uint64_t result;
};

struct Wave {
    void *sp[64];
    void *main_sp;
    uint8_t *stacks;
    Pending pend[64];
    uint64_t runnable, blocked, done, live;
    int cur;
    unsigned tid_base;
    uint3 bidx;
    dim3 bdim, gdim;
    const std::function<void()> *body;
};

static thread_local Wave *g_wave;
static std::atomic<unsigned long long> g_partial_ops{0}, g_ops{0}, g_inactive_reads{0}, g_stores{0};
// diagnostics: sites of operations resolved with part of the wave / reading a lane that is not taking part
static std::mutex g_site_mu;
struct SiteStat { void *site; int kind; unsigned long long n; };
static std::vector<SiteStat> g_sites;
static void note_site(void *site, int kind)
{
    std::lock_guard<std::mutex> g(g_site_mu);
    for (auto &s : g_sites) if (s.site == site && s.kind == kind) { s.n++; return; }
    g_sites.push_back(SiteStat{site, kind, 1});
}

uint3 thread_idx() { Wave *w = g_wave; return uint3{w->tid_base + (unsigned)w->cur, 0, 0}; }
uint3 block_idx() { return g_wave->bidx; }
dim3 block_dim() { return g_wave->bdim; }
dim3 grid_dim() { return g_wave->gdim; }

static void resolve(Wave *w)
{
    // The group to release.  Normal case: every live lane waits with the same kind of operation -- one group (the compiler
    // may have duplicated a call site into both arms of a lane-dependent branch, so the return address alone would split
    // lanes that are at the same source-level operation).  Otherwise the lanes are in different places: take the lane at
    // the lowest code address and everything that waits with the same kind of operation.
    uint64_t bl = w->blocked;
    auto same_kind = [&](const Pending &x, const Pending &y) {
        return x.op == y.op && x.ctrl == y.ctrl && (x.op != OP_DPP || x.c == y.c);
    };
    int lowest = -1;
    // Stores first (see store_hook): a lane that waits to store is released before any real cross-lane operation is resolved,
    // lowest address first, so that the lanes which skipped an `if (lane == 0) x = v;` never get past the WAVE_SYNC() after it
    // while lane 0 is still waiting to store.
    for (uint64_t m = bl; m; m &= m - 1) {
        int l = __builtin_ctzll(m);
        if (w->pend[l].op != OP_STORE) continue;
        if (lowest < 0 || (uintptr_t)w->pend[l].site < (uintptr_t)w->pend[lowest].site) lowest = l;
    }
    uint64_t grp = 0;
    void *site;
    if (lowest >= 0) {
        site = w->pend[lowest].site;
        for (uint64_t m = bl; m; m &= m - 1) {
            int l = __builtin_ctzll(m);
            if (w->pend[l].op == OP_STORE && w->pend[l].site == site) { grp |= 1ull << l; w->pend[l].result = 0; }
        }
        g_stores.fetch_add(1, std::memory_order_relaxed);
        w->blocked &= ~grp;
        w->runnable |= grp;
        return;
    }
    for (uint64_t m = bl; m; m &= m - 1) {
        int l = __builtin_ctzll(m);
        if (lowest < 0 || (uintptr_t)w->pend[l].site < (uintptr_t)w->pend[lowest].site) lowest = l;
    }
    site = w->pend[lowest].site;
    for (uint64_t m = bl; m; m &= m - 1) {
        int l = __builtin_ctzll(m);
        if (same_kind(w->pend[l], w->pend[lowest])) grp |= 1ull << l;
    }
    g_ops.fetch_add(1, std::memory_order_relaxed);
    if (grp != (w->live & ~w->done)) { g_partial_ops.fetch_add(1, std::memory_order_relaxed); note_site(site, 0); }
    const int first = __builtin_ctzll(grp);
    const int op = w->pend[first].op;
    auto active = [&](int l) { return l >= 0 && l < 64 && ((grp >> l) & 1); };
    uint64_t ballot = 0;
    if (op == OP_BALLOT)
        for (uint64_t m = grp; m; m &= m - 1) { int l = __builtin_ctzll(m); if (w->pend[l].a) ballot |= 1ull << l; }
    for (uint64_t m = grp; m; m &= m - 1) {
        const int l = __builtin_ctzll(m);
        Pending &p = w->pend[l];
        if (p.op != op) { fprintf(stderr, "wave_emu: lanes at one site disagree on the operation (%d vs %d)\n", p.op, op); abort(); }
        switch (op) {
        case OP_BARRIER: case OP_STORE: p.result = 0; break;
        case OP_BALLOT: p.result = ballot; break;
        case OP_READFIRST: p.result = w->pend[first].a; break;
        case OP_READLANE: {
            int s = (int)p.b & 63;
            if (!active(s)) { g_inactive_reads.fetch_add(1, std::memory_order_relaxed); note_site(site, 1); p.result = 0; }
            else p.result = w->pend[s].a;
            break;
        }
        case OP_BPERMUTE: {
            int s = (int)p.b & 63;
            if (!active(s)) { g_inactive_reads.fetch_add(1, std::memory_order_relaxed); note_site(site, 2); p.result = 0; }   // ds_bpermute returns 0 for a disabled source lane
            else p.result = w->pend[s].a;
            break;
        }
        case OP_SHFL_UP: { int s = l - (int)p.b; p.result = (s >= 0 && active(s)) ? w->pend[s].a : p.a; break; }
        case OP_SHFL_DOWN: { int s = l + (int)p.b; p.result = (s < 64 && active(s)) ? w->pend[s].a : p.a; break; }
        case OP_SHFL_XOR: { int s = l ^ ((int)p.b & 63); p.result = active(s) ? w->pend[s].a : p.a; break; }
        case OP_DPP: {
            const uint32_t ctrl = p.ctrl;
            const int row_mask = (int)(p.c >> 8) & 0xF, bank_mask = (int)(p.c >> 4) & 0xF, bound = (int)p.c & 1;
            const int row = l >> 4, inrow = l & 15, bank = inrow >> 2;
            const uint64_t old = (uint64_t)(uint32_t)p.b;
            if (!((row_mask >> row) & 1) || !((bank_mask >> bank) & 1)) { p.result = old; break; }
            int s = -1;
            if (ctrl >= 0x101 && ctrl <= 0x10F) { int n = ctrl & 15; s = inrow + n <= 15 ? l + n : -1; }            // row_shl:n
            else if (ctrl >= 0x111 && ctrl <= 0x11F) { int n = ctrl & 15; s = inrow >= n ? l - n : -1; }            // row_shr:n
            else if (ctrl == 0x138) s = l >= 1 ? l - 1 : -1;                                                           // wave_shr:1
            else if (ctrl == 0x130) s = l <= 62 ? l + 1 : -1;                                                          // wave_shl:1
            else if (ctrl == 0x142) s = row >= 1 ? (row - 1) * 16 + 15 : -1;                                            // row_bcast:15
            else if (ctrl == 0x143) s = l >= 32 ? 31 : -1;                                                              // row_bcast:31
            else if (ctrl <= 0xFF) s = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);                                        // quad_perm
            else { fprintf(stderr, "wave_emu: DPP control 0x%x not modelled\n", ctrl); abort(); }
            if (s >= 0 && active(s)) p.result = w->pend[s].a;
            else p.result = bound ? 0 : old;
            break;
        }
        default: abort();
        }
    }
    static const int trace = getenv("SNAPGPU_EMU_TRACE") ? atoi(getenv("SNAPGPU_EMU_TRACE")) : 0;
    if (trace) {
        Dl_info di; uintptr_t rel = (uintptr_t)site;
        if (dladdr(site, &di) && di.dli_fbase) rel -= (uintptr_t)di.dli_fbase;
        fprintf(stderr, "emu-op %llu kind %d site 0x%lx grp %016llx a0 %llx r0 %llx", (unsigned long long)g_ops.load(), op, (unsigned long)rel,
                (unsigned long long)grp, (unsigned long long)w->pend[first].a, (unsigned long long)w->pend[first].result);
        if (trace > 1) { fprintf(stderr, " a:"); for (int l = 0; l < 64; l++) fprintf(stderr, " %llx", (unsigned long long)w->pend[l].a); }
        fprintf(stderr, "\n");
    }
    w->blocked &= ~grp;
    w->runnable |= grp;
}

// called by a lane that cannot go on (blocked or finished): find somebody to run
static void schedule_from(Wave *w, int me, bool finished)
{
    for (;;) {
        if (w->runnable) {
            // Highest lane first: between two rendezvous points a lane runs on its own, so "every lane reads x, then lane 0
            // stores x" (which needs no barrier on the device: the wave executes in lockstep) only comes out right here if
            // lane 0 runs last.
            int l = 63 - __builtin_clzll(w->runnable);
            w->runnable &= ~(1ull << l);
            if (l == me && !finished) { w->cur = me; return; }
            w->cur = l;
            emu_switch(&w->sp[me], w->sp[l]);
            // somebody switched back to me: I am running again
            w->cur = me;
            return;
        }
        if (w->blocked) { resolve(w); continue; }
        // nothing runnable, nothing blocked: the wave is done
        w->cur = -1;
        emu_switch(&w->sp[me], w->main_sp);
        abort();    // a finished lane is never resumed
    }
}

uint64_t xlane(Op op, uint64_t a, int64_t b, int64_t c, uint32_t ctrl, void *site)
{
    Wave *w = g_wave;
    const int me = w->cur;
    Pending &p = w->pend[me];
    p.site = site; p.op = op; p.a = a; p.b = b; p.c = c; p.ctrl = ctrl;
    w->blocked |= 1ull << me;
    schedule_from(w, me, false);
    return w->pend[me].result;
}

static void fiber_main()
{
    Wave *w = g_wave;
    const int me = w->cur;
    (*w->body)();
    w = g_wave;
    w->done |= 1ull << me;
    schedule_from(w, me, true);
    abort();
}

static void run_wave(Wave *w, int n_lanes)
{
    w->runnable = n_lanes == 64 ? ~0ull : ((1ull << n_lanes) - 1);
    w->live = w->runnable;
    w->blocked = 0; w->done = 0;
    for (int l = 0; l < n_lanes; l++) {
        uint8_t *top = w->stacks + (size_t)(l + 1) * STACK_BYTES;
        void **sp = (void **)top;
        *--sp = nullptr;                    // fake return address of fiber_main's "caller" (keeps rsp = 8 mod 16 at entry)
        *--sp = (void *)&fiber_main;
        for (int r = 0; r < 6; r++) *--sp = nullptr;
        w->sp[l] = sp;
    }
    g_wave = w;
    int l0 = 63 - __builtin_clzll(w->runnable);
    w->runnable &= ~(1ull << l0);
    w->cur = l0;
    emu_switch(&w->main_sp, w->sp[l0]);
    w->cur = -1;
    if ((w->done & w->live) != w->live) { fprintf(stderr, "wave_emu: wave ended with unfinished lanes\n"); abort(); }
}

// diagnostics for a fault inside emulated device code: which lane, what address, where
static void segv_handler(int, siginfo_t *si, void *uc_)
{
    ucontext_t *uc = (ucontext_t *)uc_;
    Wave *w = g_wave;
    void *pc = (void *)uc->uc_mcontext.gregs[REG_RIP];
    Dl_info di; uintptr_t rel = (uintptr_t)pc;
    if (dladdr(pc, &di) && di.dli_fbase) rel -= (uintptr_t)di.dli_fbase;
    fprintf(stderr, "wave_emu: SIGSEGV at address %p, pc %p (library offset 0x%lx), lane %d, block %u\n", si->si_addr, pc, (unsigned long)rel,
            w ? w->cur : -1, w ? w->bidx.x : 0u);
    if (w && w->cur >= 0) {
        uint8_t *lo = w->stacks + (size_t)w->cur * STACK_BYTES;
        void **sp = (void **)uc->uc_mcontext.gregs[REG_RSP];
        fprintf(stderr, "wave_emu: lane stack [%p, %p), rsp %p\n", lo, lo + STACK_BYTES, (void *)sp);
        if ((uint8_t *)sp >= lo && (uint8_t *)sp < lo + STACK_BYTES) {          // words on the stack that point into this library: return addresses
            for (int i = 0, shown = 0; i < 4096 && (uint8_t *)(sp + i) < lo + STACK_BYTES && shown < 12; i++) {
                Dl_info d2;
                if (dladdr(sp[i], &d2) && d2.dli_fbase == di.dli_fbase + 0 && d2.dli_fbase) {}
                if (dladdr(sp[i], &d2) && d2.dli_fname && strstr(d2.dli_fname, "libsnapgpu_emu")) {
                    fprintf(stderr, "wave_emu:   [rsp+%d] library offset 0x%lx\n", i * 8, (unsigned long)((uintptr_t)sp[i] - (uintptr_t)d2.dli_fbase));
                    shown++;
                }
            }
        }
    }
    _exit(139);
}
static void install_segv_handler()
{
    static std::once_flag once;
    std::call_once(once, []() {
        static uint8_t alt[1 << 16];
        stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof(alt); ss.ss_flags = 0;
        sigaltstack(&ss, nullptr);
        struct sigaction sa; memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = segv_handler; sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigaction(SIGSEGV, &sa, nullptr);
    });
}

static int env_int(const char *name, int dflt) { const char *s = getenv(name); return s && *s ? atoi(s) : dflt; }

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body)
{
    if (lds_bytes > sizeof(lds)) { fprintf(stderr, "wave_emu: %zu bytes of LDS asked for\n", lds_bytes); abort(); }
    const unsigned n_blocks = grid.x;
    if (env_int("SNAPGPU_EMU_SEGV_HANDLER", 0)) install_segv_handler();
    int n_threads = env_int("SNAPGPU_EMU_THREADS", (int)std::thread::hardware_concurrency());
    if (n_threads < 1) n_threads = 1;
    if ((unsigned)n_threads > n_blocks) n_threads = (int)n_blocks;
    std::atomic<unsigned> next{0};
    auto worker = [&]() {
        Wave *w = new Wave();
        w->stacks = (uint8_t *)mmap(nullptr, 64 * STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (w->stacks == (uint8_t *)MAP_FAILED) { perror("wave_emu: mmap"); abort(); }
        w->body = &body; w->bdim = block; w->gdim = grid;
        for (;;) {
            unsigned b = next.fetch_add(1);
            if (b >= n_blocks) break;
            w->bidx = uint3{b, 0, 0};
            memset(lds, 0xA5, lds_bytes);                 // LDS is not initialised on the device either
            for (unsigned t0 = 0; t0 < block.x; t0 += 64) {
                w->tid_base = t0;
                unsigned n = block.x - t0; if (n > 64) n = 64;
                run_wave(w, (int)n);
            }
        }
        munmap(w->stacks, 64 * STACK_BYTES);
        g_wave = nullptr;                                // (the worker may be the calling thread: host code runs the store hook too)
        delete w;
    };
    if (n_threads == 1) worker();
    else {
        std::vector<std::thread> ts;
        for (int i = 0; i < n_threads; i++) ts.emplace_back(worker);
        for (auto &t : ts) t.join();
    }
}

}  // namespace emu

// ---- memory hooks.  The kernel translation units are compiled with -fsanitize=thread purely for its instrumentation: GCC
// then calls __tsan_writeN(address) before every store that is not to a non-escaping local.  A store to memory other lanes
// can see (LDS, the per-wave scratch slab, results) is made a rendezvous: every lane has done the loads that precede the store
// in program order before any lane stores.  That is what lockstep execution gives the device code for free, and what
// "all lanes read x; lane 0 stores x" and "all lanes do x += v on one wave-uniform LDS word" rely on.
namespace emu {
static inline void store_hook(void *addr, void *site)
{
    Wave *w = g_wave;
    if (!w || w->cur < 0) return;                                   // host code
    uint8_t *a = (uint8_t *)addr;
    uint8_t *lo = w->stacks + (size_t)w->cur * STACK_BYTES;
    if (a >= lo && a < lo + STACK_BYTES) return;                    // the lane's own stack
    xlane(OP_STORE, 0, 0, 0, 0, site);
}
}
#define EMU_RA() __builtin_extract_return_addr(__builtin_return_address(0))
extern "C" {
void __tsan_init() {}
void __tsan_func_entry(void *) {}
void __tsan_func_exit() {}
void __tsan_read1(void *) {}
void __tsan_read2(void *) {}
void __tsan_read4(void *) {}
void __tsan_read8(void *) {}
void __tsan_read16(void *) {}
void __tsan_unaligned_read2(void *) {}
void __tsan_unaligned_read4(void *) {}
void __tsan_unaligned_read8(void *) {}
void __tsan_unaligned_read16(void *) {}
void __tsan_read_range(void *, unsigned long) {}
void __tsan_write1(void *a) { emu::store_hook(a, EMU_RA()); }
void __tsan_write2(void *a) { emu::store_hook(a, EMU_RA()); }
void __tsan_write4(void *a) { emu::store_hook(a, EMU_RA()); }
void __tsan_write8(void *a) { emu::store_hook(a, EMU_RA()); }
void __tsan_write16(void *a) { emu::store_hook(a, EMU_RA()); }
void __tsan_unaligned_write2(void *a) { emu::store_hook(a, EMU_RA()); }
void __tsan_unaligned_write4(void *a) { emu::store_hook(a, EMU_RA()); }
void __tsan_unaligned_write8(void *a) { emu::store_hook(a, EMU_RA()); }
void __tsan_unaligned_write16(void *a) { emu::store_hook(a, EMU_RA()); }
void __tsan_write_range(void *a, unsigned long) { emu::store_hook(a, EMU_RA()); }
void __tsan_vptr_update(void **, void *) {}
void __tsan_vptr_read(void **) {}
int __tsan_atomic32_fetch_add(volatile int *p, int v, int) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
int __tsan_atomic32_fetch_sub(volatile int *p, int v, int) { return __atomic_fetch_sub(p, v, __ATOMIC_SEQ_CST); }
long __tsan_atomic64_fetch_add(volatile long *p, long v, int) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
int __tsan_atomic32_fetch_or(volatile int *p, int v, int) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
int __tsan_atomic32_exchange(volatile int *p, int v, int) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
int __tsan_atomic32_load(const volatile int *p, int) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
int __tsan_atomic32_compare_exchange_strong(volatile int *p, int *e, int d, int, int) { return __atomic_compare_exchange_n(p, e, d, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); }
int __tsan_atomic64_compare_exchange_strong(volatile long *p, long *e, long d, int, int) { return __atomic_compare_exchange_n(p, e, d, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); }
void __tsan_atomic32_store(volatile int *p, int v, int) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
void __tsan_atomic64_store(volatile long *p, long v, int) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
long __tsan_atomic64_load(const volatile long *p, int) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
unsigned long long emu_partial_ops() { return emu::g_partial_ops.load(); }
unsigned long long emu_total_ops() { return emu::g_ops.load(); }
unsigned long long emu_inactive_reads() { return emu::g_inactive_reads.load(); }
// prints "kind address-in-library count" lines (kind 0 = partial wave, 1 = readlane of an absent lane, 2 = bpermute of an absent lane);
// addresses are relative to the library's load address, ready for addr2line -e libsnapgpu_emu.so
void emu_dump_sites()
{
    std::lock_guard<std::mutex> g(emu::g_site_mu);
    Dl_info di;
    for (auto &s : emu::g_sites) {
        uintptr_t rel = (uintptr_t)s.site;
        if (dladdr(s.site, &di) && di.dli_fbase) rel -= (uintptr_t)di.dli_fbase;
        fprintf(stderr, "emu-site %d 0x%lx %llu\n", s.kind, (unsigned long)rel, s.n);
    }
}
}

// ------------------------------------------------------------------------------------------------ host runtime API
struct emu_stream { int dummy; };
struct emu_event { std::chrono::steady_clock::time_point t; };

hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "wave emulator (host)");
    strcpy(p->gcnArchName, "gfx950-emu");
    p->multiProcessorCount = emu::env_int("SNAPGPU_EMU_CUS", 1);
    p->totalGlobalMem = (size_t)16 << 30;
    p->sharedMemPerBlock = sizeof(lds);
    p->warpSize = 64;
    return hipSuccess;
}
hipError_t hipMalloc(void **p, size_t n)
{
    void *q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 1)) return hipErrorOutOfMemory;
    if (n <= ((size_t)64 << 20)) memset(q, 0xCD, n);      // device memory is not zeroed; make reliance on it visible (small blocks only)
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
// Page-locking of host ranges (the native tool pins its group buffers): nothing to lock here, but the calls are checked -- a range must not
// overlap one that is registered, an unregister must name a registered start -- and counted (SNAPGPU_EMU_PIN_REPORT=1: a line at exit).
namespace { std::mutex g_pin_mu; std::map<uintptr_t, size_t> g_pins; unsigned long long g_pin_calls = 0, g_pin_bytes = 0; bool g_pin_report_armed = false; }
hipError_t hipHostRegister(void *p, size_t n, unsigned)
{
    std::lock_guard<std::mutex> lk(g_pin_mu);
    const uintptr_t a = (uintptr_t)p;
    for (auto &r : g_pins) if (a < r.first + r.second && r.first < a + n) { fprintf(stderr, "emu: hipHostRegister of a range that overlaps a registered one\n"); abort(); }
    g_pins[a] = n; g_pin_calls++; g_pin_bytes += n;
    if (!g_pin_report_armed && getenv("SNAPGPU_EMU_PIN_REPORT")) {
        g_pin_report_armed = true;
        atexit([] { fprintf(stderr, "emu: hipHostRegister calls %llu, bytes %llu, still registered at exit %zu\n", g_pin_calls, g_pin_bytes, g_pins.size()); });
    }
    return hipSuccess;
}
hipError_t hipHostUnregister(void *p)
{
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (!g_pins.erase((uintptr_t)p)) { fprintf(stderr, "emu: hipHostUnregister of a pointer that is not registered\n"); abort(); }
    return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new emu_stream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new emu_event(); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }      // (every launch has run to its end when it returns)
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = (size_t)8 << 30; *total_b = (size_t)16 << 30; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated HIP error"; }
