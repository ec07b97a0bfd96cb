// paired_dev.h -- gfx950 instantiation of the paired-end path (paired.h): the device platform
// that maps the core's primitives onto the wavefront kernels (probe.h, lv.h, ag_win.h) and onto
// the single-end Aligner (align_single.h) for the chimeric fallback, plus the kernel itself.
//
// One wavefront owns one read pair.  LDS per wave = the single-end aligner's carve-out (its
// window, LV triangle and affine-gap rows are shared with the paired code) + both reads in both
// orientations + the four hit sets + the cold wave-uniform state (PEShared).  The candidate
// pools (ScoringCandidate / ScoringMateCandidate / MergeAnchor, sized like the reference's:
// min(-mcp, -H * seeds * 2) entries) and the Phase-4 candidate buffer live in a per-wave slab of
// HBM scratch.
#pragma once
#include "align_single.h"
#include "kernel_common.h"
#include "paired.h"
#include "paired_args.h"
#include <new>

// EXACT: the replay instantiation for pairs whose banded affine-gap traceback left the band (align_single.h: Aligner<.., EXACT>):
// ag_persist[0 / 1] are the wave's images of IntersectingPairedEndAligner's affineGap / reverseAffineGap traceback arrays, the
// single-end aligner of the chimeric fallback has its own two; the kernel zeroes all four before each pair.
template <int AGC, bool SEC = false, bool EXACT = false>
struct DevPL {
    typedef LP<DevPL> SelfPtr;
    LP<Aligner<AGC, SEC, EXACT>> al;      // single-end aligner of this wave (shares gw / lv_tri / ag_rows / ag_scratch)
    uint8_t *ag_persist0, *ag_persist1;
    uint32_t ag_hw0, ag_hw1;           // EXACT: bytes of each image written since it was last zeroed
    uint32_t ag_epoch, ag_tag;         // EXACT: pairs since the images were last cleared (1 .. 15), its tag bits (dev_common.h: bt_cell)
    // Phase-4 help (not in the exact replay: there the affine-gap calls of a pair are ordered through the traceback arrays they share)
    static const bool HELP = !EXACT;
    static const bool ALWAYS_COUNT_STALE = EXACT;
    G(PEHelpSlot) *help; uint32_t n_help; G(PEHelpSpec) *help_spec; uint32_t help_spec_cap;
    const G(uint32_t) *help_idle; bool help_eager;      // waves of the launch that have run out of pairs (nullptr: no helpers); publish regardless
    uint32_t cur_pair; int my_slot;
    G(unsigned long long) *diag;       // snapgpu_counters::reserved: [1] waits that ran into the watchdog, [2] what the last one saw
    // Cross-wave traffic is kept off the cache-wide fences (an agent-scope release writes back the whole L2 of the XCD, an acquire
    // invalidates it -- once per helped pair / per attach is fine, once per chunk of candidates is not): the speculative answers travel
    // in device-scope (sc1, write-through) stores and loads, ordered against the chunk's `done` count by a plain vmcnt(0) wait.
    // Words that are the target of read-modify-write atomics (state, next, done, helpers, the launch's done-pair count) are only ever
    // touched by read-modify-write atomics: set with an exchange whose return value is waited for, read with a compare-and-swap that
    // cannot succeed (`atomicAdd(p, 0)` is folded into an atomic LOAD by the compiler, and a load or a store may take another road to
    // memory than the atomic unit: a store followed by an add on the same word was seen to be applied after it).
    template <class P> static __device__ __forceinline__ uint32_t aload(P *p) { return first_u32(lane_id() == 0 ? atomicCAS(p, 0xFFFFFFF5u, 0xFFFFFFF5u) : 0u); }
    // (x: an lvalue in HBM, possibly typed as such -- dev_common.h: G(T))
    template <class T, class W> static __device__ __forceinline__ void spec_st(T &x, W v_in) {
        typedef typename strip_as<T>::type V;
        const V v = (V)v_in;
#ifdef SNAPGPU_WAVE_EMU
        if constexpr (sizeof(V) == 8) __atomic_store_n((uint64_t *)&x, __builtin_bit_cast(uint64_t, v), __ATOMIC_SEQ_CST);
        else __atomic_store_n((uint32_t *)&x, __builtin_bit_cast(uint32_t, v), __ATOMIC_SEQ_CST);
#else
        if constexpr (sizeof(V) == 8) __hip_atomic_store((GLB_AS uint64_t *)&x, __builtin_bit_cast(uint64_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store((GLB_AS uint32_t *)&x, __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    template <class T> static __device__ __forceinline__ typename strip_as<T>::type spec_ld(const T &x) {
        typedef typename strip_as<T>::type V;
#ifdef SNAPGPU_WAVE_EMU
        if constexpr (sizeof(V) == 8) return __builtin_bit_cast(V, first_u64(__atomic_load_n((const uint64_t *)&x, __ATOMIC_SEQ_CST)));
        else return __builtin_bit_cast(V, first_u32(__atomic_load_n((const uint32_t *)&x, __ATOMIC_SEQ_CST)));
#else
        if constexpr (sizeof(V) == 8) return __builtin_bit_cast(V, first_u64(__hip_atomic_load((const GLB_AS uint64_t *)&x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
        else return __builtin_bit_cast(V, first_u32(__hip_atomic_load((const GLB_AS uint32_t *)&x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
#endif
    }
    static __device__ __forceinline__ void stores_done() {                 // every store this wave has issued has been acknowledged
#ifndef SNAPGPU_WAVE_EMU
        __builtin_amdgcn_s_waitcnt(0x0F70);                                 // vmcnt(0), expcnt / lgkmcnt untouched
#endif
    }
    static __device__ __forceinline__ void fence_release() {
#ifdef SNAPGPU_WAVE_EMU
        __threadfence();
#else
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
    }
    static __device__ __forceinline__ void fence_acquire() {
#ifdef SNAPGPU_WAVE_EMU
        __threadfence();
#else
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    }
    static __device__ __forceinline__ void nap() {
#ifdef SNAPGPU_WAVE_EMU
        emu_yield();
#else
        __builtin_amdgcn_s_sleep(127);
#endif
    }
    // chunks of the slot's candidates, scored speculatively by this wave (which holds the pair's reads) until none are left
    // (the slot's fields are read with device-scope loads: a plain load may be served by this CU's L1 with what the slot held for an
    //  earlier pair -- even in the wave that has just stored them, if another wave of the CU had the line cached)
    // a helper's view of the slot (the owner passes its own values: it never reads back what it has just published)
    template <class Core> __device__ __forceinline__ void help_work(Core &core, G(PEHelpSlot) *slot) {
        const uint32_t n = spec_ld(slot->n);
        const int L = spec_ld(slot->limit), best = spec_ld(slot->best);
        const bool s0 = spec_ld(slot->skip0) != 0, s1 = spec_ld(slot->skip1) != 0;
        const G(snapgpu_paired_result) *agc = (const G(snapgpu_paired_result) *)(uintptr_t)spec_ld(*(const G(uint64_t) *)&slot->agc);
        const G(uint32_t) *order = (const G(uint32_t) *)(uintptr_t)spec_ld(*(const G(uint64_t) *)&slot->order);
        G(PEHelpSpec) *spec = (G(PEHelpSpec) *)(uintptr_t)spec_ld(*(const G(uint64_t) *)&slot->spec);
        help_chunks(core, slot, n, L, best, s0, s1, agc, order, spec);
    }
    template <class Core> __device__ __forceinline__ void help_chunks(Core &core, G(PEHelpSlot) *slot, uint32_t n, int L, int best, bool s0, bool s1,
                                                                      const G(snapgpu_paired_result) *agc, const G(uint32_t) *order, G(PEHelpSpec) *spec) {
        for (uint32_t it = 0; it <= n / PE_HELP_CHUNK + 1u; it++) {
            uint32_t c0 = 0;
            if (lane_id() == 0) c0 = atomicAdd(&slot->next, PE_HELP_CHUNK);
            c0 = first_u32(c0);
            if (c0 >= n) break;
            const uint32_t c1 = c0 + PE_HELP_CHUNK < n ? c0 + PE_HELP_CHUNK : n;
            for (uint32_t t = c0; t < c1; t++) core.spec_candidate(&agc[ld(order[t])], &spec[t], L, best, s0, s1);
            stores_done();
            if (lane_id() == 0) atomicAdd(&slot->done, c1 - c0);
        }
    }
    // is there anybody to publish for?  (a device-scope load of a word that only ever grows: a stale 0 just postpones the question)
    __device__ __forceinline__ bool help_wanted() const {
        if (help == nullptr) return false;
        if (help_eager) return true;
        return help_idle != nullptr && spec_ld(*help_idle) != 0u;
    }
    // candidates first .. n - 1 of the sorted list (spec is indexed by position in the list, so spec[first ..] are the ones filled in)
    template <class Core> __device__ __forceinline__ G(PEHelpSpec) *help_phase4(Core &core, uint32_t n, uint32_t first, int limit, int best, const bool skip[2]) {
        my_slot = -1;
        if (help == nullptr || n > help_spec_cap) return nullptr;
        int s = -1;
        if (lane_id() == 0) {
            for (uint32_t i = 0; i < n_help; i++) if (atomicCAS(&help[i].state, 0u, 3u) == 0u) { s = (int)i; break; }
        }
        s = (int)first_u32((uint32_t)s);
        if (s < 0) return nullptr;                          // every slot is taken: this pair goes through its list alone
        G(PEHelpSlot) *slot = &help[s];
        G(PEHelpSpec) *spec = help_spec + (size_t)s * help_spec_cap;
        if (lane_id() == 0) {
            const uint32_t w0 = atomicExch(&slot->next, first), w1 = atomicExch(&slot->done, 0u);
            if ((w0 ^ w1) == 0xFFFFFFF5u) atomicExch(&slot->done, 0u);          // (uses both return values: the exchanges have completed)
            spec_st(slot->pair, cur_pair); spec_st(slot->n, n);
            spec_st(slot->limit, (int32_t)limit); spec_st(slot->best, (int32_t)best);
            spec_st(slot->skip0, skip[0] ? 1u : 0u); spec_st(slot->skip1, skip[1] ? 1u : 0u);
            spec_st(*(G(uint64_t) *)&slot->agc, (uint64_t)(uintptr_t)core.agc); spec_st(*(G(uint64_t) *)&slot->order, (uint64_t)(uintptr_t)core.agc_order);
            spec_st(*(G(uint64_t) *)&slot->spec, (uint64_t)(uintptr_t)spec);
            fence_release();                                // the candidate records and their order, for the other XCDs
            atomicExch(&slot->state, 1u);
        }
        WAVE_SYNC();
        my_slot = s;
        help_chunks(core, slot, n, limit, best, skip[0], skip[1], core.agc, core.agc_order, spec);
        // Watchdog: a wait that lasts longer than ~1 s of shader clock is given up -- the slot is retired for the rest of the launch, the
        // pair goes through its list alone (the speculative answers are simply not used), and the event is counted; nothing can hang.
        const uint64_t t0 = wave_clock();
        bool gave_up = false;
        for (;;) {
            const uint32_t d = aload(&slot->done);
            if (d >= n - first) break;
            nap();
            if (wave_clock() - t0 > 2400000000ull) {
                if (lane_id() == 0 && diag) { atomicAdd(&diag[1], 1ull); diag[2] = 0x1000000000000000ull | ((unsigned long long)n << 32) | d; }
                gave_up = true; break;
            }
        }
        {   // close, THEN look at the attach count (see the helper loop): the exchange's return value is waited for first
            uint32_t was = 0;
            if (lane_id() == 0) was = atomicExch(&slot->state, gave_up ? 4u : 2u);
            was = first_u32(was);
            if (was != 1u) gave_up = true;                  // (cannot happen)
        }
        if (!gave_up) {
            for (;;) {
                if (aload(&slot->helpers) == 0u) break;
                nap();
                if (wave_clock() - t0 > 4800000000ull) {
                    if (lane_id() == 0 && diag) { atomicAdd(&diag[1], 1ull << 16); diag[2] = 0x2000000000000000ull | aload(&slot->helpers); }
                    gave_up = true; break;
                }
            }
            if (gave_up && lane_id() == 0) atomicExch(&slot->state, 4u);
        }
        if (gave_up) { my_slot = -1; return nullptr; }
        return spec;                                            // (read with spec_ld: no acquire fence needed)
    }
    // (diag[2], = snapgpu_counters::reserved[2]: lists published << 32 | answers the ordered walks took from speculative scoring)
    __device__ __forceinline__ void help_done(uint32_t answers_used) {
        if (my_slot >= 0 && lane_id() == 0) {
            atomicExch(&help[my_slot].state, 0u);
            if (diag) atomicAdd(&diag[2], (1ull << 32) | (unsigned long long)answers_used);
        }
        my_slot = -1;
    }
    GP<const DevTables> tab;
    AGParams agp;
    uint32_t kmax_lv;                  // the largest limit a paired-end LV call can have: what lv_big is sized for (the LDS triangle: al->cfg.kmax)
    uint16_t *lv_big;                  // per-wave HBM buffer of lv_lds_bytes(kmax_lv, RL) bytes for the calls whose limit exceeds the LDS triangle
    const uint8_t *g_bases[2], *g_quals[2];   // the pair's reads in global memory (for the single-end fallback)
    int g_len[2];
    WaveShared *ws;

    template <class T> static __device__ __forceinline__ typename strip_as<T>::type ld(const T &x) {
        typedef typename strip_as<T>::type V;
        const V v = x;
        if constexpr (sizeof(V) == 8) {
            return __builtin_bit_cast(V, first_u64(__builtin_bit_cast(uint64_t, v)));
        } else if constexpr (sizeof(V) == 4) {
            return __builtin_bit_cast(V, first_u32(__builtin_bit_cast(uint32_t, v)));
        } else {
            return (V)first_u32((uint32_t)v);
        }
    }
    template <class T, class W> static __device__ __forceinline__ void st(T &x, W v) {
        if (lane_id() == 0) x = (typename strip_as<T>::type)v;
        WAVE_SYNC();
    }
    // this pointer points into LDS (paired.h: PairedCore::L)
    static __device__ __forceinline__ void lds(const void *p) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SNAPGPU_WAVE_EMU)
        __builtin_assume(__builtin_amdgcn_is_shared(p));
#else
        (void)p;
#endif
    }
    static __device__ __forceinline__ int i32(int v) { return (int)first_u32((uint32_t)v); }
    static __device__ __forceinline__ double f64(double v) { return first_f64(v); }
    static __device__ __forceinline__ bool lane0() { return lane_id() == 0; }
    static __device__ __forceinline__ void sync() { WAVE_SYNC(); }
    // phase timers of the paired path (PECounters::cyc_*): only in a -DSNAPGPU_PHASE_TIMERS build (each clock read drains lgkmcnt)
#if defined(SNAPGPU_PHASE_TIMERS)
    static __device__ __forceinline__ uint64_t clock() { return wave_clock(); }
#else
    static __device__ __forceinline__ uint64_t clock() { return 0; }
#endif

    // ---- HashTableHitSet queries, one lookup per lane (lookups are few: <= 30).  Each reproduces the scalar loop of paired.h:
    // those loops keep the FIRST lookup (lowest index) among equal best locations and only accept locations > 0.
    // (off for the AGC == 0 variant -- reads longer than ~400 bp: with it ROCm 7.2's AMDGPU backend stops with "Illegal instruction
    //  detected: Operand has incorrect register class  V_CMP_NE_U32_e32 0, $src_shared_base"; the scalar queries are used there)
    static const bool FAST_HITSET = AGC != 0;
    static const bool SECONDARY = SEC;
    // The hits a walk is about to need are STAGED IN LDS, HS_W of them per lookup (the Landau-Vishkin block is idle during Phase 2: two window
    // blocks of max_seeds x HS_W words, one per set of the set pair being walked).  Every query below looks at hits[cur - 1 .. cur + 1] of its
    // lookups, and a walk moves `cur` forward one hit at a time, so a lane reloads its window -- one burst of HS_W independent loads from the
    // overflow table -- once per HS_W - 2 steps instead of waiting for one or two dependent HBM round trips in EVERY step (profiles/r05a: the
    // set intersection was 42 % of the paired kernel's wave cycles, ~3 600 cycles per hit).  Only the binary search of hs_next_le, which jumps,
    // probes the table directly.  Locations are 32-bit on the device (snapgpu.hip refuses larger genomes), so the arithmetic is too.
    uint32_t hs_w;                     // HS_W: 16 where 2 x max_seeds x 16 words (+ the mate ring) fit the Landau-Vishkin block, else 8
    // A walk's cursors, in REGISTERS for the length of the walk: lane n < n_used is lookup n of the set (its cursor, hit count, seed offset,
    // disjoint group, where its list is and which part of it is staged), the header's words are wave-uniform scalars.  The queries used to
    // re-read all of this from LDS -- eight loads and their waits per query, three queries per step -- and write `cur` / most_recent back.
    struct HSCursor {
        uint32_t cur, nh, so, wd, singleton; bool single, act;
        int32_t wbase; const G(uint32_t) *hits; uint32_t *win;      // per lane: staged window [wbase, wbase + hs_w) of hits[] (LDS)
        uint32_t n_used, recent; int cd;                         // wave-uniform
    };
    __device__ __forceinline__ void hs_begin_walk(PELookup *lk, PEHitSetHdr *h, int role, uint32_t max_seeds, HSCursor &c) {
        const int lane = lane_id();
        c.n_used = ld(h->n_used); c.cd = ld(h->cur_disjoint); c.recent = 0u;
        c.act = lane < (int)c.n_used;
        const PELookup *l = &lk[c.act ? lane : 0];
        c.cur = 0u; c.nh = c.act ? (uint32_t)l->n_hits : 0u; c.so = l->seed_offset; c.wd = l->which_disjoint;
        c.single = l->is_single != 0; c.singleton = l->singleton; c.hits = l->hits; c.wbase = -1;
        uint32_t *blk = (uint32_t *)(uint16_t *)al->lv_tri;
        c.win = blk + 2 * PE_MRING + (size_t)role * max_seeds * hs_w + (size_t)(c.act ? lane : 0) * hs_w;
    }
    // window of lane's lookup covers hit indices [lo, hi] (already clamped to the list)?  If not, stage [lo, lo + HS_W) -- all the loads of
    // all the lanes that need one go out together, one wait.
    template <int W> static __device__ __forceinline__ void hs_stage_w(HSCursor &c, bool need, uint32_t lo) {
        if (BALLOT(need)) {
            if (need) {
                const G(uint32_t) *src = c.hits + lo;
                const uint32_t n = c.nh - lo;
                uint32_t v[W];
#pragma unroll
                for (int j = 0; j < W; j++) v[j] = src[(uint32_t)j < n ? (uint32_t)j : n - 1u];      // (n >= 1; entries past the list repeat its last hit and are never read)
#pragma unroll
                for (int j = 0; j < W; j++) c.win[j] = v[j];
                c.wbase = (int32_t)lo;
            }
            WAVE_SYNC();
        }
    }
    __device__ __forceinline__ void hs_stage(HSCursor &c, bool want, uint32_t lo, uint32_t hi) {
        const bool need = want && hi >= lo && !c.single && (c.wbase < 0 || lo < (uint32_t)c.wbase || hi >= (uint32_t)c.wbase + hs_w);
        if (hs_w == 16u) hs_stage_w<16>(c, need, lo); else hs_stage_w<8>(c, need, lo);
    }
    // hits[i] of lane's lookup; i must be inside the staged window (or the lookup a singleton)
    static __device__ __forceinline__ uint32_t hs_get(const HSCursor &c, uint32_t i) { return c.single ? c.singleton : c.win[i - (uint32_t)c.wbase]; }
    // wave arg-max of v (> 0) over lanes with ok set; returns the winning lane (lowest lane among equals) or -1
    static __device__ __forceinline__ int wave_argmax32(bool ok, uint32_t v, uint32_t *best) {
        uint32_t m = ok ? v : 0u;
        for (int o = 16; o >= 1; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)m, o); m = t > m ? t : m; }      // (lookups live in lanes 0 .. 29)
        m = first_u32(m);
        const uint64_t who = BALLOT(ok && v == m);
        *best = m;
        return who ? __ffsll((long long)who) - 1 : -1;
    }
    __device__ __forceinline__ bool hs_first(HSCursor &c, int64_t *loc, uint32_t *seed_offset) {
        bool ok = c.act && c.nh > 0u;
        hs_stage(c, ok, 0u, c.nh > 1u ? 1u : 0u);
        const uint32_t v = ok ? hs_get(c, 0u) - c.so : 0u;
        ok = ok && v > 0u;
        uint32_t best;
        const int w = wave_argmax32(ok, v, &best);
        *loc = 0;
        if (w < 0) return true;
        *loc = (int64_t)best;
        *seed_offset = (uint32_t)__builtin_amdgcn_readlane((int)c.so, w);
        c.recent = best;
        return false;
    }
    __device__ __forceinline__ bool hs_next_lower(HSCursor &c, int64_t *loc, uint32_t *seed_offset) {
        bool live = c.act && c.cur != c.nh;
        hs_stage(c, live, c.cur > 0u ? c.cur - 1u : 0u, c.cur + 1u < c.nh ? c.cur + 1u : c.cur);
        uint32_t hv = live ? hs_get(c, c.cur) : 0u;
        if (live && hv - c.so == c.recent) {
            c.cur++;
            live = c.cur != c.nh;
            if (live) hv = hs_get(c, c.cur);
        }
        const bool ok = live && hv >= c.so && hv - c.so > 0u;
        uint32_t best;
        const int w = wave_argmax32(ok, hv - c.so, &best);
        if (w < 0) return false;
        *loc = (int64_t)best;
        *seed_offset = (uint32_t)__builtin_amdgcn_readlane((int)c.so, w);
        c.recent = best;
        return true;
    }
    __device__ __forceinline__ bool hs_next_le(HSCursor &c, int64_t max_loc, int64_t *loc, uint32_t *seed_offset) {
        const int64_t max_this = max_loc + c.so;
        const uint32_t wcap = hs_w;
        // hits[i], from the window where it holds i (the answer is often a few hits further on), else from the table
        auto peek = [&](int64_t i) -> int64_t {
            if (c.single) return (int64_t)c.singleton;
            if (c.wbase >= 0 && i >= (int64_t)c.wbase && i < (int64_t)c.wbase + (int64_t)wcap) return (int64_t)c.win[i - c.wbase];
            return (int64_t)c.hits[i];
        };
        int64_t lo = c.act ? (int64_t)c.cur : 0, hi = c.act ? (int64_t)c.nh - 1 : -1;
        if (c.act && !c.single && c.wbase >= 0) {        // the last staged hit at or below the bound: the answer is inside the window
            const int64_t wend = ((int64_t)c.wbase + (int64_t)wcap < (int64_t)c.nh ? (int64_t)c.wbase + (int64_t)wcap : (int64_t)c.nh) - 1;
            if (wend >= lo && (int64_t)c.win[wend - c.wbase] <= max_this) hi = wend;
        }
        bool found = false;
        uint32_t v = 0;
        while (BALLOT(lo <= hi && !found)) {
            if (lo <= hi && !found) {
                const int64_t probe = (lo + hi) / 2;
                const int64_t ph = peek(probe);
                if (ph <= max_this && (probe == 0 || peek(probe - 1) > max_this)) {
                    found = true; v = (uint32_t)ph - c.so;
                    c.cur = (uint32_t)probe;
                } else if (ph > max_this) lo = probe + 1; else hi = probe - 1;
            }
        }
        if (c.act && !found) c.cur = c.nh;
        uint32_t best;
        const int w = wave_argmax32(found && v > 0u, v, &best);
        if (w < 0) return false;
        *loc = (int64_t)best;
        *seed_offset = (uint32_t)__builtin_amdgcn_readlane((int)c.so, w);
        c.recent = best;
        return true;
    }
    __device__ __forceinline__ uint32_t hs_best_possible(HSCursor &c, uint32_t *exhausted) {
        const int64_t target = (int64_t)c.recent + c.so;
        hs_stage(c, c.act && c.nh > 0u, c.cur > 0u ? c.cur - 1u : 0u, c.cur < c.nh ? c.cur : c.nh - 1u);
        bool close = false;
        if (c.act) {
            if (c.cur != c.nh) { const int64_t a = (int64_t)hs_get(c, c.cur); const int64_t d = a > target ? a - target : target - a; close = d <= PE_MERGE_DIST; }
            if (!close && c.cur != 0u) { const int64_t a = (int64_t)hs_get(c, c.cur - 1u); const int64_t d = a > target ? a - target : target - a; close = d <= PE_MERGE_DIST; }
        }
        uint32_t best = 0;
        for (int d = 0; d <= c.cd; d++) {
            uint32_t m = ld(exhausted[d]) + (uint32_t)__popcll(BALLOT(c.act && !close && c.wd == (uint32_t)d));
            if (m > best) best = m;
        }
        return best;
    }

    // Phase 2a (IntersectingPairedEndAligner.cpp:743-801): entries of a strictly descending list that lie within maxKForIndels of one
    // another raise each other's largestBigIndelDetected to their distance.  The reference's two-pointer loop (paired.h has it) compares
    // entry i, as `top`, once -- with the farthest earlier entry within the distance, B(i) -- and, as `bottom`, with every later entry up to
    // the farthest within the distance, T(i), but only from the moment its predecessor stopped being `bottom`, i.e. only if T(i) > T(i - 1).
    // So big(i) = max(loc[B(i)] - loc[i], T(i) > max(i, T(i - 1)) ? loc[i] - loc[T(i)] : 0)  -- checked against the loop on 200 000 random
    // lists, and by every paired-end parity test -- which 64 entries compute side by side instead of one entry per two dependent loads.
    template <class R> __device__ __forceinline__ void hint_indels(R *r, uint32_t first, uint32_t n, int K) {
        const int lane = lane_id();
        r += first;
        uint32_t carry_t = 0;                                   // T(i - 1) for the first lane of a chunk (0 for i = 0)
        for (uint32_t base = 0; base < n; base += WAVE) {
            const uint32_t i = base + (uint32_t)lane;
            const bool act = i < n;
            const int64_t li = act ? (int64_t)r[i].loc : 0;
            uint32_t t = i, b = i;
            for (uint32_t step = 1; step < (uint32_t)K + 1u; step++) {
                const bool fw = act && t == i + step - 1u && i + step < n;
                const bool bw = act && b == i - step + 1u && i >= step;
                if (!BALLOT(fw || bw)) break;
                if (fw && li - (int64_t)r[i + step].loc < (int64_t)K) t = i + step;
                if (bw && (int64_t)r[i - step].loc - li < (int64_t)K) b = i - step;
            }
            uint32_t prev_t = (uint32_t)__shfl_up((int)t, 1);
            if (lane == 0) prev_t = carry_t;
            carry_t = (uint32_t)__builtin_amdgcn_readlane((int)t, WAVE - 1);
            int64_t big = 0;
            if (act && b < i) big = (int64_t)r[b].loc - li;
            if (act && t > i && t > prev_t) { const int64_t d = li - (int64_t)r[t].loc; big = d > big ? d : big; }
            if (act && big > 0) r[i].big_indel = (decltype(r[i].big_indel))big;
            WAVE_SYNC();
        }
    }

    __device__ __forceinline__ bool lookup(const uint8_t *text, PEHits out[2]) {
        SeedBits seed = pack_seed(text, al->ix.seed_len);
        if (!seed.valid) return false;
        HitList hl[2];
        lookup_seed(al->ix, seed, hl);
        out[0].hits = (const G(uint32_t) *)hl[0].hits; out[0].n_hits = hl[0].n_hits; out[0].singleton = hl[0].singleton;
        out[1].hits = (const G(uint32_t) *)hl[1].hits; out[1].n_hits = hl[1].n_hits; out[1].singleton = hl[1].singleton;
        al->cnt().lookups++;
        al->cnt().slots += hl[0].slots + hl[1].slots;
        return true;
    }
    __device__ __forceinline__ uint32_t wrapped_seed(uint32_t wrap) const { return tab->wrapped_seed[wrap]; }
    __device__ __forceinline__ uint32_t count_n(const uint8_t *b, int len) const {
        uint32_t n = 0;
        const int lane = lane_id();
        for (int i0 = 0; i0 < len; i0 += WAVE) {
            int i = i0 + lane;
            n += (uint32_t)__popcll(BALLOT(i < len && b[i] == 'N'));
        }
        return n;
    }
    // stage genome[loc - WIN_PAD, loc + read_len + WIN_PAD) into the shared LDS window
    __device__ __forceinline__ const uint8_t *window(int64_t loc, int read_len) {
        al->read_len = read_len;
        al->stage_window(loc);
        uint8_t *g = al->gw;
        return g + WIN_PAD;
    }
    __device__ __forceinline__ bool is_alt(int64_t loc) const { return al->is_alt(loc); }
    __device__ __forceinline__ bool substring_ok(int64_t loc, int64_t len) const { return al->substring_ok(loc, len); }
    __device__ __forceinline__ int mapq(double p_all, double p_best, int popular) const { return compute_mapq(tab, p_all, p_best, popular); }
    __device__ __forceinline__ double seed_prob() const { return tab->seed_prob; }
    __device__ __forceinline__ double phred(uint8_t q) const { return tab->phred[q]; }
    __device__ __forceinline__ double indel(int n) const { return tab->indel[n]; }
    __device__ __forceinline__ double perfect(int n) const { return tab->perfect[n]; }

    __device__ __forceinline__ LVOut lv(int st, const uint8_t *P, const uint8_t *Q, int plen, const uint8_t *T, int tlen, int k) {
        const LdsSeq Ps = lds_seq(ByteSeq{P, st}), Qs = lds_seq(ByteSeq{Q, st}), Ts = lds_seq(ByteSeq{T, st});      // (reads, qualities and the window are LDS: lv.h LdsSeq)
        // the LDS triangle serves limits up to cfg.kmax; a larger one (indel-hinted candidates only) works in the wave's HBM buffer
        LVResult r = k <= (int)al->cfg.kmax ? lv_compute(Ps, Qs, plen, Ts, tlen, k, al->lv_tri, al->cfg.kmax, tab, al->cfg.RL)
                                            : lv_compute<false>(Ps, Qs, plen, Ts, tlen, k, lv_big, kmax_lv, tab, al->cfg.RL);
        LVOut o;
        o.score = i32(r.score); o.mp = f64(r.match_probability); o.net_indel = i32(r.net_indel);
        o.total_indels = i32(r.total_indels); o.text_span = i32(r.text_span);
        return o;
    }
    __device__ __forceinline__ AGOut ag(bool banded, int st, const uint8_t *P, const uint8_t *Q, int plen, const uint8_t *T, int tlen, int lim,
                                        int read_len, bool is_rc, int use_clip) {
        const LdsSeq Ps = lds_seq(ByteSeq{P, st}), Qs = lds_seq(ByteSeq{Q, st}), Ts = lds_seq(ByteSeq{T, st});
        if constexpr (EXACT) {
            int nv, sl, ns;
            ag_dims(banded, plen, lim > 126 ? 126 : (lim < 0 ? 0 : lim), &nv, &sl, &ns);
            // (a call with a negative limit or an empty text writes nothing -- and its text length must not be trusted; nothing is ever
            //  written past the image either: ag_dispatch stops the kernel first)
            uint32_t ext = (lim >= 0 && tlen > 0) ? (uint32_t)tlen * (uint32_t)(ns * sl) : 0u;
            const uint32_t image = (uint32_t)ag_scratch_bytes(al->cfg.RL);
            if (ext > image) ext = image;
            if (st == 1) ag_hw0 = ext > ag_hw0 ? ext : ag_hw0; else ag_hw1 = ext > ag_hw1 ? ext : ag_hw1;
        }
        if (++al->ag_calls_unit == WAVE_PRIO_HEAVY_AFTER * 8) wave_set_priority(1);          // (a pair: both mates, both halves, Phases 3 and 4)
        AGResult a = ag_dispatch<AGC, EXACT>(banded, st, agp, Ps, Qs, plen, Ts, tlen, lim, read_len, is_rc, use_clip, al->ag_rows,
                                             EXACT ? (st == 1 ? ag_persist0 : ag_persist1) : (uint8_t *)al->ag_scratch, al->cfg.RL, tab, EXACT ? ag_tag : 0u);
        AGOut o;
        o.ag_score = i32(a.ag_score); o.text_offset = i32(a.text_offset); o.pattern_offset = i32(a.pattern_offset);
        o.n_edits = i32(a.n_edits); o.mp = f64(a.match_probability); o.stale = i32(a.stale_reads);
        return o;
    }
    // Stable counting sort of the Phase-4 candidates by pair score (the key kept in `reserved`, < 512): histogram and
    // running bases in LDS (the LV triangle is idle here), entries scattered 64 at a time in index order.
    __device__ __forceinline__ void sort_candidates(const G(snapgpu_paired_result) *c, uint32_t n, G(uint32_t) *order) {
        const int lane = lane_id();
        uint32_t *base = (uint32_t *)(uint16_t *)al->lv_tri;                     // the LV block of LDS: lv_lds_bytes(kmax >= 22, RL) >= 2 232 bytes; 512 counters needed
        uint32_t *hist = base;
        for (int k = lane; k < 512; k += WAVE) hist[k] = 0;
        WAVE_SYNC();
        for (uint32_t j0 = 0; j0 < n; j0 += WAVE) {
            uint32_t j = j0 + (uint32_t)lane;
            if (j < n) atomicAdd(&hist[c[j].reserved & 511u], 1u);
        }
        WAVE_SYNC();
        // exclusive prefix over 512 counters: 8 per lane, then a wave scan
        uint32_t loc[8], sum = 0;
        for (int t = 0; t < 8; t++) { loc[t] = hist[lane * 8 + t]; sum += loc[t]; }
        uint32_t incl = sum;
        for (int o = 1; o < WAVE; o <<= 1) { uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += v; }
        uint32_t run = incl - sum;
        WAVE_SYNC();
        for (int t = 0; t < 8; t++) { hist[lane * 8 + t] = run; run += loc[t]; }
        WAVE_SYNC();
        for (uint32_t j0 = 0; j0 < n; j0 += WAVE) {
            uint32_t j = j0 + (uint32_t)lane;
            const bool act = j < n;
            const uint32_t key = act ? (c[j].reserved & 511u) : 0xffffffffu;
            uint64_t todo = BALLOT(act);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
                const uint64_t same = BALLOT(act && key == k);
                const uint32_t b = base[k];
                if (act && key == k) order[b + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = j;
                WAVE_SYNC();
                if (lane == 0) base[k] = b + (uint32_t)__popcll(same);
                WAVE_SYNC();
                todo &= ~same;
            }
        }
        WAVE_SYNC();
    }
    // BaseAligner::AlignRead with setMaxK(max_k) (ChimericPairedEndAligner.cpp:278-310); hamming: the retry of :330-360
    // (AlignRead(..., useHamming) followed by BaseAligner::alignAffineGap on the candidates it collected).
    // Returns the number of secondary results the read has (SEC only); the first min(that, sec_room) are copied to sec_out.
    __device__ __forceinline__ uint32_t align_single(int r, int max_k, bool hamming, snapgpu_single_result &res, snapgpu_single_result &alt,
                                                     bool want_secondary, G(snapgpu_single_result) *sec_out, uint32_t sec_room, uint32_t room32) {
        (void)want_secondary; (void)room32;
        al->max_k = (uint32_t)max_k;
        if (!hamming) {
            al->template align_read_inner<false>(g_bases[r], g_quals[r], g_len[r]);
        } else {
            al->template align_read_inner<true>(g_bases[r], g_quals[r], g_len[r]);
            if (!al->agc_overflow) al->align_affine_gap(ws->ag_all, ws->ag_non_alt);
            al->primary().reserved = (al->ag_stale & 0x3fffffffu) | (al->ag_replay ? 0x40000000u : 0u);
        }
        WAVE_SYNC();
        res = al->primary();
        alt = al->first_alt();
        if (al->agc_overflow) res.reserved |= 0x80000000u;
        WAVE_SYNC();
        if constexpr (SEC) {
            if (al->sec_overflow) res.reserved |= 0x80000000u;                   // (sized so that it cannot happen)
            const uint32_t n = al->n_sec;
            const uint32_t n_out = n < sec_room ? n : sec_room;
            const int lane = lane_id();
            const int nd = (int)(sizeof(snapgpu_single_result) / 4);
            for (uint32_t k0 = 0; k0 < n_out; k0 += 2) {                         // two 22-dword records per pass
                const uint32_t k = k0 + (uint32_t)(lane >> 5);
                const int w = lane & 31;
                if (k < n_out && w < nd) ((G(uint32_t) *)&sec_out[k])[w] = ((const uint32_t *)&al->sec[al->sec_ord[k]])[w];
            }
            WAVE_SYNC();
            return n;
        } else {
            (void)sec_out; (void)sec_room;
            return 0;
        }
    }
    // secondary candidates the last align_single collected before finalizeSecondaryResults filtered them
    __device__ __forceinline__ uint32_t single_raw_secondary() const {
        if constexpr (SEC) return al->n_sec_raw; else return 0;
    }
};

// Scalar-heavy control flow; 3 waves per SIMD for the 192-position variant, 2 for the others: what its LDS footprint (under 10 KB per
// wave since round 3: Landau-Vishkin triangle for limits <= 22 only, the register affine-gap forms' tables instead of the LDS form's rows)
// allows and what was measured: paired_args.h, SNAPGPU_PAIRED_WAVES_PER_SIMD.
template <int AGC, bool SEC, bool EXACT = false>
__global__ __launch_bounds__(256, SNAPGPU_PAIRED_WAVES_PER_SIMD(AGC)) void k_align_paired(PairedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t wave_slot = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave_in_block;
    const LdsLayout SL = lds_layout(a.scfg.RL, a.scfg.num_weight_lists, a.scfg.kmax, a.scfg.ag_lds);
    const PairedLds PLd = paired_lds_layout(SL.total, a.scfg.RL, a.pcfg.max_seeds);
    uint8_t *my = lds + (size_t)wave_in_block * PLd.total;
    uint8_t *sc = a.scratch + (size_t)wave_slot * a.stride;

    WaveShared *ws = (WaveShared *)(my + SL.shared);
    // the wave's frame (paired_args.h: PE_FRAME_BYTES): the three objects are constructed in LDS
    typedef Aligner<AGC, SEC, EXACT> AL_T; typedef DevPL<AGC, SEC, EXACT> PL_T; typedef PairedCore<PL_T> CORE_T;
    constexpr uint32_t AL_SZ = ((uint32_t)sizeof(AL_T) + 15u) & ~15u, PL_SZ = ((uint32_t)sizeof(PL_T) + 15u) & ~15u;
    static_assert(AL_SZ + PL_SZ + sizeof(CORE_T) <= PE_FRAME_BYTES, "the wave's frame does not fit PE_FRAME_BYTES");
    uint8_t *frame = my + PLd.frame;
    AL_T &al = *new (frame) AL_T(a.ix, a.tab, a.scfg, ws);
    al.rd[0] = my + SL.rd0; al.rd[1] = my + SL.rd1;
    al.ql[0] = my + SL.ql0; al.ql[1] = my + SL.ql1;
    al.gw = my + SL.gw;
    al.seed_used = (uint32_t *)(my + SL.seed_used);
    al.wl_next = (uint16_t *)(my + SL.wl_next);
    al.wl_prev = (uint16_t *)(my + SL.wl_prev);
    al.lv_tri = (uint16_t *)(my + SL.lv);
    al.ag_rows = (int16_t *)(my + SL.ag);
    al.rp = (unsigned long long *)(my + SL.rp); al.tp = (unsigned long long *)(my + SL.tp); al.lvp = (unsigned long long *)(my + SL.lvp);
    al.heads = (uint16_t *)sc;
    al.pool = (Elem *)(sc + (size_t)a.scfg.ht_size * 2);
    al.ag_scratch = sc + (size_t)a.scfg.ht_size * 2 + (size_t)a.scfg.pool_size * sizeof(Elem);
    al.agc = a.single_agc_cap ? (snapgpu_single_result *)(sc + a.off_single_agc) : nullptr;    // no buffer without affine gap (PairedAligner.cpp:570-577)
    al.agc_cap = a.single_agc_cap;
    al.cnt() = WaveCounters{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    al.ag_calls_unit = 0;
    al.se_slots = nullptr; al.se_n_slots = 0; al.se_spec = nullptr; al.se_spec_cap = 0; al.se_ctl = nullptr; al.se_eager = 0; al.se_diag = nullptr;
    al.se_items = nullptr; al.se_first = nullptr; al.se_slot = -1; al.se_n = 0; al.se_tried = 0; al.cur_read = 0; al.se_mine = nullptr;      // (se_help.h: single-end path only)

    if constexpr (SEC) {            // secondary-result lists of the single-end aligner (as k_align_single<.., true> lays them out)
        al.sec_cfg = a.ssec_cfg;
        al.sec = (snapgpu_single_result *)(sc + a.off_ssec);
        al.sec_key = (uint32_t *)(sc + a.off_ssec + (size_t)a.ssec_cfg.cap * sizeof(snapgpu_single_result));
        al.sec_ord = al.sec_key + 2 * (size_t)a.ssec_cfg.cap;
        al.n_sec = 0; al.n_sec_raw = 0; al.sec_overflow = 0;
    }
    PL_T &pl = *new (frame + AL_SZ) PL_T;
    pl.al = &al; pl.tab = a.tab; pl.ws = ws; pl.kmax_lv = a.kmax_lv; pl.lv_big = (uint16_t *)(sc + a.off_lv_big);
    al.ag_persist0 = al.ag_persist1 = pl.ag_persist0 = pl.ag_persist1 = nullptr;
    al.ag_hw0 = al.ag_hw1 = pl.ag_hw0 = pl.ag_hw1 = 0;
    pl.ag_epoch = 0; pl.ag_tag = 0;
    pl.help = EXACT ? nullptr : (G(PEHelpSlot) *)a.help; pl.n_help = a.n_help; pl.help_spec = (G(PEHelpSpec) *)a.help_spec; pl.help_spec_cap = a.help_spec_cap;
    pl.help_idle = a.help_done ? (const G(uint32_t) *)(a.help_done + 1) : nullptr; pl.help_eager = a.help_eager != 0;
    pl.cur_pair = 0; pl.my_slot = -1; pl.diag = (G(unsigned long long) *)(a.counters + 13);
    // the Landau-Vishkin block of LDS during Phase 2: [mate ring: 2 x PE_MRING words][two window blocks of max_seeds x hs_w words]
    // (snapgpu_enable_paired: kmax >= 22 -> 2 232 bytes, max_seeds <= 30, so windows of 8 always fit)
    pl.hs_w = 8u * PE_MRING + 2u * a.pcfg.max_seeds * 16u * 4u <= lv_lds_bytes(a.scfg.kmax, a.scfg.RL) ? 16u : 8u;
    if constexpr (EXACT) {
        uint8_t *pb = a.persist + (size_t)wave_slot * a.persist_stride;
        const size_t q = (size_t)(a.persist_stride / 4);
        pl.ag_persist0 = pb; pl.ag_persist1 = pb + q; al.ag_persist0 = pb + 2 * q; al.ag_persist1 = pb + 3 * q;
    }
    pl.agp = AGParams{a.scfg.match_reward, a.scfg.sub_penalty, a.scfg.gap_open, a.scfg.gap_extend, a.scfg.five_bonus, a.scfg.three_bonus};

    CORE_T &core = *new (frame + AL_SZ + PL_SZ) CORE_T(pl, a.pcfg);
    core.help_min = a.help_min;
    core.lk = (PELookup *)(my + PLd.lk);
    core.exhausted = (uint32_t *)(my + PLd.exhausted);
    core.miss = (uint32_t *)(my + PLd.miss);
    core.hs = (PEHitSetHdr *)(my + PLd.hs);
    core.list_head = (int32_t *)(my + PLd.list_head);
    core.seed_used = (uint32_t *)(my + PLd.seed_used);
    core.sh = (PEShared *)(my + PLd.sh);
    core.mring = (uint32_t *)(my + SL.lv);
    core.cand = (G(PECand) *)(sc + a.off_cand);
    core.mate[0] = (G(PEMate) *)(sc + a.off_mate0);
    core.mate[1] = (G(PEMate) *)(sc + a.off_mate1);
    core.anchor = (G(PEAnchor) *)(sc + a.off_anchor);
    core.agc = (G(snapgpu_paired_result) *)(sc + a.off_agc);
    core.agc_order = (G(uint32_t) *)(sc + a.off_agc_order);
    core.sec = nullptr; core.sec_ord = nullptr; core.sec_key = nullptr; core.n_sec = 0;
    core.ssec_out = nullptr; core.ssec_stride = 0; core.n_ssec[0] = core.n_ssec[1] = 0; core.ref_dep = 0;
    if constexpr (SEC) {
        core.sec = (G(snapgpu_paired_result) *)(sc + a.off_sec);
        core.sec_ord = (G(uint32_t) *)(sc + a.off_sec_ord);
        core.sec_key = (G(uint32_t) *)(sc + a.off_sec_key);
    }
    core.S()->cnt = PECounters{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint8_t *prd = my + PLd.rd, *pql = my + PLd.ql;
    const uint32_t RL = a.scfg.RL;
    uint64_t n_done = 0;

    const uint32_t n_total = a.remap ? first_u32(*a.n_remap) : a.n_pairs;
    // both reads of pair i, both orientations, into this wave's LDS
    auto load_pair = [&](uint32_t i) {
        for (int r = 0; r < 2; r++) {
            const uint64_t b = first_u64(a.offsets[2 * (size_t)i + r]), e = first_u64(a.offsets[2 * (size_t)i + r + 1]);
            int len = (int)(e - b);
            if (len < 0 || len > (int)RL) len = 0;          // (the host rejects such batches; never write past the LDS buffers)
            uint8_t *f = prd + (2 * r) * RL, *rc = prd + (2 * r + 1) * RL, *qf = pql + (2 * r) * RL, *qr = pql + (2 * r + 1) * RL;
            for (int j0 = 0; j0 < len; j0 += WAVE) {
                int j = j0 + lane;
                if (j < len) {
                    uint8_t bb = a.bases[b + j], qq = a.quals[b + j];
                    f[j] = bb; qf[j] = qq;
                    rc[len - 1 - j] = rc_base(bb); qr[len - 1 - j] = qq;
                }
            }
            core.rd[r][0] = f; core.rd[r][1] = rc; core.ql[r][0] = qf; core.ql[r][1] = qr;
            core.read_len[r] = len;
            pl.g_bases[r] = a.bases + b; pl.g_quals[r] = a.quals + b; pl.g_len[r] = len;
        }
        WAVE_SYNC();
    };
    using XP = DevPL<AGC, SEC, EXACT>;
    while (true) {
        uint32_t i = 0;
        if (EXACT && a.rq_mode == 2u) {
            // pairs arrive while the main pass runs (PairedArgs::rq): take the next entry of the list, or wait for one, until the main pass
            // has finished all of its pairs and every entry has been taken
            const uint64_t t_wait0 = wave_clock();
            uint32_t got = 0xFFFFFFFFu;
            for (uint32_t round = 0;; round++) {
                const uint32_t h = XP::aload(a.rq + 1), t = XP::aload(a.rq);
                if (h < t) {
                    uint32_t was = 0;
                    if (lane == 0) was = atomicCAS(a.rq + 1, h, h + 1u);
                    if (first_u32(was) != h) continue;                    // somebody else took entry h
                    for (uint32_t w = 0;; w++) {                           // (the index lands right after the count: a very short wait)
                        got = XP::aload(a.rq_list + h);
                        if (got != 0xFFFFFFFFu || w > 4000000u) break;
                        XP::nap();
                    }
                    if (got == 0xFFFFFFFFu) { if (lane == 0) atomicAdd(a.rq + 3, 1u); continue; }      // (never seen; the pair stays flagged for the pass after)
                    break;
                }
                if (XP::aload(a.rq + 2) >= a.n_pairs) {                    // the main pass is through: is the list?  (appended before counted)
                    if (XP::aload(a.rq + 1) >= XP::aload(a.rq)) break;
                    continue;
                }
                if ((round & 63u) == 63u && wave_clock() - t_wait0 > 240000000000ull) {       // ~100 s: give up, leave a trace
                    if (lane == 0) atomicAdd(a.rq + 3, 0x10000u);
                    break;
                }
                XP::nap(); XP::nap(); XP::nap(); XP::nap();
            }
            if (got == 0xFFFFFFFFu) break;
            XP::fence_acquire();
            i = got;
        } else {
            if (lane == 0) i = atomicAdd(a.work_counter, 1u);
            i = first_u32(i);
            if (i >= n_total) break;
            if (a.remap) i = first_u32(a.remap[i]);
        }
        if constexpr (EXACT) {          // newly constructed reference aligners: all four traceback arrays read as zero -- cleared once per
            if (++pl.ag_epoch == 16u) { // fifteen pairs; in between, cells that do not carry this pair's tag read as zero (dev_common.h: bt_cell)
                if (pl.ag_hw0) wave_zero16(pl.ag_persist0, ((size_t)pl.ag_hw0 + 15) & ~(size_t)15);
                if (pl.ag_hw1) wave_zero16(pl.ag_persist1, ((size_t)pl.ag_hw1 + 15) & ~(size_t)15);
                if (al.ag_hw0) wave_zero16(al.ag_persist0, ((size_t)al.ag_hw0 + 15) & ~(size_t)15);
                if (al.ag_hw1) wave_zero16(al.ag_persist1, ((size_t)al.ag_hw1 + 15) & ~(size_t)15);
                pl.ag_hw0 = pl.ag_hw1 = al.ag_hw0 = al.ag_hw1 = 0; pl.ag_epoch = 1u;
                WAVE_SYNC();
            }
            pl.ag_tag = al.ag_tag = bt_tag_bits(pl.ag_epoch);
        }
        load_pair(i);
        pl.cur_pair = i;
        if (al.ag_calls_unit >= WAVE_PRIO_HEAVY_AFTER * 8) wave_set_priority(0);
        al.ag_calls_unit = 0;
        al.ag_obj_used0 = al.ag_obj_used1 = 0;          // the chimeric fallback's single-end aligner is one object for the whole pair
        {   // zero both results (fields the reference leaves unset read as 0 here)
            uint32_t *z0 = (uint32_t *)&core.S()->res, *z1 = (uint32_t *)&core.S()->alt;
            const int nd = (int)(sizeof(snapgpu_paired_result) / 4);
            if (lane < nd) { z0[lane] = 0; z1[lane] = 0; }
        }
        WAVE_SYNC();
        if constexpr (SEC) {
            core.ssec_out = a.ssec_out_stride ? (G(snapgpu_single_result) *)(a.single_secondary + (size_t)i * a.ssec_out_stride) : nullptr;
            core.ssec_stride = a.ssec_out_stride;
        }
        core.align_pair(a.max_k_paired, a.max_k_single);
        core.S()->res.flags = (core.overflow ? SNAPGPU_PAIR_POOL_OVERFLOW : 0) | (core.ref_dep ? SNAPGPU_PAIR_REF_BUFFER_DEPENDENT : 0) |
                             ((EXACT || core.stale_later || (a.dbg_flag_every != 0u && i % a.dbg_flag_every == 0u)) ? SNAPGPU_PAIR_EXACT_REPLAY : 0) |
                             ((EXACT && a.rq_mode == 2u) ? SNAPGPU_PAIR_REPLAYED_BESIDE : 0);
        WAVE_SYNC();
        if constexpr (SEC) {        // paired secondary results: sec[sec_ord[k]] -> secondary[i * stride + k]
            const uint32_t n_sec = core.overflow ? 0u : core.n_sec;
            if (lane == 0) {
                a.n_secondary[i] = n_sec;
                a.n_single_secondary[2 * (size_t)i] = core.overflow ? 0u : core.n_ssec[0];
                a.n_single_secondary[2 * (size_t)i + 1] = core.overflow ? 0u : core.n_ssec[1];
            }
            const uint32_t n_out = n_sec < a.sec_out_stride ? n_sec : a.sec_out_stride;
            const int nd = (int)(sizeof(snapgpu_paired_result) / 4);       // 52 dwords
            for (uint32_t k = 0; k < n_out; k++) {
                const G(uint32_t) *src = (const G(uint32_t) *)core.secondary(k);
                uint32_t *dst = (uint32_t *)&a.secondary[(size_t)i * a.sec_out_stride + k];
                if (lane < nd) dst[lane] = src[lane];
            }
            WAVE_SYNC();
        }
        {
            const uint32_t *src = (const uint32_t *)&core.S()->res, *src2 = (const uint32_t *)&core.S()->alt;
            uint32_t *dst = (uint32_t *)&a.primary[i];
            const int nd = (int)(sizeof(snapgpu_paired_result) / 4);
            if (lane < nd) dst[lane] = src[lane];
            if (a.first_alt) { uint32_t *dst2 = (uint32_t *)&a.first_alt[i]; if (lane < nd) dst2[lane] = src2[lane]; }
        }
        WAVE_SYNC();
        n_done++;
        if (!EXACT && a.rq_mode == 1u) {
            // a flagged pair goes to the exact kernel that runs beside this one: the pair's results first (the exact kernel writes the
            // same records), then the index, then the count of finished pairs (the exact kernel leaves when that count is complete
            // and the list is empty, so the index must be there before the pair is counted)
            const uint32_t fl = first_u32(core.S()->res.flags);
            if ((fl & SNAPGPU_PAIR_EXACT_REPLAY) != 0u && (fl & SNAPGPU_PAIR_POOL_OVERFLOW) == 0u) {
                XP::stores_done();
                XP::fence_release();
                if (lane == 0) {
                    const uint32_t slot = atomicAdd(a.rq, 1u);
                    const uint32_t old = atomicExch(a.rq_list + slot, i);
                    if (old != 0xFFFFFFFFu) atomicAdd(a.rq + 3, 0x1000000u);          // (uses the return value: the exchange has completed)
                }
            }
            if (lane == 0) atomicAdd(a.rq + 2, 1u);
        }
        if (!EXACT && a.help_done != nullptr && lane == 0) atomicAdd(a.help_done, 1u);
    }
    if constexpr (!EXACT) {
        // Out of pairs: until every pair of the launch is done, score Phase-4 candidates of the pairs that asked for help.
        if (a.help != nullptr && a.help_done != nullptr) {
            if (lane == 0) atomicAdd(a.help_done + 1, 1u);                // one more idle wave: pairs in Phase 4 start publishing
            const uint64_t t_idle0 = wave_clock();
            for (uint32_t round = 0;; round++) {
                if ((round & 7u) == 0u) {
                    const uint32_t dn = DevPL<AGC, SEC, EXACT>::aload(a.help_done);
                    if (dn >= n_total) break;
                    if (wave_clock() - t_idle0 > 24000000000ull) {        // ~10 s without the launch finishing: stop waiting, leave a trace
                        if (lane == 0) { atomicAdd(&a.counters[14], 1ull << 32); a.counters[15] = 0x3000000000000000ull | ((unsigned long long)n_total << 32) | dn; }
                        break;
                    }
                }
                bool any = false;
                for (uint32_t s = 0; s < a.n_help; s++) {
                    G(PEHelpSlot) *slot = (G(PEHelpSlot) *)&a.help[s];
                    if (DevPL<AGC, SEC, EXACT>::aload(&slot->pair) >= a.n_pairs) continue;
                    // (every cross-wave read of the slot goes through an L2 atomic until the acquire fence below: a plain load may be
                    //  served by this CU's L1 with what the slot held for an earlier pair)
                    if (DevPL<AGC, SEC, EXACT>::aload(&slot->state) != 1u) continue;
                    if (DevPL<AGC, SEC, EXACT>::aload(&slot->next) >= DevPL<AGC, SEC, EXACT>::aload(&slot->n)) continue;
                    // attach, THEN look at the state again -- and the look must not overtake the attach (the owner does the mirror image:
                    // close, then look at the attach count): the increment's return value is waited for before the state is read
                    uint32_t before = 0;
                    if (lane == 0) before = atomicAdd(&slot->helpers, 1u);
                    before = first_u32(before);
                    if (before < 0x7fffffffu && DevPL<AGC, SEC, EXACT>::aload(&slot->state) == 1u) {
                        DevPL<AGC, SEC, EXACT>::fence_acquire();
                        load_pair(DevPL<AGC, SEC, EXACT>::aload(&slot->pair));
                        pl.help_work(core, slot);
                        any = true;
                    }
                    if (lane == 0) atomicSub(&slot->helpers, 1u);
                }
#ifdef SNAPGPU_WAVE_EMU
                // (emulator: the waves of a block run one after the other and blocks beyond the host-thread pool wait for a free thread, so a
                //  wave that waits can keep the pairs it waits for from ever starting; SNAPGPU_EMU_HELP_SPIN=1 for grids that fit the pool)
                if (!getenv("SNAPGPU_EMU_HELP_SPIN")) break;
#endif
                if (!any) { DevPL<AGC, SEC, EXACT>::nap(); DevPL<AGC, SEC, EXACT>::nap(); }
            }
        }
    }
    if constexpr (EXACT) {              // leave the images zeroed for the next launch (the high-water marks live in registers)
        if (pl.ag_hw0) wave_zero16(pl.ag_persist0, ((size_t)pl.ag_hw0 + 15) & ~(size_t)15);
        if (pl.ag_hw1) wave_zero16(pl.ag_persist1, ((size_t)pl.ag_hw1 + 15) & ~(size_t)15);
        if (al.ag_hw0) wave_zero16(al.ag_persist0, ((size_t)al.ag_hw0 + 15) & ~(size_t)15);
        if (al.ag_hw1) wave_zero16(al.ag_persist1, ((size_t)al.ag_hw1 + 15) & ~(size_t)15);
    }
    if (lane == 0 && !(EXACT && a.is_replay)) {          // (a pair redone by the exact replay was already counted)
        if (!a.is_replay) atomicAdd(&a.counters[0], (unsigned long long)(2 * n_done));
        atomicAdd(&a.counters[1], (unsigned long long)al.cnt().lookups);
        atomicAdd(&a.counters[2], (unsigned long long)al.cnt().slots);
        atomicAdd(&a.counters[3], (unsigned long long)(al.cnt().hits + core.S()->cnt.hits));
        atomicAdd(&a.counters[4], (unsigned long long)(al.cnt().overflow_lists + core.S()->cnt.overflow_lists));
        atomicAdd(&a.counters[5], (unsigned long long)(al.cnt().lv + core.S()->cnt.lv));
        atomicAdd(&a.counters[6], (unsigned long long)(al.cnt().ag + core.S()->cnt.ag));
        atomicAdd(&a.counters[7], (unsigned long long)(al.cnt().lv_ref_bytes + core.S()->cnt.lv_ref_bytes));
        // phase cycles of the paired path: lookup = Phase 1, hits = Phase 2 (intersection), lv / ag = paired scoring only,
        // reserved[0] = the single-end fallback as a whole
        atomicAdd(&a.counters[8], (unsigned long long)core.S()->cnt.cyc_lookup);
        atomicAdd(&a.counters[9], (unsigned long long)core.S()->cnt.cyc_intersect);
        atomicAdd(&a.counters[10], (unsigned long long)core.S()->cnt.cyc_lv);
        atomicAdd(&a.counters[11], (unsigned long long)core.S()->cnt.cyc_ag);
        atomicAdd(&a.counters[12], (unsigned long long)core.S()->cnt.cyc_total);
        atomicAdd(&a.counters[13], (unsigned long long)core.S()->cnt.cyc_single);
    }
}

// pairs flagged SNAPGPU_PAIR_POOL_OVERFLOW by the first pass -> list for the second pass
// (stale != 0: instead the pairs whose traceback left the band and whose pools did not overflow -> list for the exact pass)
template <int UNUSED>
__global__ void k_collect_flagged(snapgpu_paired_result *primary, uint32_t n, uint32_t *list, uint32_t *count, int stale)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t fl = primary[i].flags;
    const bool ov = (fl & SNAPGPU_PAIR_POOL_OVERFLOW) != 0;
    if (stale && (fl & SNAPGPU_PAIR_REPLAYED_BESIDE)) { primary[i].flags = fl & ~SNAPGPU_PAIR_REPLAYED_BESIDE; return; }   // already redone beside the main pass
    if (stale ? (!ov && (fl & SNAPGPU_PAIR_EXACT_REPLAY) != 0) : ov) list[atomicAdd(count, 1u)] = i;
}
