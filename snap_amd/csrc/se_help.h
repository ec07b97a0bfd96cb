// se_help.h -- a heavy read's remaining candidates, scored by wavefronts that have run out of reads (single-end path).
//
// Why: a launch lasts as long as its slowest read, and the slowest reads are slow by two orders of magnitude -- profiles/r03c: of 1 M
// reads 65 take 110-220 ms each (400-600 affine-gap scorings against the copies of a diverged repeat family; the average read takes
// 0.6 ms) while the launch's work, spread evenly, would be done in ~100 ms.  Heavy-first dequeue (order.h) starts them first; this
// shortens them.
//
// What is shared: BaseAligner::score (BaseAligner.cpp:918-1534) under force_result walks every element left in the weight lists, in a fixed
// order, and evaluates every unscored candidate of each: Landau-Vishkin on both sides of the seed, then -- if that found more than
// maxKSame edits and the element can still matter -- affine gap on both sides.  The evaluation of one candidate (Aligner::eval_candidate)
// is a pure function of (read, location, seed offset, score limit, whether affine gap runs); everything that is NOT pure -- score sets,
// FP64 probability sums, the nearby-bucket merge, which limit the next candidate gets -- happens in the owner's ordered walk, untouched.
// So: the owner, once somebody is idle, lists the candidates it still has to visit, in order, and publishes the list in a slot with the
// limits it has now; idle waves attach, load the read, and evaluate candidates from the list into a per-item record; the owner keeps
// walking and, at each candidate, takes the stored evaluation if it is there AND was made under the limit / affine-gap decision the owner
// arrives with (same inputs, same answer -- bit-identical by construction), waits for it if somebody is at it, or evaluates it itself.
// Traceback steps outside the band of a speculative evaluation are all counted as "later call" steps (the helper cannot know what the
// owner's aligner object scored before), which sends the read to the exact replay exactly as a locally computed one would be.
// Not in the EXACT instantiations (the replay): there the calls of a read are ordered through the traceback arrays they share.
//
// Cross-wave traffic follows paired_dev.h (which learnt it the hard way): protocol words are only touched by read-modify-write atomics;
// the item records travel in device-scope stores / loads ordered against their state word by a vmcnt(0) wait; one agent-scope release per
// publication (the owner's candidate table and item list), one acquire per attach; every wait has a watchdog that turns into "go on alone".
#pragma once
#include "dev_common.h"

#define SE_HELP_SLOTS 384u            // lists open at a time: a launch of 1 M reads has ~300 reads of 50 ms and more (profiles/r03c)
#define SE_HELP_MIN_ITEMS 48u          // fewer candidates left than this: not worth a publication
#define SE_HELP_LOOKAHEAD 4u           // helpers start this many items ahead of the owner
#define SE_HELP_CHUNK 2u
#define SE_HELP_MAX_HELPERS 16u        // idle waves attached to one list at a time
#define SE_HELP_ITEMS_CAP 4096u        // candidates per published list (what is beyond stays with the owner)

struct __attribute__((aligned(16))) SESpec {       // one candidate's evaluation
    uint32_t state;                    // 0 untouched, 1 somebody is evaluating it, 2 done
    int32_t  limit;                    // score limit it was evaluated under
    uint32_t best_all;                 // all.best_score the affine-gap decision was taken with
    uint32_t lv_sum_high;              // both Landau-Vishkin halves succeeded with more than maxKSame edits in total (the decision depended on best_all)
    uint32_t sc;
    int32_t  ag_score, used_ag, clip_before, clip_after;
    uint32_t n_lv, n_ag, stale;
    int64_t  loc;
    double   mp;
    uint64_t lv_ref_bytes;
    int32_t  lv1, lv2;                 // Landau-Vishkin's own answers for the two sides (-1: above the limit; lv2 -2: not run)
};
static_assert(sizeof(SESpec) == 80, "SESpec layout");

struct SEHelpSlot {
    uint32_t state;                    // 0 free, 3 being filled, 1 open, 2 closing (no new helpers), 4 retired by a watchdog
    uint32_t read, n, next, helpers, owner_pos;
    int32_t  lim_alt, lim_non_alt;
    uint32_t best_all, pad0;
    const uint32_t *items;             // [n] element index << 6 | candidate index, in the owner's visiting order
    const void *pool;                  // the owner's candidate table (Elem[])
    SESpec *spec;                      // [n]
};

// device-scope accessors (see paired_dev.h: DevPL for why each one is spelled the way it is)
struct XW {
    static __device__ __forceinline__ uint32_t aload(uint32_t *p) { return first_u32(lane_id() == 0 ? atomicCAS(p, 0xFFFFFFF5u, 0xFFFFFFF5u) : 0u); }
    template <class T> static __device__ __forceinline__ void st(T &x, T v) {
#ifdef SNAPGPU_WAVE_EMU
        if constexpr (sizeof(T) == 8) __atomic_store_n((uint64_t *)&x, __builtin_bit_cast(uint64_t, v), __ATOMIC_SEQ_CST);
        else __atomic_store_n((uint32_t *)&x, __builtin_bit_cast(uint32_t, v), __ATOMIC_SEQ_CST);
#else
        if constexpr (sizeof(T) == 8) __hip_atomic_store((uint64_t *)&x, __builtin_bit_cast(uint64_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store((uint32_t *)&x, __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    template <class T> static __device__ __forceinline__ T ld(const T &x) {
#ifdef SNAPGPU_WAVE_EMU
        if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T, first_u64(__atomic_load_n((const uint64_t *)&x, __ATOMIC_SEQ_CST)));
        else return __builtin_bit_cast(T, first_u32(__atomic_load_n((const uint32_t *)&x, __ATOMIC_SEQ_CST)));
#else
        if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T, first_u64(__hip_atomic_load((const uint64_t *)&x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
        else return __builtin_bit_cast(T, first_u32(__hip_atomic_load((const uint32_t *)&x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
#endif
    }
    // the same load for code that only one lane runs (no broadcast)
    static __device__ __forceinline__ uint32_t ld_lane(const uint32_t &x) {
#ifdef SNAPGPU_WAVE_EMU
        return __atomic_load_n(&x, __ATOMIC_SEQ_CST);
#else
        return __hip_atomic_load(&x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    static __device__ __forceinline__ void stores_done() {
#ifndef SNAPGPU_WAVE_EMU
        __builtin_amdgcn_s_waitcnt(0x0F70);                                 // vmcnt(0)
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
};
