// paired_args.h -- argument struct / LDS carve-out of k_align_paired and the per-variant launchers
// (paired_k.hip is compiled once per affine-gap variant; see __graft_entry__.build()).
#pragma once
#include <hip/hip_runtime.h>
#include "kernel_common.h"
#include "paired.h"

// waves per SIMD k_align_paired is compiled for (its __launch_bounds__) and the host sizes its grid for.  Round 3 shrank the LDS
// footprint of the 192-position variant to under 10 KB per wave (paired_dev.h), which allows up to 4, and measured them on the bench
// batch (profiles/r03c, r03l):
//   4 (128 VGPRs, +360 spilled dwords per lane)   one context 6.58 s per batch against 5.66 s at 2: slower -- twice the slabs behind the same L2
//   3 (168 VGPRs)                                  one context 5.62 s against 5.44 s; TWO feeders 4.99 s against 5.72 s, THREE 3.98 s against 4.2-4.4 s
// i.e. the kernel is bound by its scratch / slab traffic, not by latency hiding, but a third wave per SIMD gives overlapping launches room.
// 3 for that variant (the default runs three feeders); the variants with more affine-gap state in registers and the LDS form stay at 2.
#ifndef SNAPGPU_PAIRED_WAVES_PER_SIMD
#define SNAPGPU_PAIRED_WAVES_PER_SIMD(AGC) ((AGC) != 0 ? 3 : 2)     // (round 6: AGC 4 / 6 keep three chunks in registers like AGC 3, ag_win.h: ag_dispatch_inl)
#endif

// PE_FRAME_BYTES: the wave's FRAME -- the three objects of the kernel (the single-end Aligner, DevPL, PairedCore: wave-uniform state, one pair
// per wave) live in LDS, not in per-lane private memory: as locals their addresses escape (dynamic indices, the objects point at one another),
// the compiler keeps them in scratch, and 1.9 KB per LANE of scratch -- 120 KB per wave, behind the same L2 as the candidate pools -- is what
// 94 % of the kernel's wave cycles waited for (profiles/r05zz).  k_align_paired static_asserts that they fit.
#define PE_FRAME_BYTES 2048u
struct PairedLds { uint32_t single_total, rd, ql, lk, exhausted, miss, hs, list_head, seed_used, sh, frame, total; };
static __host__ __device__ __forceinline__ PairedLds paired_lds_layout(uint32_t single_total, uint32_t RL, uint32_t max_seeds) {
    PairedLds L; uint32_t o = (single_total + 15) & ~15u;
    L.single_total = o;
    L.rd = o; o += 4 * RL;                                   // [read][dir]
    L.ql = o; o += 4 * RL;
    L.lk = o; o += (4 * max_seeds * (uint32_t)sizeof(PELookup) + 15) & ~15u;
    L.exhausted = o; o += (4 * max_seeds * 4 + 15) & ~15u;
    L.miss = o; o += (max_seeds * 4 + 15) & ~15u;
    L.hs = o; o += (4 * (uint32_t)sizeof(PEHitSetHdr) + 15) & ~15u;
    L.list_head = o; o += ((SNAPGPU_MAX_K + 1) * 4 + 15) & ~15u;
    L.seed_used = o; o += (((RL + 31) / 32) * 4 + 15) & ~15u;
    L.sh = o; o += ((uint32_t)sizeof(PEShared) + 15) & ~15u;
    L.frame = o; o += PE_FRAME_BYTES;
    L.total = o;
    return L;
}

// A heavy pair's Phase 4 offered to idle wavefronts (paired.h: PEHelpSpec).  One slot per pair being helped; the owner publishes it, takes
// chunks of candidates like everyone else, waits until all are scored and then walks the list in order.  Waves that have run out of pairs
// poll the slots until every pair of the launch is done.  All cross-wave traffic goes through device-scope atomics on the slot plus a
// __threadfence() between the data and the flag on both sides.
struct PEHelpSlot {
    uint32_t state;                    // 0 free, 3 being filled, 1 open, 2 closing (no new helpers)
    uint32_t pair, n, next, done, helpers;
    int32_t  limit, best;
    uint32_t skip0, skip1, pad0, pad1;
    const snapgpu_paired_result *agc;
    const uint32_t *order;
    PEHelpSpec *spec;
};
#define PE_HELP_CHUNK 16u

struct PairedArgs {
    DevIndex ix;
    AlignCfg scfg;                     // the single-end aligner of the chimeric fallback
    PECfg pcfg;
    const DevTables *tab;
    uint8_t *scratch;                  // n_wave_slots * stride
    uint64_t stride;
    uint64_t off_single_agc, off_cand, off_mate0, off_mate1, off_anchor, off_agc, off_agc_order;   // offsets inside a wave's slab (single-end scratch first)
    uint64_t off_lv_big;               // Landau-Vishkin working set for limits beyond the LDS triangle (paired_dev.h: DevPL::lv_big)
    uint32_t single_agc_cap;
    const uint8_t *bases, *quals;
    const uint64_t *offsets;           // [2n+1]
    uint32_t n_pairs;
    int32_t max_k_paired, max_k_single;
    snapgpu_paired_result *primary, *first_alt;
    uint32_t *work_counter;
    unsigned long long *counters;      // snapgpu_counters layout
    uint32_t kmax_lv;
    // second pass over the pairs whose candidate buffers overflowed in the first (see launch_paired): work item i is pair remap[i]
    const uint32_t *remap, *n_remap;
    uint32_t is_replay;                // the pairs of this launch were already counted by an earlier launch of the same call (second / third pass)
    // secondary results (k_align_paired<.., true> only): extra sections of a wave's slab, and the caller's buffers
    uint64_t off_sec, off_sec_ord, off_sec_key, off_ssec;
    SecCfg ssec_cfg;                   // the single-end aligner's lists
    snapgpu_paired_result *secondary; uint32_t sec_out_stride; uint32_t *n_secondary;                     // [n * stride], [n]
    snapgpu_single_result *single_secondary; uint32_t ssec_out_stride; uint32_t *n_single_secondary;      // [n * stride], [2n]
    // exact replay of pairs whose affine-gap traceback left the band (k_align_paired<0, SEC, true>): 4 x ag_scratch_bytes(RL) per wave slot
    uint8_t *persist; uint64_t persist_stride;
    // Phase-4 help (paired_dev.h: PEHelpSlot): slots, one PEHelpSpec array of help_spec_cap entries per slot, the launch's done-pair counter
    // (help_done[0]: pairs of the launch that are done; help_done[1]: waves that have run out of pairs -- the idle count a pair looks at
    //  before it publishes; help_eager: publish whether or not anybody is idle)
    struct PEHelpSlot *help; uint32_t n_help; PEHelpSpec *help_spec; uint32_t help_spec_cap; uint32_t *help_done; uint32_t help_min;
    uint32_t help_eager;
    // Exact replay that runs BESIDE the main pass (launch_paired): the main kernel (rq_mode 1) appends a pair to rq_list as soon as it has
    // flagged it, the exact kernel (rq_mode 2, its own stream) takes pairs from the list while the main pass is still going and leaves
    // when the main pass has finished every pair and the list is empty.  rq[0] entries appended, rq[1] entries taken, rq[2] pairs the main
    // pass has finished, rq[3] diagnostics.  rq_list entries start as 0xFFFFFFFF.
    uint32_t *rq, *rq_list; uint32_t rq_mode;
    uint32_t dbg_flag_every;           // tests (SNAPGPU_DEBUG_PAIRED_FLAG_EVERY=<k>): the fast pass flags every k-th pair for the exact kernel as well
};
#define SNAPGPU_PAIR_REPLAYED_BESIDE 0x40000000u   // (internal, never leaves launch_paired: the pair was redone by the exact kernel beside the main pass)


extern "C" {
void snapgpu_launch_paired_3(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_paired_4(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_paired_6(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_paired_0(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_paired_sec_3(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_paired_sec_0(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_collect_flagged(snapgpu_paired_result *primary, uint32_t n, uint32_t *list, uint32_t *count, int stale, hipStream_t s);
void snapgpu_launch_paired_exact_3(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_paired_exact_0(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_paired_exact_4(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_paired_exact_6(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_paired_sec_exact_3(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_paired_sec_exact_0(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
}
