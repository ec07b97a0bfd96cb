// align_single.h -- BaseAligner::AlignRead for one read per wavefront.
//
// Restates (file:line in the reference tree):
//   BaseAligner::AlignRead                 SNAPLib/BaseAligner.cpp:273-763   adaptive seed loop
//   BaseAligner::score                     SNAPLib/BaseAligner.cpp:918-1534  ordered candidate scoring
//   ScoreSet::updateBestScore & friends    SNAPLib/BaseAligner.cpp:2132-2323
//   findElement/findCandidate/allocateNewCandidate/incrementWeight/clearCandidates
//                                          SNAPLib/BaseAligner.cpp:1811-1973, 2332-2379
//   scoreLimit                             SNAPLib/BaseAligner.cpp:2556-2570
//   Genome::getSubstring / isGenomeLocationALT / getContigAtLocation
//                                          SNAPLib/Genome.h:339-367, 436; Genome.cpp:574-602
//   computeMAPQ                            SNAPLib/mapq.h:31-68
// (exact semantics that matter for bit-identical results are collected in SURVEY.md App. A.1-A.3)
//
// The algorithm is adaptive and sequential per read (which seed is looked up next, which
// candidate is scored next and with what limit all depend on earlier scores), so one
// wavefront owns one read and executes the control flow wave-uniformly, the way one CPU
// thread owns one BaseAligner in the reference.  The 64 lanes are used inside the
// primitives: 2x32 lanes walk the two hash-probe sequences, 64 lanes fetch an overflow list
// line-coalesced, lanes are LV diagonals / affine-gap SSE lanes, and the reference window is
// staged once per candidate into LDS with one coalesced load.
//
// Per-read state lives in LDS (read, its reverse complement, qualities, seed-used bitmap,
// weight-list heads, LV triangle, reference window).  The candidate table -- a few hundred
// bytes per 48-base bucket, up to max_hits*max_seeds buckets -- lives in a per-wave slab of
// HBM scratch that stays L2-resident; it is indexed by a small open hash (u16 heads) that is
// un-done bucket by bucket at the end of each read instead of the reference's epoch trick.
#pragma once
#include "dev_common.h"
#include "probe.h"
#include "lv.h"
#include "ag_win.h"
#include "se_help.h"
#include "adjust.h"
#include <stddef.h>
#include "../../include/snapgpu.h"

#define BUCKET 48                      // hashTableElementSize == maxMergeDist, BaseAligner.h:177,213
#define SENT_MIN 0xFC00u               // link values >= SENT_MIN denote weight-list sentinels
#define WIN_PAD 128                    // reference bytes staged on either side of [loc, loc+readLen)

struct AlignCfg {
    uint32_t max_hits, max_k, num_seeds, min_weight, extra_depth, use_ag;
    int32_t  match_reward, sub_penalty, gap_open, gap_extend, five_bonus, three_bonus;
    uint32_t alt_aware, emit_alt;
    int32_t  max_gap_alt;
    double   seed_coverage;
    uint32_t num_weight_lists;         // ctor: maxSeedsToUse + 1, BaseAligner.cpp:173-180
    uint32_t RL;                       // per-wave read buffer length (max_read_len rounded up to 16)
    uint32_t kmax;                     // largest score limit: min(126, max_k + extra_depth)
    uint32_t pool_size;                // candidate buckets per wave
    uint32_t ht_size;                  // power of two, u16 heads
    uint64_t scratch_stride;           // bytes of HBM scratch per wave
    uint32_t lds_per_wave;
    uint32_t ag_numvec_max;            // ceil(RL/8)
    // The affine-gap LDS rows and traceback slab exist: use_ag, or the paired-end path with soft clipping, whose Hamming retry calls
    // BaseAligner::alignAffineGap whatever useAffineGap says (ChimericPairedEndAligner.cpp:330-360: the _ASSERT(useAffineGap) there is
    // compiled out of a release build).
    uint32_t ag_buffers;
    // LDS bytes per wave for the affine-gap code: ag_lds_bytes(RL) where a kernel of this context can run the LDS form (AGC == 0: reads
    // beyond the register variants, and the exact replay behind the 256- / 384-position variants), ag_lds_bytes_reg(RL, 3) where only the
    // 192-position register forms run; 0 without affine-gap buffers.
    uint32_t ag_lds;
    // se_help.h: a wave's list of candidates still to visit (se_items_cap words) and, per candidate-table element, where its candidates
    // start in that list (pool_size words), at byte se_off of the wave's scratch slab; se_items_cap == 0: no help in this context
    uint32_t se_items_cap; uint64_t se_off;
    // -f / -x (AlignerOptions.cpp:571-574; BaseAligner::setStopOnFirstHit / setExplorePopularSeeds, SingleAligner.cpp:179-180): stop at the
    // first location within maxK and report it as MultipleHits with MAPQ 0 (BaseAligner.cpp:1490-1505); apply the first maxHits hits of a
    // seed with more than maxHits of them instead of skipping it (:574)
    uint32_t stop_on_first_hit, explore_popular_seeds;
};

struct __attribute__((aligned(16))) Elem {   // HashTableElement, BaseAligner.h:223-258
    uint64_t used;                     // candidatesUsed
    uint64_t scored;                   // candidatesScored
    int64_t  base;                     // baseGenomeLocation
    int64_t  best_loc;                 // bestScoreGenomeLocation
    double   match_prob;               // matchProbabilityForBestScore
    uint32_t weight;
    uint32_t lps;                      // lowestPossibleScore
    uint32_t best_score;               // unsigned, may hold (unsigned)ScoreAboveLimit
    int32_t  ag_score;
    int32_t  clip_before, clip_after, seed_offset;
    uint16_t wnext, wprev, hnext;
    uint8_t  dir;
    uint8_t  flags;                    // bit0 allExtantCandidatesScored, bit1 usedAffineGapScoring
    uint32_t pad0;
    uint16_t cand_seed_offset[BUCKET]; // Candidate::seedOffset
};
static_assert(sizeof(Elem) == 176, "Elem layout");

struct ScoreSet {                      // BaseAligner.h:260-329
    int32_t  best_score;
    int64_t  best_loc, best_orig_loc;
    int32_t  dir;
    int32_t  used_ag, clip_before, clip_after, ag_score, seed_offset;
    double   best_match_prob;
    double   p_all, p_best;
    __device__ __forceinline__ void init() {
        best_score = SNAPGPU_UnusedScoreValue; best_loc = SNAPGPU_InvalidGenomeLocation32;
        best_orig_loc = SNAPGPU_InvalidGenomeLocation32; dir = 0; used_ag = 0; clip_before = 0;
        clip_after = 0; ag_score = -1; seed_offset = 0; best_match_prob = 0.0; p_all = 0.0; p_best = 0.0;
    }
};

struct WaveCounters {
    uint64_t lookups, slots, hits, overflow_lists, lv, ag, lv_ref_bytes;
    // shader-clock cycles spent per phase (s_memtime), for the profile breakdown in profiles/
    uint64_t cyc_lookup, cyc_hits, cyc_lv, cyc_ag, cyc_total;
};
static __device__ __forceinline__ uint64_t wave_clock() { return __builtin_amdgcn_s_memtime(); }

// computeMAPQ (mapq.h:31-68) with the log10 replaced by a threshold table built from the
// host's log10 at context creation (DevTables::mapq_threshold), so (int)(-10*log10(x)) is
// reproduced exactly.
static __device__ __forceinline__ int compute_mapq(const DevTables *tab, double p_all, double p_best, int popular_skipped) {
    if (p_all < p_best) p_all = p_best;
    double correctness = p_best / p_all;
    int base_mapq;
    if (correctness >= 1) {
        base_mapq = 70;
    } else {
        double x = 1 - correctness;
        // largest m in [0,70] with x <= threshold[m]; thresholds are decreasing in m
        int lo = 0, hi = 70;
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (x <= tab->mapq_threshold[mid]) lo = mid; else hi = mid - 1;
        }
        base_mapq = lo;
    }
    int pen = popular_skipped - 10; if (pen < 0) pen = 0;
    base_mapq -= pen / 2;
    if (base_mapq < 0) base_mapq = 0;
    return (int)first_u32((uint32_t)base_mapq);
}

struct WaveShared {                    // per-wave LDS block (see Aligner)
    ScoreSet all, non_alt;
    ScoreSet ag_all, ag_non_alt;       // the local score sets of BaseAligner::alignAffineGap (paired-end fallback only)
    snapgpu_single_result primary, first_alt;
    WaveCounters cnt;
};

// Secondary alignments (-om / -omax / -mpc): what AlignRead's secondaryResults buffer and finalizeSecondaryResults do
// (BaseAligner.cpp:2174-2200, 2245-2271, 2423-2553).  Per-wave HBM scratch, sized so that it cannot overflow.
struct SecCfg {
    int32_t  om;                       // maxEditDistanceForSecondaryResults (>= 0 when enabled)
    int32_t  mpc;                      // maxSecondaryAlignmentsPerContig (-1: unlimited)
    int64_t  omax;                     // maxSecondaryResults
    uint32_t cap;                      // entries of per-wave scratch
    uint32_t adjust;                   // -ae: AlignmentAdjuster on the primary and on every secondary result before the filter (adjust.h)
    uint64_t adj_off;                  // offset of adjust.h's scratch inside the wave's secondary-result slab
};

// EXACT: the replay instantiation for reads whose banded affine-gap traceback left the band (`reserved` != 0 after the fast pass): every
// affine-gap call goes through the layout-literal form of ag.h over ag_persist[0 / 1], the wave's images of the reference aligner's
// affineGap / reverseAffineGap traceback arrays, zeroed by the kernel before the read -- the answer of a newly constructed reference aligner.
// TIMED: the s_memtime phase timers (WaveCounters::cyc_*) are compiled in.  Each read of the clock drains lgkmcnt, ~40 of them per read, so
// the kernels that are timed for throughput are built without (SNAPGPU_PHASE_TIMERS=1 selects the timed instantiation for a breakdown run).
// PLANES: the plane Landau-Vishkin (planes.h; SNAPGPU_LV_PLANES=1) is compiled in.  Its own instantiations (single_planes_k.hip): carried by
// every kernel it cost the default ones 110-120 bytes of scratch per lane (exact form 520 -> 408, fast form 712 -> 592) for an option that is off.
// REGDIR: the candidate table's DIRECTORY in a vector register -- lane j holds the key (bucket number * 2 + direction) of element j, j < 64 -- so that
// findElement, which the reference does once per seed hit and which was two dependent HBM loads here (the u16 head, then the chained element), is one compare and
// a ballot while the read has at most 64 buckets (nearly every read); the HBM hash is built when the 65th arrives and used from then on.  Only where this
// object lives in registers (k_align_single): in the paired-end kernel it is one LDS object per wave and has no per-lane members.
struct AlignerDirState { uint32_t dirkey; };
struct AlignerNoDirState {};
template <bool REGDIR> struct AlignerDirBase { using type = AlignerNoDirState; };
template <> struct AlignerDirBase<true> { using type = AlignerDirState; };

template <int AGC, bool SEC = false, bool EXACT = false, bool TIMED = false, bool PLANES = false, bool REGDIR = false>
struct Aligner : AlignerDirBase<REGDIR>::type {
    // ---- constant for the launch
    // held by value: a reference member would make the kernel-argument struct escape through a
    // flat pointer and pin this whole object (ScoreSets, results, counters) in scratch memory
    const DevIndex ix;
    GP<const DevTables> tab;
    const AlignCfg cfg;
    // ---- LDS carve-out for this wave
    // (LP<T> / GP<T>, dev_common.h: pointers stored with their address space)
    LP<uint8_t>  rd[2];     // bases, forward / reverse complement
    LP<uint8_t>  ql[2];     // qualities in the same orientation
    LP<uint8_t>  gw;        // reference window: gw[x] = genome[win_loc - WIN_PAD + x]
    LP<uint32_t> seed_used;
    LP<uint16_t> wl_next, wl_prev;
    LP<uint16_t> lv_tri;
    LP<int16_t>  ag_rows;   // H, H-1, E rows of the affine-gap DP
    LP<unsigned long long> rp; // bit planes of the read, both directions (planes.h); [dir][plane][read_plane_words(RL)]
    LP<unsigned long long> tp; // bit planes of the candidate's reference window; [plane][text_plane_blocks(RL, WIN_PAD)]
    int tp_org;             // bit of tp that is genome[loc] of the staged candidate
    LP<unsigned long long> lvp; // LDS work area of the prepared plane form (planes.h: lv_plane_work_words)
    uint32_t rd_plain;      // bit dir: rd[dir] is all ACGT (its 'N' / other planes are empty)
    // ---- HBM scratch for this wave
    GP<uint16_t> heads;
    GP<Elem>     pool;
    GP<uint8_t>  ag_scratch;
    uint8_t  *ag_persist0, *ag_persist1;   // EXACT only: images of the forward object's (affineGap) and the backward object's (reverseAffineGap) array
    uint32_t ag_hw0, ag_hw1;               // EXACT only: bytes of each image written since it was last zeroed (what the next read must clear)
    uint32_t ag_epoch, ag_tag;             // EXACT only: reads since the images were last cleared (1 .. 15) and its tag bits (dev_common.h: bt_cell): cells of other reads read as zero
    // ---- per-read state (wave-uniform)
    // (not a stored value: in the paired-end kernel this object lives in LDS, one per WAVE -- paired_args.h: PE_FRAME_BYTES)
    struct LaneId { __device__ __forceinline__ operator int() const { return lane_id(); } };
    LaneId lane;
    int read_len;
    uint32_t n_used;
    uint32_t highest_used_weight_list;
    uint32_t wrap_count;
    uint32_t lps_unseen[2];            // lowestPossibleScoreOfAnyUnseenLocation
    uint32_t cur_round_lps[2];         // currRoundLowestPossibleScoreOfAnyUnseenLocation
    uint32_t n_seeds_applied[2];
    uint32_t popular_seeds_skipped;
    uint32_t ag_stale;                 // affine-gap traceback steps outside the computed band (see ag.h)
    uint32_t ag_replay;                // ... of which in a call that was not its object's first: those can matter (note_ag_call)
    uint32_t ag_obj_used0, ag_obj_used1;   // has affineGap / reverseAffineGap scored anything yet for this read (this pair, for the fallback)?
                                       // (two scalars, not an array: an index that is not a compile-time constant would pin the object in scratch)
    uint32_t max_k;                    // BaseAligner::maxK: cfg.max_k, or what setMaxK() last said (ChimericPairedEndAligner.cpp:278,301)
    uint32_t ag_calls_unit;            // affine-gap calls since the unit (read; the paired kernel: pair) began -- see wave_set_priority
    // ---- help for heavy reads (se_help.h); all NULL / 0 where there is none (the paired-end kernel, the exact replay)
    SEHelpSlot *se_slots; uint32_t se_n_slots; SESpec *se_spec; uint32_t se_spec_cap;
    uint32_t *se_ctl;                  // [0] reads of the launch that are done, [1] waves that have run out of reads, [2] lists open now
    uint32_t se_eager;                 // publish whether or not anybody is idle (tests)
    uint32_t *se_items, *se_first;     // this wave's list / per-element start (HBM slab)
    unsigned long long *se_diag;       // snapgpu_counters::reserved[1 .. 2]
    int se_slot; uint32_t se_n, se_tried, cur_read; SESpec *se_mine;
    // candidates for BaseAligner::alignAffineGap, collected by the Hamming pass only (BaseAligner.cpp:1445-1456)
    GP<snapgpu_single_result> agc;
    uint32_t agc_cap, n_agc, agc_overflow;
    // secondary results (SEC only): the list AlignRead appends to, then finalizeSecondaryResults' working arrays
    SecCfg sec_cfg;
    snapgpu_single_result *sec;        // [sec_cfg.cap]
    uint32_t *sec_key, *sec_ord;       // [sec_cfg.cap] each
    uint32_t n_sec, n_sec_raw, sec_overflow;
    uint8_t *adj_scratch;              // adjust.h (sec_cfg.adjust only)
    // Cold, wave-uniform state lives in LDS (WaveShared), not in registers: it is touched a few
    // times per candidate / per read, and keeping ~150 dwords of it live across the LV and
    // affine-gap code is what pushed the kernel to 1-2 waves per SIMD.  Every lane executes the
    // same stores with the same values (uniform control flow), so no lane guard is needed.
    // (Accessors over ONE stored LDS address, not reference members: where this object itself lives in memory -- the paired-end kernel keeps it
    //  in LDS -- a reference member is a pointer loaded back from there, i.e. a generic one, and every access through it a FLAT instruction.)
    LDS_AS WaveShared *ws_;
    __device__ __forceinline__ ScoreSet &all() const { return *(ScoreSet *)&ws_->all; }
    __device__ __forceinline__ ScoreSet &non_alt() const { return *(ScoreSet *)&ws_->non_alt; }
    __device__ __forceinline__ snapgpu_single_result &primary() const { return *(snapgpu_single_result *)&ws_->primary; }
    __device__ __forceinline__ snapgpu_single_result &first_alt() const { return *(snapgpu_single_result *)&ws_->first_alt; }
    __device__ __forceinline__ WaveCounters &cnt() const { return *(WaveCounters *)&ws_->cnt; }

    __device__ __forceinline__ Aligner(const DevIndex &ix_, const DevTables *tab_, const AlignCfg &cfg_, WaveShared *ws)
        : ix(ix_), tab(tab_), cfg(cfg_), tp_org(0), rd_plain(0), ag_hw0(0), ag_hw1(0), ag_epoch(0), ag_tag(0), read_len(0), popular_seeds_skipped(0),
          ag_stale(0), ag_replay(0), ag_obj_used0(0), ag_obj_used1(0), max_k(cfg_.max_k), ag_calls_unit(0),
          agc(nullptr), agc_cap(0), n_agc(0), agc_overflow(0), n_sec(0), n_sec_raw(0), sec_overflow(0),
          ws_((LDS_AS WaveShared *)ws) {}

    static __device__ __forceinline__ uint64_t clk() { if constexpr (TIMED) return wave_clock(); else return 0; }

    // A traceback step outside the band reads what the reference object's array holds there.  For the FIRST call an object serves after
    // its construction that is zero -- exactly what the kernels read -- so only steps of later calls make the answer depend on the
    // earlier ones and send the read to the exact replay (in the replay itself the array is real and every step is simply counted).
    __device__ __forceinline__ void note_ag_call(int obj, uint32_t stale_steps) {
        ag_stale += stale_steps;
        const uint32_t used = obj == 0 ? ag_obj_used0 : ag_obj_used1;
        if (EXACT || used) ag_replay += stale_steps;
        if (obj == 0) ag_obj_used0 = 1; else ag_obj_used1 = 1;
    }

    // EXACT: a new read = newly constructed reference aligners, whose traceback arrays read as zero.  The images are cleared once per
    // fifteen reads; in between a read's cells carry its tag and everybody else's read as zero (dev_common.h: bt_cell).
    __device__ __forceinline__ void new_read_images() {
        if constexpr (EXACT) {
            if (++ag_epoch == 16u) {
                if (ag_hw0) wave_zero16(ag_persist0, ((size_t)ag_hw0 + 15) & ~(size_t)15);
                if (ag_hw1) wave_zero16(ag_persist1, ((size_t)ag_hw1 + 15) & ~(size_t)15);
                ag_hw0 = ag_hw1 = 0; ag_epoch = 1u;
                WAVE_SYNC();
            }
            ag_tag = bt_tag_bits(ag_epoch);
        }
    }

    // EXACT: how far into its object's array a call can write: text_len rows of numVec * numSeg * 8 bytes (ag_dims)
    __device__ __forceinline__ void note_ag_extent(int obj, bool banded, int plen, int w, int tlen) {
        if constexpr (EXACT) {
            int nv, sl, ns;
            ag_dims(banded, plen, w > 126 ? 126 : (w < 0 ? 0 : w), &nv, &sl, &ns);
            uint32_t ext = (w >= 0 && tlen > 0) ? (uint32_t)tlen * (uint32_t)(ns * sl) : 0u;      // (see DevPL::ag)
            const uint32_t image = (uint32_t)ag_scratch_bytes(cfg.RL);
            if (ext > image) ext = image;
            if (obj == 0) ag_hw0 = ext > ag_hw0 ? ext : ag_hw0; else ag_hw1 = ext > ag_hw1 ? ext : ag_hw1;
        }
    }

    // ------------------------------------------------------------------ helpers
    __device__ __forceinline__ bool is_alt(int64_t loc) const { return (uint64_t)loc >= ix.first_alt_location && loc >= 0; }

    __device__ __forceinline__ int score_limit(bool for_alt) const {          // BaseAligner.cpp:2556-2570
        int64_t inner;
        if (for_alt) {
            int64_t g = cfg.max_gap_alt < non_alt().best_score ? cfg.max_gap_alt : non_alt().best_score;
            int64_t b = (int64_t)non_alt().best_score - g;
            inner = all().best_score < b ? all().best_score : b;
        } else {
            int64_t a = (int64_t)all().best_score + cfg.max_gap_alt;
            inner = a < non_alt().best_score ? a : non_alt().best_score;
        }
        int64_t m = (int64_t)max_k < inner ? (int64_t)max_k : inner;
        int64_t v = (int64_t)cfg.extra_depth + m;
        return (int)(v < 126 ? v : 126);
    }

    // Genome::getSubstring(location, lengthNeeded) != NULL  (Genome.h:339-367)
    __device__ __forceinline__ bool substring_ok(int64_t loc, int64_t len) const {
        if (!substring_in_range(loc, len)) return false;
        return substring_ok_known(loc, len, first_u32(((const G(uint8_t) *)ix.genome)[loc]) == 'n');
    }
    // the part of getSubstring that needs no memory: the window lies inside what the genome (and its padding) holds
    __device__ __forceinline__ bool substring_in_range(int64_t loc, int64_t len) const {
        int64_t nb = (int64_t)ix.n_bases;
        if (loc > nb || loc + len > nb + 1000) return false;
        if (loc < -(int64_t)ix.genome_pad + WIN_PAD) return false;            // (cannot happen; keeps loads in bounds)
        return true;
    }
    // ... and the rest, given whether genome[loc] is padding ('n'): the caller has usually just staged the window and reads that byte
    // from LDS instead of making a dependent trip to HBM for it
    __device__ __forceinline__ bool substring_ok_known(int64_t loc, int64_t len, bool first_is_pad) const {
        int64_t nb = (int64_t)ix.n_bases;
        if (len <= (int64_t)ix.chromosome_padding && !first_is_pad) return true;
        if (len == 0) return true;
        // getContigAtLocation: last contig whose beginning <= loc
        int lo = 0, hi = (int)ix.n_contigs - 1, found = -1;
        while (lo <= hi) {
            int mid = (lo + hi) >> 1;
            uint64_t b = first_u64(((const G(uint64_t) *)ix.contig_begin)[mid]);
            if ((int64_t)b <= loc) { found = mid; lo = mid + 1; } else { hi = mid - 1; }
        }
        if (found < 0) return false;
        int64_t cbeg = (int64_t)first_u64(((const G(uint64_t) *)ix.contig_begin)[found]);
        int64_t cend = found == (int)ix.n_contigs - 1 ? nb : (int64_t)first_u64(((const G(uint64_t) *)ix.contig_begin)[found + 1]);
        if (cend <= loc + len) return false;
        (void)cbeg;
        return true;
    }

    // ---- weight lists: doubly linked, FIFO, sentinel per weight (BaseAligner.cpp:1954-1957, 2366-2378)
    __device__ __forceinline__ uint16_t sent(uint32_t w) const { return (uint16_t)(0xFFFFu - w); }
    __device__ __forceinline__ uint16_t get_next(uint16_t i) const {
        return i >= SENT_MIN ? wl_next[0xFFFFu - i] : (uint16_t)first_u32(pool[i].wnext);
    }
    __device__ __forceinline__ uint16_t get_prev(uint16_t i) const {
        return i >= SENT_MIN ? wl_prev[0xFFFFu - i] : (uint16_t)first_u32(pool[i].wprev);
    }
    __device__ __forceinline__ void set_next(uint16_t i, uint16_t v) {
        if (lane == 0) { if (i >= SENT_MIN) wl_next[0xFFFFu - i] = v; else pool[i].wnext = v; }
        WAVE_SYNC();
    }
    __device__ __forceinline__ void set_prev(uint16_t i, uint16_t v) {
        if (lane == 0) { if (i >= SENT_MIN) wl_prev[0xFFFFu - i] = v; else pool[i].wprev = v; }
        WAVE_SYNC();
    }
    __device__ __forceinline__ void list_unlink(uint16_t e) {
        uint16_t n = get_next(e), p = get_prev(e);
        set_prev(n, p);
        set_next(p, n);
    }
    __device__ __forceinline__ void list_push_tail(uint32_t w, uint16_t e) {
        uint16_t s = sent(w);
        uint16_t tail = get_prev(s);
        if (lane == 0) { pool[e].wnext = s; pool[e].wprev = tail; }
        WAVE_SYNC();
        set_prev(s, e);
        set_next(tail, e);
    }

    // ---- candidate hash: (bucket base, direction) -> element index
    __device__ __forceinline__ uint32_t head_slot(int64_t base, int dir) const {
        uint64_t k = ((uint64_t)base / BUCKET) * 2 + (uint64_t)dir;
        k *= 0x9E3779B97F4A7C15ull;
        return (uint32_t)(k >> 40) & (cfg.ht_size - 1);
    }
    // findElement (BaseAligner.cpp:1811-1840): returns element index or 0xFFFF
    static __device__ __forceinline__ uint32_t dir_key(int64_t base, int dir) { return (uint32_t)(((uint64_t)base / BUCKET) * 2 + (uint64_t)dir); }   // (locations fit 32 bits: < 2^28)
#ifndef SNAPGPU_DIR_CAP
#define SNAPGPU_DIR_CAP 64            // (test builds lower it so that ordinary fixtures cross the directory-to-hash transition)
#endif
    static constexpr uint32_t DIR_CAP = SNAPGPU_DIR_CAP;
    __device__ __forceinline__ uint16_t find_element(int64_t loc, int dir) const {
        int64_t low = (int64_t)((uint64_t)loc % BUCKET);
        int64_t base = loc - low;
        if constexpr (REGDIR) {
            if (n_used <= DIR_CAP) {                                          // the whole table is in the directory
                const unsigned long long m = BALLOT(this->dirkey == dir_key(base, dir));
                return m ? (uint16_t)__builtin_ctzll(m) : (uint16_t)0xFFFF;
            }
        }
        uint16_t h = (uint16_t)first_u32(heads[head_slot(base, dir)]);
        while (h != 0) {
            // base (dwords 4,5) and hnext|dir|flags (dword 18) of the chained element in one load instead of three dependent ones
            const uint32_t *ew = (const uint32_t *)&pool[h - 1];
            const uint32_t v = ew[lane == 0 ? 4 : (lane == 1 ? 5 : 18)];
            const int64_t b = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)v, 1) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 0));
            const uint32_t w18 = (uint32_t)__builtin_amdgcn_readlane((int)v, 2);
            if (b == base && (int)((w18 >> 16) & 0xffu) == dir) return (uint16_t)(h - 1);
            h = (uint16_t)(w18 & 0xffffu);
        }
        return 0xFFFF;
    }

    // ------------------------------------------------------------------ per read
    __device__ __forceinline__ void clear_candidates() {                     // BaseAligner.cpp:2332-2339
        n_used = 0;
        highest_used_weight_list = 0;
        if constexpr (REGDIR) this->dirkey = 0xFFFFFFFFu;
        for (uint32_t i = lane; i < cfg.num_weight_lists; i += WAVE) {
            wl_next[i] = sent(i); wl_prev[i] = sent(i);
        }
        WAVE_SYNC();
    }

    // undo the head-table entries of this read (lane-parallel)
    __device__ __forceinline__ void release_candidates() {
        if (REGDIR && n_used <= DIR_CAP) return;                              // (the hash was never built)
        for (uint32_t i = lane; i < n_used; i += WAVE) {
            heads[head_slot(pool[i].base, pool[i].dir)] = 0;
        }
        WAVE_SYNC();
    }

    __device__ __forceinline__ void increment_weight(uint16_t ei) {          // BaseAligner.cpp:2342-2379
        Elem *e = &pool[ei];
        uint32_t flags = first_u32(e->flags);
        if (flags & 1) return;
        uint32_t w = first_u32(e->weight);
        if (w >= cfg.num_weight_lists - 1) return;
        list_unlink(ei);
        w++;
        if (lane == 0) e->weight = w;
        WAVE_SYNC();
        if (w > highest_used_weight_list) highest_used_weight_list = w;
        list_push_tail(w, ei);
    }

    __device__ __forceinline__ void allocate_new_candidate(int64_t loc, int dir, uint32_t lps, int seed_offset) {
        // BaseAligner.cpp:1885-1973
        int64_t low = (int64_t)((uint64_t)loc % BUCKET);
        int64_t base = loc - low;
        uint16_t ei = (uint16_t)n_used;
        n_used++;
        Elem *e = &pool[ei];
        uint32_t hs = head_slot(base, dir);
        uint16_t old_head = 0;
        bool hashed = true;
        if constexpr (REGDIR) {
            if (ei < DIR_CAP) { this->dirkey = lane == (int)ei ? dir_key(base, dir) : this->dirkey; hashed = false; }
            else if (ei == DIR_CAP) {                                         // the directory is full: the hash takes over, starting with what the directory holds
                for (uint32_t j = 0; j < DIR_CAP; j++) {
                    uint64_t k = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)this->dirkey, (int)j) * 0x9E3779B97F4A7C15ull;
                    const uint32_t hj = (uint32_t)(k >> 40) & (cfg.ht_size - 1);
                    const uint16_t oh = (uint16_t)first_u32(heads[hj]);
                    if (lane == 0) { pool[j].hnext = oh; heads[hj] = (uint16_t)(j + 1); }
                    WAVE_SYNC();
                }
            }
        }
        if (hashed) old_head = (uint16_t)first_u32(heads[hs]);
        if (lane == 0) {
            e->used = 1ull << low;
            e->scored = 0;
            e->lps = lps;
            e->dir = (uint8_t)dir;
            e->weight = 1;
            e->base = base;
            e->best_score = SNAPGPU_UnusedScoreValue;
            e->flags = 0;
            e->match_prob = 0;
            e->clip_before = 0; e->clip_after = 0; e->ag_score = 0;
            e->best_loc = 0; e->seed_offset = 0;
            e->cand_seed_offset[low] = (uint16_t)seed_offset;
            e->hnext = old_head;
            if (hashed) heads[hs] = (uint16_t)(ei + 1);
        }
        WAVE_SYNC();
        list_push_tail(1, ei);
        if (highest_used_weight_list < 1) highest_used_weight_list = 1;
    }

    // One hit of a seed lookup (body of the loop at BaseAligner.cpp:629-667).
    __device__ __forceinline__ void apply_hit(uint32_t hit, uint32_t offset, int dir) {
        int64_t loc = (int64_t)(uint32_t)(hit - offset);                      // unsigned 32-bit arithmetic, :637
        uint16_t ei = find_element(loc, dir);
        if (ei != 0xFFFF) {
            // findCandidate (:1873-1878) + :648-652
            Elem *e = &pool[ei];
            uint32_t low = (uint32_t)((uint64_t)loc % BUCKET);
            uint64_t bit = 1ull << low;
            uint64_t used = first_u64(e->used);
            uint32_t flags = first_u32(e->flags);
            uint32_t new_flags = (flags & ~1u) | (((flags & 1u) && (used & bit)) ? 1u : 0u);
            if (lane == 0) {
                e->flags = (uint8_t)new_flags;
                e->used = used | bit;
                e->cand_seed_offset[low] = (uint16_t)offset;
            }
            WAVE_SYNC();
            increment_weight(ei);
        } else {
            bool cand_alt = cfg.alt_aware && is_alt(loc);
            const uint32_t lps_d = dir ? lps_unseen[1] : lps_unseen[0];          // (no dynamic indexing: keeps the state in registers)
            if ((int64_t)lps_d <= (int64_t)score_limit(cand_alt)) {
                allocate_new_candidate(loc, dir, lps_d, (int)offset);
            }
        }
    }

    // one entry of candidatesForAffineGap (BaseAligner.cpp:2208-2225 / :2279-2296)
    __device__ __forceinline__ void record_candidate(int dir, int64_t loc, int64_t orig_loc, int score, int used_ag, int clip_before,
                                                     int clip_after, int ag_score, double mp, int seed_offset) {
        if (agc == nullptr) return;                                        // no buffer: nothing is kept (:2202 NULL != candidatesForAffineGap)
        if (n_agc >= agc_cap) { agc_overflow = 1; return; }
        snapgpu_single_result *r = &agc[n_agc];
        if (lane == 0) {
            r->direction = dir; r->location = loc; r->orig_location = orig_loc; r->mapq = 0; r->score = score;
            r->status = SNAPGPU_MultipleHits; r->clipping_for_read_adjustment = 0; r->used_affine_gap_scoring = used_ag;
            r->bases_clipped_before = clip_before; r->bases_clipped_after = clip_after; r->ag_score = ag_score;
            r->match_probability = mp; r->seed_offset = seed_offset; r->reserved = (uint32_t)score;      // reserved = sort key of alignAffineGap
        }
        WAVE_SYNC();
        n_agc++;
    }

    // one entry of secondaryResults (BaseAligner.cpp:2182-2199 / :2253-2270).  The scratch holds 2 entries per scored candidate,
    // which is all updateBestScore can produce, so there is no overflow path (the reference doubles its buffer and re-runs).
    __device__ __forceinline__ void record_secondary(int dir, int64_t loc, int64_t orig_loc, int score, int used_ag, int clip_before,
                                                     int clip_after, int ag_score, double mp, int seed_offset) {
        if (n_sec >= sec_cfg.cap) { sec_overflow = 1; return; }            // (sized so that it cannot happen, see snapgpu_enable_secondary)
        snapgpu_single_result *r = &sec[n_sec];
        if (lane == 0) {
            r->status = SNAPGPU_MultipleHits; r->direction = dir; r->location = loc; r->orig_location = orig_loc; r->score = score;
            r->score_prior_to_clipping = 0; r->mapq = 0; r->clipping_for_read_adjustment = 0; r->used_affine_gap_scoring = used_ag;
            r->bases_clipped_before = clip_before; r->bases_clipped_after = clip_after; r->ag_score = ag_score; r->supplementary = 0;
            r->seed_offset = seed_offset; r->match_probability = mp; r->probability_all_candidates = 0.0; r->popular_seeds_skipped = 0;
            r->reserved = 0;
        }
        WAVE_SYNC();
        n_sec++;
    }

    // ScoreSet::updateBestScore (BaseAligner.cpp:2143-2299); SEC keeps secondary results, HAM keeps affine-gap candidates
    template <bool HAM>
    __device__ __forceinline__ bool update_best(ScoreSet &ss, int64_t loc, int64_t orig_loc, uint32_t score,
                                                int ag_score, double mp, const Elem *e, int e_dir,
                                                int e_used_ag, int e_clip_before, int e_clip_after, int e_seed_offset,
                                                double e_match_prob) {
        bool seen_new;
        if (cfg.use_ag) {
            seen_new = (ag_score > ss.ag_score) || (ss.ag_score == ag_score && mp > ss.p_best);
        } else {
            seen_new = (score < (uint32_t)ss.best_score) || (score == (uint32_t)ss.best_score && mp > ss.p_best);
        }
        if constexpr (SEC) {
            const uint32_t best = (uint32_t)ss.best_score;
            if (seen_new) {
                if (best >= score && (int)(best - score) <= sec_cfg.om) {                              // the displaced best, :2176
                    record_secondary(ss.dir, ss.best_loc, ss.best_orig_loc, ss.best_score, ss.used_ag, ss.clip_before, ss.clip_after,
                                     ss.ag_score, ss.best_match_prob, ss.seed_offset);
                }
            } else if ((int)(best - score) <= sec_cfg.om && score != (uint32_t)SNAPGPU_ScoreAboveLimit && best >= score) {              // :2247
                record_secondary(e_dir, loc, orig_loc, (int)score, e_used_ag, e_clip_before, e_clip_after, ag_score, e_match_prob, e_seed_offset);
            }
        }
        if constexpr (HAM) {
            const uint32_t best = (uint32_t)ss.best_score;
            if (seen_new) {
                if (best >= score && (int)(best - score) <= (int)cfg.extra_depth) {                    // the displaced best, :2202
                    record_candidate(ss.dir, ss.best_loc, ss.best_orig_loc, ss.best_score, ss.used_ag, ss.clip_before, ss.clip_after,
                                     ss.ag_score, ss.best_match_prob, ss.seed_offset);
                    if (agc_overflow) return false;
                }
            } else if ((int)(best - score) <= (int)cfg.extra_depth && score != (uint32_t)SNAPGPU_ScoreAboveLimit && best >= score) {   // :2273
                record_candidate(e_dir, loc, orig_loc, (int)score, e_used_ag, e_clip_before, e_clip_after, ag_score, e_match_prob, e_seed_offset);
            }
        }
        if (seen_new) {
            ss.best_score = (int32_t)score;
            ss.ag_score = ag_score;
            ss.p_best = mp;
            ss.best_loc = loc;
            ss.best_orig_loc = orig_loc;
            ss.dir = e_dir;
            ss.used_ag = e_used_ag;
            ss.clip_before = e_clip_before;
            ss.clip_after = e_clip_after;
            ss.seed_offset = e_seed_offset;
            ss.best_match_prob = e_match_prob;
        }
        (void)e;
        return seen_new;
    }

    __device__ __forceinline__ void fill_result(const ScoreSet &ss, snapgpu_single_result &r) const {  // :2301-2323
        r.ag_score = ss.ag_score;
        r.bases_clipped_after = ss.clip_after;
        r.bases_clipped_before = ss.clip_before;
        r.clipping_for_read_adjustment = 0;
        r.direction = ss.dir;
        r.location = ss.best_loc;
        r.orig_location = ss.best_orig_loc;
        r.mapq = compute_mapq(tab, ss.p_all, ss.p_best, (int)popular_seeds_skipped);
        r.score = ss.best_score;
        r.used_affine_gap_scoring = ss.used_ag;
        r.seed_offset = ss.seed_offset;
        r.match_probability = ss.best_match_prob;
        r.popular_seeds_skipped = popular_seeds_skipped;
        r.status = r.mapq >= 10 ? SNAPGPU_SingleHit : SNAPGPU_MultipleHits;       // MAPQ_LIMIT_FOR_SINGLE_HIT
        r.probability_all_candidates = ss.p_all;
    }

    // AffineGapVectorized::computeGaplessScore (AffineGapVectorized.h:139-254): Hamming walk away from the seed, the
    // best-scoring prefix is kept and the rest of the pattern is clipped.  st = +1 / -1; T, P, Q address the first byte.
    // Wave-uniform scalar code (all operands are LDS bytes); only used for reads nothing else could place.
    __device__ __forceinline__ int gapless_score(int st, const uint8_t *T, const uint8_t *P, const uint8_t *Q, int plen, int score_init,
                                                 int limit, int *n_edits, int *pattern_offset, double *mp, int *n_gapless) const {
        *mp = 1.0;
        if (limit < 0) { *n_edits = -1; *n_gapless = -1; return -1; }
        int sc = score_init, best = score_init, best_i = 0;
        for (int i = 0; i < plen; i++) {
            sc += (P[i * st] == T[i * st]) ? cfg.match_reward : -cfg.sub_penalty;
            if (sc > best) { best = sc; best_i = i; }
        }
        best = (int)first_u32((uint32_t)best); best_i = (int)first_u32((uint32_t)best_i);
        if (best > score_init) {
            int ne = 0, nm = 0;
            double p = 1.0;
            for (int i = 0; i <= best_i; i++) {
                if (P[i * st] != T[i * st]) { ne++; p *= tab->phred[Q[i * st]]; } else nm++;
            }
            p *= tab->perfect[nm];
            const int clipped = plen - (best_i + 1);
            *pattern_offset = clipped;
            ne = (int)first_u32((uint32_t)ne);
            *n_gapless = ne <= limit ? ne : -1;
            *n_edits = ne + clipped;
            p *= tab->indel[clipped];
            *mp = first_f64(p);
            return best;
        }
        *n_edits = -1; *n_gapless = -1;
        return -1;
    }

    // stage genome[loc - WIN_PAD, loc + read_len + WIN_PAD) into LDS with coalesced loads
    // (this pointer is LDS: said out loud for the kernels in which this object lives in memory -- the paired-end one, which passes its
    //  address around -- and its pointers come back from there as generic ones: flat stores that wait for every load and store in flight)
    template <class T> static __device__ __forceinline__ T *lds_ptr(T *p) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SNAPGPU_WAVE_EMU)
        __builtin_assume(__builtin_amdgcn_is_shared((const void *)p));
#endif
        return p;
    }
    __device__ __forceinline__ void stage_window(int64_t loc) {
        uint8_t *const gw = this->gw;
        const int total = read_len + 2 * WIN_PAD;
        const G(uint8_t) *src = (const G(uint8_t) *)ix.genome + (loc - WIN_PAD);       // (HBM, said out loud: where this object lives in LDS the pointer comes back generic)
        // 4 bytes per lane where the source is 4-byte aligned; byte loads at the ragged ends
        uintptr_t a = (uintptr_t)src;
        int head = (int)((4 - (a & 3)) & 3);
        if (head > total) head = total;
        if (lane < head) gw[lane] = src[lane];
        int words = (total - head) >> 2;
        const G(uint32_t) *s32 = (const G(uint32_t) *)(src + head);
        for (int w = lane; w < words; w += WAVE) {
            uint32_t v = s32[w];
            uint8_t *d = gw + head + 4 * w;
            d[0] = (uint8_t)v; d[1] = (uint8_t)(v >> 8); d[2] = (uint8_t)(v >> 16); d[3] = (uint8_t)(v >> 24);
        }
        int done = head + 4 * words;
        if (done + lane < total) gw[done + lane] = src[done + lane];
        WAVE_SYNC();
    }

    // the candidate's window as bit planes: the blocks that hold genome[loc - WIN_PAD, loc + read_len + WIN_PAD), one coalesced load
    __device__ __forceinline__ void stage_planes(int64_t loc) {
        const int64_t bit0 = loc - WIN_PAD + (int64_t)ix.genome_pad;         // position in the padded genome = bit of the plane array
        const int64_t blk0 = bit0 >> 6;
        tp_org = (int)(bit0 & 63) + WIN_PAD;
        const int nb = (int)text_plane_blocks(cfg.RL, WIN_PAD);
        for (int l = lane; l < 3 * nb; l += WAVE) {
            const int b = l / 3, pl = l - 3 * b;
            tp[pl * nb + b] = ix.planes[(blk0 + b) * 3 + pl];
        }
        WAVE_SYNC();
    }
    // bit planes of rd[dir][0 .. len): code bits, 'N' / 'n', any other byte (planes.h: LvPlanes)
    __device__ __forceinline__ void build_read_planes(int len) {
        const int rpw = (int)read_plane_words(cfg.RL);
        uint32_t not_plain = 0;
        for (int dir = 0; dir < 2; dir++) {
            for (int w = 0; w < rpw; w++) {
                const int i = w * 64 + lane;
                const uint8_t c = i < len ? rd[dir][i] : (uint8_t)0;
                const uint32_t v = base_value(c);
                const bool in = i < len, non = v > 3u, nn = c == 'N' || c == 'n';
                const unsigned long long p0 = BALLOT(in && (non ? c == 'n' : (v & 1u))), p1 = BALLOT(in && !non && (v & 2u));
                const unsigned long long pn = BALLOT(in && nn), po = BALLOT(in && non && !nn);
                if (lane == 0) {
                    unsigned long long *base = rp + (size_t)dir * 4 * rpw;
                    base[w] = p0; base[rpw + w] = p1; base[2 * rpw + w] = pn; base[3 * rpw + w] = po;
                }
                if (pn | po) not_plain |= 1u << dir;
            }
        }
        rd_plain = ~not_plain & 3u;
        WAVE_SYNC();
    }

    // One candidate location, evaluated: what the body of BaseAligner::score computes for it before any bookkeeping (BaseAligner.cpp:
    // 1108-1347) -- Landau-Vishkin on both sides of the seed, affine gap on both sides when that found more than maxKSame edits and the
    // element can still matter.  A pure function of (read, location, direction, seed offset, limit, best_all); its only side effects are
    // the work counters and the traceback-step notes (note_ag_call), which is what lets another wave do it (se_help.h).
    struct CandEval { uint32_t sc; double mp; int64_t loc; int used_ag, clip_before, clip_after, ag_score; uint32_t lv_sum_high;
                      int lv1, lv2; };     // what Landau-Vishkin alone said for the two sides (-1: above the limit, lv2 -2: not run)
    template <bool HAM>
    __device__ __forceinline__ CandEval eval_candidate(int64_t loc, int e_dir, uint32_t e_lps, int cand_seed_offset, int limit_e, uint32_t best_all) {
        uint32_t lv_sum_high = 0;
        int lv1 = -1, lv2 = -2;
        uint32_t sc = (uint32_t)SNAPGPU_ScoreAboveLimit;
        double mp = 0.0;
        const int64_t glen = (int64_t)read_len + SNAPGPU_MAX_K;
        int used_ag = 0, clip_before = 0, clip_after = 0, ag_score = -1;

        // Landau-Vishkin works on bit planes (planes.h) when the context has them and the limit's 2k + 1 diagonals fit
        // the wave; the byte window is only staged for what reads bytes: the gapless walk, affine gap
        const bool lv_planes = PLANES && !HAM && ix.planes != nullptr && limit_e <= 31;
        bool sub_ok = false;
        if (substring_in_range(loc, glen)) {                 // Genome::getSubstring (Genome.h:339-367): its "is this padding" byte comes with the window
            bool first_is_pad;
            stage_window(loc);                               // (the first levels of Landau-Vishkin, the gapless walk and affine gap read bytes)
            if (lv_planes) stage_planes(loc);
            first_is_pad = first_u32(gw[WIN_PAD]) == 'n';
            sub_ok = substring_ok_known(loc, glen, first_is_pad);
        }
        if (sub_ok) {
            const uint8_t *data = gw + WIN_PAD;                   // data[i] = genome[loc + i]
            const int seed_len = (int)ix.seed_len;
            const int seed_offset = cand_seed_offset;
            const int tail_start = seed_offset + seed_len;
            int ag1 = seed_len, ag2 = 0;
            int score1 = 0, score2 = 0;
            double mp1 = 1.0, mp2 = 1.0;
            int loc_offset = 0;
            const int text_len = read_len + SNAPGPU_MAX_K - tail_start;

            // Two halves, one call site each for LV and AG (half 0: forward from the end of the seed
            // over the tail of the read, :1160; half 1: backwards from the start of the seed over the
            // reversed head of the read, :1169).  The backward text/pattern are the same bytes walked
            // with stride -1, so LandauVishkin<1> and <-1> are one instantiation.
            const uint8_t *rdd = e_dir ? rd[1] : rd[0], *qld = e_dir ? ql[1] : ql[0];
            int g1 = 0, g2 = 0;                                     // score1Gapless / score2Gapless
            if constexpr (HAM) {                                    // :1177-1199
                if (tail_start != read_len) {
                    int po;
                    ag1 = gapless_score(+1, data + tail_start, rdd + tail_start, qld + tail_start, read_len - tail_start, read_len,
                                        limit_e, &score1, &po, &mp1, &g1);
                    ag1 += seed_len - read_len;
                }
                if (g1 != -1 && seed_offset != 0) {
                    int po = 0;
                    ag2 = gapless_score(-1, data + seed_offset - 1, rdd + seed_offset - 1, qld + seed_offset - 1, seed_offset, read_len,
                                        limit_e - g1, &score2, &po, &mp2, &g2);
                    ag2 -= read_len;
                    loc_offset = g2 != -1 ? po : 0;
                }
            }
            const uint64_t t_lv0 = clk();
            for (int half = 0; half < 2 && !HAM; half++) {
                if (half == 1 && score1 == -1) break;
                const int st = half == 0 ? 1 : -1;
                const int org = half == 0 ? tail_start : seed_offset - 1;
                const int plen = half == 0 ? read_len - tail_start : seed_offset;
                const int tlen = half == 0 ? text_len : seed_offset + SNAPGPU_MAX_K;
                const int lim = half == 0 ? limit_e : limit_e - score1;
                const LdsSeq P = lds_seq(ByteSeq{rdd + org, st}), Q = lds_seq(ByteSeq{qld + org, st}), T = lds_seq(ByteSeq{data + org, st});     // (read, qualities, window: LDS)
                LvPlanes lp;
                if (lv_planes) {
                    const int rpw = (int)read_plane_words(cfg.RL);
                    const LDS_AS unsigned long long *rb = (const LDS_AS unsigned long long *)(unsigned long long *)rp + (e_dir ? 4 * rpw : 0);
                    const LDS_AS unsigned long long *tb = (const LDS_AS unsigned long long *)(unsigned long long *)tp;
                    lp.p0 = rb; lp.p1 = rb + rpw; lp.pn = rb + 2 * rpw; lp.po = rb + 3 * rpw;
                    const int tpb = (int)text_plane_blocks(cfg.RL, WIN_PAD);
                    lp.t0 = tb; lp.t1 = tb + tpb; lp.tn = tb + 2 * tpb;
                    lp.p_org = org; lp.t_org = tp_org + org; lp.st = st; lp.p_words = rpw; lp.t_words = tpb;
                    lp.work = (LDS_AS unsigned long long *)(unsigned long long *)lvp; lp.plain = ((rd_plain >> e_dir) & 1u) != 0;
                }
                LVResult r = lv_compute(P, Q, plen, T, tlen, lim, lv_tri, cfg.kmax, tab, cfg.RL, lv_planes ? &lp : nullptr);
                // results are wave-uniform; say so, so they (and everything derived from them) live in SGPRs
                r.score = (int)first_u32((uint32_t)r.score); r.net_indel = (int)first_u32((uint32_t)r.net_indel);
                r.match_probability = first_f64(r.match_probability);
                if (half == 0) {
                    score1 = r.score; mp1 = r.match_probability;
                    ag1 = (seed_len + read_len - tail_start - score1) * cfg.match_reward - score1 * cfg.sub_penalty;
                    cnt().lv_ref_bytes += (uint64_t)plen + (uint64_t)(2 * (limit_e < 0 ? 0 : limit_e));
                } else {
                    score2 = r.score; mp2 = r.match_probability; loc_offset = r.net_indel;
                    ag2 = (seed_offset - score2) * cfg.match_reward - score2 * cfg.sub_penalty;
                    cnt().lv_ref_bytes += (uint64_t)plen;
                }
            }
            if (!HAM) cnt().lv++;
            cnt().cyc_lv += clk() - t_lv0;
            lv1 = score1; lv2 = score1 == -1 ? -2 : score2;

            if (!HAM && score1 != -1 && score2 != -1) {
                int max_k_same = cfg.gap_open / (cfg.sub_penalty - cfg.gap_extend);     // :1148
                lv_sum_high = score1 + score2 > max_k_same ? 1u : 0u;
                if (cfg.use_ag && (lv_sum_high && e_lps <= best_all)) {   // :1203
                    if (lv_planes) stage_window(loc);             // affine gap reads bytes
                    score1 = 0; score2 = 0; ag1 = seed_len; ag2 = 0;
                    used_ag = 1;
                    cnt().ag++;
                    if (++ag_calls_unit == WAVE_PRIO_HEAVY_AFTER) wave_set_priority(1);
                    const uint64_t t_ag0 = clk();
                    AGParams agp{cfg.match_reward, cfg.sub_penalty, cfg.gap_open, cfg.gap_extend, cfg.five_bonus, cfg.three_bonus};
                    for (int half = 0; half < 2; half++) {
                        if (half == 0 && tail_start == read_len) continue;               // :1208
                        if (half == 1 && (score1 == -1 || seed_offset == 0)) break;      // :1244-1245
                        const int st = half == 0 ? 1 : -1;
                        const int org = half == 0 ? tail_start : seed_offset - 1;
                        const int plen = half == 0 ? read_len - tail_start : seed_offset;
                        const int lim = half == 0 ? limit_e : limit_e - score1;
                        const int tlen = half == 0 ? text_len : seed_offset + lim;
                        const bool banded = plen >= 3 * (2 * lim + 1);                   // :1213 / :1251
                        const LdsSeq P = lds_seq(ByteSeq{rdd + org, st}), Q = lds_seq(ByteSeq{qld + org, st}), T = lds_seq(ByteSeq{data + org, st});
                        note_ag_extent(half, banded, plen, lim, tlen);
                        AGResult a = ag_dispatch<AGC, EXACT>(banded, st, agp, P, Q, plen, T, tlen, lim, read_len, e_dir != 0,
                                                      false, ag_rows, EXACT ? (half == 0 ? ag_persist0 : ag_persist1) : (uint8_t *)ag_scratch, cfg.RL, tab, EXACT ? ag_tag : 0u);
                        a.ag_score = (int)first_u32((uint32_t)a.ag_score); a.n_edits = (int)first_u32((uint32_t)a.n_edits);
                        a.pattern_offset = (int)first_u32((uint32_t)a.pattern_offset); a.text_offset = (int)first_u32((uint32_t)a.text_offset);
                        a.stale_reads = (int)first_u32((uint32_t)a.stale_reads); a.match_probability = first_f64(a.match_probability);
                        note_ag_call(half, (uint32_t)a.stale_reads);
                        if (half == 0) {
                            ag1 = a.ag_score + (seed_len - read_len); clip_after = a.pattern_offset;
                            score1 = a.n_edits; mp1 = a.match_probability;
                        } else {
                            ag2 = a.ag_score - read_len; clip_before = a.pattern_offset;
                            score2 = a.n_edits; mp2 = a.match_probability; loc_offset = a.text_offset;
                        }
                    }
                    cnt().cyc_ag += clk() - t_ag0;
                }
            }

            bool found = HAM ? (g1 != -1 && g2 != -1) : (score1 != -1 && score2 != -1);               // :1293
            if (found && loc_offset != 0 && !substring_ok(loc + loc_offset, glen)) found = false;   // :1295-1301
            if (found) {
                sc = (uint32_t)(score1 + score2);
                mp = mp1 * mp2 * tab->seed_prob;                  // :1314
                loc += loc_offset;
                ag_score = ag1 + ag2;
            } else {
                sc = (uint32_t)SNAPGPU_ScoreAboveLimit;
                ag_score = SNAPGPU_ScoreAboveLimit;
                mp = 0.0;
            }
        } else {
            mp = 0.0;
        }

        CandEval ce; ce.sc = sc; ce.mp = mp; ce.loc = loc; ce.used_ag = used_ag; ce.clip_before = clip_before; ce.clip_after = clip_after;
        ce.ag_score = ag_score; ce.lv_sum_high = lv_sum_high; ce.lv1 = lv1; ce.lv2 = lv2;
        return ce;
    }

    // ------------------------------------------------------------------ help for heavy reads (se_help.h)
    __device__ __forceinline__ bool se_wanted() const { return se_eager || XW::ld(se_ctl[1]) != 0u; }

    // List what the forced walk still has to visit -- weight lists from wl down, first in first out, every unscored candidate of each
    // element -- and offer it to the idle waves.
    __device__ __forceinline__ void se_publish(uint32_t wl) {
        // a slot first (the list is only worth building if there is one); none free: look again some elements further on
        int s = -1;
        if (lane == 0) {
            const uint32_t start = (cur_read * 7u) % se_n_slots;
            for (uint32_t k = 0; k < se_n_slots; k++) {
                const uint32_t i = start + k < se_n_slots ? start + k : start + k - se_n_slots;
                if (XW::ld_lane(se_slots[i].state) != 0u) continue;
                if (atomicCAS(&se_slots[i].state, 0u, 3u) == 0u) { s = (int)i; break; }
            }
        }
        s = (int)first_u32((uint32_t)s);
        if (s < 0) { se_tried = 0x80000000u | 32u; return; }
        se_tried = 1;
        for (uint32_t i = (uint32_t)lane; i < n_used; i += WAVE) se_first[i] = 0xffffffffu;
        WAVE_SYNC();
        uint32_t n = 0;
        for (uint32_t w = wl; w >= 1 && n + BUCKET <= cfg.se_items_cap; w--) {
            uint16_t ei = get_next(sent(w));
            while (ei < SENT_MIN && n + BUCKET <= cfg.se_items_cap) {
                const uint32_t *ep = (const uint32_t *)&pool[ei];
                const uint32_t v = lane < 4 ? ep[lane] : (lane == 4 ? ep[17] : 0u);          // used, scored, wnext | wprev << 16
                const uint64_t used = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)v, 1) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
                const uint64_t scored = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)v, 3) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 2);
                const uint16_t nxt = (uint16_t)((uint32_t)__builtin_amdgcn_readlane((int)v, 4) & 0xffffu);
                const uint64_t m = used & ~scored;
                if (lane == 0) se_first[ei] = n;
                if (lane < BUCKET && ((m >> lane) & 1ull)) se_items[n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)ei << 6) | (uint32_t)lane;
                n += (uint32_t)__popcll(m);
                ei = nxt;
            }
        }
        WAVE_SYNC();
        if (n < SE_HELP_MIN_ITEMS || n > se_spec_cap) { if (lane == 0) atomicExch(&se_slots[s].state, 0u); return; }     // not worth it: give the slot back
        SEHelpSlot *slot = &se_slots[s];
        SESpec *spec = se_spec + (size_t)s * se_spec_cap;
        for (uint32_t t = (uint32_t)lane; t < n; t += WAVE) spec[t].state = 0u;
        WAVE_SYNC();
        if (lane == 0) {
            const uint32_t w0 = atomicExch(&slot->next, SE_HELP_LOOKAHEAD), w1 = atomicExch(&slot->helpers, 0u);
            if ((w0 ^ w1) == 0xFFFFFFF5u) atomicExch(&slot->helpers, 0u);      // (uses both return values: the exchanges have completed)
            XW::st(slot->read, cur_read); XW::st(slot->n, n); XW::st(slot->owner_pos, 0u);
            XW::st(slot->lim_alt, (int32_t)score_limit(true)); XW::st(slot->lim_non_alt, (int32_t)score_limit(false));
            XW::st(slot->best_all, (uint32_t)all().best_score);
            XW::st(*(uint64_t *)&slot->items, (uint64_t)(uintptr_t)se_items); XW::st(*(uint64_t *)&slot->pool, (uint64_t)(uintptr_t)(Elem *)pool);
            XW::st(*(uint64_t *)&slot->spec, (uint64_t)(uintptr_t)spec);
            XW::fence_release();                            // the candidate table, the list and the cleared records, for the other XCDs
            atomicExch(&slot->state, 1u);
            atomicAdd(&se_ctl[2], 1u);
            if (se_diag) atomicAdd(&se_diag[1], 1ull << 32);
        }
        WAVE_SYNC();
        se_slot = s; se_n = n; se_mine = spec;
    }

    // The owner arrives at candidate idx of element ei: take what an idle wave stored for it, if that was computed from the inputs the
    // owner has now; claim it otherwise (false: evaluate it here).
    __device__ __forceinline__ bool se_take(uint16_t ei, int idx, uint64_t listed0, int limit_e, uint32_t e_lps, int64_t loc0, int cand_seed_offset, CandEval &ce) {
        const uint32_t t0 = first_u32(se_first[ei]);
        if (t0 == 0xffffffffu) return false;
        const uint32_t t = t0 + (uint32_t)__popcll(listed0 & ((1ull << idx) - 1ull));
        if (t >= se_n) return false;
        SEHelpSlot *slot = &se_slots[se_slot];
        SESpec *sp = &se_mine[t];
        if (lane == 0) XW::st(slot->owner_pos, t + 1u);
        const uint64_t t_w0 = wave_clock();
        for (;;) {
            uint32_t st = 0;
            if (lane == 0) st = atomicCAS(&sp->state, 0u, 1u);
            st = first_u32(st);
            if (st == 0u) return false;                     // untouched: the owner's now
            if (st == 2u) break;
            XW::nap();
            if (wave_clock() - t_w0 > 1200000000ull) {      // ~0.5 s on one candidate: stop relying on the slot, leave a trace
                if (lane == 0 && se_diag) atomicAdd(&se_diag[0], 1ull);
                se_abandon();
                return false;
            }
        }
        const int lim = XW::ld(sp->limit);
        const uint32_t high = XW::ld(sp->lv_sum_high), best_then = XW::ld(sp->best_all);
        const uint32_t n_lv = XW::ld(sp->n_lv);
        uint32_t n_ag = XW::ld(sp->n_ag), stale = XW::ld(sp->stale);
        uint64_t bytes = XW::ld(sp->lv_ref_bytes);
        if (lim == limit_e) {
            if (high && ((e_lps <= best_then) != (e_lps <= (uint32_t)all().best_score))) { if (lane == 0 && se_diag) atomicAdd(&se_diag[-1], 1ull << 16); return false; }  // the affine-gap decision (:1203) would differ
            ce.sc = XW::ld(sp->sc); ce.mp = XW::ld(sp->mp); ce.loc = XW::ld(sp->loc); ce.used_ag = XW::ld(sp->used_ag);
            ce.clip_before = XW::ld(sp->clip_before); ce.clip_after = XW::ld(sp->clip_after); ce.ag_score = XW::ld(sp->ag_score);
        } else {
            // Evaluated under a larger limit (limits only fall while a read is scored).  Landau-Vishkin's answers for e <= k do not depend on
            // k (LandauVishkin.h:100-351; SURVEY.md Appendix C: 12 M comparisons), so what it says under the owner's limit follows: the
            // same when the edits of both sides still fit, "above the limit" otherwise.  Affine gap is another matter -- its band is the limit --
            // so an evaluation in which it ran, or would run now, is redone.
            if (lim < limit_e || n_lv == 0u) { if (lane == 0 && se_diag) atomicAdd(&se_diag[-1], 1ull); return false; }
            const int lv1 = XW::ld(sp->lv1), lv2 = XW::ld(sp->lv2);
            const int plen0 = read_len - (cand_seed_offset + (int)ix.seed_len), plen1 = cand_seed_offset;
            const bool half1_runs = lv1 >= 0 && lv1 <= limit_e;
            bytes = (uint64_t)plen0 + (uint64_t)(2 * (limit_e < 0 ? 0 : limit_e)) + (half1_runs ? (uint64_t)plen1 : 0ull);
            if (half1_runs && lv2 >= 0 && lv1 + lv2 <= limit_e) {                 // both sides still fit
                if (n_ag != 0u || (high && e_lps <= (uint32_t)all().best_score)) {      // affine gap ran under another band, or would run now
                    if (lane == 0 && se_diag) atomicAdd(&se_diag[-1], 1ull << 16);
                    return false;
                }
                ce.sc = XW::ld(sp->sc); ce.mp = XW::ld(sp->mp); ce.loc = XW::ld(sp->loc); ce.used_ag = 0;
                ce.clip_before = 0; ce.clip_after = 0; ce.ag_score = XW::ld(sp->ag_score);
            } else {                                                             // :1293-1347 with score1 or score2 == -1
                ce.sc = (uint32_t)SNAPGPU_ScoreAboveLimit; ce.mp = 0.0; ce.loc = loc0; ce.used_ag = 0; ce.clip_before = 0; ce.clip_after = 0;
                ce.ag_score = SNAPGPU_ScoreAboveLimit;
                n_ag = 0; stale = 0;
            }
        }
        ce.lv_sum_high = high; ce.lv1 = 0; ce.lv2 = 0;
        cnt().lv += n_lv; cnt().ag += n_ag; cnt().lv_ref_bytes += bytes;
        // a speculative evaluation cannot know what this read's aligner objects scored before: all of its out-of-band steps count as
        // later-call ones, and from here on the objects count as used
        ag_stale += stale; ag_replay += stale;
        if (n_ag) { ag_obj_used0 = 1; ag_obj_used1 = 1; ag_calls_unit += n_ag; }
        if (lane == 0 && se_diag) atomicAdd(&se_diag[1], 1ull);
        return true;
    }

    __device__ __forceinline__ void se_abandon() {          // after a watchdog: the slot is retired for the rest of the launch
        if (lane == 0) { atomicExch(&se_slots[se_slot].state, 4u); atomicSub(&se_ctl[2], 1u); }
        se_slot = -1;
    }

    // end of the read: no new helpers, wait for the attached ones to leave (they read this wave's candidate table), free the slot
    __device__ __forceinline__ void se_close() {
        SEHelpSlot *slot = &se_slots[se_slot];
        uint32_t was = 0;
        if (lane == 0) { was = atomicExch(&slot->state, 2u); atomicSub(&se_ctl[2], 1u); }
        was = first_u32(was);
        bool gave_up = was != 1u;
        const uint64_t t0 = wave_clock();
        while (!gave_up) {
            if (XW::aload(&slot->helpers) == 0u) break;
            XW::nap();
            if (wave_clock() - t0 > 4800000000ull) { if (lane == 0 && se_diag) atomicAdd(&se_diag[0], 1ull << 16); gave_up = true; }
        }
        if (lane == 0) atomicExch(&slot->state, gave_up ? 4u : 0u);
        se_slot = -1;
    }

    // A wave that has run out of reads, attached to `slot` (the read is in this wave's LDS): evaluate candidates of the list until none
    // are left or the owner closes.
    __device__ __forceinline__ void se_help_slot(SEHelpSlot *slot) {
        const uint32_t n = XW::ld(slot->n);
        const int lim_alt = XW::ld(slot->lim_alt), lim_non_alt = XW::ld(slot->lim_non_alt);
        const uint32_t best = XW::ld(slot->best_all);
        const uint32_t *items = (const uint32_t *)(uintptr_t)XW::ld(*(const uint64_t *)&slot->items);
        const Elem *opool = (const Elem *)(uintptr_t)XW::ld(*(const uint64_t *)&slot->pool);
        SESpec *spec = (SESpec *)(uintptr_t)XW::ld(*(const uint64_t *)&slot->spec);
        for (uint32_t it = 0; it <= n; it++) {
            if (XW::aload(&slot->state) != 1u) break;
            uint32_t c0 = 0;
            if (lane == 0) c0 = atomicAdd(&slot->next, SE_HELP_CHUNK);
            c0 = first_u32(c0);
            if (c0 >= n) break;
            const uint32_t c1 = c0 + SE_HELP_CHUNK < n ? c0 + SE_HELP_CHUNK : n;
            for (uint32_t t = c0; t < c1; t++) {
                if (t < XW::ld(slot->owner_pos)) continue;  // the owner is past it
                uint32_t st = 1;
                if (lane == 0) st = atomicCAS(&spec[t].state, 0u, 1u);
                if (first_u32(st) != 0u) continue;
                // (after a watchdog the owner retires the slot and reuses its list and pool for its next read while a stalled helper may
                //  still be here: look at the state again before every item, and never follow an element index outside the pool)
                if (XW::aload(&slot->state) != 1u) break;
                const uint32_t item = first_u32(items[t]);
                const uint32_t ei = item >> 6; const int idx = (int)(item & 63u);
                if (ei >= cfg.pool_size) continue;
                const uint32_t *ep = (const uint32_t *)&opool[ei];
                const uint32_t ew = lane < (int)(sizeof(Elem) / 4) ? ep[lane] : 0u;
                auto EW = [&](int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)ew, i); };
                const int64_t e_base = (int64_t)(((uint64_t)EW(5) << 32) | EW(4));
                const int e_dir = (int)((EW(18) >> 16) & 0xffu);
                const uint32_t e_lps = EW(11);
                const int cso = (int)((EW(20 + (idx >> 1)) >> (16 * (idx & 1))) & 0xffffu);
                const int limit_e = (cfg.alt_aware && is_alt(e_base)) ? lim_alt : lim_non_alt;
                CandEval ce; ce.sc = 0; ce.mp = 0; ce.loc = 0; ce.used_ag = 0; ce.clip_before = 0; ce.clip_after = 0; ce.ag_score = 0; ce.lv_sum_high = 0;
                ce.lv1 = -1; ce.lv2 = -2;
                uint32_t d_lv = 0, d_ag = 0, stale = 0; uint64_t d_bytes = 0;
                int rec_limit = (int)0x80000000;            // (an element the owner would skip under this limit: a record nobody can use)
                if ((int64_t)e_lps <= (int64_t)limit_e && e_base >= 0 && (uint64_t)(e_base + idx) < (uint64_t)ix.n_bases && idx < 48 && cso >= 0 && cso <= read_len) {
                    const uint64_t lv0 = cnt().lv, ag0 = cnt().ag, b0 = cnt().lv_ref_bytes;
                    ag_stale = 0; ag_replay = 0; ag_obj_used0 = 1; ag_obj_used1 = 1;
                    ce = eval_candidate<false>(e_base + idx, e_dir, e_lps, cso, limit_e, best);
                    d_lv = (uint32_t)(cnt().lv - lv0); d_ag = (uint32_t)(cnt().ag - ag0); d_bytes = cnt().lv_ref_bytes - b0; stale = ag_stale;
                    cnt().lv = lv0; cnt().ag = ag0; cnt().lv_ref_bytes = b0;        // (the owner counts what it uses)
                    WAVE_SYNC();
                    rec_limit = limit_e;
                }
                if (lane == 0) {
                    SESpec *sp = &spec[t];
                    XW::st(sp->limit, (int32_t)rec_limit); XW::st(sp->best_all, best); XW::st(sp->lv_sum_high, ce.lv_sum_high);
                    XW::st(sp->sc, ce.sc); XW::st(sp->ag_score, (int32_t)ce.ag_score); XW::st(sp->used_ag, (int32_t)ce.used_ag);
                    XW::st(sp->clip_before, (int32_t)ce.clip_before); XW::st(sp->clip_after, (int32_t)ce.clip_after);
                    XW::st(sp->n_lv, d_lv); XW::st(sp->n_ag, d_ag); XW::st(sp->stale, stale);
                    XW::st(sp->loc, ce.loc); XW::st(sp->mp, ce.mp); XW::st(sp->lv_ref_bytes, d_bytes);
                    XW::st(sp->lv1, (int32_t)ce.lv1); XW::st(sp->lv2, (int32_t)ce.lv2);
                    XW::stores_done();
                    atomicExch(&sp->state, 2u);
                    if (se_diag) atomicAdd(&se_diag[-1], 1ull << 32);       // (snapgpu_counters::reserved[0]: evaluations stored << 32 | refused: limit | AG << 16)
#if defined(SNAPGPU_WAVE_EMU)
                    if (getenv("SNAPGPU_DEBUG_SE_HELP")) fprintf(stderr, "HELPED read %u item %u limit %d sc %u n_ag %u\n", XW::ld(slot->read), t, rec_limit, ce.sc, d_ag);
#endif
                }
                WAVE_SYNC();
            }
        }
    }

    // ------------------------------------------------------------------ score()  (BaseAligner.cpp:918-1534)
    // returns true when a final answer has been written to `primary`.  HAM: the useHamming variant (gapless scoring with
    // clipping, candidates kept for alignAffineGap), used only by the paired-end fallback (ChimericPairedEndAligner.cpp:340).
    template <bool HAM>
    __device__ __forceinline__ bool score(bool force_result) {
        // :995-1007 (EXACT_DISJOINT_MISS_COUNT)
        if (cur_round_lps[0] > lps_unseen[0]) lps_unseen[0] = cur_round_lps[0];
        if (cur_round_lps[1] > lps_unseen[1]) lps_unseen[1] = cur_round_lps[1];
        uint32_t wl = highest_used_weight_list;
        do {
            while (wl > 0 && get_next(sent(wl)) == sent(wl)) {
                wl--;
                highest_used_weight_list = wl;
            }
            int lim_t = score_limit(true), lim_f = score_limit(false);
            int lim_max = lim_t > lim_f ? lim_t : lim_f;
            uint32_t lps_min = lps_unseen[0] < lps_unseen[1] ? lps_unseen[0] : lps_unseen[1];
            if ((int64_t)lps_min > (int64_t)lim_max || force_result) {
                if (wl < cfg.min_weight) {
                    // :1034-1056
                    bool fin_all;          // (a flag, not a pointer: selecting between &all and &non_alt would pin both in scratch)
                    first_alt().status = SNAPGPU_NotFound;
                    if (!cfg.alt_aware || non_alt().best_score > all().best_score + cfg.max_gap_alt) {
                        fin_all = true;
                    } else {
                        fin_all = false;
                        if (cfg.emit_alt && all().best_score <= non_alt().best_score && all().best_loc != non_alt().best_loc) {
                            fill_result(all(), first_alt());
                        }
                    }
                    const int fin_best = fin_all ? all().best_score : non_alt().best_score;
                    primary().score = fin_best;
                    if ((uint32_t)fin_best <= max_k || (HAM && fin_best != SNAPGPU_UnusedScoreValue)) {          // :1048
                        if (fin_all) fill_result(all(), primary()); else fill_result(non_alt(), primary());
                        primary().supplementary = 0;
                    } else {
                        primary().status = SNAPGPU_NotFound;
                        primary().mapq = 0;
                    }
                    return true;
                }
                force_result = true;
            } else if (wl == 0) {
                return false;
            }

            if constexpr (!EXACT && !HAM) {             // idle waves can take over part of what is left of a forced walk (se_help.h)
                if (force_result && se_slots != nullptr && se_slot < 0) {
                    if (se_tried & 0x80000000u) { se_tried = (se_tried & 0x7fffffffu) > 1u ? se_tried - 1u : 0u; }     // (no slot was free: count down to the next look)
                    else if (!se_tried && se_wanted()) se_publish(wl);
                }
            }
            uint16_t ei = get_next(sent(wl));
            Elem *e = &pool[ei];
            // The element (44 dwords) comes in with ONE coalesced load, lane i holding dword i; its fields are then lane reads.
            // (Field by field it was ~10 dependent L2 round trips per candidate.)  Fields this loop changes are tracked in
            // registers next to the stores.
            const uint32_t ew = lane < (int)(sizeof(Elem) / 4) ? ((const uint32_t *)e)[lane] : 0u;
            auto EW = [&](int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)ew, i); };
            static_assert(offsetof(Elem, wnext) == 68 && offsetof(Elem, base) == 16 && offsetof(Elem, match_prob) == 32 && offsetof(Elem, lps) == 44 &&
                          offsetof(Elem, best_score) == 48 && offsetof(Elem, hnext) == 72 && offsetof(Elem, dir) == 74 &&
                          offsetof(Elem, flags) == 75 && offsetof(Elem, cand_seed_offset) == 80, "Elem layout");
            int64_t e_base = (int64_t)(((uint64_t)EW(5) << 32) | EW(4));
            int e_dir = (int)((EW(18) >> 16) & 0xffu);
            uint32_t e_lps = EW(11);
            uint64_t e_scored = ((uint64_t)EW(3) << 32) | EW(2);
            uint32_t e_best_cur = EW(12);
            double e_mp_cur = __longlong_as_double((long long)(((uint64_t)EW(9) << 32) | EW(8)));
            uint32_t e_flags_cur = EW(18) >> 24;
            int limit_e = score_limit(cfg.alt_aware && is_alt(e_base));      // :1084
            if ((int64_t)e_lps <= (int64_t)limit_e) {
                uint64_t mask = ((uint64_t)EW(1) << 32) | EW(0);              // snapshot, :1088
                const uint64_t listed0 = mask & ~e_scored;                    // the candidates this visit evaluates (what se_publish listed)
                while (mask) {
                    int idx = __ffsll((long long)mask) - 1;
                    uint64_t bit = 1ull << idx;
                    mask &= ~bit;
                    uint64_t scored = e_scored;
                    if (scored & bit) continue;
                    bool any_nearby = scored != 0;
                    e_scored = scored | bit;
                    if (lane == 0) e->scored = e_scored;
                    WAVE_SYNC();

                    int64_t loc = e_base + idx;
                    const int64_t orig_loc = loc, elem_loc = loc;
                    bool loc_non_alt = !cfg.alt_aware || !is_alt(loc);

                    int cand_seed_offset = (int)((EW(20 + (idx >> 1)) >> (16 * (idx & 1))) & 0xffffu);
                    CandEval ce;
                    bool have = false;
                    if constexpr (!EXACT && !HAM) { if (se_slot >= 0) have = se_take(ei, idx, listed0, limit_e, e_lps, loc, cand_seed_offset, ce); }
                    if (!have) ce = eval_candidate<HAM>(loc, e_dir, e_lps, cand_seed_offset, limit_e, (uint32_t)all().best_score);
                    const uint32_t sc = ce.sc; const double mp = ce.mp; loc = ce.loc;
                    const int used_ag = ce.used_ag, clip_before = ce.clip_before, clip_after = ce.clip_after, ag_score = ce.ag_score;

                    // ---- bookkeeping after scoring one candidate (:1349-1519)
                    uint32_t e_best = e_best_cur;
                    double e_mp = e_mp_cur;
                    if (any_nearby) {
                        if (HAM && mp <= e_mp) continue;                                      // :1362
                        if (e_best < sc || (e_best == sc && mp <= e_mp)) continue;            // :1366
                    }
                    const uint32_t e_flags = e_flags_cur;
                    e_flags_cur = (e_flags & 1u) | (used_ag ? 2u : 0u);
                    if (lane == 0) {
                        e->best_loc = loc;
                        e->flags = (uint8_t)e_flags_cur;
                        e->clip_before = clip_before;
                        e->clip_after = clip_after;
                        e->ag_score = ag_score;
                        e->seed_offset = cand_seed_offset;
                    }
                    WAVE_SYNC();

                    // nearby bucket (the other half-bucket neighbour), :1396-1435
                    uint16_t ni = 0xFFFF;
                    if ((uint32_t)SNAPGPU_ScoreAboveLimit != sc && sc < 2) {
                        int64_t half = (int64_t)(((uint64_t)elem_loc % BUCKET) / (BUCKET / 2));
                        int64_t nearby_loc = elem_loc + (2 * half - 1) * (BUCKET / 2);
                        ni = find_element(nearby_loc, e_dir);
                    }
                    if (ni != 0xFFFF) {
                        Elem *ne = &pool[ni];
                        uint64_t n_scored = first_u64(ne->scored);
                        if (n_scored != 0) {
                            int64_t n_best_loc = (int64_t)first_u64((uint64_t)ne->best_loc);
                            int64_t dist = loc > n_best_loc ? loc - n_best_loc : n_best_loc - loc;
                            if (dist <= BUCKET) {                                             // genomeLocationIsWithin(..., maxMergeDist)
                                uint32_t n_best = first_u32(ne->best_score);
                                double n_mp = first_f64(ne->match_prob);
                                if (HAM && n_mp >= mp) continue;                             // :1418
                                if (n_best < sc || (n_best == sc && n_mp >= mp)) continue;   // :1421
                                double v = all().p_all - n_mp; all().p_all = v > 0.0 ? v : 0.0;    // updateProbabilitiesForNearbyMatch
                                if (loc_non_alt) { double u = non_alt().p_all - n_mp; non_alt().p_all = u > 0.0 ? u : 0.0; }
                                any_nearby = true;
                                if (lane == 0) ne->match_prob = 0;
                                WAVE_SYNC();
                            }
                        }
                    }

                    // updateProbabilitiesForNewMatch (:2137-2141): two separate FP64 operations
                    {
                        double v = all().p_all - e_mp; v = v > 0.0 ? v : 0.0; all().p_all = v + mp;
                        if (loc_non_alt) { double u = non_alt().p_all - e_mp; u = u > 0.0 ? u : 0.0; non_alt().p_all = u + mp; }
                    }
                    e_mp_cur = mp; e_best_cur = sc;
                    if (lane == 0) { e->match_prob = mp; e->best_score = sc; }
                    WAVE_SYNC();

                    update_best<HAM>(all(), loc, orig_loc, sc, ag_score, mp, e, e_dir, used_ag, clip_before, clip_after, cand_seed_offset, mp);
                    if (loc_non_alt) {
                        update_best<HAM>(non_alt(), loc, orig_loc, sc, ag_score, mp, e, e_dir, used_ag, clip_before, clip_after, cand_seed_offset, mp);
                    }
                    if (HAM && agc != nullptr && n_agc >= agc_cap) { agc_overflow = 1; return true; }   // :1475 (the caller grows the buffer and retries)

                    // -f (:1490-1505): the first location within maxK ends the search; MultipleHits and MAPQ 0, because nothing says it is the best
                    if (cfg.stop_on_first_hit && ((uint32_t)all().best_score <= max_k || (HAM && all().best_score != SNAPGPU_UnusedScoreValue))) {
                        if (cfg.alt_aware) fill_result(non_alt(), primary()); else fill_result(all(), primary());
                        primary().status = SNAPGPU_MultipleHits; primary().mapq = 0;
                        first_alt().status = SNAPGPU_NotFound;
                        WAVE_SYNC();
                        return true;
                    }
                    // early out: nothing can rescue MAPQ once the candidates' total probability reaches 4.9 (:1512)
                    double p_chk = cfg.alt_aware ? non_alt().p_all : all().p_all;
                    if (!SEC && p_chk >= 4.9) {                                                 // (&& -1 == maxEditDistanceForSecondaryResults)
                        if (cfg.alt_aware) fill_result(non_alt(), primary()); else fill_result(all(), primary());
                        first_alt().status = SNAPGPU_NotFound;
                        return true;
                    }
                }
            }

            // remove the element from its weight list (:1526-1529)
            {
                if (lane == 0) e->flags = (uint8_t)(e_flags_cur | 1u);
                WAVE_SYNC();
            }
            list_unlink(ei);
            if (lane == 0) { e->wnext = ei; e->wprev = ei; }
            WAVE_SYNC();
        } while (force_result);
        return false;
    }

    // ------------------------------------------------------------------ AlignRead (BaseAligner.cpp:273-763)
    __device__ __forceinline__ void seed_set_used(uint32_t i) {
        if (lane == 0) seed_used[i >> 5] |= (1u << (i & 31));
        WAVE_SYNC();
    }
    __device__ __forceinline__ bool seed_is_used(uint32_t i) const { return (seed_used[i >> 5] >> (i & 31)) & 1u; }

    __device__ __forceinline__ void align_read(const uint8_t *g_bases, const uint8_t *g_quals, int len) {
        const uint64_t t_read0 = clk();
        ag_obj_used0 = ag_obj_used1 = 0;                                      // a newly constructed aligner for every read
        ag_calls_unit = 0;
        align_read_inner<false>(g_bases, g_quals, len);
        if (ag_calls_unit >= WAVE_PRIO_HEAVY_AFTER) wave_set_priority(0);
        cnt().cyc_total += clk() - t_read0;
    }
    // the read into LDS, forward and reverse complement (BaseAligner.cpp:388-396); returns its number of 'N's
    __device__ __forceinline__ uint32_t load_read(const uint8_t *g_bases, const uint8_t *g_quals, int len) {
        uint32_t n_count = 0;
        for (int i0 = 0; i0 < len; i0 += WAVE) {
            int i = i0 + lane;
            uint8_t b = 0, q = 0;
            if (i < len) {
                b = g_bases[i]; q = g_quals[i];
                rd[0][i] = b; ql[0][i] = q;
                rd[1][len - 1 - i] = rc_base(b);
                ql[1][len - 1 - i] = q;
            }
            n_count += (uint32_t)__popcll(BALLOT(i < len && b == 'N'));
        }
        return n_count;
    }
    template <bool HAM>
    __device__ __forceinline__ void align_read_inner(const uint8_t *g_bases, const uint8_t *g_quals, int len) {
        read_len = len;
        if constexpr (SEC) { n_sec = 0; n_sec_raw = 0; sec_overflow = 0; }    // *nSecondaryResults = 0, :318-320
        // what the callers read after ANY return, the early ones included (the Hamming retry of the chimeric fallback builds `reserved`
        // from the step notes and looks at the candidate count: ChimericPairedEndAligner.cpp:273-274 starts both counts at 0 per read)
        ag_stale = 0; ag_replay = 0; n_agc = 0; agc_overflow = 0;
        // result = NotFound (:334-344); remaining fields as a zero-initialised struct
        primary().status = SNAPGPU_NotFound; primary().direction = 0;
        primary().location = SNAPGPU_InvalidGenomeLocation32; primary().orig_location = 0;
        primary().score = SNAPGPU_UnusedScoreValue; primary().score_prior_to_clipping = 0; primary().mapq = 0;
        primary().clipping_for_read_adjustment = 0; primary().used_affine_gap_scoring = 0;
        primary().bases_clipped_before = 0; primary().bases_clipped_after = 0; primary().ag_score = 0;
        primary().supplementary = 0; primary().seed_offset = 0; primary().match_probability = 0.0;
        primary().probability_all_candidates = 0.0; primary().popular_seeds_skipped = 0; primary().reserved = 0;
        first_alt() = primary();
        first_alt().location = 0; first_alt().score = 0;

        const int seed_len = (int)ix.seed_len;
        if (len < seed_len || len > (int)cfg.RL) return;                      // :360 (too long is rejected on the host)

        // load the read, build the reverse complement (:388-396)
        const uint32_t n_count = load_read(g_bases, g_quals, len);
        for (uint32_t i = lane; i < (cfg.RL + 31) / 32; i += WAVE) seed_used[i] = 0;
        WAVE_SYNC();
        if (n_count > max_k) return;                                          // :398
        if (PLANES && !HAM && ix.planes != nullptr) build_read_planes(len);

        if (n_count > 0) {                                                    // :407-420 block seeds containing a non-ACGT base
            int min_seed = 0;
            for (int i = 0; i < len; i++) {
                if (base_value(rd[0][i]) > 3) {
                    int limit = i + seed_len - 1 < len - 1 ? i + seed_len - 1 : len - 1;
                    int j0 = i - seed_len + 1; if (j0 < min_seed) j0 = min_seed;
                    for (int j = j0; j <= limit; j++) seed_set_used((uint32_t)j);
                    min_seed = limit + 1;
                    if (min_seed >= len) break;
                }
            }
        }

        uint32_t max_seeds_to_use = cfg.num_seeds != 0 ? cfg.num_seeds
                                  : (uint32_t)(int)(2 * cfg.seed_coverage * len / seed_len);    // :327-332
        clear_candidates();
        const uint32_t n_possible_seeds = (uint32_t)(len - seed_len + 1);
        uint32_t next_seed = 0;
        wrap_count = 0;
        lps_unseen[0] = lps_unseen[1] = 0;
        cur_round_lps[0] = cur_round_lps[1] = 0;
        all().init();
        non_alt().init();
        if (!cfg.alt_aware) non_alt().best_score = SNAPGPU_TooBigScoreValue;    // :325 (never re-initialised without ALT awareness)
        n_seeds_applied[0] = n_seeds_applied[1] = 0;
        popular_seeds_skipped = 0;
        se_tried = 0;

        // ONE call site for score(): it is inlined (a call would pin this object in memory), and the reference's three sites -- after a seed's
        // hits, when the seeds have wrapped seedLen times (:466), after the loop (:734) -- were three copies of the whole scoring code in the
        // kernel (12 400 instructions against a 64 KB instruction cache).  `force`: this is one of the two final calls.
        while (true) {
            bool force = false, applied_either = false;
            if (n_seeds_applied[0] + n_seeds_applied[1] >= max_seeds_to_use) force = true;                                     // the loop's condition failed: :734
            else if (next_seed >= n_possible_seeds && wrap_count + 1 >= (uint32_t)seed_len) { wrap_count++; force = true; }    // :455-470
            if (!force) do {                                                  // (one seed; `continue` = on to the next one, nothing to score)
            if (next_seed >= n_possible_seeds) {                              // wrapping, :455-504
                wrap_count++;
                next_seed = tab->wrapped_seed[wrap_count];
                cur_round_lps[0] = cur_round_lps[1] = 0;
            }
            while (next_seed < n_possible_seeds && seed_is_used(next_seed)) next_seed++;
            if (next_seed >= n_possible_seeds) continue;
            seed_set_used(next_seed);

            SeedBits seed = pack_seed(rd[0] + next_seed, (uint32_t)seed_len);
            if (!seed.valid) continue;                                        // :524

            HitList hl[2];
            const uint64_t t_lk0 = clk();
            lookup_seed(ix, seed, hl);
            const uint64_t t_lk1 = clk();
            cnt().cyc_lookup += t_lk1 - t_lk0;
            cnt().lookups++;
            cnt().slots += hl[0].slots + hl[1].slots;

            for (int dir = 0; dir < 2; dir++) {
                const int64_t dir_n_hits = dir ? hl[1].n_hits : hl[0].n_hits;
                const uint32_t dir_singleton = dir ? hl[1].singleton : hl[0].singleton;
                const uint32_t *dir_hits = dir ? hl[1].hits : hl[0].hits;
                if (dir_n_hits > (int64_t)cfg.max_hits && !cfg.explore_popular_seeds) {                 // too popular, :574-579
                    popular_seeds_skipped++;
                } else {
                    uint32_t offset = dir == 0 ? next_seed : (uint32_t)(len - seed_len) - next_seed;   // :591-606
                    int64_t limit = dir_n_hits < (int64_t)cfg.max_hits ? dir_n_hits : (int64_t)cfg.max_hits;      // :625 (only -x gets here with more than maxHits)
                    if (limit > 1) cnt().overflow_lists++;
                    cnt().hits += (uint64_t)limit;
                    for (int64_t c0 = 0; c0 < limit; c0 += WAVE) {
                        // one coalesced load of up to 64 hits, then consume them in stored order
                        uint32_t mine = 0;
                        int64_t i = c0 + lane;
                        if (i < limit) mine = (dir_n_hits == 1) ? dir_singleton : dir_hits[i];
                        int n = (int)(limit - c0 < WAVE ? limit - c0 : WAVE);
                        for (int j = 0; j < n; j++) {
                            uint32_t h = first_u32((uint32_t)__shfl((int)mine, j));
                            apply_hit(h, offset, dir);
                        }
                    }
                    if (dir) { n_seeds_applied[1]++; cur_round_lps[1]++; } else { n_seeds_applied[0]++; cur_round_lps[0]++; }
                    applied_either = true;
                }
            }
            cnt().cyc_hits += clk() - t_lk1;
            next_seed += (uint32_t)seed_len;                                  // :676
            } while (0);
            if (force || applied_either) {
                if (score<HAM>(force) || force) break;
            }
        }
        if constexpr (!EXACT && !HAM) { if (se_slot >= 0) se_close(); }       // (before the candidate table is released: helpers read it)
        primary().score_prior_to_clipping = primary().score;                      // finalizeSecondaryResults, :2442
        primary().reserved = (ag_stale & 0x3fffffffu) | (ag_replay ? 0x40000000u : 0u);      // bit 30: the exact pass must redo this read
        release_candidates();
        if constexpr (SEC) { n_sec_raw = n_sec; finalize_secondary(); }
    }

    // ------------------------------------------------------------------ BaseAligner::finalizeSecondaryResults (BaseAligner.cpp:2423-2553)
    // with ignoreAlignmentAdjustmentsForOm (the default, AlignerOptions.cpp:96).  Works on an index list (sec_ord) and a key per
    // entry (sec_key); the records themselves move once, when the kernel copies sec[sec_ord[k]] out.
    __device__ __forceinline__ int contig_of(int64_t loc) const {             // Genome::getContigNumAtLocation (Genome.cpp:560-600)
        int lo = 0, hi = (int)ix.n_contigs - 1;                               // last contig whose beginning <= loc
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if ((int64_t)ix.contig_begin[mid] <= loc) lo = mid; else hi = mid - 1;
        }
        return lo;
    }
    // stable sort of sec_ord[0..n) by sec_key[sec_ord[.]] -- what glibc's merge-sorting qsort leaves (:2516, :2550).
    // Rank by counting: O(n^2 / 64); n is a handful outside repeats.
    __device__ __forceinline__ void sec_stable_sort(uint32_t n) {
        uint32_t *tmp = sec_key + sec_cfg.cap;                                // second half of the key array: [cap, 2*cap)
        for (uint32_t i0 = 0; i0 < n; i0 += WAVE) {
            const uint32_t i = i0 + (uint32_t)lane;
            if (i < n) {
                const uint32_t me = sec_ord[i], k = sec_key[me];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < n; j++) {
                    const uint32_t kj = sec_key[sec_ord[j]];
                    rank += (kj < k || (kj == k && j < i)) ? 1u : 0u;
                }
                tmp[rank] = me;
            }
        }
        WAVE_SYNC(); __threadfence_block();
        for (uint32_t i = (uint32_t)lane; i < n; i += WAVE) sec_ord[i] = tmp[i];
        WAVE_SYNC(); __threadfence_block();
    }
    __device__ __forceinline__ void finalize_secondary() {
        int best = (int)first_u32((uint32_t)primary().score);
        if (sec_cfg.adjust) {                                                          // :2444-2463 (-ae)
            const AdjustScratch asc = adjust_scratch_at(adj_scratch, cfg.RL);
            const AdjustIx aix = adjust_ix(ix);
            {
                const AdjustOut o = adjust_alignment(aix, rd[0], rd[1], read_len, (int)first_u32((uint32_t)primary().status), (int)first_u32((uint32_t)primary().direction),
                                                     (long long)first_u64((uint64_t)primary().location), best, SNAPGPU_InvalidGenomeLocation32, asc);
                primary().status = o.status; primary().location = o.location; primary().score = o.score; primary().clipping_for_read_adjustment = o.clipping;
                best = o.status != SNAPGPU_NotFound ? o.score : SNAPGPU_TooBigScoreValue;
            }
            WAVE_SYNC(); __threadfence_block();
            for (uint32_t i = 0; i < n_sec; i++) {
                snapgpu_single_result *r = &sec[i];
                const int sc0 = (int)first_u32((uint32_t)r->score);
                const AdjustOut o = adjust_alignment(aix, rd[0], rd[1], read_len, (int)first_u32((uint32_t)r->status), (int)first_u32((uint32_t)r->direction),
                                                     (long long)first_u64((uint64_t)r->location), sc0, SNAPGPU_InvalidGenomeLocation32, asc);
                if (lane == 0) {
                    r->score_prior_to_clipping = sc0;
                    r->status = o.status; r->location = o.location; r->score = o.score; r->clipping_for_read_adjustment = o.clipping;
                }
                if (o.status != SNAPGPU_NotFound && o.score < best) best = o.score;
            }
            WAVE_SYNC(); __threadfence_block();
        }
        uint32_t n = n_sec;
        if (n == 0) return;
        int worst = best + sec_cfg.om; if (worst > (int)max_k) worst = (int)max_k;     // :2465
        for (uint32_t i = (uint32_t)lane; i < n; i += WAVE) { sec_ord[i] = i; sec_key[i] = (uint32_t)sec[i].score; }
        WAVE_SYNC(); __threadfence_block();
        // :2467-2485: drop what is now too far from the best, moving the last entry into the hole (order matters downstream)
        if (lane == 0) {
            uint32_t i = 0;
            while (i < n) {
                if ((int)sec_key[sec_ord[i]] > worst) { sec_ord[i] = sec_ord[n - 1]; n--; }
                else i++;
            }
        }
        n = first_u32(n);
        WAVE_SYNC(); __threadfence_block();
        for (uint32_t i = (uint32_t)lane; i < n; i += WAVE) {
            snapgpu_single_result *r = &sec[sec_ord[i]];
            if (!sec_cfg.adjust) r->score_prior_to_clipping = r->score;                 // :2478-2480
            r->supplementary = (cfg.alt_aware && is_alt(r->location)) ? 1 : 0;
        }
        WAVE_SYNC(); __threadfence_block();
        if (sec_cfg.mpc > 0 && primary().status != SNAPGPU_NotFound && n > 0) {           // :2487-2547
            const int primary_contig = contig_of(primary().location);
            // key = (contig, score); scores here are <= max_k <= 127
            for (uint32_t i = (uint32_t)lane; i < n; i += WAVE) {
                const uint32_t me = sec_ord[i];
                sec_key[me] = ((uint32_t)contig_of(sec[me].location) << 8) | (uint32_t)(sec[me].score & 0xff);
            }
            WAVE_SYNC(); __threadfence_block();
            // does any contig hold more than mpc (the primary counts for its own)?
            uint32_t too_many = 0;
            for (uint32_t i0 = 0; i0 < n; i0 += WAVE) {
                const uint32_t i = i0 + (uint32_t)lane;
                bool over = false;
                if (i < n) {
                    const uint32_t c = sec_key[sec_ord[i]] >> 8;
                    int count = (int)c == primary_contig ? 1 : 0;
                    for (uint32_t j = 0; j < n; j++) count += (sec_key[sec_ord[j]] >> 8) == c ? 1 : 0;
                    over = count > sec_cfg.mpc;
                }
                too_many |= BALLOT(over) != 0 ? 1u : 0u;
            }
            if (too_many) {
                sec_stable_sort(n);                                           // compareByContigAndScore
                if (lane == 0) {
                    int cur = -1, cur_count = 0; uint32_t dest = 0;
                    for (uint32_t src = 0; src < n; src++) {
                        const uint32_t me = sec_ord[src];
                        const int c = (int)(sec_key[me] >> 8);
                        if (c != cur) { cur = c; cur_count = c == primary_contig ? 1 : 0; }
                        cur_count++;
                        if (cur_count <= sec_cfg.mpc) sec_ord[dest++] = me;
                    }
                    n = dest;
                }
                n = first_u32(n);
                WAVE_SYNC(); __threadfence_block();
            }
        }
        if ((int64_t)n > sec_cfg.omax) {                                      // :2549-2552
            for (uint32_t i = (uint32_t)lane; i < n; i += WAVE) { const uint32_t me = sec_ord[i]; sec_key[me] = (uint32_t)sec[me].score; }
            WAVE_SYNC(); __threadfence_block();
            sec_stable_sort(n);                                               // compareByScore
            n = (uint32_t)sec_cfg.omax;
        }
        n_sec = n;
    }

    // ------------------------------------------------------------------ BaseAligner::scoreLocationWithAffineGap (BaseAligner.cpp:766-915)
    // (the forward half asks for the clipping optimisations, the backward half does not: :827 vs :857)
    __device__ __forceinline__ void score_location_ag(int dir, int64_t loc, int seed_offset, int limit, int *score, double *mp, int *offset,
                                                      int *clip_before, int *clip_after, int *ag_score) {
        const int64_t glen = (int64_t)read_len + SNAPGPU_MAX_K;
        *offset = 0;
        if (!substring_ok(loc, glen)) { *score = -1; *mp = 0; *ag_score = -1; return; }
        *clip_before = 0; *clip_after = 0;
        stage_window(loc);
        const uint8_t *data = gw + WIN_PAD;
        const int seed_len = (int)ix.seed_len;
        const int tail_start = seed_offset + seed_len;
        const uint8_t *rdd = dir ? rd[1] : rd[0], *qld = dir ? ql[1] : ql[0];
        int score1 = 0, score2 = 0, ag1 = seed_len, ag2 = 0;
        double mp1 = 1.0, mp2 = 1.0;
        AGParams agp{cfg.match_reward, cfg.sub_penalty, cfg.gap_open, cfg.gap_extend, cfg.five_bonus, cfg.three_bonus};
        for (int half = 0; half < 2; half++) {
            if (half == 0 && tail_start == read_len) continue;
            if (half == 1 && (score1 == -1 || seed_offset == 0)) break;
            const int st = half == 0 ? 1 : -1;
            const int org = half == 0 ? tail_start : seed_offset - 1;
            const int plen = half == 0 ? read_len - tail_start : seed_offset;
            const int lim = half == 0 ? limit : limit - score1;
            const int tlen = half == 0 ? (int)(glen - tail_start) : seed_offset + lim;
            const bool banded = plen >= 3 * (2 * lim + 1);
            const LdsSeq P = lds_seq(ByteSeq{rdd + org, st}), Q = lds_seq(ByteSeq{qld + org, st}), T = lds_seq(ByteSeq{data + org, st});
            note_ag_extent(half, banded, plen, lim, tlen);
            AGResult a = ag_dispatch<AGC, EXACT>(banded, st, agp, P, Q, plen, T, tlen, lim, read_len, dir != 0, half == 0, ag_rows,
                                                 EXACT ? (half == 0 ? ag_persist0 : ag_persist1) : (uint8_t *)ag_scratch, cfg.RL, tab, EXACT ? ag_tag : 0u);
            a.ag_score = (int)first_u32((uint32_t)a.ag_score); a.n_edits = (int)first_u32((uint32_t)a.n_edits);
            a.pattern_offset = (int)first_u32((uint32_t)a.pattern_offset); a.text_offset = (int)first_u32((uint32_t)a.text_offset);
            a.stale_reads = (int)first_u32((uint32_t)a.stale_reads); a.match_probability = first_f64(a.match_probability);
            note_ag_call(half, (uint32_t)a.stale_reads);
            if (half == 0) {
                ag1 = a.ag_score + (seed_len - read_len); *clip_after = a.pattern_offset; score1 = a.n_edits; mp1 = a.match_probability;
            } else {
                ag2 = a.ag_score - read_len; *clip_before = a.pattern_offset; score2 = a.n_edits; mp2 = a.match_probability; *offset = a.text_offset;
                if (score2 == -1) *offset = 0;
            }
        }
        if (score1 != -1 && score2 != -1) {
            *score = score1 + score2;
            *mp = mp1 * mp2 * tab->seed_prob_pow;             // :907: pow(double, unsigned) -- libm's pow, not the powi of :1314 (dev_common.h: DevTables)
            *ag_score = ag1 + ag2;
        } else {
            *score = -1; *ag_score = -1; *mp = 0.0;
        }
    }

    // ScoreSet::init(SingleAlignmentResult*) / updateBestScore(SingleAlignmentResult*) (BaseAligner.cpp:2116-2130, BaseAligner.h:310-328)
    __device__ __forceinline__ void set_from_result(ScoreSet &ss, const snapgpu_single_result &r) const {
        ss.best_score = r.score; ss.best_loc = r.location; ss.best_orig_loc = r.orig_location; ss.dir = r.direction;
        ss.used_ag = r.used_affine_gap_scoring; ss.clip_before = r.bases_clipped_before; ss.clip_after = r.bases_clipped_after;
        ss.ag_score = r.ag_score; ss.seed_offset = r.seed_offset; ss.best_match_prob = r.match_probability;
        ss.p_all = r.probability_all_candidates; ss.p_best = r.match_probability;
    }
    __device__ __forceinline__ bool set_update_from(ScoreSet &ss, int64_t loc, int64_t orig, int dir, int score, int used_ag, int cb, int ca,
                                                    int ag, int so, double mp) const {
        ss.p_all += mp;
        if (ag > ss.ag_score || (ag == ss.ag_score && mp > ss.best_match_prob)) {
            ss.best_score = score; ss.ag_score = ag; ss.best_match_prob = mp; ss.best_loc = loc; ss.best_orig_loc = orig; ss.dir = dir;
            ss.used_ag = used_ag; ss.clip_before = cb; ss.clip_after = ca; ss.seed_offset = so;
            return true;
        }
        return false;
    }
    static __device__ __forceinline__ void set_sub_all(ScoreSet &ss, double old) { double v = ss.p_all - old; ss.p_all = v > 0.0 ? v : 0.0; }

    // ------------------------------------------------------------------ BaseAligner::alignAffineGap (BaseAligner.cpp:1537-1792)
    // Runs on `primary` / `first_alt` and the candidates the preceding Hamming pass collected; the read is still in LDS.
    __device__ __forceinline__ void align_affine_gap(ScoreSet &A, ScoreSet &N) {
        if (primary().status == SNAPGPU_NotFound) return;
        uint32_t n_count = 0;
        for (int i0 = 0; i0 < read_len; i0 += WAVE) {
            int i = i0 + lane;
            n_count += (uint32_t)__popcll(BALLOT(i < read_len && rd[0][i] == 'N'));
        }
        if (n_count > max_k) return;
        const int best_score = (int)first_u32((uint32_t)primary().score);
        int limit = SNAPGPU_MAX_K - 1, limit_alt = SNAPGPU_MAX_K - 1;
        int g_off = 0;
        bool skip = false;
        const double old_p = primary().match_probability;
        const double old_p_alt = first_alt().status != SNAPGPU_NotFound ? first_alt().match_probability : 0.0;
        const int max_k_same = cfg.gap_open / (cfg.sub_penalty - cfg.gap_extend);

        primary().used_affine_gap_scoring = 0;
        if (primary().score > max_k_same) {
            primary().used_affine_gap_scoring = 1;
            int sc, cb = primary().bases_clipped_before, ca = primary().bases_clipped_after, ag = primary().ag_score;
            double mp = primary().match_probability;
            score_location_ag(primary().direction, primary().orig_location, primary().seed_offset, limit, &sc, &mp, &g_off, &cb, &ca, &ag);
            primary().score = sc; primary().match_probability = mp; primary().bases_clipped_before = cb; primary().bases_clipped_after = ca; primary().ag_score = ag;
            if (sc != -1) primary().location = primary().orig_location + g_off; else primary().status = SNAPGPU_NotFound;
        } else {
            skip = true;
        }
        if (first_alt().status != SNAPGPU_NotFound && first_alt().score > max_k_same) {
            first_alt().used_affine_gap_scoring = 1;
            int sc, cb = first_alt().bases_clipped_before, ca = first_alt().bases_clipped_after, ag = first_alt().ag_score;
            double mp = first_alt().match_probability;
            score_location_ag(first_alt().direction, first_alt().orig_location, first_alt().seed_offset, limit_alt, &sc, &mp, &g_off, &cb, &ca, &ag);
            first_alt().score = sc; first_alt().match_probability = mp; first_alt().bases_clipped_before = cb; first_alt().bases_clipped_after = ca; first_alt().ag_score = ag;
            if (sc != -1) first_alt().location = first_alt().orig_location + g_off; else first_alt().status = SNAPGPU_NotFound;
        }
        if (primary().status == SNAPGPU_NotFound || primary().score > SNAPGPU_MAX_K - 1) {
            primary().location = SNAPGPU_InvalidGenomeLocation32; primary().mapq = 0; primary().score = -1; primary().status = SNAPGPU_NotFound;
            primary().clipping_for_read_adjustment = 0; primary().used_affine_gap_scoring = 0; primary().bases_clipped_before = 0;
            primary().bases_clipped_after = 0; primary().ag_score = -1; primary().seed_offset = 0; primary().match_probability = 0.0;
            first_alt().status = SNAPGPU_NotFound;
            return;
        }

        bool non_alt_aln = !cfg.alt_aware || !is_alt(primary().location);
        set_from_result(A, primary());
        bool alt_best = false;
        if (first_alt().status != SNAPGPU_NotFound) {
            alt_best = set_update_from(A, first_alt().location, first_alt().orig_location, first_alt().direction, first_alt().score,
                                       first_alt().used_affine_gap_scoring, first_alt().bases_clipped_before, first_alt().bases_clipped_after,
                                       first_alt().ag_score, first_alt().seed_offset, first_alt().match_probability);
        }
        if (non_alt_aln) set_from_result(N, primary()); else N.init();
        if (!skip) {
            const double new_p = primary().match_probability;
            if (alt_best) { set_sub_all(A, old_p_alt); A.p_best = first_alt().match_probability; A.p_all += first_alt().match_probability; }
            else          { set_sub_all(A, old_p); A.p_best = new_p; A.p_all += new_p; }
            if (non_alt_aln) { set_sub_all(N, old_p); N.p_best = new_p; N.p_all += new_p; }
        }

        if (n_agc > 0 && !skip) {
            limit = (int)((max_k < (uint32_t)best_score ? max_k : (uint32_t)best_score) + cfg.extra_depth);      // :1714 (unsigned min, as in the reference)
            // qsort(compareByScore): glibc's merge sort is stable, so candidates are visited by (score, insertion index)
            int last_score = -0x7fffffff, last_idx = -1;
            for (uint32_t done_n = 0; done_n < n_agc; done_n++) {
                int bi = -1, bs = 0x7fffffff;
                for (uint32_t j0 = 0; j0 < n_agc; j0 += WAVE) {
                    uint32_t j = j0 + (uint32_t)lane;
                    int sj = j < n_agc ? (int)agc[j].reserved : 0x7fffffff;
                    bool ok = j < n_agc && (sj > last_score || (sj == last_score && (int)j > last_idx));
                    int key = ok ? sj : 0x7fffffff;
                    // wave minimum of (key, j)
                    for (int o = 32; o >= 1; o >>= 1) {
                        int ok2 = __shfl_xor(key, o), oj = __shfl_xor((int)j, o);
                        if (ok2 < key || (ok2 == key && oj < (int)j)) { key = ok2; j = (uint32_t)oj; }
                    }
                    key = (int)first_u32((uint32_t)key); j = first_u32(j);
                    if (key < bs) { bs = key; bi = (int)j; }
                }
                last_score = bs; last_idx = bi;
                snapgpu_single_result *c = &agc[bi];
                const int64_t c_loc = (int64_t)first_u64((uint64_t)c->location), c_orig = (int64_t)first_u64((uint64_t)c->orig_location);
                const int c_dir = (int)first_u32((uint32_t)c->direction), c_so = (int)first_u32((uint32_t)c->seed_offset);
                const bool c_non_alt = !cfg.alt_aware || !is_alt(c_loc);
                const double c_old_p = first_f64(c->match_probability);
                int sc, cb = (int)first_u32((uint32_t)c->bases_clipped_before), ca = (int)first_u32((uint32_t)c->bases_clipped_after);
                int ag = (int)first_u32((uint32_t)c->ag_score);
                double mp = c_old_p;
                score_location_ag(c_dir, c_orig, c_so, limit, &sc, &mp, &g_off, &cb, &ca, &ag);
                sc = (int)first_u32((uint32_t)sc);
                if (sc != -1 && sc <= SNAPGPU_MAX_K - 1) {
                    const int64_t new_loc = c_orig + g_off;
                    if (primary().location == new_loc) continue;                                   // same alignment again: do not lower MAPQ
                    set_sub_all(A, c_old_p);
                    set_update_from(A, new_loc, c_orig, c_dir, sc, 1, cb, ca, ag, c_so, mp);
                    if (c_non_alt) { set_sub_all(N, c_old_p); set_update_from(N, new_loc, c_orig, c_dir, sc, 1, cb, ca, ag, c_so, mp); }
                    limit = score_limit(cfg.alt_aware && !c_non_alt);                             // the *member* score sets, as the reference does (:1756)
                }
            }
        }

        const bool emit_all = !cfg.alt_aware || N.best_score > A.best_score + cfg.max_gap_alt;
        const uint32_t pop = primary().popular_seeds_skipped, pop_alt = first_alt().popular_seeds_skipped;
        const uint32_t saved_pop = popular_seeds_skipped;
        popular_seeds_skipped = pop;
        if (emit_all) fill_result(A, primary()); else fill_result(N, primary());
        if (cfg.alt_aware && !emit_all && A.best_loc != N.best_loc) {
            popular_seeds_skipped = pop_alt;
            fill_result(A, first_alt());
            first_alt().supplementary = 1;
        } else {
            first_alt().status = SNAPGPU_NotFound;
        }
        popular_seeds_skipped = saved_pop;
    }
};
