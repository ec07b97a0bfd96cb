// paired.h -- the paired-end path: IntersectingPairedEndAligner::align wrapped by
// ChimericPairedEndAligner::align, one read pair per wavefront.
//
// Restates (file:line in the reference tree):
//   IntersectingPairedEndAligner::align               SNAPLib/IntersectingPairedEndAligner.cpp:169-251
//     alignLandauVishkin  (Phases 1-3)                :254-1435
//     alignHamming        (same walk, gapless scoring) :1441-2487
//     alignAffineGap      (Phase 4)                   :2489-2970, incl. the ALT liftover of :2866-2968 / :2971-3117
//   scoreLocation / scoreLocationWithAffineGap / scoreLocationWithHammingDistance   :3283-3399 / :3119-3280 / :3402-3513
//   HashTableHitSet::{init,recordLookup,getFirstHit,getNextLowerHit,getNextHitLessThanOrEqualTo,
//                    computeBestPossibleScoreForCurrentHit}                          :3516-3810
//   MergeAnchor::checkMerge :3820-3876, ScoreSet :3878-3972 (+ .h:614-700), computeScoreLimit :3975-3988
//   AffineGapVectorized::computeGaplessScore          SNAPLib/AffineGapVectorized.h:139-254
//   ChimericPairedEndAligner::align                   SNAPLib/ChimericPairedEndAligner.cpp:126-448
//
// The walk is adaptive and sequential per pair (what is scored next and with which limit depends on
// every earlier score), so -- as for single-end reads -- one wavefront owns one pair and executes
// this control flow wave-uniformly; the 64 lanes work inside the primitives (hash probes, LV
// diagonals, affine-gap columns, window staging).  Everything here is written against a small
// platform interface PL so the identical control flow can also be compiled for the host by the
// test suite (oracle/paired_host.cpp, which plugs in the CPU restatement's primitives); the product
// instantiation is the device one in snapgpu.hip.
//
// PL provides:
//   static ld(x)/st(x,v)     wave-uniform load / single-lane store of a scalar in LDS or scratch
//   static i32/i64/f64(v)    assert a value is wave-uniform (readfirstlane on the device)
//   lookup(text, out[2])     GenomeIndex::lookupSeed32 of the seed at `text`; false if not a seed
//   window(loc, read_len)    pointer d with d[i] = genome[loc+i] for i in [-128, read_len+128)
//   lv(...), ag(...)         LandauVishkin / AffineGapVectorized primitives
//   substring_ok(loc,len)    Genome::getSubstring(loc,len) != NULL;   is_alt(loc)
//   mapq(pAll,pBest,popular) computeMAPQ;   seed_prob()  pow(1-SNP_PROB, seedLen);  phred/indel/perfect tables
//   align_single(...)        the single-end aligner of the chimeric fallback
//   FAST_HITSET + hs_*(...)  optional lane-parallel versions of the four HashTableHitSet queries (the scalar ones below are
//                            the definition; the device runs one lookup per lane, the hits it is about to need staged in LDS)
//   hs_begin_walk(...)       a set starts its Phase-2 walk (role 0: the read with fewer hits, 1: its mate); no-op on the host
#pragma once
#include <stdint.h>
#include "../../include/snapgpu.h"

#ifdef SNAP_PE_HOST
#define PE_FN inline
#else
#define PE_FN __device__ __forceinline__
#endif

#ifndef G
#define G(T) T                         // (device builds: dev_common.h -- a type known to live in HBM)
#endif

#define PE_MAX_SEEDS       30          // MAX_MAX_SEEDS, IntersectingPairedEndAligner.h:216
#define PE_NOT_YET_SCORED  (-2)        // ScoringMateCandidate::LocationNotYetScored
#define PE_MERGE_DIST      31          // maxMergeDistance, IntersectingPairedEndAligner.cpp:3990
#define PE_MAXK1           (SNAPGPU_MAX_K - 1)
#define PE_MRING           32          // mates the Phase-2 walk can look back at without going to the pool (device: LDS)

struct PEHits {                        // one direction of a seed lookup
    const G(uint32_t) *hits;              // overflow list when n_hits > 1 (descending)
    int64_t  n_hits;
    uint32_t singleton;                // the hit when n_hits == 1
};

struct PEProj {                        // contig table with the ALT-to-primary projections (Genome.h:386-400), in the platform's memory
    const uint64_t *contig_begin, *proj_begin;
    const uint8_t  *proj_rc;
    const uint32_t *cigar_start, *cigar_ops;       // ops: (count << 8) | action
    uint32_t n_contigs;
};

struct PECfg {
    PEProj   proj;
    int32_t  max_k;                    // maxK of the current align() call
    int32_t  extra_depth, max_k_for_indels, max_gap_alt;
    uint32_t use_ag, alt_aware, emit_alt, use_soft_clip, force_spacing;
    int32_t  match_reward, sub_penalty, gap_open, gap_extend, five_bonus, three_bonus;
    int32_t  min_spacing; uint32_t max_spacing;
    uint32_t max_big_hits, num_seeds;
    double   seed_coverage;
    uint32_t min_read_length;
    int32_t  flatten_mapq, min_score_realign, min_score_gap_realign_alt, min_ag_improve;
    uint32_t enable_hamming_base;
    int32_t  seed_len;
    uint32_t pool_size;                // scoringCandidatePoolSize, :141
    uint32_t ag_cand_cap;              // capacity of the Phase-4 candidate buffer (PairedAligner.cpp:571)
    uint32_t max_seeds;                // lookups per hit set the LDS arrays are sized for
    // secondary results (-om / -omax / -mpc; all zero / om = -1 when off)
    int32_t  om;                       // maxEditDistanceForSecondaryResults, -1 = none
    int32_t  mpc;                      // maxSecondaryAlignmentsPerContig, -1 = no limit
    int64_t  omax;                     // maxSecondaryResultsToReturn
    uint32_t sec_cap;                  // capacity of the paired secondary-result list of a wave
};

struct PELookup {                      // HashTableLookup<unsigned>, IntersectingPairedEndAligner.h:247
    const G(uint32_t) *hits;
    int64_t  n_hits;                   // after trimming (:3562)
    int64_t  cur;                      // currentHitForIntersection
    uint32_t seed_offset;
    uint32_t singleton;
    uint32_t is_single;
    uint32_t which_disjoint;
};
struct PEHitSetHdr { int64_t most_recent; uint32_t n_used; int32_t cur_disjoint; };

struct PECand {                        // ScoringCandidate, .h:560-611
    int64_t  loc;                      // readWithFewerHitsGenomeLocation
    double   match_prob;
    int32_t  next;                     // scoreListNext (pool index, -1 = end)
    int32_t  anchor;                   // mergeAnchor (pool index, -1 = none)
    uint32_t mate_index;               // scoringMateCandidateIndex
    uint32_t set_pair;
    uint32_t seed_offset;
    uint32_t best_possible;
    int32_t  big_indel;                // largestBigIndelDetected
    int32_t  clip_before, clip_after, ag_score, lv_indels, ref_span;
    uint32_t used_gapless;
    uint32_t pad;
};
struct PEMate {                        // ScoringMateCandidate, .h:515-558
    int64_t  loc;                      // readWithMoreHitsGenomeLocation
    double   match_prob;
    int64_t  big_indel;
    int32_t  best_possible, score, score_limit;
    uint32_t seed_offset;
    int32_t  genome_offset;
    int32_t  clip_before, clip_after, ag_score, lv_indels, ref_span;
    uint32_t used_gapless;
    uint32_t pad;
};
struct PEAnchor { double match_prob; int64_t loc_more, loc_fewer; int32_t pair_score, pair_ag; };   // MergeAnchor, .h:474-513

struct PESet {                         // ScoreSet, .h:614-700
    int64_t  loc[2], orig[2];
    double   mp[2];
    double   p_best, p_all;
    int32_t  dir[2];
    uint32_t score[2];
    int32_t  used_ag[2], clip_before[2], clip_after[2], ag[2], seed_off[2], lv_indels[2], gapless[2], ref_span[2];
    int32_t  best_pair_score, best_pair_ag;
};

// Phase 4 of a heavy pair, shared with idle wavefronts (paired_dev.h: the help slots).  The two affine-gap problems of a Phase-4 candidate
// depend on the other candidates only through the running score limit, so a candidate can be scored SPECULATIVELY under a predicted limit
// by any wave that holds the pair's reads; PEHelpSpec keeps what scoreLocationWithAffineGap returned for each mate together with the limit
// argument the call was made with.  The owner then walks the candidates in the reference's order exactly as before and, wherever the limit it
// arrives with equals the one a speculative call used, takes the stored answer instead of computing it again -- same inputs, same answer;
// anything else it computes itself.  FP64 sums and score-set updates happen only in that ordered walk.
struct PEHelpSpec {
    int32_t  lim[2];                   // limit argument of the call for mate r; PE_SPEC_NONE: no call was made
    int32_t  score[2], g_off[2], cb[2], ca[2], ag[2], span[2];
    uint32_t stale_fw[2], stale_bw[2];  // out-of-band traceback steps of the call's affineGap / reverseAffineGap object
    uint32_t calls[2];                 // bit 0: affineGap was called, bit 1: reverseAffineGap was called
    uint32_t n_ag[2];
    double   mp[2];
};
#define PE_SPEC_NONE (-0x7fffffff)

struct PECounters { uint64_t lv, ag, lookups, hits, overflow_lists, lv_ref_bytes; uint64_t cyc_lookup, cyc_intersect, cyc_lv, cyc_ag, cyc_single, cyc_total; };   // cycles: PL::clock()

struct PEShared {                      // cold wave-uniform state (LDS on the device)
    PESet all, non_alt;
    snapgpu_paired_result res, alt, saved;
    snapgpu_single_result single[2], single_alt[2];
    PECounters cnt;
};

struct LVOut { int score; double mp; int net_indel, total_indels, text_span; };
struct AGOut { int ag_score, text_offset, pattern_offset, n_edits; double mp; int stale; };

template <class PL>
struct PairedCore {
    typename PL::SelfPtr pl;           // the platform object (device: an LDS address, stored as such -- dev_common.h: LP)
    const PECfg cfg;
    // hot per-pair arrays (LDS on the device)
    PELookup    *lk;                   // [4][cfg.max_seeds]   set = 2*whichRead + dir
    uint32_t    *exhausted;            // [4][cfg.max_seeds]   DisjointHitSet::countOfExhaustedHits
    uint32_t    *miss;                 // [cfg.max_seeds]      DisjointHitSet::missCount (scratch)
    PEHitSetHdr *hs;                   // [4]
    int32_t     *list_head;            // [SNAPGPU_MAX_K + 1]   scoringCandidates[]
    uint32_t    *seed_used;            // bitmap over seed start offsets
    uint32_t    *mring = nullptr;      // [2 * PE_MRING] (location, best possible score) of the last PE_MRING mates of the set pair being walked; NULL: none
    PEShared    *sh;
    // the two reads, both orientations (LDS on the device).  reversedRead[][] of the reference is rd walked with stride -1.
    const uint8_t *rd[2][2], *ql[2][2];
    int read_len[2];
    // per-wave pools in HBM scratch
    // (G(T): HBM -- the device build's accesses through these are global loads / stores, not flat ones: dev_common.h)
    G(PECand)   *cand;                 // [cfg.pool_size]
    G(PEMate)   *mate[2];              // [cfg.pool_size / 2] each
    G(PEAnchor) *anchor;               // [cfg.pool_size]
    G(snapgpu_paired_result) *agc;     // [cfg.ag_cand_cap]   lvCandidatesForAffineGap
    G(uint32_t) *agc_order;            // [cfg.ag_cand_cap]   the order Phase 4 visits them in
    G(snapgpu_paired_result) *sec;     // [cfg.sec_cap]       secondaryResults of IntersectingPairedEndAligner::align
    G(uint32_t) *sec_ord, *sec_key;    // [cfg.sec_cap], [2 * cfg.sec_cap]   index list / sort keys of the final filtering
    uint32_t n_sec;
    // single-end secondary results of the chimeric fallback go straight to the caller's buffer (read 0's, then read 1's)
    G(snapgpu_single_result) *ssec_out;   // [ssec_stride] for this pair, or NULL
    uint32_t ssec_stride;
    uint32_t n_ssec[2];
    uint32_t ref_dep;                  // SNAPGPU_PAIR_REF_BUFFER_DEPENDENT
    // per-pair scalars
    uint32_t n_cand, n_mate[2], n_anchor;
    uint32_t n_agc;
    uint32_t popular[2];
    int more, fewer;                   // readWithMoreHits / readWithFewerHits
    uint32_t stale, overflow;
    uint32_t stale_later = 0;          // steps outside the band in calls that were not their object's first (or speculative): the pair needs the exact pass
    uint32_t ag_obj_used0 = 0, ag_obj_used1 = 0;  // has affineGap / reverseAffineGap of the intersecting aligner scored anything yet for this pair?
    bool spec_mode = false;            // speculative scoring of Phase-4 candidates: work counters are left to the ordered walk
    uint32_t spec_n_ag = 0, spec_calls = 0, spec_stale_fw = 0, spec_stale_bw = 0;
    uint32_t spec_used = 0;            // answers the ordered walk took from speculative scoring (this pair)
    uint32_t help_min = 0xffffffffu;   // Phase-4 lists at least this long are offered to idle waves (PL::HELP only)

    PE_FN PairedCore(PL &pl_, const PECfg &c) : pl(&pl_), cfg(c) {}

    template <class T> static PE_FN auto ld(const T &x) -> decltype(PL::ld(x)) { return PL::ld(x); }
    template <class T, class V> static PE_FN void st(T &x, V v) { PL::st(x, (decltype(PL::ld(x)))v); }

    // secondary results wanted?  Compile-time off in the default kernel (PL::SECONDARY), so that it carries none of that code.
    PE_FN bool want_sec() const { return PL::SECONDARY && cfg.om != -1; }

    static PE_FN int64_t dist(int64_t a, int64_t b) { return a > b ? a - b : b - a; }                 // DistanceBetweenGenomeLocations
    static PE_FN bool within(int64_t a, int64_t b, int64_t d) { return dist(a, b) <= d; }             // genomeLocationIsWithin
    static PE_FN int set_dir(int set_pair, int which_read) { return (set_pair == 0) ? which_read : 1 - which_read; }   // setPairDirection

    // ------------------------------------------------------------------ HashTableHitSet
    // The hot per-pair arrays are LDS on the device, and every access to them must be an LDS instruction: the pointers live in this object
    // (which sits in scratch memory: its address escapes), and a pointer loaded back from memory is a GENERIC one -- a flat_load / flat_store
    // that waits for vmcnt(0) AND lgkmcnt(0), i.e. for every store to the candidate pools in HBM that is still on its way (profiles/r05b: the
    // set intersection at 40 % of the kernel's wave cycles with windows or without).  PL::lds() tells the compiler what the pointer is
    // (the device: llvm.assume(llvm.amdgcn.is.shared(p)), which address-space inference follows; the host: nothing).
    template <class T> static PE_FN T *L(T *p) { PL::lds((const void *)p); return p; }
    PE_FN PELookup *lks(int s) const { return L(lk) + (size_t)s * cfg.max_seeds; }
    PE_FN uint32_t *exh(int s) const { return L(exhausted) + (size_t)s * cfg.max_seeds; }
    PE_FN PEHitSetHdr *HS() const { return L(hs); }
    PE_FN int32_t *LH() const { return L(list_head); }
    PE_FN uint32_t *MISS() const { return L(miss); }
    PE_FN uint32_t *SU() const { return L(seed_used); }
    PE_FN uint32_t *MR() const { return L(mring); }
    PE_FN PEShared *S() const { return L(sh); }
    PE_FN uint32_t hit(const PELookup *l, int64_t i) const {
        return ld(l->is_single) ? ld(l->singleton) : ld(l->hits[i]);
    }

    PE_FN void hs_init(int s) { st(HS()[s].n_used, 0); st(HS()[s].cur_disjoint, -1); }                    // :3516-3527

    PE_FN void hs_record(int s, uint32_t seed_offset, const PEHits &h, bool begins) {                  // recordLookup, :3536-3579
        int cd = ld(HS()[s].cur_disjoint);
        if (begins) { cd++; st(HS()[s].cur_disjoint, cd); st(exh(s)[cd], 0); }
        if (h.n_hits == 0) {
            st(exh(s)[cd], ld(exh(s)[cd]) + 1);
            return;
        }
        uint32_t n = ld(HS()[s].n_used);
        PELookup *l = &lks(s)[n];
        st(l->hits, h.hits); st(l->singleton, h.singleton); st(l->is_single, h.n_hits == 1 ? 1u : 0u);
        st(l->seed_offset, seed_offset); st(l->cur, 0); st(l->which_disjoint, (uint32_t)cd);
        int64_t nh = h.n_hits;
        while (nh > 0 && hit(l, nh - 1) < seed_offset) nh--;      // hits before the start of the genome are meaningless (:3562)
        st(l->n_hits, nh);
        st(HS()[s].n_used, n + 1);
    }

    // returns true when the set is EMPTY (the reference returns !anyFound), :3720-3746
#if defined(SNAPGPU_PT_PHASE2)
    // diagnostic build (with -DSNAPGPU_PHASE_TIMERS): where Phase 2's cycles go -- cycles_lookup = the hit-set queries, cycles_lv = the
    // best-possible-score computations, cycles_ag = mate / candidate records, cycles_single_fallback = Phase 2a; the counters' usual
    // meanings are switched off (PT2_OFF)
#define PT2_T0() const uint64_t pt2_t0 = PL::clock()
#define PT2_ADD(f) S()->cnt.f += PL::clock() - pt2_t0
#define PT2_OFF(x) ((void)0)
#else
#define PT2_T0() ((void)0)
#define PT2_ADD(f) ((void)0)
#define PT2_OFF(x) x
#endif
    // (FAST_HITSET: the walk's cursors live in registers for the length of the walk -- PL::HSCursor, filled by hs_begin_walk; the lookups'
    //  records in LDS are not written back: nothing reads a hit set after its walk)
    PE_FN bool hs_first(int s, int64_t *loc, uint32_t *seed_offset, typename PL::HSCursor &c) {
        if constexpr (PL::FAST_HITSET) { PT2_T0(); const bool r = pl->hs_first(c, loc, seed_offset); PT2_ADD(cyc_lookup); return r; }
        bool any = false;
        *loc = 0;
        const uint32_t n = ld(HS()[s].n_used);
        for (uint32_t i = 0; i < n; i++) {
            const PELookup *l = &lks(s)[i];
            if (ld(l->n_hits) > 0) {
                uint32_t so = ld(l->seed_offset);
                int64_t v = (int64_t)(uint32_t)(hit(l, 0) - so);
                if (v > *loc) { *loc = v; *seed_offset = so; any = true; }
            }
        }
        if (any) st(HS()[s].most_recent, *loc);
        return !any;
    }

    PE_FN bool hs_next_lower(int s, int64_t *loc, uint32_t *seed_offset, typename PL::HSCursor &c) {     // getNextLowerHit, :3750-3816
        if constexpr (PL::FAST_HITSET) { PT2_T0(); const bool r = pl->hs_next_lower(c, loc, seed_offset); PT2_ADD(cyc_lookup); return r; }
        int64_t found = 0;
        bool any = false;
        const uint32_t n = ld(HS()[s].n_used);
        const int64_t recent = ld(HS()[s].most_recent);
        for (uint32_t i = 0; i < n; i++) {
            PELookup *l = &lks(s)[i];
            int64_t cur = ld(l->cur);
            const int64_t nh = ld(l->n_hits);
            const uint32_t so = ld(l->seed_offset);
            if (cur == nh) continue;
            int64_t h = (int64_t)hit(l, cur);
            if (h - so == recent) {
                cur++;
                st(l->cur, cur);
                if (cur == nh) continue;
                h = (int64_t)hit(l, cur);
            }
            if (found < h - so && h >= (int64_t)so) {
                *loc = found = h - so;
                *seed_offset = so;
                any = true;
            }
        }
        if (any) st(HS()[s].most_recent, found);
        return any;
    }

    PE_FN bool hs_next_le(int s, int64_t max_loc, int64_t *loc, uint32_t *seed_offset, typename PL::HSCursor &c) {   // getNextHitLessThanOrEqualTo, :3628-3717
        if constexpr (PL::FAST_HITSET) { PT2_T0(); const bool r = pl->hs_next_le(c, max_loc, loc, seed_offset); PT2_ADD(cyc_lookup); return r; }
        bool any = false;
        int64_t best = 0;
        const uint32_t n = ld(HS()[s].n_used);
        for (uint32_t i = 0; i < n; i++) {
            PELookup *l = &lks(s)[i];
            int64_t lo = ld(l->cur), hi = ld(l->n_hits) - 1;
            const uint32_t so = ld(l->seed_offset);
            const int64_t max_this = max_loc + so;
            while (lo <= hi) {
                int64_t probe = (lo + hi) / 2;
                int64_t ph = (int64_t)hit(l, probe);
                bool c1 = ph <= max_this;
                if (c1 && (probe == 0 || (int64_t)hit(l, probe - 1) > max_this)) {
                    if (ph - so > best) {
                        any = true;
                        best = ph - so;
                        *loc = best;
                        *seed_offset = so;
                    }
                    st(l->cur, probe);
                    break;
                }
                if (ph > max_this) lo = probe + 1; else hi = probe - 1;
            }
            if (lo > hi) st(l->cur, ld(l->n_hits));
        }
        if (any) st(HS()[s].most_recent, best);
        return any;
    }

    PE_FN uint32_t hs_best_possible(int s, typename PL::HSCursor &c) {                                   // computeBestPossibleScoreForCurrentHit, :3585-3625
        if constexpr (PL::FAST_HITSET) { PT2_T0(); const uint32_t r = pl->hs_best_possible(c, exh(s)); PT2_ADD(cyc_lv); return r; }
        const int cd = ld(HS()[s].cur_disjoint);
        for (int i = 0; i <= cd; i++) st(MISS()[i], ld(exh(s)[i]));
        const uint32_t n = ld(HS()[s].n_used);
        const int64_t recent = ld(HS()[s].most_recent);
        for (uint32_t i = 0; i < n; i++) {
            const PELookup *l = &lks(s)[i];
            const int64_t cur = ld(l->cur), nh = ld(l->n_hits);
            const int64_t target = recent + ld(l->seed_offset);
            bool close = (cur != nh && within((int64_t)hit(l, cur), target, PE_MERGE_DIST)) ||
                         (cur != 0 && within((int64_t)hit(l, cur - 1), target, PE_MERGE_DIST));
            if (!close) { uint32_t w = ld(l->which_disjoint); st(MISS()[w], ld(MISS()[w]) + 1); }
        }
        uint32_t best = 0;
        for (int i = 0; i <= cd; i++) { uint32_t m = ld(MISS()[i]); if (m > best) best = m; }
        return best;
    }

    // ------------------------------------------------------------------ ScoreSet
    PE_FN void set_init(PESet &s) {                                                                       // .h:619-639
        for (int i = 0; i < 2; i++) {
            s.loc[i] = SNAPGPU_InvalidGenomeLocation32; s.orig[i] = SNAPGPU_InvalidGenomeLocation32;
            s.score[i] = (uint32_t)SNAPGPU_ScoreAboveLimit; s.dir[i] = 0; s.used_ag[i] = 0; s.clip_before[i] = 0; s.clip_after[i] = 0;
            s.ag[i] = 0; s.seed_off[i] = 0; s.lv_indels[i] = 0; s.mp[i] = 0.0; s.gapless[i] = 0; s.ref_span[i] = 0;
        }
        s.p_best = 0; s.p_all = 0; s.best_pair_score = SNAPGPU_TooBigScoreValue; s.best_pair_ag = 0;
    }
    PE_FN void set_init_from(PESet &s, const snapgpu_paired_result &r) {                                 // .h:641-662
        for (int i = 0; i < 2; i++) {
            s.loc[i] = r.location[i]; s.orig[i] = r.orig_location[i]; s.score[i] = (uint32_t)r.score[i]; s.dir[i] = r.direction[i];
            s.used_ag[i] = r.used_affine_gap_scoring[i]; s.clip_before[i] = r.bases_clipped_before[i]; s.clip_after[i] = r.bases_clipped_after[i];
            s.ag[i] = r.ag_score[i]; s.seed_off[i] = r.seed_offset[i]; s.lv_indels[i] = r.lv_indels[i]; s.mp[i] = r.match_probability[i];
            s.gapless[i] = r.used_gapless_clipping[i]; s.ref_span[i] = r.ref_span[i];
        }
        s.p_best = r.match_probability[0] * r.match_probability[1];
        s.p_all = r.probability_all_pairs;
        s.best_pair_score = r.score[0] + r.score[1];
        s.best_pair_ag = r.ag_score[0] + r.ag_score[1];
    }
    static PE_FN void set_sub_all(PESet &s, double old) { double v = s.p_all - old; s.p_all = v > 0 ? v : 0; }   // updateProbabilityOfAllPairs

    // updateBestHitIfNeeded(candidate, mate), :3883-3926
    PE_FN bool set_update(PESet &s, int pair_score, int pair_ag, double pair_p, int fewer_score, int fewer_off, int ci, int mi, int set_pair) {
        s.p_all += pair_p;
        if (pair_ag > s.best_pair_ag || (pair_ag == s.best_pair_ag && pair_p > s.p_best)) {
            const G(PECand) *c = &cand[ci];
            const G(PEMate) *m = &mate[set_pair][mi];
            s.best_pair_score = pair_score; s.best_pair_ag = pair_ag; s.p_best = pair_p;
            const int f = fewer, mo = more;
            s.loc[f] = ld(c->loc) + fewer_off;            s.loc[mo] = ld(m->loc) + ld(m->genome_offset);
            s.orig[f] = ld(c->loc);                       s.orig[mo] = ld(m->loc);
            s.score[f] = (uint32_t)fewer_score;           s.score[mo] = (uint32_t)ld(m->score);
            s.dir[f] = set_dir(set_pair, f);              s.dir[mo] = set_dir(set_pair, mo);
            s.used_ag[f] = 0;                             s.used_ag[mo] = 0;       // scoreLocation never sets usedAffineGapScoring
            s.gapless[f] = (int)ld(c->used_gapless);      s.gapless[mo] = (int)ld(m->used_gapless);
            s.clip_before[f] = ld(c->clip_before);        s.clip_before[mo] = ld(m->clip_before);
            s.clip_after[f] = ld(c->clip_after);          s.clip_after[mo] = ld(m->clip_after);
            s.ag[f] = ld(c->ag_score);                    s.ag[mo] = ld(m->ag_score);
            s.seed_off[f] = (int)ld(c->seed_offset);      s.seed_off[mo] = (int)ld(m->seed_offset);
            s.mp[f] = ld(c->match_prob);                  s.mp[mo] = ld(m->match_prob);
            s.lv_indels[f] = ld(c->lv_indels);            s.lv_indels[mo] = ld(m->lv_indels);
            s.ref_span[f] = ld(c->ref_span);              s.ref_span[mo] = ld(m->ref_span);
            return true;
        }
        return false;
    }
    // updateBestHitIfNeeded(PairedAlignmentResult*), :3928-3954.  `r` lives in the Phase-4 candidate buffer.
    PE_FN bool set_update_from(PESet &s, int pair_score, int pair_ag, double pair_p, const G(snapgpu_paired_result) *r) {
        s.p_all += pair_p;
        if (pair_ag > s.best_pair_ag || (pair_ag == s.best_pair_ag && pair_p > s.p_best)) {
            s.best_pair_score = pair_score; s.best_pair_ag = pair_ag; s.p_best = pair_p;
            for (int i = 0; i < 2; i++) {
                s.loc[i] = ld(r->location[i]); s.orig[i] = ld(r->orig_location[i]); s.score[i] = (uint32_t)ld(r->score[i]);
                s.dir[i] = ld(r->direction[i]); s.used_ag[i] = ld(r->used_affine_gap_scoring[i]); s.gapless[i] = ld(r->used_gapless_clipping[i]);
                s.clip_before[i] = ld(r->bases_clipped_before[i]); s.clip_after[i] = ld(r->bases_clipped_after[i]); s.ag[i] = ld(r->ag_score[i]);
                s.seed_off[i] = ld(r->seed_offset[i]); s.mp[i] = ld(r->match_probability[i]); s.lv_indels[i] = ld(r->lv_indels[i]);
                s.ref_span[i] = ld(r->ref_span[i]);
            }
            return true;
        }
        return false;
    }
    PE_FN void set_fill(const PESet &s, snapgpu_paired_result &r, const uint32_t pop[2]) {             // fillInResult, :3957-3972
        for (int i = 0; i < 2; i++) {
            r.location[i] = s.loc[i]; r.orig_location[i] = s.orig[i]; r.direction[i] = s.dir[i];
            r.mapq[i] = pl->mapq(s.p_all, s.p_best, (int)(pop[0] + pop[1]));
            r.status[i] = r.mapq[i] > 10 ? SNAPGPU_SingleHit : SNAPGPU_MultipleHits;                  // MAPQ_LIMIT_FOR_SINGLE_HIT
            r.score[i] = (int32_t)s.score[i]; r.clipping_for_read_adjustment[i] = 0; r.used_affine_gap_scoring[i] = s.used_ag[i];
            r.used_gapless_clipping[i] = s.gapless[i]; r.bases_clipped_before[i] = s.clip_before[i]; r.bases_clipped_after[i] = s.clip_after[i];
            r.ag_score[i] = s.ag[i]; r.seed_offset[i] = s.seed_off[i]; r.lv_indels[i] = s.lv_indels[i]; r.match_probability[i] = s.mp[i];
            r.popular_seeds_skipped[i] = pop[i]; r.ref_span[i] = s.ref_span[i];
        }
        r.probability_all_pairs = s.p_all;
    }

    PE_FN int score_limit(bool non_alt, int64_t big_indel) const {                                       // computeScoreLimit, :3975-3988
        const PESet &a = S()->all, &n = S()->non_alt;
        int64_t inner;
        if (non_alt) {
            int64_t x = (int64_t)a.best_pair_score + cfg.max_gap_alt;
            inner = x < n.best_pair_score ? x : n.best_pair_score;
        } else {
            int64_t x = (int64_t)n.best_pair_score - cfg.max_gap_alt;
            inner = a.best_pair_score < x ? a.best_pair_score : x;
        }
        int64_t m = (int64_t)cfg.max_k + big_indel;
        if (inner < m) m = inner;
        int64_t v = (int64_t)cfg.extra_depth + m;
        return PL::i32((int)(v < PE_MAXK1 ? v : PE_MAXK1));
    }

    // ------------------------------------------------------------------ location scoring
    // computeGaplessScore (AffineGapVectorized.h:139-254): Hamming walk away from the seed, best prefix kept, rest clipped.
    // st = +1 forward / -1 backward.  Returns the affine score or -1; *n_edits / *n_gapless / *pattern_offset / *mp as in the reference.
    PE_FN int gapless(int st, const uint8_t *T, const uint8_t *P, const uint8_t *Q, int plen, int score_init, int limit,
                      int *n_edits, int *pattern_offset, double *mp, int *n_gapless) {
        *mp = 1.0;                                       // callers pre-set 1.0; the early return below leaves it alone in the reference too
        if (limit < 0) { *n_edits = -1; *n_gapless = -1; return -1; }
        int sc = score_init, best = score_init, best_i = 0;
        for (int i = 0; i < plen; i++) {
            sc += (P[i * st] == T[i * st]) ? cfg.match_reward : -cfg.sub_penalty;
            if (sc > best) { best = sc; best_i = i; }
        }
        if (best > score_init) {
            int ne = 0, nm = 0;
            double p = 1.0;
            for (int i = 0; i <= best_i; i++) {
                if (P[i * st] != T[i * st]) { ne++; p *= pl->phred(Q[i * st]); } else nm++;
            }
            p *= pl->perfect(nm);
            int clipped = plen - (best_i + 1);
            *pattern_offset = clipped;
            *n_gapless = ne <= limit ? ne : -1;
            *n_edits = ne + clipped;
            p *= pl->indel(clipped);
            *mp = PL::f64(p);
            return PL::i32(best);
        }
        *n_edits = -1; *n_gapless = -1;
        return -1;
    }

    struct LocScore {                 // outputs of the scoreLocation* family
        int score; double mp; int offset; int clip_before, clip_after, ag_score, lv_indels, ref_span; bool gapless; int score_gapless;
        PE_FN void reset(int cb, int ca, int ag, int ind) {       // the fields a failed call leaves as the caller had them
            score = -1; mp = 0.0; offset = 0; clip_before = cb; clip_after = ca; ag_score = ag; lv_indels = ind; ref_span = 0; gapless = false; score_gapless = -1;
        }
    };

    // scoreLocation (:3283-3399): Landau-Vishkin both ways from the seed.
    PE_FN void score_lv(int which, int dir, int64_t loc, int seed_offset, int limit, LocScore &o) {
        const int rl = read_len[which], sl = cfg.seed_len;
        const int64_t glen = (int64_t)rl + SNAPGPU_MAX_K;
        o.offset = 0; o.ref_span = 0; o.gapless = false;
        if (!pl->substring_ok(loc, glen)) { o.score = -1; o.mp = 0; o.ag_score = -1; return; }
        o.clip_before = 0; o.clip_after = 0;
        const uint8_t *data = pl->window(loc, rl);
        const uint8_t *R = L(rd[which][dir]), *Qd = L(ql[which][dir]);
        const int tail = seed_offset + sl;
        S()->cnt.lv++;
        const uint64_t t_lv = PL::clock();
        LVOut a = pl->lv(+1, R + tail, Qd + tail, rl - tail, data + tail, (int)(glen - tail), limit);
        int score1 = a.score, score2 = 0, off = 0, ind2 = 0, span2 = 0;
        double mp1 = a.mp, mp2 = 1.0;
        int ag1 = (sl + rl - tail - score1) * cfg.match_reward - score1 * cfg.sub_penalty, ag2 = 0;
        if (score1 != -1) {
            LVOut b = pl->lv(-1, R + seed_offset - 1, Qd + seed_offset - 1, seed_offset, data + seed_offset - 1, seed_offset + SNAPGPU_MAX_K, limit - score1);
            score2 = b.score; mp2 = b.mp; off = b.net_indel; ind2 = b.total_indels; span2 = b.text_span;
            ag2 = (seed_offset - score2) * cfg.match_reward - score2 * cfg.sub_penalty;
        } else {
            mp1 = 1.0;                               // (unused)
        }
        PT2_OFF(S()->cnt.cyc_lv += PL::clock() - t_lv);
        S()->cnt.lv_ref_bytes += (uint64_t)(rl - tail) + (uint64_t)(2 * (limit < 0 ? 0 : limit)) + (uint64_t)seed_offset;
        o.offset = off;
        if (off != 0 && !pl->substring_ok(loc + off, glen)) score2 = -1;                               // :3364-3375
        if (score1 != -1 && score2 != -1) {
            o.score = score1 + score2;
            o.mp = mp1 * mp2 * pl->seed_prob();
            o.ag_score = ag1 + ag2;
            o.ref_span = a.text_span + sl + span2;
            o.lv_indels = a.total_indels + ind2;
        } else {
            o.score = -1; o.ag_score = -1; o.mp = 0.0;
        }
    }

    // scoreLocationWithHammingDistance (:3402-3513)
    PE_FN void score_hamming(int which, int dir, int64_t loc, int seed_offset, int limit, LocScore &o) {
        const int rl = read_len[which], sl = cfg.seed_len;
        const int64_t glen = (int64_t)rl + SNAPGPU_MAX_K;
        o.offset = 0; o.gapless = false;
        if (!pl->substring_ok(loc, glen)) { o.score = -1; o.mp = 0; o.ag_score = -1; return; }
        o.clip_before = 0; o.clip_after = 0;
        const uint8_t *data = pl->window(loc, rl);
        const uint8_t *R = L(rd[which][dir]), *Qd = L(ql[which][dir]);
        const int tail = seed_offset + sl;
        int score1 = 0, score2 = 0, g1 = 0, g2 = 0, ag1 = sl, ag2 = 0, po = 0;
        double mp1 = 1.0, mp2 = 1.0;
        if (tail != rl) {
            ag1 = gapless(+1, data + tail, R + tail, Qd + tail, rl - tail, rl, limit, &score1, &po, &mp1, &g1);
            ag1 += sl - rl;
        }
        if (g1 != -1 && seed_offset != 0) {
            int off = 0;
            ag2 = gapless(-1, data + seed_offset - 1, R + seed_offset - 1, Qd + seed_offset - 1, seed_offset, rl, limit - g1, &score2, &off, &mp2, &g2);
            ag2 -= rl;
            o.offset = g2 != -1 ? off : 0;               // o_textOffset = bases clipped at the read's start (:3480, AffineGapVectorized.h:243)
        }
        if (g1 != -1 && g2 != -1) {
            o.score = score1 + score2;
            o.mp = mp1 * mp2 * pl->seed_prob();
            o.ag_score = ag1 + ag2;
            o.score_gapless = g1 + g2;
            o.gapless = true;
        } else {
            o.score = -1; o.ag_score = -1; o.mp = 0.0; o.score_gapless = -1;
        }
    }

    // (see Aligner::note_ag_call: the first call an affine-gap object serves for a pair reads zeros outside its band, as the kernels do;
    //  speculative calls have no place in the order of calls, so their steps always count)
    PE_FN void note_ag_call(int obj, uint32_t stale_steps) {
        stale += stale_steps;
        const uint32_t used = obj == 0 ? ag_obj_used0 : ag_obj_used1;
        if (spec_mode) {                // no place in the order of calls yet: the owner books the call when it takes the answer (take_spec_calls)
            spec_calls |= 1u << obj;
            if (obj == 0) spec_stale_fw = stale_steps; else spec_stale_bw = stale_steps;
            return;
        }
        if (PL::ALWAYS_COUNT_STALE || used) stale_later += stale_steps;
        if (obj == 0) ag_obj_used0 = 1; else ag_obj_used1 = 1;
    }
    // An answer scored ahead of time is taken: its calls enter the order of the pair's calls here, exactly as if this wave had made them
    // (which object had scored something before, which steps count) -- the flags of a pair do not depend on who scored its candidates.
    PE_FN void take_spec_calls(const G(PEHelpSpec) *sp, int r) {
        const uint32_t calls = PL::spec_ld(sp->calls[r]);
        if (calls & 1u) note_ag_call(0, PL::spec_ld(sp->stale_fw[r]));
        if (calls & 2u) note_ag_call(1, PL::spec_ld(sp->stale_bw[r]));
    }

    // scoreLocationWithAffineGap (:3119-3280).  clip_before/clip_after/ag/ref_span are in/out like the reference's pointers.
    PE_FN void score_ag(int which, int dir, int64_t loc, int seed_offset, int limit, int *score, double *mp, int *offset,
                        int *clip_before, int *clip_after, int *ag_score, int *ref_span) {
        const int rl = read_len[which], sl = cfg.seed_len;
        const int64_t glen = (int64_t)rl + SNAPGPU_MAX_K;
        *offset = 0; *ref_span = 0;
        if (!pl->substring_ok(loc, glen)) { *score = -1; *mp = 0; *ag_score = -1; return; }
        *clip_before = 0; *clip_after = 0;
        const uint8_t *data = pl->window(loc, rl);
        const uint8_t *R = L(rd[which][dir]), *Qd = L(ql[which][dir]);
        const int tail = seed_offset + sl;
        const int clip = cfg.use_soft_clip ? 1 : 0;
        int score1 = 0, score2 = 0, ag1 = sl, ag2 = 0;
        double mp1 = 1.0, mp2 = 1.0;
        int text_rem = rl - tail;
        const uint64_t t_ag = PL::clock();
#ifdef PE_AG_STATS
        pl->note_ag(which, dir, loc, seed_offset, limit);
#endif
        if (tail != rl) {
            const int plen = rl - tail;
            const bool banded = plen >= 3 * (2 * limit + 1);
            AGOut a = pl->ag(banded, +1, R + tail, Qd + tail, plen, data + tail, (int)(glen - tail), limit, rl, dir != 0, clip);
            note_ag_call(0, (uint32_t)a.stale);
            ag1 = a.ag_score + (sl - rl); text_rem = a.text_offset; *clip_after = a.pattern_offset; score1 = a.n_edits; mp1 = a.mp;
            if (spec_mode) spec_n_ag++; else S()->cnt.ag++;
        }
        if (score1 != -1) {
            if (seed_offset != 0) {
                const int left = limit - score1;
                const bool banded = seed_offset >= 3 * (2 * left + 1);
                AGOut b = pl->ag(banded, -1, R + seed_offset - 1, Qd + seed_offset - 1, seed_offset, data + seed_offset - 1, seed_offset + left, left,
                                rl, dir != 0, clip);
                note_ag_call(1, (uint32_t)b.stale);
                ag2 = b.ag_score - rl; *offset = b.text_offset; *clip_before = b.pattern_offset; score2 = b.n_edits; mp2 = b.mp;
                if (score2 == -1) *offset = 0;
            }
        }
        PT2_OFF(S()->cnt.cyc_ag += PL::clock() - t_ag);
        if (score1 != -1 && score2 != -1) {
            *score = score1 + score2;
            *mp = mp1 * mp2 * pl->seed_prob();
            *ref_span = (seed_offset - *offset) + sl + (rl - tail - text_rem);
            *ag_score = ag1 + ag2;
        } else {
            *score = -1; *ag_score = -1; *mp = 0.0;
        }
    }

    // ------------------------------------------------------------------ results
    PE_FN void res_not_found(snapgpu_paired_result &r, snapgpu_paired_result &alt, const uint32_t *pop) {   // :1206-1236 / :2655-2683
        for (int w = 0; w < 2; w++) {
            r.location[w] = SNAPGPU_InvalidGenomeLocation32; r.orig_location[w] = SNAPGPU_InvalidGenomeLocation32;
            r.mapq[w] = 0; r.score[w] = -1; r.status[w] = SNAPGPU_NotFound; r.clipping_for_read_adjustment[w] = 0;
            r.used_affine_gap_scoring[w] = 0; r.used_gapless_clipping[w] = 0; r.bases_clipped_before[w] = 0; r.bases_clipped_after[w] = 0;
            r.ag_score[w] = -1; r.seed_offset[w] = 0; r.lv_indels[w] = 0;
            if (pop) r.popular_seeds_skipped[w] = pop[w];
            r.match_probability[w] = 0.0;
            alt.status[w] = SNAPGPU_NotFound;
        }
        r.probability_all_pairs = 0.0;
    }

    // copy of the current best of `s` into a Phase-4 candidate slot (:1032-1058)
    PE_FN void agc_from_set(G(snapgpu_paired_result) *e, const PESet &s) {
        if (PL::lane0()) {
            e->aligned_as_pair = 1;
            for (int r = 0; r < 2; r++) {
                e->direction[r] = s.dir[r]; e->location[r] = s.loc[r]; e->orig_location[r] = s.orig[r]; e->mapq[r] = 0;
                e->score[r] = (int32_t)s.score[r]; e->status[r] = SNAPGPU_MultipleHits; e->used_affine_gap_scoring[r] = s.used_ag[r];
                e->bases_clipped_before[r] = s.clip_before[r]; e->bases_clipped_after[r] = s.clip_after[r]; e->ag_score[r] = s.ag[r];
                e->seed_offset[r] = s.seed_off[r]; e->popular_seeds_skipped[r] = popular[r]; e->lv_indels[r] = s.lv_indels[r];
                e->match_probability[r] = s.mp[r]; e->used_gapless_clipping[r] = s.gapless[r]; e->ref_span[r] = s.ref_span[r];
            }
            e->reserved = s.score[0] + s.score[1];         // sort key of Phase 4: the pair score at the time of the sort
        }
        PL::sync();
    }
    // the pair just scored into a Phase-4 candidate slot (:1134-1164)
    PE_FN void agc_from_pair(G(snapgpu_paired_result) *e, int ci, int mi, int set_pair, int fewer_score, int fewer_off) {
        const G(PECand) *c = &cand[ci];
        const G(PEMate) *m = &mate[set_pair][mi];
        const int f = fewer, mo = more;
        int64_t c_loc = ld(c->loc), m_loc = ld(m->loc);
        int m_off = ld(m->genome_offset), m_score = ld(m->score);
        uint32_t cg = ld(c->used_gapless), mg = ld(m->used_gapless);
        int ccb = ld(c->clip_before), cca = ld(c->clip_after), mcb = ld(m->clip_before), mca = ld(m->clip_after);
        int cag = ld(c->ag_score), mag = ld(m->ag_score), cli = ld(c->lv_indels), mli = ld(m->lv_indels);
        uint32_t cso = ld(c->seed_offset), mso = ld(m->seed_offset);
        double cmp = ld(c->match_prob), mmp = ld(m->match_prob);
        if (PL::lane0()) {
            e->aligned_as_pair = 1;
            e->direction[mo] = set_dir(set_pair, mo); e->direction[f] = set_dir(set_pair, f);
            e->location[mo] = m_loc + m_off;          e->location[f] = c_loc + fewer_off;
            e->orig_location[mo] = m_loc;             e->orig_location[f] = c_loc;
            e->mapq[0] = e->mapq[1] = 0;
            e->score[mo] = m_score;                   e->score[f] = fewer_score;
            e->status[0] = e->status[1] = SNAPGPU_MultipleHits;
            e->used_affine_gap_scoring[mo] = 0;       e->used_affine_gap_scoring[f] = 0;
            e->used_gapless_clipping[mo] = (int)mg;   e->used_gapless_clipping[f] = (int)cg;
            e->bases_clipped_before[f] = ccb;         e->bases_clipped_after[f] = cca;
            e->bases_clipped_before[mo] = mcb;        e->bases_clipped_after[mo] = mca;
            e->ag_score[mo] = mag;                    e->ag_score[f] = cag;
            e->seed_offset[mo] = (int)mso;            e->seed_offset[f] = (int)cso;
            e->lv_indels[mo] = mli;                   e->lv_indels[f] = cli;
            e->match_probability[mo] = mmp;           e->match_probability[f] = cmp;
            e->popular_seeds_skipped[mo] = popular[mo]; e->popular_seeds_skipped[f] = popular[f];
            e->ref_span[0] = e->ref_span[1] = 0;      // (left unset by the reference here; only used by ALT liftover)
            e->reserved = (uint32_t)(m_score + fewer_score);
        }
        PL::sync();
    }

    PE_FN void mring_put(uint32_t i, int64_t loc, uint32_t bp) {
        if (mring == nullptr) return;
        st(MR()[2 * (i % PE_MRING)], (uint32_t)loc); st(MR()[2 * (i % PE_MRING) + 1], bp);
    }
    PE_FN bool seed_is_used(int i) const { return (ld(SU()[i >> 5]) >> (i & 31)) & 1u; }
    PE_FN void seed_set_used(int i) { st(SU()[i >> 5], ld(SU()[i >> 5]) | (1u << (i & 31))); }

    // ------------------------------------------------------------------ Phases 1-3 (alignLandauVishkin / alignHamming)
    PE_FN void phases123(bool hamming) {
        snapgpu_paired_result &res = S()->res, &alt = S()->alt;
        PESet &all = S()->all, &non_alt = S()->non_alt;
        alt.status[0] = alt.status[1] = SNAPGPU_NotFound;
        if (!hamming) { alt.ref_span[0] = alt.ref_span[1] = 0; res.ref_span[0] = res.ref_span[1] = 0; res.liftover[0] = res.liftover[1] = 0; }
        for (int r = 0; r < 2; r++) {
            res.clipping_for_read_adjustment[r] = 0; res.used_affine_gap_scoring[r] = 0; res.bases_clipped_before[r] = 0;
            res.bases_clipped_after[r] = 0; res.ag_score[r] = 0; res.used_gapless_clipping[r] = 0;
        }
        n_agc = 0;
        n_sec = 0;                                                                                                 // :294 / :1472

        const int sl = cfg.seed_len;
        int max_seeds;
        if (cfg.num_seeds != 0) max_seeds = (int)cfg.num_seeds;
        else max_seeds = (int)((read_len[0] > read_len[1] ? read_len[0] : read_len[1]) * cfg.seed_coverage / sl);      // :301-307
        if (max_seeds > (int)cfg.max_seeds) max_seeds = (int)cfg.max_seeds;

        n_cand = 0; n_mate[0] = n_mate[1] = 0; n_anchor = 0;
        for (int k = 0; k <= cfg.max_k + cfg.extra_depth; k++) st(LH()[k], -1);
        set_init(all); set_init(non_alt);

        if (read_len[0] < sl || read_len[1] < sl) return;                                                             // :343

        uint32_t n_count = 0;
        for (int w = 0; w < 2; w++) {
            popular[w] = 0;
            for (int d = 0; d < 2; d++) hs_init(2 * w + d);
            n_count += pl->count_n(L(rd[w][0]), read_len[w]);
        }
        if ((int)n_count > cfg.max_k) return;                                                                         // :385

        // ---- Phase 1: seed lookups into the four hit sets (:417-502)
        const uint64_t t_p1 = PL::clock();
        int64_t total_hits[2][2] = {{0, 0}, {0, 0}};
        for (int w = 0; w < 2; w++) {
            int next_seed = 0, lookups = 0;
            uint32_t wrap = 0;
            const int n_possible = read_len[w] - sl + 1;
            const int mx = read_len[0] > read_len[1] ? read_len[0] : read_len[1];
            for (int i = 0; i < (mx + 31) / 32; i++) st(SU()[i], 0);
            bool begins[2] = {true, true};
            while (lookups < n_possible && lookups < max_seeds) {
                if (next_seed >= n_possible) {
                    wrap++;
                    begins[0] = begins[1] = true;
                    if (wrap >= (uint32_t)sl) break;
                    next_seed = (int)pl->wrapped_seed(wrap);
                }
                while (next_seed < n_possible && seed_is_used(next_seed)) next_seed++;
                if (next_seed >= n_possible) continue;
                seed_set_used(next_seed);
                PEHits h[2];
                if (!pl->lookup(L(rd[w][0]) + next_seed, h)) { next_seed++; continue; }                                // seed with an N, :454
                S()->cnt.lookups++;
                lookups++;
                for (int d = 0; d < 2; d++) {
                    const int offset = d == 0 ? next_seed : read_len[w] - sl - next_seed;
                    if (h[d].n_hits < (int64_t)cfg.max_big_hits) {
                        total_hits[w][d] += h[d].n_hits;
                        S()->cnt.hits += (uint64_t)h[d].n_hits;                 // the hit lists this pair is entitled to read (roofline byte model)
                        if (h[d].n_hits > 1) S()->cnt.overflow_lists++;
                        hs_record(2 * w + d, (uint32_t)offset, h[d], begins[d]);
                        begins[d] = false;
                    } else {
                        popular[w]++;
                    }
                }
                if ((max_seeds - lookups + 1) * sl + next_seed < n_possible) {
                    next_seed += (n_possible - next_seed - 1) / (max_seeds - lookups + 1);                         // space the rest out evenly, :494
                } else {
                    next_seed += sl;
                }
            }
        }
        more = (total_hits[0][0] + total_hits[0][1] > total_hits[1][0] + total_hits[1][1]) ? 0 : 1;               // :513
        fewer = 1 - more;

        // ---- Phase 2: walk both set pairs from high to low locations, collect candidates (:527-741)
        const uint64_t t_p2 = PL::clock();
        PT2_OFF(S()->cnt.cyc_lookup += t_p2 - t_p1);
        int max_used_list = 0;
        uint32_t n_cand0 = 0;                       // candidates of set pair 0 (they come first in cand[])
        for (int sp = 0; sp < 2; sp++) {
            // set pair 0 = read0 FORWARD + read1 RC, set pair 1 = read0 RC + read1 FORWARD
            const int s_fewer = 2 * fewer + set_dir(sp, fewer), s_more = 2 * more + set_dir(sp, more);
            int64_t loc_f, loc_m = SNAPGPU_InvalidGenomeLocation32;
            uint32_t so_f = 0, so_m = 0;
            bool out_of_more = false;
            int64_t last_mate_loc = 0;              // mate[sp][n_mate[sp] - 1].loc (the walk looks back at it in every step)
            if (sp == 1) n_cand0 = n_cand;
            typename PL::HSCursor cf, cm;
            pl->hs_begin_walk(lks(s_fewer), &HS()[s_fewer], 0, cfg.max_seeds, cf); pl->hs_begin_walk(lks(s_more), &HS()[s_more], 1, cfg.max_seeds, cm);
            if (hs_first(s_fewer, &loc_f, &so_f, cf)) continue;
            for (;;) {
                if (loc_m > loc_f + (int64_t)cfg.max_spacing) {
                    if (!hs_next_le(s_more, loc_f + (int64_t)cfg.max_spacing, &loc_m, &so_m, cm)) break;
                }
                if ((loc_m + (int64_t)cfg.max_spacing < loc_f || out_of_more) &&
                    (0 == n_mate[sp] || !within(last_mate_loc, loc_f, cfg.max_spacing))) {
                    if (out_of_more) break;
                    if (!hs_next_le(s_fewer, loc_m + (int64_t)cfg.max_spacing, &loc_f, &so_f, cf)) break;
                    continue;
                }
                while (loc_m + (int64_t)cfg.max_spacing >= loc_f && !out_of_more) {
                    uint32_t bp = hs_best_possible(s_more, cm);
                    if (n_mate[sp] >= cfg.pool_size / 2) { overflow = 1; return; }
                    PT2_T0();
                    G(PEMate) *m = &mate[sp][n_mate[sp]];
                    if (PL::lane0()) {                                                                             // ScoringMateCandidate::init
                        m->loc = loc_m; m->best_possible = (int32_t)bp; m->seed_offset = so_m; m->score = PE_NOT_YET_SCORED;
                        m->score_limit = -1; m->match_prob = 0; m->genome_offset = 0; m->used_gapless = 0; m->clip_before = 0;
                        m->clip_after = 0; m->ag_score = 0; m->lv_indels = 0; m->big_indel = 0; m->ref_span = 0;
                    }
                    PL::sync();
                    mring_put(n_mate[sp], loc_m, bp);
                    last_mate_loc = loc_m;
                    n_mate[sp]++;
                    PT2_ADD(cyc_ag);
                    if (!hs_next_lower(s_more, &loc_m, &so_m, cm)) { loc_m = 0; out_of_more = true; break; }
                }
                const int bp_f = (int)hs_best_possible(s_fewer, cf);
                PT2_T0();
                int lowest_mate = cfg.max_k + cfg.extra_depth;
                for (int i = (int)n_mate[sp] - 1; i >= 0; i--) {
                    int64_t ml; int b;
                    if (mring != nullptr && (int)n_mate[sp] - 1 - i < PE_MRING) { ml = (int64_t)ld(MR()[2 * (i % PE_MRING)]); b = (int)ld(MR()[2 * (i % PE_MRING) + 1]); }
                    else { ml = ld(mate[sp][i].loc); b = ld(mate[sp][i].best_possible); }
                    if (ml > loc_f + (int64_t)cfg.max_spacing) break;
                    if (b < lowest_mate) lowest_mate = b;
                }
                if (lowest_mate + bp_f <= cfg.max_k + cfg.extra_depth) {
                    if (n_cand >= cfg.pool_size) { overflow = 1; return; }
                    const int list = lowest_mate + bp_f;
                    G(PECand) *c = &cand[n_cand];
                    const int32_t old_head = ld(LH()[list]);
                    if (PL::lane0()) {                                                                             // ScoringCandidate::init
                        c->loc = loc_f; c->set_pair = (uint32_t)sp; c->mate_index = n_mate[sp] - 1; c->seed_offset = so_f;
                        c->best_possible = (uint32_t)bp_f; c->next = old_head; c->anchor = -1; c->used_gapless = 0; c->clip_before = 0;
                        c->clip_after = 0; c->ag_score = 0; c->lv_indels = 0; c->match_prob = 1.0; c->big_indel = 0; c->ref_span = 0;
                    }
                    PL::sync();
                    st(LH()[list], (int32_t)n_cand);
                    n_cand++;
                    if (list > max_used_list) max_used_list = list;
                }
                PT2_ADD(cyc_ag);
                if (!hs_next_lower(s_fewer, &loc_f, &so_f, cf)) break;
            }
        }

        // ---- Phase 2a: seed-hinted indels raise the limit for candidates that sit close together (:743-801); not in alignHamming
        PT2_T0();
#if defined(SNAPGPU_WAVE_EMU)
        const bool dbg_seq_hints = getenv("SNAPGPU_DEBUG_SEQ_HINTS") != nullptr;      // (emulator: the reference's two-pointer loops instead of the closed form)
#else
        const bool dbg_seq_hints = false;
#endif
        if (!hamming && PL::FAST_HITSET && !dbg_seq_hints) {
            // (each list is strictly descending, which gives the loops below a closed form: paired_dev.h: hint_indels)
            for (int sp = 0; sp < 2; sp++) pl->hint_indels(mate[sp], 0u, n_mate[sp], cfg.max_k_for_indels);
            pl->hint_indels(cand, 0u, n_cand0, cfg.max_k_for_indels);
            pl->hint_indels(cand, n_cand0, n_cand - n_cand0, cfg.max_k_for_indels);
        } else if (!hamming) {
            for (int sp = 0; sp < 2; sp++) {
                int bottom = 0, top = 1;
                while (top < (int)n_mate[sp]) {
                    int64_t spread = dist(ld(mate[sp][bottom].loc), ld(mate[sp][top].loc));
                    if (spread < cfg.max_k_for_indels) {
                        int64_t b = ld(mate[sp][bottom].big_indel), t = ld(mate[sp][top].big_indel);
                        st(mate[sp][bottom].big_indel, spread > b ? spread : b);
                        st(mate[sp][top].big_indel, spread > t ? spread : t);
                        top++;
                    } else if (bottom < top - 1) {
                        bottom++;
                    } else {
                        bottom++; top++;
                    }
                }
            }
            int bottom = 0, top = 1;
            while (top < (int)n_cand) {
                if (ld(cand[bottom].set_pair) != ld(cand[top].set_pair)) { bottom = top; top = top + 1; continue; }
                int64_t spread = dist(ld(cand[bottom].loc), ld(cand[top].loc));
                if (spread < cfg.max_k_for_indels) {
                    int b = ld(cand[bottom].big_indel), t = ld(cand[top].big_indel);
                    st(cand[bottom].big_indel, (int)spread > b ? (int)spread : b);
                    st(cand[top].big_indel, (int)spread > t ? (int)spread : t);
                    top++;
                } else if (bottom < top - 1) {
                    bottom++;
                } else {
                    bottom++; top++;
                }
            }
        }

        // ---- Phase 3: score candidates in order of their best possible score (:803-1190)
        PT2_ADD(cyc_single);
        S()->cnt.cyc_intersect += PL::clock() - t_p2;
        int cur_list = 0;
        bool done = false;
        while (!done && cur_list <= max_used_list) {
            {
                int a = all.best_pair_score, n = non_alt.best_pair_score;
                int x = a < n - cfg.max_gap_alt ? a : n - cfg.max_gap_alt;
                int y = a + cfg.max_gap_alt < n ? a + cfg.max_gap_alt : n;
                int z = x > y ? x : y;
                int lim = cfg.extra_depth + (cfg.max_k < z ? cfg.max_k : z);
                if (cur_list > PL::i32(lim)) break;
            }
            const int ci = ld(LH()[cur_list]);
            if (ci < 0) { cur_list++; continue; }
            G(PECand) *c = &cand[ci];
            const int64_t c_loc = ld(c->loc);
            const int sp = (int)ld(c->set_pair);
            const int c_big = ld(c->big_indel);
            const uint32_t c_so = ld(c->seed_offset);
            const bool non_alt_aln = !cfg.alt_aware || !pl->is_alt(c_loc);
            int limit = score_limit(non_alt_aln, hamming ? 0 : c_big);
            if (cur_list > limit) { st(LH()[cur_list], ld(c->next)); continue; }

            LocScore f;
            f.reset(0, 0, 0, 0);
            if (hamming) score_hamming(fewer, set_dir(sp, fewer), c_loc, (int)c_so, limit, f);
            else         score_lv(fewer, set_dir(sp, fewer), c_loc, (int)c_so, limit, f);
            const int fewer_score = PL::i32(f.score), fewer_off = PL::i32(f.offset);
            const double fewer_mp = PL::f64(f.mp);
            const bool c_gapless = f.gapless;
            if (PL::lane0()) {
                c->match_prob = fewer_mp;
                c->clip_before = f.clip_before; c->clip_after = f.clip_after; c->ag_score = f.ag_score; c->used_gapless = f.gapless ? 1u : 0u;
                if (!hamming) { c->lv_indels = f.lv_indels; c->ref_span = f.ref_span; }
            }
            PL::sync();

            if (fewer_score != -1) {
                uint32_t mi = ld(c->mate_index);
                for (;;) {
                    G(PEMate) *m = &mate[sp][mi];
                    const int64_t m_loc = ld(m->loc);
                    if (!hamming) {
                        int64_t mb = ld(m->big_indel);
                        int64_t cb = c_big < fewer_score ? c_big : fewer_score;
                        limit = score_limit(non_alt_aln, mb > cb ? mb : cb);
                    }
                    if (!within(m_loc, c_loc, (int64_t)cfg.min_spacing - 1) &&
                        ((ld(m->best_possible) <= limit - fewer_score) || (hamming && c_gapless))) {
                        const int mate_limit = (hamming && c_gapless) ? limit : limit - fewer_score;
                        int m_score = ld(m->score);
                        if (m_score == PE_NOT_YET_SCORED || (m_score == -1 && ld(m->score_limit) < limit - fewer_score) || (hamming && c_gapless)) {
                            LocScore g;
                            g.reset(ld(m->clip_before), ld(m->clip_after), ld(m->ag_score), ld(m->lv_indels));
                            if (hamming) score_hamming(more, set_dir(sp, more), m_loc, (int)ld(m->seed_offset), mate_limit, g);
                            else         score_lv(more, set_dir(sp, more), m_loc, (int)ld(m->seed_offset), mate_limit, g);
                            m_score = PL::i32(g.score);
                            const double g_mp = PL::f64(g.mp);
                            if (PL::lane0()) {
                                m->score = g.score; m->match_prob = g_mp; m->genome_offset = g.offset; m->clip_before = g.clip_before;
                                m->clip_after = g.clip_after; m->ag_score = g.ag_score; m->used_gapless = g.gapless ? 1u : 0u;
                                if (!hamming) { m->lv_indels = g.lv_indels; m->ref_span = g.ref_span; }
                                m->score_limit = mate_limit;
                            }
                            PL::sync();
                        }
                        const bool m_gapless = ld(m->used_gapless) != 0;
                        if (m_score != -1 && ((fewer_score + m_score <= limit) || (hamming && (c_gapless || m_gapless)))) {
                            const double m_mp = ld(m->match_prob);
                            const double pair_p = m_mp * fewer_mp;
                            const int pair_score = m_score + fewer_score;
                            const int pair_ag = ld(m->ag_score) + ld(c->ag_score);
                            const int64_t new_more = m_loc + ld(m->genome_offset), new_fewer = c_loc + fewer_off;

                            // merge anchors: treat alignments within 50 bases of one another as one (:925-987)
                            int ai = ld(c->anchor);
                            if (ai < 0) {
                                for (int j = ci - 1; j >= 0 && within(ld(cand[j].loc), new_fewer, 50) && (int)ld(cand[j].set_pair) == sp; j--) {
                                    int aj = ld(cand[j].anchor);
                                    if (aj >= 0) { ai = aj; st(c->anchor, ai); break; }
                                }
                                if (ai < 0) {
                                    for (int j = ci + 1; j < (int)n_cand && within(ld(cand[j].loc), new_fewer, 50) && (int)ld(cand[j].set_pair) == sp; j++) {
                                        int aj = ld(cand[j].anchor);
                                        if (aj >= 0) { ai = aj; st(c->anchor, ai); break; }
                                    }
                                }
                            }
                            bool eliminated, replaced = false;
                            double old_p;
                            if (ai < 0) {
                                if (n_anchor >= cfg.pool_size) { overflow = 1; return; }
                                ai = (int)n_anchor++;
                                G(PEAnchor) *an = &anchor[ai];
                                if (PL::lane0()) { an->loc_more = new_more; an->loc_fewer = new_fewer; an->match_prob = pair_p; an->pair_score = pair_score; an->pair_ag = pair_ag; }
                                PL::sync();
                                eliminated = false; old_p = 0;
                                st(c->anchor, ai);
                            } else {                                                                               // MergeAnchor::checkMerge, :3820-3876
                                G(PEAnchor) *an = &anchor[ai];
                                const int64_t a_more = ld(an->loc_more), a_fewer = ld(an->loc_fewer);
                                if (a_more == SNAPGPU_InvalidGenomeLocation32 || !(dist(a_more, new_more) < 50 && dist(a_fewer, new_fewer) < 50)) {
                                    if (PL::lane0()) { an->loc_more = new_more; an->loc_fewer = new_fewer; an->match_prob = pair_p; an->pair_score = pair_score; an->pair_ag = pair_ag; }
                                    PL::sync();
                                    old_p = 0.0; eliminated = false;
                                } else {
                                    const int a_ag = ld(an->pair_ag);
                                    const double a_p = ld(an->match_prob);
                                    if (pair_ag > a_ag || (pair_ag == a_ag && pair_p > a_p)) {
                                        old_p = a_p;
                                        if (PL::lane0()) { an->match_prob = pair_p; an->pair_score = pair_score; an->pair_ag = pair_ag; }
                                        PL::sync();
                                        replaced = true; eliminated = false;
                                    } else {
                                        old_p = 0; eliminated = true;
                                    }
                                }
                            }

                            if (!eliminated) {
                                set_sub_all(all, old_p);
                                if (non_alt_aln) set_sub_all(non_alt, old_p);

                                // keep the displaced best as a Phase-4 candidate (:1027-1062)
                                bool close = hamming ? (pair_score <= all.best_pair_score && cfg.extra_depth >= all.best_pair_score - pair_score)
                                                     : (cfg.extra_depth >= all.best_pair_score - pair_score);
                                // ... and as a secondary result (:999-1034 / :2077-2112)
                                if (want_sec() && pair_p > all.p_best && (!hamming || pair_score <= all.best_pair_score) &&
                                    cfg.om >= all.best_pair_score - pair_score) {
                                    if (n_sec >= cfg.sec_cap) { overflow = 1; return; }
                                    agc_from_set(&sec[n_sec], all);
                                    n_sec++;
                                }
                                if (!replaced && pair_p > all.p_best && cfg.ag_cand_cap > 0 && close) {
                                    if (n_agc >= cfg.ag_cand_cap) { overflow = 1; return; }
                                    agc_from_set(&agc[n_agc], all);
                                    n_agc++;
                                }
                                if (non_alt_aln) set_update(non_alt, pair_score, pair_ag, pair_p, fewer_score, fewer_off, ci, (int)mi, sp);
                                const bool updated = set_update(all, pair_score, pair_ag, pair_p, fewer_score, fewer_off, ci, (int)mi, sp);

                                bool near = hamming ? (pair_score >= all.best_pair_score && cfg.extra_depth >= pair_score - all.best_pair_score)
                                                    : (pair_score <= cfg.max_k + cfg.extra_depth && cfg.extra_depth >= pair_score - all.best_pair_score);
                                const bool near_sec = hamming ? (pair_score >= all.best_pair_score && cfg.om >= pair_score - all.best_pair_score)
                                                              : (pair_score <= cfg.max_k + cfg.extra_depth && cfg.om >= pair_score - all.best_pair_score);
                                if (!updated && want_sec() && near_sec) {                                        // :1082-1124 / :2157-2199
                                    if (n_sec >= cfg.sec_cap) { overflow = 1; return; }
                                    agc_from_pair(&sec[n_sec], ci, (int)mi, sp, fewer_score, fewer_off);
                                    n_sec++;
                                }
                                if (!updated && cfg.ag_cand_cap > 0 && near) {                                     // :1126-1166
                                    if (n_agc >= cfg.ag_cand_cap) { overflow = 1; return; }
                                    agc_from_pair(&agc[n_agc], ci, (int)mi, sp, fewer_score, fewer_off);
                                    n_agc++;
                                }
                                if ((cfg.alt_aware ? non_alt.p_all : all.p_all) >= 4.9 && !want_sec()) { done = true; break; }    // :1181
                            }
                        }
                    }
                    if (mi == 0 || !within(ld(mate[sp][mi - 1].loc), c_loc, cfg.max_spacing)) break;
                    mi--;
                }
            }
            if (done) break;
            st(LH()[cur_list], ld(c->next));
        }

        // ---- emit (:1192-1262)
        const bool emit_all = !cfg.alt_aware || non_alt.best_pair_score > all.best_pair_score + cfg.max_gap_alt;
        const int emit_best = emit_all ? all.best_pair_score : non_alt.best_pair_score;
        if (emit_best == SNAPGPU_TooBigScoreValue) {
            res_not_found(res, alt, popular);
        } else {
            if (emit_all) set_fill(all, res, popular); else set_fill(non_alt, res, popular);
            if (cfg.alt_aware && !emit_all && (all.loc[0] != non_alt.loc[0] || all.loc[1] != non_alt.loc[1])) {
                set_fill(all, alt, popular);
                alt.supplementary[0] = alt.supplementary[1] = 1;
            } else {
                alt.status[0] = alt.status[1] = SNAPGPU_NotFound;
            }
        }
        for (int w = 0; w < 2; w++) res.score_prior_to_clipping[w] = res.score[w];                                 // :1273-1275
        if (want_sec()) finalize_secondary(emit_best, res);
    }

    // ------------------------------------------------------------------ the tail of alignLandauVishkin / alignHamming (:1289-1411 / :2364-2483)
    // with ignoreAlignmentAdjustmentsForOm (the default).  Works on an index list; the records move once, when they are emitted.
    // stable sort of sec_ord[0..n) by sec_key[sec_ord[.]]: what glibc's merge-sorting qsort leaves.
    PE_FN void sec_stable_sort(uint32_t n) {
        G(uint32_t) *tmp = sec_key + cfg.sec_cap;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t me = ld(sec_ord[i]), k = ld(sec_key[me]);
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; j++) {
                const uint32_t kj = ld(sec_key[ld(sec_ord[j])]);
                rank += (kj < k || (kj == k && j < i)) ? 1u : 0u;
            }
            st(tmp[rank], me);
        }
        for (uint32_t i = 0; i < n; i++) st(sec_ord[i], ld(tmp[i]));
    }
    PE_FN void finalize_secondary(int best_pair_score, const snapgpu_paired_result &res) {
        uint32_t n = n_sec;
        for (uint32_t i = 0; i < n; i++) {
            G(snapgpu_paired_result) *e = &sec[i];
            const int s0 = ld(e->score[0]), s1 = ld(e->score[1]);
            if (PL::lane0()) { e->score_prior_to_clipping[0] = s0; e->score_prior_to_clipping[1] = s1; }
            st(sec_ord[i], i); st(sec_key[i], (uint32_t)(s0 + s1));
        }
        PL::sync();
        {   // too far from the best now: the last entry moves into the hole (:1320-1331)
            uint32_t i = 0;
            while (i < n) {
                if ((int)ld(sec_key[ld(sec_ord[i])]) > best_pair_score + cfg.om) { st(sec_ord[i], ld(sec_ord[n - 1])); n--; }
                else i++;
            }
        }
        if (cfg.mpc > 0 && res.status[0] != SNAPGPU_NotFound && n > 0) {                                            // :1336-1404
            const int primary_contig = contig_num(res.location[0]);
            // PairedAlignmentResult::compareByContigAndScore compares the ADDRESSES of the score arrays (AlignmentResult.cpp:82-85),
            // and qsort sorts records this large through pointers to the originals: contig, then position in the list
            bool too_many = false;
            for (uint32_t i = 0; i < n; i++) st(sec_key[ld(sec_ord[i])], (uint32_t)contig_num(ld(sec[ld(sec_ord[i])].location[0])));
            for (uint32_t i = 0; i < n && !too_many; i++) {
                const uint32_t c = ld(sec_key[ld(sec_ord[i])]);
                int count = (int)c == primary_contig ? 1 : 0;
                for (uint32_t j = 0; j < n; j++) count += ld(sec_key[ld(sec_ord[j])]) == c ? 1 : 0;
                if (count > cfg.mpc) too_many = true;
            }
            if (too_many) {
                sec_stable_sort(n);
                int cur = -1, cur_count = 0; uint32_t dest = 0;
                for (uint32_t src = 0; src < n; src++) {
                    const uint32_t me = ld(sec_ord[src]);
                    const int c = (int)ld(sec_key[me]);
                    if (c != cur) { cur = c; cur_count = c == primary_contig ? 1 : 0; }
                    cur_count++;
                    if (cur_count <= cfg.mpc) { st(sec_ord[dest], me); dest++; }
                }
                n = dest;
            }
        }
        if ((int64_t)n > cfg.omax) {                                                                                // :1407-1410
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t me = ld(sec_ord[i]);
                st(sec_key[me], (uint32_t)(ld(sec[me].score[0]) + ld(sec[me].score[1])));
            }
            sec_stable_sort(n);                                                                                    // compareByScore
            n = (uint32_t)cfg.omax;
        }
        n_sec = n;
    }
    // secondary result k of this pair, after the filtering
    PE_FN const G(snapgpu_paired_result) *secondary(uint32_t k) const { return &sec[ld(sec_ord[k])]; }

    // ------------------------------------------------------------------ Phase 4 (alignAffineGap, :2489-2970)
    PE_FN void phase4() {
        snapgpu_paired_result &res = S()->res, &alt = S()->alt;
        PESet &all = S()->all, &non_alt = S()->non_alt;
        if (res.status[0] == SNAPGPU_NotFound || res.status[1] == SNAPGPU_NotFound) return;
        const int sl = cfg.seed_len;
        if (read_len[0] < sl || read_len[1] < sl) return;
        uint32_t n_count = pl->count_n(L(rd[0][0]), read_len[0]) + pl->count_n(L(rd[1][0]), read_len[1]);
        if ((int)n_count > cfg.max_k) return;

        const int max_k_same = cfg.gap_open / (cfg.sub_penalty - cfg.gap_extend);
        const int best_pair_score = PL::i32(res.score[0] + res.score[1]);
        int limit, limit_alt;
        if (res.used_gapless_clipping[0] || res.used_gapless_clipping[1]) limit = limit_alt = PE_MAXK1;
        else limit = limit_alt = cfg.max_k + cfg.extra_depth;
        int g_off[2] = {0, 0};
        bool skip[2] = {false, false};
        const double old_p_best = res.match_probability[0] * res.match_probability[1];
        const double old_p_best_alt = (alt.status[0] != SNAPGPU_NotFound) ? alt.match_probability[0] * alt.match_probability[1] : 0.0;

        for (int r = 0; r < 2; r++) {
            if (res.used_gapless_clipping[r] || res.score[r] > max_k_same) {
                res.used_affine_gap_scoring[r] = 1;
                if (!res.used_gapless_clipping[r]) limit = limit > res.score[r] ? limit : res.score[r];
                int sc, cb = res.bases_clipped_before[r], ca = res.bases_clipped_after[r], ag = res.ag_score[r], span;
                double mp = res.match_probability[r];
                score_ag(r, res.direction[r], res.orig_location[r], res.seed_offset[r], PL::i32(limit), &sc, &mp, &g_off[r], &cb, &ca, &ag, &span);
                res.score[r] = PL::i32(sc); res.match_probability[r] = PL::f64(mp); res.bases_clipped_before[r] = cb; res.bases_clipped_after[r] = ca;
                res.ag_score[r] = ag; res.ref_span[r] = span;
                if (res.score[r] != -1) { res.location[r] = res.orig_location[r] + g_off[r]; limit -= res.score[r]; }
                else res.status[r] = SNAPGPU_NotFound;

                if (alt.status[r] != SNAPGPU_NotFound) {
                    if (alt.used_gapless_clipping[r] || alt.score[r] > max_k_same) {
                        alt.used_affine_gap_scoring[r] = 1;
                        if (!alt.used_gapless_clipping[r]) limit_alt = limit_alt > alt.score[r] ? limit_alt : alt.score[r];
                        int sc2, cb2 = alt.bases_clipped_before[r], ca2 = alt.bases_clipped_after[r], ag2 = alt.ag_score[r], span2;
                        double mp2 = alt.match_probability[r];
                        score_ag(r, alt.direction[r], alt.orig_location[r], alt.seed_offset[r], PL::i32(limit_alt), &sc2, &mp2, &g_off[r], &cb2, &ca2, &ag2, &span2);
                        alt.score[r] = PL::i32(sc2); alt.match_probability[r] = PL::f64(mp2); alt.bases_clipped_before[r] = cb2; alt.bases_clipped_after[r] = ca2;
                        alt.ag_score[r] = ag2; alt.ref_span[r] = span2;
                        if (alt.score[r] != -1) { alt.location[r] = alt.orig_location[r] + g_off[r]; limit_alt -= alt.score[r]; }
                        else alt.status[r] = SNAPGPU_NotFound;
                    }
                }
            } else {
                res.used_affine_gap_scoring[r] = 0;
                skip[r] = true;
            }
        }

        if (res.status[0] == SNAPGPU_NotFound || res.status[1] == SNAPGPU_NotFound || res.score[0] > PE_MAXK1 || res.score[1] > PE_MAXK1) {
            res_not_found(res, alt, 0);
            return;
        }

        PESet &A = all, &N = non_alt;
        bool non_alt_aln = !cfg.alt_aware || !pl->is_alt(res.location[0]);
        set_init_from(A, res);
        bool alt_best = false;
        if (alt.status[0] != SNAPGPU_NotFound && alt.status[1] != SNAPGPU_NotFound) {
            // updateBestHitIfNeeded(PairedAlignmentResult*) on a result that lives in LDS
            double pp = alt.match_probability[0] * alt.match_probability[1];
            int ps = alt.score[0] + alt.score[1], pa = alt.ag_score[0] + alt.ag_score[1];
            A.p_all += pp;
            if (pa > A.best_pair_ag || (pa == A.best_pair_ag && pp > A.p_best)) {
                A.best_pair_score = ps; A.best_pair_ag = pa; A.p_best = pp;
                for (int i = 0; i < 2; i++) {
                    A.loc[i] = alt.location[i]; A.orig[i] = alt.orig_location[i]; A.score[i] = (uint32_t)alt.score[i]; A.dir[i] = alt.direction[i];
                    A.used_ag[i] = alt.used_affine_gap_scoring[i]; A.gapless[i] = alt.used_gapless_clipping[i]; A.clip_before[i] = alt.bases_clipped_before[i];
                    A.clip_after[i] = alt.bases_clipped_after[i]; A.ag[i] = alt.ag_score[i]; A.seed_off[i] = alt.seed_offset[i]; A.mp[i] = alt.match_probability[i];
                    A.lv_indels[i] = alt.lv_indels[i]; A.ref_span[i] = alt.ref_span[i];
                }
                alt_best = true;
            }
        }
        if (non_alt_aln) set_init_from(N, res); else set_init(N);

        if (!skip[0] || !skip[1]) {
            double new_p = res.match_probability[0] * res.match_probability[1];
            if (alt_best) {
                double new_p_alt = alt.match_probability[0] * alt.match_probability[1];
                set_sub_all(A, old_p_best_alt);
                A.p_best = new_p_alt;                                  // updateProbabilityOfBestPair(.., false)
            } else {
                set_sub_all(A, old_p_best);
                A.p_best = new_p; A.p_all += new_p;
            }
            if (non_alt_aln) { set_sub_all(N, old_p_best); N.p_best = new_p; N.p_all += new_p; }
        }

        if (n_agc > 0 && (!skip[0] || !skip[1])) {
            limit = (cfg.max_k < best_pair_score ? cfg.max_k : best_pair_score) + cfg.extra_depth;
            // qsort(compareByScore) is glibc's stable merge sort here: visit candidates by (pair score, insertion index).
            // pl.sort_candidates writes that order (a stable counting sort on the key kept in `reserved`) to agc_order.
            pl->sort_candidates(agc, n_agc, agc_order);
            G(PEHelpSpec) *spec = nullptr;
            uint32_t spec_from = 0;
            spec_used = 0;
            for (uint32_t t = 0; t < n_agc; t++) {
                if constexpr (PL::HELP) {
                    // A long rest of the list is published ON DEMAND -- once waves that have run out of pairs exist (or at once, in the
                    // eager mode the tests use) -- so that the help slots are held by the pairs that ARE the tail of the launch, not by
                    // whichever long pair came first while every wave was still busy.  Candidates t .. n_agc - 1 are then scored
                    // speculatively under the limit the walk has arrived with, by this wave and the idle ones, before the walk goes on
                    // and consumes the answers (struct PEHelpSpec).  Looked at every 16 candidates.
                    if (spec == nullptr && (t & 15u) == 0u && n_agc - t >= help_min && pl->help_wanted()) {
                        spec = pl->help_phase4(*this, n_agc, t, PL::i32(limit), best_pair_score, skip);
                        spec_from = t;
                    }
                }
                G(snapgpu_paired_result) *e = &agc[ld(agc_order[t])];
                phase4_candidate(e, limit, best_pair_score, skip, g_off, (spec && t >= spec_from) ? &spec[t] : nullptr);
            }
            if constexpr (PL::HELP) { if (spec) pl->help_done(spec_used); }
        }

        const bool emit_all = !cfg.alt_aware || N.best_pair_score > A.best_pair_score + cfg.max_gap_alt;
        uint32_t pop[2] = {res.popular_seeds_skipped[0], res.popular_seeds_skipped[1]};
        uint32_t pop_alt[2] = {alt.popular_seeds_skipped[0], alt.popular_seeds_skipped[1]};
        if (emit_all) set_fill(A, res, pop); else set_fill(N, res, pop);
        if (cfg.alt_aware && !emit_all && (A.loc[0] != N.loc[0] || A.loc[1] != N.loc[1])) {
            set_fill(A, alt, pop_alt);
            alt.supplementary[0] = alt.supplementary[1] = 1;
        } else {
            alt.status[0] = alt.status[1] = SNAPGPU_NotFound;
        }
        if (cfg.use_soft_clip) alt_liftover();
    }

    // ------------------------------------------------------------------ ALT liftover (Genome.cpp:574-716, IntersectingPairedEndAligner.cpp:2870-2968)
    // Genome::getContigNumAtLocation.  The reference returns pointer garbage for a location no contig holds (before the first
    // contig, InvalidGenomeLocation); what matters to its callers is only that it differs from every real contig number.
    PE_FN int contig_num(int64_t loc) const {
        if (loc == SNAPGPU_InvalidGenomeLocation32) return -2;
        int lo = 0, hi = (int)cfg.proj.n_contigs - 1;
        while (lo <= hi) {
            const int mid = (lo + hi) / 2;
            const int64_t b = (int64_t)ld(cfg.proj.contig_begin[mid]);
            if (b <= loc && (mid == (int)cfg.proj.n_contigs - 1 || (int64_t)ld(cfg.proj.contig_begin[mid + 1]) > loc)) return mid;
            if (b <= loc) lo = mid + 1; else hi = mid - 1;
        }
        return -1;
    }
    PE_FN int64_t proj_span(int c) const {                                                    // getContigProjSpan, :640-659
        const uint32_t a = ld(cfg.proj.cigar_start[c]), b = ld(cfg.proj.cigar_start[c + 1]);
        int64_t off = 0;
        for (uint32_t i = a; i < b; i++) {
            const uint32_t op = ld(cfg.proj.cigar_ops[i]);
            const char act = (char)(op & 0xff);
            const int64_t cnt = (int64_t)(op >> 8);
            if (i == a) { if (act != 'S' && act != 'H') off += cnt; }
            else if (act == 'M' || act == 'I') off += cnt;
        }
        return off;
    }
    PE_FN int64_t proj_location(int64_t loc, int span) const {                                // getProjLocation, :661-710
        const int c = contig_num(loc);
        const int64_t cb = (int64_t)ld(cfg.proj.contig_begin[c]);
        const bool rc = ld(cfg.proj.proj_rc[c]) != 0;
        const int64_t offset = !rc ? loc - cb : proj_span(c) - (loc - cb + span);
        int64_t proj_off = 0;
        const uint32_t a = ld(cfg.proj.cigar_start[c]), b = ld(cfg.proj.cigar_start[c + 1]);
        uint32_t start = a;
        if (b > a) { const char act = (char)(ld(cfg.proj.cigar_ops[a]) & 0xff); if (act == 'S' || act == 'H') start = a + 1; }
        int64_t x = 0, y = 0;                                  // x: primary, y: ALT
        for (uint32_t i = start; i < b; i++) {
            const uint32_t op = ld(cfg.proj.cigar_ops[i]);
            const char act = (char)(op & 0xff);
            const int64_t cnt = (int64_t)(op >> 8);
            if (act == 'M') {
                if (y <= offset && offset < y + cnt) { proj_off = x + (offset - y); break; }
                x += cnt; y += cnt;
            } else if (act == 'D') {
                x += cnt;
            } else if (act == 'I') {
                if (y <= offset && offset < y + cnt) { proj_off = x; break; }
                y += cnt;
            } else if (act == 'S' || act == 'H') {
                if (y <= offset && offset < y + cnt) { proj_off = -1; break; }
                y += cnt;
            }
        }
        if (proj_off == -1) return SNAPGPU_InvalidGenomeLocation32;
        return (int64_t)ld(cfg.proj.proj_begin[c]) + proj_off;
    }

    // scoreLocationWithAffineGapLiftover (:2971-3117): the whole read against the projected location, no seed to anchor on
    PE_FN void score_ag_liftover(int which, int dir, int64_t loc, int limit, int *score, double *mp, int *offset,
                                 int *clip_before, int *clip_after, int *ag_score, int *ref_span) {
        const int rl = read_len[which];
        const int64_t glen = (int64_t)rl + SNAPGPU_MAX_K;
        *offset = 0; *ref_span = 0;
        if (!pl->substring_ok(loc, glen)) { *score = -1; *mp = 0; *ag_score = -1; return; }
        *clip_before = 0; *clip_after = 0;
        const uint8_t *data = pl->window(loc, rl);
        const uint8_t *R = L(rd[which][dir]), *Qd = L(ql[which][dir]);
        const int clip = 2;                                    // useSoftClip (the caller's condition) + useAltLiftover
        int score1 = 0, score2 = 0;
        double mp2 = 1.0;
        AGOut a = pl->ag(rl >= 3 * (2 * limit + 1), +1, R, Qd, rl, data, (int)glen, limit, rl, dir != 0, clip);
        note_ag_call(0, (uint32_t)a.stale);
        const int text_rem = a.text_offset;
        *clip_after = a.pattern_offset; score1 = a.n_edits;
        if (score1 != -1 && score1 <= PE_MAXK1) {
            const int left = score1;
            const int plen = rl - *clip_after;
            AGOut b = pl->ag(plen >= 3 * (2 * left + 1), -1, R + (rl - 1 - *clip_after), Qd + (rl - 1 - *clip_after), plen,
                            data + (rl - text_rem - 1), rl - text_rem, left, rl, dir != 0, clip);
            note_ag_call(1, (uint32_t)b.stale);
            *clip_before = b.pattern_offset; score2 = b.n_edits; mp2 = b.mp;
            if (score2 == -1 || score2 > PE_MAXK1) {
                *score = -1; *offset = 0; *ag_score = -1;
            } else {
                *score = score2;
                *ag_score = b.ag_score - rl;
                *mp = mp2;
                *offset = *clip_after + b.text_offset - text_rem;
                *ref_span = rl - text_rem - *offset;
            }
        } else {
            *score = -1; *offset = 0; *ag_score = -1;
        }
    }

    // the tail of alignAffineGap (:2866-2968): project an ALT alignment onto the primary assembly and rescore it there
    PE_FN void alt_liftover() {
        snapgpu_paired_result &res = S()->res, &alt = S()->alt;
        int best_alt = 0x7fffffff, best_res = 0x7fffffff, alt_proj_contig = -1, res_contig = -1;
        bool res_is_alt = false;
        if (alt.status[0] != SNAPGPU_NotFound && alt.status[1] != SNAPGPU_NotFound) {
            best_alt = alt.score[0] + alt.score[1];
            const int c = contig_num(alt.location[0]);
            alt_proj_contig = c < 0 ? -3 : contig_num((int64_t)ld(cfg.proj.proj_begin[c]));                   // getProjContigNumAtLocation
        }
        if (res.status[0] != SNAPGPU_NotFound && res.status[1] != SNAPGPU_NotFound) {
            best_res = res.score[0] + res.score[1];
            res_contig = contig_num(res.location[0]);
            res_is_alt = cfg.alt_aware && pl->is_alt(res.location[0]) && pl->is_alt(res.location[1]);
        }
        const bool use = cfg.alt_aware && ((best_alt < best_res && alt_proj_contig != res_contig) || res_is_alt);
        if (!PL::i32(use ? 1 : 0)) return;

        bool found[2] = {true, true};
        int new_dir[2], new_so[2], new_mapq[2];
        int64_t new_loc[2];
        S()->saved = res;
        const snapgpu_paired_result &src = res_is_alt ? res : alt;
        const int cc = contig_num(src.location[0]);
        const int proj_dir = (cc >= 0 && ld(cfg.proj.proj_rc[cc])) ? 1 : 0;                                   // isProjContigRC
        for (int r = 0; r < 2; r++) {
            new_dir[r] = (src.direction[r] != proj_dir) ? 1 : 0;
            new_so[r] = new_dir[r] != src.direction[r] ? read_len[r] - src.seed_offset[r] - 1 : src.seed_offset[r];
            new_loc[r] = proj_location(src.location[r], src.ref_span[r]);
            found[r] = new_loc[r] != SNAPGPU_InvalidGenomeLocation32;
            new_mapq[r] = res_is_alt ? (src.mapq[r] <= 3 ? 70 : src.mapq[r]) : src.mapq[r];
        }
        if (!(found[0] && found[1])) return;
        for (int r = 0; r < 2; r++) {
            res.used_affine_gap_scoring[r] = 1; res.liftover[r] = 1;
            res.direction[r] = new_dir[r]; res.location[r] = new_loc[r]; res.seed_offset[r] = new_so[r]; res.mapq[r] = new_mapq[r];
            int sc = res.score[r], off = 0, cb = res.bases_clipped_before[r], ca = res.bases_clipped_after[r], ag = res.ag_score[r], span = 0;
            double mp = res.match_probability[r];
            score_ag_liftover(r, new_dir[r], new_loc[r], PE_MAXK1, &sc, &mp, &off, &cb, &ca, &ag, &span);
            sc = PL::i32(sc);
            res.score[r] = sc; res.match_probability[r] = PL::f64(mp); res.bases_clipped_before[r] = cb; res.bases_clipped_after[r] = ca;
            res.ag_score[r] = ag; res.ref_span[r] = span;
            if (sc != -1 && sc <= PE_MAXK1) res.location[r] += off; else res.status[r] = SNAPGPU_NotFound;
        }
        if (res.status[0] == SNAPGPU_NotFound || res.status[1] == SNAPGPU_NotFound) res = S()->saved;          // no liftover alignment: keep the ALT one
    }

    // body of the candidate loop of alignAffineGap (:2736-2823)
    // One candidate scored ahead of the ordered walk: the limit bookkeeping of phase4_candidate up to its two scoreLocationWithAffineGap
    // calls, entered with limit `L`; nothing but *sp is written (the candidate record, the score sets and the work counters stay as they are).
    PE_FN void spec_candidate(const G(snapgpu_paired_result) *e, G(PEHelpSpec) *sp, int L, int best_pair_score, bool skip0, bool skip1) {
        int limit = L;
        int s0 = ld(e->score[0]), s1 = ld(e->score[1]);
        const int lv_pair_score = s0 + s1;
        const int lv_pair_indels = ld(e->lv_indels[0]) + ld(e->lv_indels[1]);
        const bool gl0 = ld(e->used_gapless_clipping[0]) != 0, gl1 = ld(e->used_gapless_clipping[1]) != 0;
        int lim0 = PE_SPEC_NONE, lim1 = PE_SPEC_NONE;
        int o_s[2] = {0, 0}, o_g[2] = {0, 0}, o_cb[2] = {0, 0}, o_ca[2] = {0, 0}, o_ag[2] = {0, 0}, o_span[2] = {0, 0};
        uint32_t o_sfw[2] = {0, 0}, o_sbw[2] = {0, 0}, o_calls[2] = {0, 0}, o_nag[2] = {0, 0};
        double o_mp[2] = {0.0, 0.0};
        if (gl0 || gl1) limit = PE_MAXK1;
        else if (lv_pair_score > best_pair_score + cfg.extra_depth && lv_pair_indels > 1) limit = cfg.max_k + cfg.extra_depth;
        if (lv_pair_score <= best_pair_score + cfg.extra_depth || lv_pair_indels > 1 || gl0 || gl1) {
            const uint32_t stale_keep = stale, later_keep = stale_later;
            const bool mode_keep = spec_mode;
            spec_mode = true;
            if (!skip0) {
                if (!gl0) limit = limit > s0 ? limit : s0;
                int cb = ld(e->bases_clipped_before[0]), ca = ld(e->bases_clipped_after[0]), span = 0, ag = ld(e->ag_score[0]), off = 0;
                double mp = ld(e->match_probability[0]);
                lim0 = PL::i32(limit);
                spec_n_ag = 0; spec_calls = 0; spec_stale_fw = 0; spec_stale_bw = 0;
                score_ag(0, ld(e->direction[0]), ld(e->orig_location[0]), ld(e->seed_offset[0]), lim0, &s0, &mp, &off, &cb, &ca, &ag, &span);
                s0 = PL::i32(s0);
                o_s[0] = s0; o_g[0] = PL::i32(off); o_cb[0] = PL::i32(cb); o_ca[0] = PL::i32(ca); o_ag[0] = PL::i32(ag); o_span[0] = PL::i32(span);
                o_mp[0] = PL::f64(mp); o_sfw[0] = spec_stale_fw; o_sbw[0] = spec_stale_bw; o_calls[0] = spec_calls; o_nag[0] = spec_n_ag;
            }
            if (s0 != -1 && s0 <= PE_MAXK1 && !skip1) {
                limit = limit - s0;
                if (!gl1) limit = limit > s1 ? limit : s1;
                int cb = ld(e->bases_clipped_before[1]), ca = ld(e->bases_clipped_after[1]), span = 0, ag = ld(e->ag_score[1]), off = 0;
                double mp = ld(e->match_probability[1]);
                lim1 = PL::i32(limit);
                spec_n_ag = 0; spec_calls = 0; spec_stale_fw = 0; spec_stale_bw = 0;
                score_ag(1, ld(e->direction[1]), ld(e->orig_location[1]), ld(e->seed_offset[1]), lim1, &s1, &mp, &off, &cb, &ca, &ag, &span);
                o_s[1] = PL::i32(s1); o_g[1] = PL::i32(off); o_cb[1] = PL::i32(cb); o_ca[1] = PL::i32(ca); o_ag[1] = PL::i32(ag); o_span[1] = PL::i32(span);
                o_mp[1] = PL::f64(mp); o_sfw[1] = spec_stale_fw; o_sbw[1] = spec_stale_bw; o_calls[1] = spec_calls; o_nag[1] = spec_n_ag;
            }
            stale = stale_keep; stale_later = later_keep; spec_mode = mode_keep;
        }
        if (PL::lane0()) {              // (PL::spec_st / spec_ld: stores and loads that other wavefronts see without a cache-wide fence)
            for (int r = 0; r < 2; r++) {
                PL::spec_st(sp->score[r], (int32_t)o_s[r]); PL::spec_st(sp->g_off[r], (int32_t)o_g[r]); PL::spec_st(sp->cb[r], (int32_t)o_cb[r]);
                PL::spec_st(sp->ca[r], (int32_t)o_ca[r]); PL::spec_st(sp->ag[r], (int32_t)o_ag[r]); PL::spec_st(sp->span[r], (int32_t)o_span[r]);
                PL::spec_st(sp->stale_fw[r], o_sfw[r]); PL::spec_st(sp->stale_bw[r], o_sbw[r]); PL::spec_st(sp->calls[r], o_calls[r]); PL::spec_st(sp->n_ag[r], o_nag[r]); PL::spec_st(sp->mp[r], o_mp[r]);
            }
            PL::spec_st(sp->lim[0], (int32_t)lim0); PL::spec_st(sp->lim[1], (int32_t)lim1);
        }
        PL::sync();
    }

    PE_FN void phase4_candidate(G(snapgpu_paired_result) *e, int &limit, int best_pair_score, const bool skip[2], int g_off[2],
                                const G(PEHelpSpec) *sp = nullptr) {
        snapgpu_paired_result &res = S()->res;
        PESet &A = S()->all, &N = S()->non_alt;
        int s0 = ld(e->score[0]), s1 = ld(e->score[1]);
        const int lv_pair_score = s0 + s1;
        const int lv_pair_indels = ld(e->lv_indels[0]) + ld(e->lv_indels[1]);
        const bool gl0 = ld(e->used_gapless_clipping[0]) != 0, gl1 = ld(e->used_gapless_clipping[1]) != 0;
        if (gl0 || gl1) limit = PE_MAXK1;
        else if (lv_pair_score > best_pair_score + cfg.extra_depth && lv_pair_indels > 1) limit = cfg.max_k + cfg.extra_depth;
        if (!(lv_pair_score <= best_pair_score + cfg.extra_depth || lv_pair_indels > 1 || gl0 || gl1)) return;

        const bool non_alt_aln = !cfg.alt_aware || !pl->is_alt(ld(e->location[0]));
        double mp0 = ld(e->match_probability[0]), mp1 = ld(e->match_probability[1]);
        const double old_p = mp0 * mp1;
        int ag0 = ld(e->ag_score[0]), ag1 = ld(e->ag_score[1]);
        int64_t loc0 = ld(e->location[0]), loc1 = ld(e->location[1]);
        if (!skip[0]) {
            st(e->used_affine_gap_scoring[0], 1);
            if (!gl0) limit = limit > s0 ? limit : s0;
            int cb = ld(e->bases_clipped_before[0]), ca = ld(e->bases_clipped_after[0]), span = 0;
            if (sp != nullptr && PL::spec_ld(sp->lim[0]) == PL::i32(limit)) {      // scored ahead of time with exactly these arguments
                s0 = PL::spec_ld(sp->score[0]); mp0 = PL::spec_ld(sp->mp[0]); g_off[0] = PL::spec_ld(sp->g_off[0]); cb = PL::spec_ld(sp->cb[0]);
                ca = PL::spec_ld(sp->ca[0]); ag0 = PL::spec_ld(sp->ag[0]); span = PL::spec_ld(sp->span[0]); take_spec_calls(sp, 0);
                S()->cnt.ag += PL::spec_ld(sp->n_ag[0]); spec_used++;
            } else {
                score_ag(0, ld(e->direction[0]), ld(e->orig_location[0]), ld(e->seed_offset[0]), PL::i32(limit), &s0, &mp0, &g_off[0], &cb, &ca, &ag0, &span);
            }
            s0 = PL::i32(s0); mp0 = PL::f64(mp0);
            if (PL::lane0()) { e->score[0] = s0; e->match_probability[0] = mp0; e->bases_clipped_before[0] = cb; e->bases_clipped_after[0] = ca; e->ag_score[0] = ag0; e->ref_span[0] = span; }
            PL::sync();
        }
        if (s0 != -1 && s0 <= PE_MAXK1) {
            loc0 = ld(e->orig_location[0]) + g_off[0];
            st(e->location[0], loc0);
            if (!skip[1]) {
                st(e->used_affine_gap_scoring[1], 1);
                limit = limit - s0;
                if (!gl1) limit = limit > s1 ? limit : s1;
                int cb = ld(e->bases_clipped_before[1]), ca = ld(e->bases_clipped_after[1]), span = 0;
                if (sp != nullptr && PL::spec_ld(sp->lim[1]) == PL::i32(limit)) {
                    s1 = PL::spec_ld(sp->score[1]); mp1 = PL::spec_ld(sp->mp[1]); g_off[1] = PL::spec_ld(sp->g_off[1]); cb = PL::spec_ld(sp->cb[1]);
                    ca = PL::spec_ld(sp->ca[1]); ag1 = PL::spec_ld(sp->ag[1]); span = PL::spec_ld(sp->span[1]); take_spec_calls(sp, 1);
                    S()->cnt.ag += PL::spec_ld(sp->n_ag[1]); spec_used++;
                } else {
                    score_ag(1, ld(e->direction[1]), ld(e->orig_location[1]), ld(e->seed_offset[1]), PL::i32(limit), &s1, &mp1, &g_off[1], &cb, &ca, &ag1, &span);
                }
                s1 = PL::i32(s1); mp1 = PL::f64(mp1);
                if (PL::lane0()) { e->score[1] = s1; e->match_probability[1] = mp1; e->bases_clipped_before[1] = cb; e->bases_clipped_after[1] = ca; e->ag_score[1] = ag1; e->ref_span[1] = span; }
                PL::sync();
            }
            if (s1 != -1 && s1 <= PE_MAXK1) {
                loc1 = ld(e->orig_location[1]) + g_off[1];
                st(e->location[1], loc1);
                const double pair_p = mp0 * mp1;
                const int pair_score = s0 + s1, pair_ag = ag0 + ag1;
                if (res.location[0] == loc0 && res.location[1] == loc1) return;                        // same alignment again: do not lower MAPQ
                set_sub_all(A, old_p);
                set_update_from(A, pair_score, pair_ag, pair_p, e);
                if (non_alt_aln) { set_sub_all(N, old_p); set_update_from(N, pair_score, pair_ag, pair_p, e); }
                limit = score_limit(non_alt_aln, 0);
            }
        }
    }

    // ------------------------------------------------------------------ IntersectingPairedEndAligner::align (:169-251)
    PE_FN void intersecting_align() {
        stale = 0;
        phases123(false);
        if (overflow) return;
        if (!cfg.use_ag) return;
        if (cfg.use_soft_clip && (S()->res.status[0] == SNAPGPU_NotFound || S()->res.status[1] == SNAPGPU_NotFound)) {
            phases123(true);
            if (overflow) return;
        }
        phase4();
    }

    // ------------------------------------------------------------------ ChimericPairedEndAligner::align (ChimericPairedEndAligner.cpp:126-448)
    // max_k_paired = maxKPairedEnd, max_k_single = maxKSingleEnd = maxK / 2 (:81).  Results in S()->res / S()->alt.
    PE_FN void read_not_aligned(snapgpu_paired_result &res, snapgpu_paired_result &alt, int r, bool touch_pair_flag) {      // :281-296 / :392-406
        res.status[r] = SNAPGPU_NotFound; res.mapq[r] = 0; res.direction[r] = 0; res.location[r] = SNAPGPU_InvalidGenomeLocation32;
        res.score[r] = 0; res.used_affine_gap_scoring[r] = 0; res.bases_clipped_before[r] = 0; res.bases_clipped_after[r] = 0;
        res.ag_score[r] = 0; res.clipping_for_read_adjustment[r] = 0;
        if (touch_pair_flag) res.aligned_as_pair = 0;
        alt.status[r] = SNAPGPU_NotFound;
    }

    PE_FN void align_pair(int max_k_paired, int max_k_single) {
        overflow = 0; stale = 0; stale_later = 0; ag_obj_used0 = ag_obj_used1 = 0;
        n_sec = 0; n_ssec[0] = n_ssec[1] = 0; ref_dep = 0;                                                              // :151-153
        const uint64_t t_all = PL::clock();
        align_pair_inner(max_k_paired, max_k_single);
        S()->cnt.cyc_total += PL::clock() - t_all;
        S()->res.reserved = stale;                           // not in the reference: see snapgpu_paired_result.reserved
    }

    PE_FN void align_pair_inner(int max_k_paired, int max_k_single) {
        snapgpu_paired_result &res = S()->res, &alt = S()->alt;
        res.status[0] = res.status[1] = SNAPGPU_NotFound;
        for (int r = 0; r < 2; r++) {
            res.used_affine_gap_scoring[r] = 0; res.bases_clipped_before[r] = 0; res.bases_clipped_after[r] = 0;
            res.clipping_for_read_adjustment[r] = 0; res.ag_score[r] = 0; res.liftover[r] = 0;
        }
        res.ag_forced_single_aligner_call = 0;
        alt.status[0] = alt.status[1] = SNAPGPU_NotFound;

        const int min_len = (int)cfg.min_read_length;
        if (read_len[0] < min_len && read_len[1] < min_len) {                                                           // :169-181
            for (int r = 0; r < 2; r++) { res.location[r] = SNAPGPU_InvalidGenomeLocation32; res.mapq[r] = 0; res.score[r] = 0; res.status[r] = SNAPGPU_NotFound; }
            res.aligned_as_pair = 0;
            return;
        }

        int pair_ag = 0, sum_pair_score = 0;
        bool compare_single = false;
        if (read_len[0] >= min_len && read_len[1] >= min_len) {
            (void)max_k_paired;                               // cfg.max_k is the paired limit for this call (:247 maxK = maxK_)
            intersecting_align();
            if (overflow) return;
            res.aligned_as_pair = 1;
            if (cfg.force_spacing) {                                                                                    // :208-215
                if (res.status[0] == SNAPGPU_NotFound) res.aligned_as_pair = 0;
                return;
            }
            const int max_score = res.score[0] > res.score[1] ? res.score[0] : res.score[1];
            sum_pair_score = res.score[0] + res.score[1];
            const int sum_pair_score_alt = alt.score[0] + alt.score[1];
            res.mapq[0] = res.mapq[0] <= cfg.flatten_mapq ? 0 : res.mapq[0];
            res.mapq[1] = res.mapq[1] <= cfg.flatten_mapq ? 0 : res.mapq[1];
            const bool better_alt = alt.status[0] != SNAPGPU_NotFound && alt.status[1] != SNAPGPU_NotFound &&
                                    sum_pair_score_alt <= sum_pair_score - cfg.min_score_gap_realign_alt;
            const bool lifted = res.liftover[0] && res.liftover[1];
            if ((res.used_affine_gap_scoring[0] || res.used_affine_gap_scoring[1]) && max_score >= cfg.min_score_realign && !better_alt && !lifted) {
                compare_single = true;
            }
            if (res.status[0] != SNAPGPU_NotFound && res.status[1] != SNAPGPU_NotFound && !compare_single) return;     // not chimeric
        }

        int limit_left = max_k_single;
        if (compare_single) {
            limit_left = sum_pair_score;
            if (res.status[0] != SNAPGPU_NotFound && res.status[1] != SNAPGPU_NotFound) res.ag_forced_single_aligner_call = 1;
        }

        snapgpu_single_result *single = S()->single, *single_alt = S()->single_alt;
        for (int r = 0; r < 2; r++) { single[r].status = SNAPGPU_NotFound; single[r].mapq = 0; single[r].score = 0; single[r].ag_score = 0; }
        int single_ag = 0;
        bool choose_single_mapq = true;
        for (int r = 0; r < 2; r++) {
            if (compare_single) pair_ag += res.ag_score[r];
            int max_k_read = max_k_single;
            if (read_len[r] < min_len) {
                read_not_aligned(res, alt, r, true);
                choose_single_mapq = false;
            } else {
                if (compare_single) {
                    if (limit_left < 0) break;
                    int a = res.score[r] < limit_left ? res.score[r] : limit_left;
                    max_k_read = max_k_single < a ? max_k_single : a;
                }
                const uint64_t t_s = PL::clock();
                // single-end secondary results land behind read 0's (ChimericPairedEndAligner.cpp:308-312: "it's either 0 or all we've seen")
                const uint32_t sec_base = n_ssec[0];
                G(snapgpu_single_result) *sec_dst = nullptr; uint32_t sec_room = 0;
                if (want_sec() && ssec_out != nullptr && sec_base < ssec_stride) { sec_dst = ssec_out + sec_base; sec_room = ssec_stride - sec_base; }
                const uint32_t room32 = sec_base < 32u ? 32u - sec_base : 0u;       // what PairedAligner.cpp:566's initial buffer would have left
                uint32_t n_this = pl->align_single(r, PL::i32(max_k_read), false, single[r], single_alt[r], want_sec(), sec_dst, sec_room, room32);
                PT2_OFF(S()->cnt.cyc_single += PL::clock() - t_s);
                stale += single[r].reserved & 0x3fffffffu; stale_later += (single[r].reserved & 0x40000000u) ? 1u : 0u;
                bool used_hamming = false;
                if (cfg.use_soft_clip && cfg.enable_hamming_base) {
                    if (single[r].status == SNAPGPU_NotFound && res.status[r] == SNAPGPU_NotFound) {                  // :330-360
                        used_hamming = true;
                        n_this = pl->align_single(r, PL::i32(max_k_read), true, single[r], single_alt[r], want_sec(), sec_dst, sec_room, room32);
                        // the reference drops this call's "buffer too small" on the floor (:339-343): see SNAPGPU_PAIR_REF_BUFFER_DEPENDENT
                        if (want_sec() && pl->single_raw_secondary() > room32) ref_dep = 1;
                        stale += single[r].reserved & 0x3fffffffu; stale_later += (single[r].reserved & 0x40000000u) ? 1u : 0u;
                        if (single[r].reserved & 0x80000000u) { overflow = 1; return; }      // candidate buffer of the single-end aligner overflowed
                    }
                }
                n_ssec[r] = n_this;                                                                                   // :365
                if (compare_single) {
                    if (!used_hamming) {
                        if (single[r].score != -1 && single[r].score != SNAPGPU_UnusedScoreValue) limit_left -= single[r].score;
                        else limit_left = -1;
                    }
                    single_ag += single[r].ag_score;
                    if (res.ag_score[r] >= single[r].ag_score) choose_single_mapq = false;
                }
            }
        }

        if (choose_single_mapq) {
            for (int r = 0; r < 2; r++) {
                res.mapq[r] = res.mapq[r] < single[r].mapq ? res.mapq[r] : single[r].mapq;
                if (res.mapq[r] <= cfg.flatten_mapq) res.mapq[r] = 0;
            }
        }

        if (!compare_single || single_ag >= pair_ag + cfg.min_ag_improve) {
            for (int r = 0; r < 2; r++) {
                if (read_len[r] < min_len) {
                    read_not_aligned(res, alt, r, false);
                } else {
                    res.status[r] = single[r].status;
                    res.mapq[r] = single[r].mapq / 3;                                    // heavy penalty for chimeric reads
                    res.mapq[r] = res.mapq[r] <= 3 ? 0 : res.mapq[r];
                    res.direction[r] = single[r].direction;
                    res.location[r] = single[r].location;
                    res.score[r] = single[r].score;
                    res.score_prior_to_clipping[r] = single[r].score_prior_to_clipping;
                    res.used_affine_gap_scoring[r] = single[r].used_affine_gap_scoring;
                    res.bases_clipped_before[r] = single[r].bases_clipped_before;
                    res.bases_clipped_after[r] = single[r].bases_clipped_after;
                    res.ag_score[r] = single[r].ag_score;
                }
            }
            res.aligned_as_pair = 0;
        }
    }
};
