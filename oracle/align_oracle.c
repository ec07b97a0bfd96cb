#include <stdio.h>
#include <stdlib.h>
/*
 * align_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded restatement of BaseAligner::AlignRead (SNAPLib/BaseAligner.cpp:273-763) and what it calls:
 * the seed loop, the candidate table (findElement / findCandidate / allocateNewCandidate / incrementWeight, :1811-1973,
 * :2332-2379), score() (:918-1534), ScoreSet::updateBestScore / fillInSingleAlignmentResult (:2143-2323), scoreLimit
 * (:2556-2570), Genome::getSubstring (Genome.h:339-367) and, with -om, the secondary-result recording and
 * finalizeSecondaryResults (:2423-2553).  Lookup, Landau-Vishkin, affine gap and MAPQ are the primitives of snap_oracle.c.
 *
 * Written from the reference's text, scalar and obvious rather than fast; independent of the device code in snap_amd/csrc
 * (which it is a checker for).  Pinned by tests/test_oracle.py against what the compiled reference answered
 * (tests/golden/tiny_reads.npz, secondary_reads.npz).
 *
 * Nothing under snap_amd/ may include, link or call this file; only tests/, bench.py's cpu_baseline leg and
 * __graft_entry__.smoke() may, and only as the checker.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/snapgpu.h"
#include "snap_oracle.h"

#define BUCKET 48                       /* hashTableElementSize = maxMergeDist, BaseAligner.h:177, 213 */
#define MAXK   SNAPGPU_MAX_K

/* oracle_genome (what Genome holds for this path): snap_oracle.h */

typedef struct {                        /* HashTableElement, BaseAligner.h:223-258 */
    uint64_t used, scored;
    int64_t  base, best_loc;
    double   match_prob;
    uint32_t weight, lps, best_score;
    int      ag_score, clip_before, clip_after, seed_offset, used_ag;
    int      wnext, wprev, hnext;       /* indices; weight-list sentinels are negative: -(w + 1) */
    int      dir, all_scored;
    int      cand_seed_offset[BUCKET];
} Elem;

typedef struct {                        /* ScoreSet, BaseAligner.h:260-329 */
    int      best_score;
    int64_t  best_loc, best_orig_loc;
    int      dir, used_ag, clip_before, clip_after, ag_score, seed_offset;
    double   best_match_prob, p_all, p_best;
} ScoreSet;

typedef struct {
    const oracle_index *ix; const oracle_genome *g; const snapgpu_params *p;
    int max_k;
    const char *rd[2], *ql[2];          /* read / qualities, forward and reverse complement */
    int read_len;
    Elem *pool; int n_used, pool_cap;
    int *heads; int ht_size;            /* (bucket, direction) -> element index + 1 */
    int *wl_next, *wl_prev; int n_wl;   /* weight-list sentinels */
    int highest_wl;
    uint32_t lps_unseen[2], cur_round_lps[2];
    uint32_t popular_skipped;
    ScoreSet all, non_alt;
    snapgpu_single_result *primary, *first_alt;
    /* secondary results */
    int om, mpc; int64_t omax;
    snapgpu_single_result *sec; uint32_t n_sec, sec_cap; int sec_overflow;
    /* useHamming: gapless scoring, candidates kept for alignAffineGap (only the paired-end fallback asks for it) */
    int ham;
    snapgpu_single_result *agc; uint32_t n_agc, agc_cap; int keep_agc;
    uint32_t stale;
} A;

static void set_init(ScoreSet *s) {                                     /* ScoreSet::init, :2099-2114 */
    s->best_score = SNAPGPU_UnusedScoreValue; s->best_loc = SNAPGPU_InvalidGenomeLocation32; s->best_orig_loc = SNAPGPU_InvalidGenomeLocation32;
    s->dir = 0; s->used_ag = 0; s->clip_before = 0; s->clip_after = 0; s->ag_score = -1; s->seed_offset = 0; s->best_match_prob = 0.0;
    s->p_all = 0.0; s->p_best = 0.0;
}

static int is_alt(const A *a, int64_t loc) { return loc >= 0 && (uint64_t)loc >= a->g->first_alt_location; }   /* Genome.h:436 */

static int score_limit(const A *a, int for_alt) {                       /* :2556-2570 */
    int64_t inner;
    const int64_t gap = a->p->max_score_gap_to_prefer_non_alt;
    if (for_alt) {
        int64_t m = gap < a->non_alt.best_score ? gap : a->non_alt.best_score;
        int64_t b = (int64_t)a->non_alt.best_score - m;
        inner = a->all.best_score < b ? a->all.best_score : b;
    } else {
        int64_t x = (int64_t)a->all.best_score + gap;
        inner = x < a->non_alt.best_score ? x : a->non_alt.best_score;
    }
    int64_t m = (int64_t)a->max_k < inner ? (int64_t)a->max_k : inner;
    int64_t v = (int64_t)a->p->extra_search_depth + m;
    return (int)(v < MAXK - 1 ? v : MAXK - 1);
}

/* -f / -x (BaseAligner::setStopOnFirstHit / setExplorePopularSeeds, SingleAligner.cpp:179-180): set for the calls that follow, like
 * oracle/ref_driver.cpp's snapref_set_aligner_flags.  Single-end AlignRead only (the chimeric fallback's aligner never gets them). */
static int g_stop_on_first_hit = 0, g_explore_popular_seeds = 0;
static __thread int t_flags_apply = 0;                                  /* 1 inside oracle_align_read (the single-end AlignRead proper) */
void oracle_set_aligner_flags(int stop_on_first_hit, int explore_popular_seeds) { g_stop_on_first_hit = stop_on_first_hit; g_explore_popular_seeds = explore_popular_seeds; }

static int contig_at(const oracle_genome *g, int64_t loc) {             /* Genome::getContigAtLocation, Genome.cpp:574 */
    int lo = 0, hi = (int)g->n_contigs - 1, found = -1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        if ((int64_t)g->contig_begin[mid] <= loc) { found = mid; lo = mid + 1; } else hi = mid - 1;
    }
    return found;
}

static int substring_ok(const oracle_genome *g, int64_t loc, int64_t len) {   /* Genome::getSubstring != NULL, Genome.h:339-367 */
    const int64_t nb = (int64_t)g->n_bases;
    if (loc > nb || loc + len > nb + 1000) return 0;
    if (loc < -(int64_t)g->genome_pad) return 0;
    if (len <= (int64_t)g->chromosome_padding && g->genome[loc] != 'n') return 1;
    if (len == 0) return 1;
    int c = contig_at(g, loc);
    if (c < 0) return 0;
    int64_t cend = c == (int)g->n_contigs - 1 ? nb : (int64_t)g->contig_begin[c + 1];
    return cend > loc + len;
}

/* ---- weight lists: doubly linked, FIFO within a weight (:1954-1957, :2366-2378) */
static int *nextp(A *a, int i) { return i < 0 ? &a->wl_next[-i - 1] : &a->pool[i].wnext; }
static int *prevp(A *a, int i) { return i < 0 ? &a->wl_prev[-i - 1] : &a->pool[i].wprev; }
static void list_unlink(A *a, int e) { int n = *nextp(a, e), p = *prevp(a, e); *prevp(a, n) = p; *nextp(a, p) = n; }
static void list_push_tail(A *a, int w, int e) {
    int s = -(w + 1), tail = *prevp(a, s);
    a->pool[e].wnext = s; a->pool[e].wprev = tail;
    *prevp(a, s) = e; *nextp(a, tail) = e;
}

static int head_slot(const A *a, int64_t base, int dir) {
    uint64_t k = ((uint64_t)base / BUCKET) * 2 + (uint64_t)dir;
    k *= 0x9E3779B97F4A7C15ull;
    return (int)((k >> 40) & (uint64_t)(a->ht_size - 1));
}

static int find_element(const A *a, int64_t loc, int dir) {             /* findElement, :1811-1840 */
    int64_t base = loc - (int64_t)((uint64_t)loc % BUCKET);
    for (int h = a->heads[head_slot(a, base, dir)]; h != 0; h = a->pool[h - 1].hnext)
        if (a->pool[h - 1].base == base && a->pool[h - 1].dir == dir) return h - 1;
    return -1;
}

static void increment_weight(A *a, int ei) {                            /* :2342-2379 */
    Elem *e = &a->pool[ei];
    if (e->all_scored) return;
    if (e->weight >= (uint32_t)a->n_wl - 1) return;
    list_unlink(a, ei);
    e->weight++;
    if ((int)e->weight > a->highest_wl) a->highest_wl = (int)e->weight;
    list_push_tail(a, (int)e->weight, ei);
}

static void allocate_new_candidate(A *a, int64_t loc, int dir, uint32_t lps, int seed_offset) {   /* :1885-1973 */
    int low = (int)((uint64_t)loc % BUCKET);
    if (a->n_used >= a->pool_cap) return;                               /* (sized so that it cannot happen: maxHits * seeds * 2) */
    int ei = a->n_used++;
    Elem *e = &a->pool[ei];
    memset(e, 0, sizeof(*e));
    e->used = 1ull << low; e->lps = lps; e->dir = dir; e->weight = 1; e->base = loc - low;
    e->best_score = SNAPGPU_UnusedScoreValue;
    e->cand_seed_offset[low] = seed_offset;
    int hs = head_slot(a, e->base, dir);
    e->hnext = a->heads[hs];
    a->heads[hs] = ei + 1;
    list_push_tail(a, 1, ei);
    if (a->highest_wl < 1) a->highest_wl = 1;
}

static void apply_hit(A *a, uint32_t hit, uint32_t offset, int dir) {   /* body of the loop at :629-667 */
    int64_t loc = (int64_t)(uint32_t)(hit - offset);                    /* 32-bit GenomeLocation arithmetic */
    int ei = find_element(a, loc, dir);
    if (ei >= 0) {                                                      /* findCandidate, :1844-1880, then :648-652 */
        Elem *e = &a->pool[ei];
        int low = (int)((uint64_t)loc % BUCKET);
        uint64_t bit = 1ull << low;
        e->all_scored = e->all_scored && (e->used & bit) != 0;
        e->used |= bit;
        e->cand_seed_offset[low] = (int)offset;
        increment_weight(a, ei);
    } else {
        int cand_alt = a->p->alt_awareness && is_alt(a, loc);
        if ((int64_t)a->lps_unseen[dir] <= (int64_t)score_limit(a, cand_alt)) allocate_new_candidate(a, loc, dir, a->lps_unseen[dir], (int)offset);
    }
}

/* ---- ScoreSet::updateBestScore, :2143-2299 (the secondary-result part; no affine-gap candidate buffer here) */
static void record_secondary(A *a, int dir, int64_t loc, int64_t orig, int score, int used_ag, int cb, int ca, int ag, double mp, int so) {
    if (a->n_sec >= a->sec_cap) { a->sec_overflow = 1; return; }
    snapgpu_single_result *r = &a->sec[a->n_sec++];
    memset(r, 0, sizeof(*r));
    r->status = SNAPGPU_MultipleHits; r->direction = dir; r->location = loc; r->orig_location = orig; r->score = score;
    r->used_affine_gap_scoring = used_ag; r->bases_clipped_before = cb; r->bases_clipped_after = ca; r->ag_score = ag;
    r->match_probability = mp; r->seed_offset = so;
}

static void record_candidate(A *a, int dir, int64_t loc, int64_t orig, int score, int used_ag, int cb, int ca, int ag, double mp, int so) {
    if (!a->keep_agc) return;                                           /* no buffer: nothing is kept (:2202 NULL != candidatesForAffineGap) */
    if (a->n_agc >= a->agc_cap) { a->agc_cap *= 2; a->agc = (snapgpu_single_result *)realloc(a->agc, sizeof(*a->agc) * a->agc_cap); }   /* (the caller doubles and re-calls) */
    snapgpu_single_result *r = &a->agc[a->n_agc++];
    memset(r, 0, sizeof(*r));
    r->status = SNAPGPU_MultipleHits; r->direction = dir; r->location = loc; r->orig_location = orig; r->score = score;
    r->used_affine_gap_scoring = used_ag; r->bases_clipped_before = cb; r->bases_clipped_after = ca; r->ag_score = ag;
    r->match_probability = mp; r->seed_offset = so;
}

static void update_best(A *a, ScoreSet *s, int64_t loc, int64_t orig, uint32_t score, int ag, double mp, int e_dir, int used_ag,
                        int cb, int ca, int so) {
    int seen_new;
    if (a->p->use_affine_gap) seen_new = ag > s->ag_score || (s->ag_score == ag && mp > s->p_best);
    else seen_new = score < (uint32_t)s->best_score || (score == (uint32_t)s->best_score && mp > s->p_best);
    const uint32_t best = (uint32_t)s->best_score;
    if (a->om >= 0) {
        if (seen_new) {
            if (best >= score && (int)(best - score) <= a->om)                                                     /* :2176 */
                record_secondary(a, s->dir, s->best_loc, s->best_orig_loc, s->best_score, s->used_ag, s->clip_before, s->clip_after,
                                 s->ag_score, s->best_match_prob, s->seed_offset);
        } else if ((int)(best - score) <= a->om && score != (uint32_t)SNAPGPU_ScoreAboveLimit && best >= score) {   /* :2247 */
            record_secondary(a, e_dir, loc, orig, (int)score, used_ag, cb, ca, ag, mp, so);
        }
    }
    if (a->ham) {                                                                                                   /* :2202-2228, :2273-2297 */
        if (seen_new) {
            if (best >= score && (int)(best - score) <= (int)a->p->extra_search_depth)
                record_candidate(a, s->dir, s->best_loc, s->best_orig_loc, s->best_score, s->used_ag, s->clip_before, s->clip_after,
                                 s->ag_score, s->best_match_prob, s->seed_offset);
        } else if ((int)(best - score) <= (int)a->p->extra_search_depth && score != (uint32_t)SNAPGPU_ScoreAboveLimit && best >= score) {
            record_candidate(a, e_dir, loc, orig, (int)score, used_ag, cb, ca, ag, mp, so);
        }
    }
    if (seen_new) {
        s->best_score = (int)score; s->ag_score = ag; s->p_best = mp; s->best_loc = loc; s->best_orig_loc = orig; s->dir = e_dir;
        s->used_ag = used_ag; s->clip_before = cb; s->clip_after = ca; s->seed_offset = so; s->best_match_prob = mp;
    }
}

static void fill_result(const A *a, const ScoreSet *s, snapgpu_single_result *r) {   /* fillInSingleAlignmentResult, :2301-2323 */
    r->ag_score = s->ag_score; r->bases_clipped_after = s->clip_after; r->bases_clipped_before = s->clip_before;
    r->clipping_for_read_adjustment = 0; r->direction = s->dir; r->location = s->best_loc; r->orig_location = s->best_orig_loc;
    r->mapq = oracle_compute_mapq(s->p_all, s->p_best, s->best_score, (int)a->popular_skipped);
    r->score = s->best_score; r->used_affine_gap_scoring = s->used_ag; r->seed_offset = s->seed_offset;
    r->match_probability = s->best_match_prob; r->popular_seeds_skipped = a->popular_skipped;
    r->status = r->mapq >= 10 ? SNAPGPU_SingleHit : SNAPGPU_MultipleHits;                      /* MAPQ_LIMIT_FOR_SINGLE_HIT */
    r->probability_all_candidates = s->p_all;
}

static void reversed(const char *src, int n, char *dst) { for (int i = 0; i < n; i++) dst[i] = src[n - 1 - i]; }

/* AffineGapVectorized::computeGaplessScore (AffineGapVectorized.h:139-254): Hamming walk away from the seed; the best-scoring
 * prefix is kept, the rest of the pattern is clipped.  st = +1 / -1; T, P, Q address the first byte compared. */
static int gapless_score(const A *a, int st, const char *T, const char *P, const char *Q, int plen, int score_init, int limit,
                         int *n_edits, int *pattern_offset, double *mp, int *n_gapless) {
    *mp = 1.0;
    if (limit < 0) { *n_edits = -1; *n_gapless = -1; return -1; }
    int sc = score_init, best = score_init, best_i = 0;
    for (int i = 0; i < plen; i++) {
        sc += P[i * st] == T[i * st] ? (int)a->p->match_reward : -(int)a->p->sub_penalty;
        if (sc > best) { best = sc; best_i = i; }
    }
    if (best > score_init) {
        int ne = 0, nm = 0;
        double pr = 1.0;
        for (int i = 0; i <= best_i; i++) {
            if (P[i * st] != T[i * st]) { ne++; pr *= oracle_phred_table()[(uint8_t)Q[i * st]]; } else nm++;
        }
        pr *= oracle_perfect_table()[nm];
        const int clipped = plen - (best_i + 1);
        *pattern_offset = clipped;
        *n_gapless = ne <= limit ? ne : -1;
        *n_edits = ne + clipped;
        pr *= oracle_indel_table()[clipped];
        *mp = pr;
        return best;
    }
    *n_edits = -1; *n_gapless = -1;
    return -1;
}

/* ---- score(), :918-1534.  Returns 1 when a final answer has been written. */
static int score(A *a, int force_result) {
    const snapgpu_params *p = a->p;
    const int seed_len = (int)a->ix->seed_len, read_len = a->read_len;
    for (int d = 0; d < 2; d++) if (a->cur_round_lps[d] > a->lps_unseen[d]) a->lps_unseen[d] = a->cur_round_lps[d];   /* :995-1007 */
    oracle_ag_params agp = { (int)p->match_reward, (int)p->sub_penalty, (int)p->gap_open_penalty, (int)p->gap_extend_penalty,
                             (int)p->five_prime_end_bonus, (int)p->three_prime_end_bonus };
    int wl = a->highest_wl;
    char pbuf[1024 + 16], qbuf[1024 + 16];
    do {
        while (wl > 0 && a->wl_next[wl] == -(wl + 1)) { wl--; a->highest_wl = wl; }                                    /* :1015-1025 */
        int lim_t = score_limit(a, 1), lim_f = score_limit(a, 0);
        int lim_max = lim_t > lim_f ? lim_t : lim_f;
        uint32_t lps_min = a->lps_unseen[0] < a->lps_unseen[1] ? a->lps_unseen[0] : a->lps_unseen[1];
        if ((int64_t)lps_min > (int64_t)lim_max || force_result) {
            if (wl < (int)(p->min_weight_to_check > 1 ? p->min_weight_to_check : 1)) {                                  /* :1028-1057 */
                const ScoreSet *fin;
                a->first_alt->status = SNAPGPU_NotFound;
                if (!p->alt_awareness || a->non_alt.best_score > a->all.best_score + p->max_score_gap_to_prefer_non_alt) {
                    fin = &a->all;
                } else {
                    fin = &a->non_alt;
                    if (p->emit_alt_alignments && a->all.best_score <= a->non_alt.best_score && a->all.best_loc != a->non_alt.best_loc)
                        fill_result(a, &a->all, a->first_alt);
                }
                a->primary->score = fin->best_score;
                if ((uint32_t)fin->best_score <= (uint32_t)a->max_k || (a->ham && fin->best_score != SNAPGPU_UnusedScoreValue)) {   /* :1048 */
                    fill_result(a, fin, a->primary);
                    a->primary->supplementary = 0;
                } else {
                    a->primary->status = SNAPGPU_NotFound;
                    a->primary->mapq = 0;
                }
                return 1;
            }
            force_result = 1;
        } else if (wl == 0) {
            return 0;
        }

        int ei = a->wl_next[wl];                                                                                        /* head of the list, :1071 */
        Elem *e = &a->pool[ei];
        int limit_e = score_limit(a, p->alt_awareness && is_alt(a, e->base));                                           /* :1084 */
        if ((int64_t)e->lps <= (int64_t)limit_e) {
            uint64_t mask = e->used;                                                                                    /* snapshot, :1088 */
            while (mask) {
                int idx = __builtin_ctzll(mask);
                uint64_t bit = 1ull << idx;
                mask &= ~bit;
                if (e->scored & bit) continue;
                int any_nearby = e->scored != 0;
                e->scored |= bit;

                int64_t loc = e->base + idx;
                const int64_t orig_loc = loc, elem_loc = loc;
                const int loc_non_alt = !p->alt_awareness || !is_alt(a, loc);
                uint32_t sc = (uint32_t)SNAPGPU_ScoreAboveLimit;
                double mp = 0.0;
                const int64_t glen = (int64_t)read_len + MAXK;
                int used_ag = 0, clip_before = 0, clip_after = 0, ag_score = -1;
                const int seed_offset = e->cand_seed_offset[idx];

                if (substring_ok(a->g, loc, glen)) {
                    const char *data = (const char *)a->g->genome + loc;
                    const int tail_start = seed_offset + seed_len;
                    const char *rdd = a->rd[e->dir], *qld = a->ql[e->dir];
                    int score1 = 0, score2 = 0, ag1 = seed_len, ag2 = 0, loc_offset = 0;
                    double mp1 = 1.0, mp2 = 1.0;
                    const int text_len = read_len + MAXK - tail_start;
                    int g1 = 0, g2 = 0;                                                                                       /* score1Gapless / score2Gapless */
                    if (a->ham) {                                                                                              /* :1177-1199 */
                        if (tail_start != read_len) {
                            int po;
                            ag1 = gapless_score(a, +1, data + tail_start, rdd + tail_start, qld + tail_start, read_len - tail_start, read_len, limit_e,
                                                &score1, &po, &mp1, &g1);
                            ag1 += seed_len - read_len;
                        }
                        if (g1 != -1 && seed_offset != 0) {
                            int po = 0;
                            ag2 = gapless_score(a, -1, data + seed_offset - 1, rdd + seed_offset - 1, qld + seed_offset - 1, seed_offset, read_len,
                                                limit_e - g1, &score2, &po, &mp2, &g2);
                            ag2 -= read_len;
                            loc_offset = g2 != -1 ? po : 0;
                        }
                    }
                    /* Landau-Vishkin forward over the tail of the read (:1160), then backward over the reversed head (:1169) */
                    if (!a->ham) {
                        int net = 0, tot = 0, span = 0;
                        score1 = oracle_lv(1, data + tail_start, text_len, rdd + tail_start, qld + tail_start, read_len - tail_start, limit_e, &mp1, &net, &tot, &span);
                        ag1 = (seed_len + read_len - tail_start - score1) * (int)p->match_reward - score1 * (int)p->sub_penalty;
                        if (score1 != -1) {
                            reversed(rdd, seed_offset, pbuf); reversed(qld, seed_offset, qbuf);
                            score2 = oracle_lv(-1, data + seed_offset, seed_offset + MAXK, pbuf, qbuf, seed_offset, limit_e - score1, &mp2, &net, &tot, &span);
                            loc_offset = net;
                            ag2 = (seed_offset - score2) * (int)p->match_reward - score2 * (int)p->sub_penalty;
                        }
                    }
                    if (!a->ham && score1 != -1 && score2 != -1) {
                        const int max_k_same = (int)p->gap_open_penalty / ((int)p->sub_penalty - (int)p->gap_extend_penalty);    /* :1148 */
                        if (p->use_affine_gap && score1 + score2 > max_k_same && e->lps <= (uint32_t)a->all.best_score) {          /* :1203 */
                            score1 = 0; score2 = 0; ag1 = seed_len; ag2 = 0; used_ag = 1;
                            if (tail_start != read_len) {                                                                      /* :1208-1242 */
                                const int plen = read_len - tail_start;
                                int to = -1, po = -1, ne = -1, st = 0; double m = 1.0;
                                int s1 = oracle_ag(1, plen >= 3 * (2 * limit_e + 1), &agp, data + tail_start, text_len, rdd + tail_start, qld + tail_start,
                                                   plen, limit_e, read_len, e->dir, 0, &to, &po, &ne, &m, &st);
                                a->stale += (uint32_t)st;
                                ag1 = s1 + (seed_len - read_len); clip_after = po; score1 = ne; mp1 = m;
                            }
                            if (score1 != -1 && seed_offset != 0) {                                                            /* :1244-1281 */
                                const int lim = limit_e - score1;
                                int to = -1, po = -1, ne = -1, st = 0; double m = 1.0;
                                reversed(rdd, seed_offset, pbuf); reversed(qld, seed_offset, qbuf);
                                int s2 = oracle_ag(-1, seed_offset >= 3 * (2 * lim + 1), &agp, data + seed_offset, seed_offset + lim, pbuf, qbuf,
                                                   seed_offset, lim, read_len, e->dir, 0, &to, &po, &ne, &m, &st);
                                a->stale += (uint32_t)st;
                                ag2 = s2 - read_len; clip_before = po; score2 = ne; mp2 = m; loc_offset = to;
                            }
                        }
                    }
                    int found = a->ham ? (g1 != -1 && g2 != -1) : (score1 != -1 && score2 != -1);                              /* :1293 */
                    if (found && loc_offset != 0 && !substring_ok(a->g, loc + loc_offset, glen)) found = 0;                    /* :1295-1301 */
                    if (found) {
                        sc = (uint32_t)(score1 + score2);
                        mp = mp1 * mp2 * oracle_seed_prob(seed_len);                                                           /* :1314 */
                        loc += loc_offset;
                        ag_score = ag1 + ag2;
                    } else {
                        sc = (uint32_t)SNAPGPU_ScoreAboveLimit; ag_score = SNAPGPU_ScoreAboveLimit; mp = 0.0;
                    }
                }

                /* ---- bookkeeping after scoring one candidate, :1349-1519 */
                if (any_nearby) {
                    if (a->ham && mp <= e->match_prob) continue;                                                               /* :1362 */
                    if (e->best_score < sc || (e->best_score == sc && mp <= e->match_prob)) continue;                          /* :1366 */
                }
                e->best_loc = loc; e->used_ag = used_ag; e->clip_before = clip_before; e->clip_after = clip_after;
                e->ag_score = ag_score; e->seed_offset = seed_offset;
                if (sc != (uint32_t)SNAPGPU_ScoreAboveLimit && sc < 2) {                                                       /* nearby bucket, :1396-1435 */
                    int64_t half = (int64_t)(((uint64_t)elem_loc % BUCKET) / (BUCKET / 2));
                    int64_t nearby_loc = elem_loc + (2 * half - 1) * (BUCKET / 2);
                    int ni = find_element(a, nearby_loc, e->dir);
                    if (ni >= 0) {
                        Elem *ne = &a->pool[ni];
                        if (ne->scored != 0) {
                            int64_t d = loc > ne->best_loc ? loc - ne->best_loc : ne->best_loc - loc;
                            if (d <= BUCKET) {                                                                                /* genomeLocationIsWithin(.., maxMergeDist) */
                                if (a->ham && ne->match_prob >= mp) continue;                                                   /* :1418 */
                                if (ne->best_score < sc || (ne->best_score == sc && ne->match_prob >= mp)) continue;           /* :1421 */
                                double v = a->all.p_all - ne->match_prob; a->all.p_all = v > 0.0 ? v : 0.0;                   /* updateProbabilitiesForNearbyMatch */
                                if (loc_non_alt) { double u = a->non_alt.p_all - ne->match_prob; a->non_alt.p_all = u > 0.0 ? u : 0.0; }
                                any_nearby = 1;
                                ne->match_prob = 0;
                            }
                        }
                    }
                }
                {   /* updateProbabilitiesForNewMatch, :2137-2141 */
                    double v = a->all.p_all - e->match_prob; v = v > 0.0 ? v : 0.0; a->all.p_all = v + mp;
                    if (loc_non_alt) { double u = a->non_alt.p_all - e->match_prob; u = u > 0.0 ? u : 0.0; a->non_alt.p_all = u + mp; }
                }
                e->match_prob = mp; e->best_score = sc;
                update_best(a, &a->all, loc, orig_loc, sc, ag_score, mp, e->dir, used_ag, clip_before, clip_after, seed_offset);
                if (loc_non_alt) update_best(a, &a->non_alt, loc, orig_loc, sc, ag_score, mp, e->dir, used_ag, clip_before, clip_after, seed_offset);
                if (g_stop_on_first_hit && !(!t_flags_apply) && ((uint32_t)a->all.best_score <= (uint32_t)a->max_k)) {          /* -f, :1490-1505 */
                    fill_result(a, p->alt_awareness ? &a->non_alt : &a->all, a->primary);
                    a->primary->status = SNAPGPU_MultipleHits; a->primary->mapq = 0;
                    a->first_alt->status = SNAPGPU_NotFound;
                    return 1;
                }
                {   /* nothing can rescue MAPQ once the candidates' total probability reaches 4.9 -- unless secondary results are wanted, :1512 */
                    double chk = p->alt_awareness ? a->non_alt.p_all : a->all.p_all;
                    if (chk >= 4.9 && a->om < 0) {
                        fill_result(a, p->alt_awareness ? &a->non_alt : &a->all, a->primary);
                        a->first_alt->status = SNAPGPU_NotFound;
                        return 1;
                    }
                }
            }
        }
        e->all_scored = 1;                                                                                              /* :1526-1529 */
        list_unlink(a, ei);
        e->wnext = e->wprev = ei;
    } while (force_result);
    return 0;
}

/* ---- finalizeSecondaryResults, :2423-2553 (ignoreAlignmentAdjustmentsForOm, the default) */
static int cmp_contig_score(const void *x, const void *y, void *ctx) {
    const A *a = (const A *)ctx;
    const snapgpu_single_result *f = (const snapgpu_single_result *)x, *s = (const snapgpu_single_result *)y;
    int fc = contig_at(a->g, f->location), scn = contig_at(a->g, s->location);
    if (fc != scn) return fc < scn ? -1 : 1;
    if (f->score != s->score) return f->score < s->score ? -1 : 1;
    return 0;
}
static int cmp_score(const void *x, const void *y, void *ctx) {
    (void)ctx;
    const snapgpu_single_result *f = (const snapgpu_single_result *)x, *s = (const snapgpu_single_result *)y;
    return f->score < s->score ? -1 : f->score > s->score ? 1 : 0;
}
/* glibc's qsort is a merge sort, i.e. stable; the reference's results depend on that.  Insertion sort is stable too. */
static void stable_sort(snapgpu_single_result *v, uint32_t n, int (*cmp)(const void *, const void *, void *), void *ctx) {
    for (uint32_t i = 1; i < n; i++) {
        snapgpu_single_result t = v[i];
        uint32_t j = i;
        while (j > 0 && cmp(&v[j - 1], &t, ctx) > 0) { v[j] = v[j - 1]; j--; }
        v[j] = t;
    }
}

static void finalize_secondary(A *a) {
    uint32_t n = a->n_sec;
    int best = a->primary->score;
    int worst = best + a->om; if (worst > a->max_k) worst = a->max_k;                                                   /* :2465 */
    uint32_t i = 0;
    while (i < n) {                                                                                                     /* :2467-2485 */
        if (a->sec[i].score > worst) { a->sec[i] = a->sec[n - 1]; n--; }
        else {
            a->sec[i].score_prior_to_clipping = a->sec[i].score;
            a->sec[i].supplementary = a->p->alt_awareness && is_alt(a, a->sec[i].location);
            i++;
        }
    }
    if (a->mpc > 0 && a->primary->status != SNAPGPU_NotFound) {                                                         /* :2487-2547 */
        const int pc = contig_at(a->g, a->primary->location);
        int too_many = 0;
        for (i = 0; i < n && !too_many; i++) {
            int c = contig_at(a->g, a->sec[i].location), count = c == pc ? 1 : 0;
            for (uint32_t j = 0; j < n; j++) count += contig_at(a->g, a->sec[j].location) == c;
            if (count > a->mpc) too_many = 1;
        }
        if (too_many) {
            stable_sort(a->sec, n, cmp_contig_score, a);
            int cur = -1, cur_count = 0; uint32_t dest = 0;
            for (uint32_t src = 0; src < n; src++) {
                int c = contig_at(a->g, a->sec[src].location);
                if (c != cur) { cur = c; cur_count = c == pc ? 1 : 0; }
                cur_count++;
                if (cur_count <= a->mpc) a->sec[dest++] = a->sec[src];
            }
            n = dest;
        }
    }
    if ((int64_t)n > a->omax) { stable_sort(a->sec, n, cmp_score, a); n = (uint32_t)a->omax; }                          /* :2549-2552 */
    a->n_sec = n;
}

static char rc_base(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }

/* ---- BaseAligner::scoreLocationWithAffineGap (:766-915): the forward half asks for the clipping optimisations, the backward half
 * does not (:827 vs :857) */
static void score_location_ag(A *a, int dir, int64_t loc, int seed_offset, int limit, int *score, double *mp, int *offset,
                              int *clip_before, int *clip_after, int *ag_score) {
    const snapgpu_params *p = a->p;
    const int read_len = a->read_len, seed_len = (int)a->ix->seed_len;
    const int64_t glen = (int64_t)read_len + MAXK;
    *offset = 0;
    if (!substring_ok(a->g, loc, glen)) { *score = -1; *mp = 0; *ag_score = -1; return; }
    *clip_before = 0; *clip_after = 0;
    const char *data = (const char *)a->g->genome + loc;
    const int tail_start = seed_offset + seed_len;
    const char *rdd = a->rd[dir], *qld = a->ql[dir];
    oracle_ag_params agp = { (int)p->match_reward, (int)p->sub_penalty, (int)p->gap_open_penalty, (int)p->gap_extend_penalty,
                             (int)p->five_prime_end_bonus, (int)p->three_prime_end_bonus };
    int score1 = 0, score2 = 0, ag1 = seed_len, ag2 = 0;
    double mp1 = 1.0, mp2 = 1.0;
    char pbuf[1024 + 16], qbuf[1024 + 16];
    if (tail_start != read_len) {
        const int plen = read_len - tail_start;
        int to = -1, po = -1, ne = -1, st = 0; double m = 1.0;
        int s1 = oracle_ag(1, plen >= 3 * (2 * limit + 1), &agp, data + tail_start, (int)(glen - tail_start), rdd + tail_start, qld + tail_start,
                           plen, limit, read_len, dir, 1, &to, &po, &ne, &m, &st);
        a->stale += (uint32_t)st;
        ag1 = s1 + (seed_len - read_len); *clip_after = po; score1 = ne; mp1 = m;
    }
    if (score1 != -1 && seed_offset != 0) {
        const int lim = limit - score1;
        int to = -1, po = -1, ne = -1, st = 0; double m = 1.0;
        reversed(rdd, seed_offset, pbuf); reversed(qld, seed_offset, qbuf);
        int s2 = oracle_ag(-1, seed_offset >= 3 * (2 * lim + 1), &agp, data + seed_offset, seed_offset + lim, pbuf, qbuf, seed_offset, lim, read_len,
                           dir, 0, &to, &po, &ne, &m, &st);
        a->stale += (uint32_t)st;
        ag2 = s2 - read_len; *clip_before = po; score2 = ne; mp2 = m; *offset = to;
        if (score2 == -1) *offset = 0;
    }
    if (score1 != -1 && score2 != -1) {
        /* :907: seedLen is the unsigned member here, so pow() is libm's pow(double, double) -- not the powi of :1314 (snap_oracle.c) */
        *score = score1 + score2; *mp = mp1 * mp2 * oracle_seed_prob_pow(seed_len); *ag_score = ag1 + ag2;
    } else {
        *score = -1; *ag_score = -1; *mp = 0.0;
    }
}

/* ScoreSet::init(SingleAlignmentResult*) / updateBestScore(SingleAlignmentResult*), :2116-2130, BaseAligner.h:310-328 */
static void set_from_result(ScoreSet *s, const snapgpu_single_result *r) {
    s->best_score = r->score; s->best_loc = r->location; s->best_orig_loc = r->orig_location; s->dir = r->direction;
    s->used_ag = r->used_affine_gap_scoring; s->clip_before = r->bases_clipped_before; s->clip_after = r->bases_clipped_after;
    s->ag_score = r->ag_score; s->seed_offset = r->seed_offset; s->best_match_prob = r->match_probability;
    s->p_all = r->probability_all_candidates; s->p_best = r->match_probability;
}
static int set_update_from(ScoreSet *s, int64_t loc, int64_t orig, int dir, int score, int used_ag, int cb, int ca, int ag, int so, double mp) {
    s->p_all += mp;
    if (ag > s->ag_score || (ag == s->ag_score && mp > s->best_match_prob)) {
        s->best_score = score; s->ag_score = ag; s->best_match_prob = mp; s->best_loc = loc; s->best_orig_loc = orig; s->dir = dir;
        s->used_ag = used_ag; s->clip_before = cb; s->clip_after = ca; s->seed_offset = so;
        return 1;
    }
    return 0;
}
static void set_sub_all(ScoreSet *s, double old) { double v = s->p_all - old; s->p_all = v > 0.0 ? v : 0.0; }

static int cmp_agc(const void *x, const void *y, void *ctx) {
    (void)ctx;
    const snapgpu_single_result *f = (const snapgpu_single_result *)x, *s = (const snapgpu_single_result *)y;
    return f->score < s->score ? -1 : f->score > s->score ? 1 : 0;
}
static void stable_sort(snapgpu_single_result *v, uint32_t n, int (*cmp)(const void *, const void *, void *), void *ctx);

/* ---- BaseAligner::alignAffineGap (:1537-1792): affine-gap rescoring of the result of a Hamming pass and of the candidates it kept */
static void align_affine_gap(A *a) {
    const snapgpu_params *p = a->p;
    snapgpu_single_result *primary = a->primary, *first_alt = a->first_alt;
    if (primary->status == SNAPGPU_NotFound) return;
    int n_count = 0;
    for (int i = 0; i < a->read_len; i++) n_count += a->rd[0][i] == 'N';
    if (n_count > a->max_k) return;
    const int best_score = primary->score;
    int limit = MAXK - 1, limit_alt = MAXK - 1, g_off = 0, skip = 0;
    const double old_p = primary->match_probability;
    const double old_p_alt = first_alt->status != SNAPGPU_NotFound ? first_alt->match_probability : 0.0;
    const int max_k_same = (int)p->gap_open_penalty / ((int)p->sub_penalty - (int)p->gap_extend_penalty);

    primary->used_affine_gap_scoring = 0;
    if (primary->score > max_k_same) {
        primary->used_affine_gap_scoring = 1;
        int sc, cb = primary->bases_clipped_before, ca = primary->bases_clipped_after, ag = primary->ag_score;
        double mp = primary->match_probability;
        score_location_ag(a, primary->direction, primary->orig_location, primary->seed_offset, limit, &sc, &mp, &g_off, &cb, &ca, &ag);
        primary->score = sc; primary->match_probability = mp; primary->bases_clipped_before = cb; primary->bases_clipped_after = ca; primary->ag_score = ag;
        if (sc != -1) primary->location = primary->orig_location + g_off; else primary->status = SNAPGPU_NotFound;
    } else {
        skip = 1;
    }
    if (first_alt->status != SNAPGPU_NotFound && first_alt->score > max_k_same) {
        first_alt->used_affine_gap_scoring = 1;
        int sc, cb = first_alt->bases_clipped_before, ca = first_alt->bases_clipped_after, ag = first_alt->ag_score;
        double mp = first_alt->match_probability;
        score_location_ag(a, first_alt->direction, first_alt->orig_location, first_alt->seed_offset, limit_alt, &sc, &mp, &g_off, &cb, &ca, &ag);
        first_alt->score = sc; first_alt->match_probability = mp; first_alt->bases_clipped_before = cb; first_alt->bases_clipped_after = ca; first_alt->ag_score = ag;
        if (sc != -1) first_alt->location = first_alt->orig_location + g_off; else first_alt->status = SNAPGPU_NotFound;
    }
    if (primary->status == SNAPGPU_NotFound || primary->score > MAXK - 1) {
        primary->location = SNAPGPU_InvalidGenomeLocation32; primary->mapq = 0; primary->score = -1; primary->status = SNAPGPU_NotFound;
        primary->clipping_for_read_adjustment = 0; primary->used_affine_gap_scoring = 0; primary->bases_clipped_before = 0;
        primary->bases_clipped_after = 0; primary->ag_score = -1; primary->seed_offset = 0; primary->match_probability = 0.0;
        first_alt->status = SNAPGPU_NotFound;
        return;
    }

    /* the function's own score sets (:1670-1690); scoreLimit below still reads the aligner's members, as the reference does (:1756) */
    ScoreSet SA, SN;
    const int non_alt_aln = !p->alt_awareness || !is_alt(a, primary->location);
    set_from_result(&SA, primary);
    int alt_best = 0;
    if (first_alt->status != SNAPGPU_NotFound)
        alt_best = set_update_from(&SA, first_alt->location, first_alt->orig_location, first_alt->direction, first_alt->score, first_alt->used_affine_gap_scoring,
                                   first_alt->bases_clipped_before, first_alt->bases_clipped_after, first_alt->ag_score, first_alt->seed_offset,
                                   first_alt->match_probability);
    if (non_alt_aln) set_from_result(&SN, primary); else set_init(&SN);
    if (!skip) {
        const double new_p = primary->match_probability;
        if (alt_best) { set_sub_all(&SA, old_p_alt); SA.p_best = first_alt->match_probability; SA.p_all += first_alt->match_probability; }
        else          { set_sub_all(&SA, old_p); SA.p_best = new_p; SA.p_all += new_p; }
        if (non_alt_aln) { set_sub_all(&SN, old_p); SN.p_best = new_p; SN.p_all += new_p; }
    }
    if (a->n_agc > 0 && !skip) {
        limit = (int)(((uint32_t)a->max_k < (uint32_t)best_score ? (uint32_t)a->max_k : (uint32_t)best_score) + p->extra_search_depth);   /* :1714 */
        stable_sort(a->agc, a->n_agc, cmp_agc, NULL);                                                                    /* qsort(compareByScore), :1719 */
        for (uint32_t t = 0; t < a->n_agc; t++) {
            snapgpu_single_result *c = &a->agc[t];
            const int c_non_alt = !p->alt_awareness || !is_alt(a, c->location);
            const double c_old_p = c->match_probability;
            int sc, cb = c->bases_clipped_before, ca = c->bases_clipped_after, ag = c->ag_score;
            double mp = c_old_p;
            score_location_ag(a, c->direction, c->orig_location, c->seed_offset, limit, &sc, &mp, &g_off, &cb, &ca, &ag);
            if (sc != -1 && sc <= MAXK - 1) {
                const int64_t new_loc = c->orig_location + g_off;
                if (primary->location == new_loc) continue;                                                              /* same alignment again */
                set_sub_all(&SA, c_old_p);
                set_update_from(&SA, new_loc, c->orig_location, c->direction, sc, 1, cb, ca, ag, c->seed_offset, mp);
                if (c_non_alt) { set_sub_all(&SN, c_old_p); set_update_from(&SN, new_loc, c->orig_location, c->direction, sc, 1, cb, ca, ag, c->seed_offset, mp); }
                limit = score_limit(a, p->alt_awareness && !c_non_alt);
            }
        }
    }
    const int emit_all = !p->alt_awareness || SN.best_score > SA.best_score + p->max_score_gap_to_prefer_non_alt;
    const uint32_t pop = primary->popular_seeds_skipped, pop_alt = first_alt->popular_seeds_skipped, saved = a->popular_skipped;
    a->popular_skipped = pop;
    fill_result(a, emit_all ? &SA : &SN, primary);
    if (p->alt_awareness && !emit_all && SA.best_loc != SN.best_loc) {
        a->popular_skipped = pop_alt;
        fill_result(a, &SA, first_alt);
        first_alt->supplementary = 1;
    } else {
        first_alt->status = SNAPGPU_NotFound;
    }
    a->popular_skipped = saved;
}

/*
 * BaseAligner::AlignRead (:273-763) for one read.  sp == NULL: no secondary results (the reference default).
 * *n_secondary = how many the read has; the first min(that, sec_room) are stored.  *stale = affine-gap traceback steps
 * through cells this call never wrote (the reference's own answer for such a read depends on its history).
 * max_k: BaseAligner::maxK for this call (setMaxK; p->max_k for the plain single-end path).  ctor_max_k: the maxK the object was
 * built with (it sizes nothing here, but documents the chimeric fallback's maxK / 2).  hamming: AlignRead(..., useHamming = true)
 * followed by alignAffineGap on the candidates it kept, as ChimericPairedEndAligner.cpp:339-360 does.
 * Returns 0, or -1 for a read longer than 1000 bases / an invalid option set.
 */
static int align_read_ex(const oracle_index *ix, const oracle_genome *g, const snapgpu_params *p, int max_k, int hamming,
                         const char *bases, const char *quals, int len,
                         snapgpu_single_result *primary, snapgpu_single_result *first_alt,
                         const snapgpu_secondary_params *sp, snapgpu_single_result *secondary, uint32_t sec_room, uint32_t *n_secondary,
                         uint32_t *stale, uint32_t *raw_secondary);

int oracle_align_read(const oracle_index *ix, const oracle_genome *g, const snapgpu_params *p, const char *bases, const char *quals, int len,
                      snapgpu_single_result *primary, snapgpu_single_result *first_alt,
                      const snapgpu_secondary_params *sp, snapgpu_single_result *secondary, uint32_t sec_room, uint32_t *n_secondary,
                      uint32_t *stale)
{
    /* a NEWLY CONSTRUCTED BaseAligner for this read: its affineGap / reverseAffineGap traceback arrays start zero-filled and persist
       over the calls of the read (unless the caller bound the arrays of a longer-lived aligner itself) */
    uint8_t *f0, *b0; size_t c0;
    oracle_ag_bound_objects(&f0, &b0, &c0);
    uint8_t *mine = NULL;
    if (!f0) {
        const size_t cap = ((size_t)len + 128) * ((size_t)len + 264);
        mine = calloc(2 * cap, 1);
        oracle_ag_bind_objects(mine, mine + cap, cap);
    }
    t_flags_apply = 1;
    int rc = align_read_ex(ix, g, p, (int)p->max_k, 0, bases, quals, len, primary, first_alt, sp, secondary, sec_room, n_secondary, stale, NULL);
    t_flags_apply = 0;
    if (mine) { oracle_ag_bind_objects(NULL, NULL, 0); free(mine); }
    return rc;
}

/* the single-end aligner inside ChimericPairedEndAligner: `p` holds its constructor arguments (maxK / 2, maxSeedsSingleEnd) */
int oracle_align_read_chimeric(const oracle_index *ix, const oracle_genome *g, const snapgpu_params *p, int max_k, int hamming,
                               const char *bases, const char *quals, int len, snapgpu_single_result *primary, snapgpu_single_result *first_alt,
                               const snapgpu_secondary_params *sp, snapgpu_single_result *secondary, uint32_t sec_room, uint32_t *n_secondary,
                               uint32_t *stale, uint32_t *raw_secondary)
{
    return align_read_ex(ix, g, p, max_k, hamming, bases, quals, len, primary, first_alt, sp, secondary, sec_room, n_secondary, stale, raw_secondary);
}

static int align_read_ex(const oracle_index *ix, const oracle_genome *g, const snapgpu_params *p, int max_k, int hamming,
                         const char *bases, const char *quals, int len,
                         snapgpu_single_result *primary, snapgpu_single_result *first_alt,
                         const snapgpu_secondary_params *sp, snapgpu_single_result *secondary, uint32_t sec_room, uint32_t *n_secondary,
                         uint32_t *stale, uint32_t *raw_secondary)
{
    oracle_init();
    if (len > 1000 || len < 0) return -1;
    A a;
    memset(&a, 0, sizeof(a));
    a.ix = ix; a.g = g; a.p = p; a.max_k = max_k; a.read_len = len; a.primary = primary; a.first_alt = first_alt;
    a.ham = hamming; a.keep_agc = hamming && p->use_affine_gap;        /* no candidate buffer without affine gap (PairedAligner.cpp:570-577) */
    if (raw_secondary) *raw_secondary = 0;
    a.om = sp ? sp->max_edit_distance : -1; a.mpc = sp ? sp->max_per_contig : -1; a.omax = sp ? sp->max_results : 0x7fffffff;
    if (n_secondary) *n_secondary = 0;
    if (stale) *stale = 0;

    memset(primary, 0, sizeof(*primary));                                                                               /* :334-344 */
    primary->status = SNAPGPU_NotFound; primary->location = SNAPGPU_InvalidGenomeLocation32; primary->score = SNAPGPU_UnusedScoreValue;
    *first_alt = *primary; first_alt->location = 0; first_alt->score = 0;

    const int seed_len = (int)ix->seed_len;
    if (len < seed_len) return 0;                                                                                       /* :360 */
    char fwd_q[1024], rc[1024], rc_q[1024];
    int n_count = 0;
    for (int i = 0; i < len; i++) {                                                                                     /* :388-396 */
        n_count += bases[i] == 'N';
        rc[len - 1 - i] = rc_base(bases[i]); rc_q[len - 1 - i] = quals[i]; fwd_q[i] = quals[i];
    }
    if (n_count > a.max_k) return 0;                                                                                    /* :398 */
    a.rd[0] = bases; a.rd[1] = rc; a.ql[0] = fwd_q; a.ql[1] = rc_q;

    uint8_t used[1024];
    memset(used, 0, sizeof(used));
    if (n_count > 0) {                                                                                                  /* :407-420 */
        int min_seed = 0;
        for (int i = 0; i < len; i++) {
            if (bases[i] != 'A' && bases[i] != 'C' && bases[i] != 'G' && bases[i] != 'T') {
                int limit = i + seed_len - 1 < len - 1 ? i + seed_len - 1 : len - 1;
                int j0 = i - seed_len + 1; if (j0 < min_seed) j0 = min_seed;
                for (int j = j0; j <= limit; j++) used[j] = 1;
                min_seed = limit + 1;
                if (min_seed >= len) break;
            }
        }
    }

    uint32_t max_seeds = p->num_seeds != 0 ? p->num_seeds : (uint32_t)(int)(2 * p->seed_coverage * len / seed_len);     /* :327-332 */
    uint32_t ctor_seeds = p->num_seeds != 0 ? p->num_seeds : (uint32_t)(int)(p->seed_coverage * 1000 / seed_len);       /* the constructor's, :173-180 */
    a.n_wl = (int)ctor_seeds + 1;
    if (a.n_wl < 2) a.n_wl = 2;
    a.pool_cap = (int)((uint64_t)p->max_hits * (ctor_seeds + 1) * 2 + 64);
    a.pool = (Elem *)malloc(sizeof(Elem) * (size_t)a.pool_cap);
    a.ht_size = 1; while (a.ht_size < 2 * a.pool_cap) a.ht_size <<= 1;
    a.heads = (int *)calloc((size_t)a.ht_size, sizeof(int));
    a.wl_next = (int *)malloc(sizeof(int) * (size_t)a.n_wl); a.wl_prev = (int *)malloc(sizeof(int) * (size_t)a.n_wl);
    for (int w = 0; w < a.n_wl; w++) a.wl_next[w] = a.wl_prev[w] = -(w + 1);                                            /* clearCandidates, :2332-2339 */
    if (sp) { a.sec_cap = (uint32_t)(2 * (uint64_t)(max_seeds + 1) * p->max_hits + 2); a.sec = (snapgpu_single_result *)malloc(sizeof(*a.sec) * a.sec_cap); }
    if (a.keep_agc) { a.agc_cap = 4096; a.agc = (snapgpu_single_result *)malloc(sizeof(*a.agc) * a.agc_cap); }

    const uint32_t n_possible = (uint32_t)(len - seed_len + 1);
    uint32_t next_seed = 0, wrap_count = 0, n_applied[2] = {0, 0};
    set_init(&a.all); set_init(&a.non_alt);
    if (!p->alt_awareness) a.non_alt.best_score = SNAPGPU_TooBigScoreValue;       /* :325: only bestScore is reset without ALT awareness (a fresh aligner) */
    int finished = 0;

    while (n_applied[0] + n_applied[1] < max_seeds) {                                                                   /* :451 */
        if (next_seed >= n_possible) {                                                                                  /* wrapping, :455-504 */
            wrap_count++;
            if (wrap_count >= (uint32_t)seed_len) { score(&a, 1); finished = 1; break; }
            next_seed = oracle_wrapped_next_seed((unsigned)seed_len, wrap_count);
            a.cur_round_lps[0] = a.cur_round_lps[1] = 0;
        }
        while (next_seed < n_possible && used[next_seed]) next_seed++;                                                  /* :506-512 */
        if (next_seed >= n_possible) continue;
        used[next_seed] = 1;
        uint64_t sb, srcv;
        if (!oracle_pack_seed(bases + next_seed, (unsigned)seed_len, &sb, &srcv)) continue;                             /* :522-526 */
        int64_t n_hits[2]; const uint32_t *hits[2]; uint32_t single_v[2], slots[2];
        oracle_lookup_seed(ix, sb, srcv, n_hits, hits, single_v, slots);
        int applied_either = 0;
        for (int dir = 0; dir < 2; dir++) {
            if (n_hits[dir] > (int64_t)p->max_hits && !(g_explore_popular_seeds && !(!t_flags_apply))) {             /* too popular, :574-579 (-x: not skipped) */
                a.popular_skipped++;
            } else {
                uint32_t offset = dir == 0 ? next_seed : (uint32_t)(len - seed_len) - next_seed;                        /* :591-606 */
                const int64_t limit_hits = n_hits[dir] < (int64_t)p->max_hits ? n_hits[dir] : (int64_t)p->max_hits;    /* :625 */
                for (int64_t i = 0; i < limit_hits; i++) apply_hit(&a, n_hits[dir] == 1 ? single_v[dir] : hits[dir][i], offset, dir);
                n_applied[dir]++; a.cur_round_lps[dir]++;
                applied_either = 1;
            }
        }
        next_seed += (uint32_t)seed_len;                                                                                /* :676 */
        if (applied_either && score(&a, 0)) { finished = 1; break; }
    }
    if (!finished) score(&a, 1);                                                                                        /* :734 */
    primary->score_prior_to_clipping = primary->score;                                                                  /* finalizeSecondaryResults, :2442 */
    int rc_ret = 0;
    if (sp) {
        if (a.sec_overflow) rc_ret = -1;
        if (raw_secondary) *raw_secondary = a.n_sec;
        finalize_secondary(&a);
        if (n_secondary) *n_secondary = a.n_sec;
        for (uint32_t k = 0; k < a.n_sec && k < sec_room; k++) secondary[k] = a.sec[k];
        free(a.sec);
    }
    if (hamming) align_affine_gap(&a);                                                                                  /* ChimericPairedEndAligner.cpp:358-360 */
    primary->reserved = a.stale;
    if (stale) *stale = a.stale;
    if (a.agc) free(a.agc);
    free(a.pool); free(a.heads); free(a.wl_next); free(a.wl_prev);
    return rc_ret;
}

/* a batch of reads, for the tests */
int oracle_align_reads(const oracle_index *ix, const oracle_genome *g, const snapgpu_params *p, uint32_t n, const char *bases, const char *quals,
                       const uint64_t *offsets, snapgpu_single_result *primary, snapgpu_single_result *first_alt,
                       const snapgpu_secondary_params *sp, snapgpu_single_result *secondary, uint32_t sec_stride, uint32_t *n_secondary)
{
    for (uint32_t i = 0; i < n; i++) {
        uint32_t ns = 0, stale = 0;
        int rc = oracle_align_read(ix, g, p, bases + offsets[i], quals + offsets[i], (int)(offsets[i + 1] - offsets[i]), &primary[i], &first_alt[i],
                                   sp, secondary ? secondary + (size_t)i * sec_stride : NULL, sec_stride, &ns, &stale);
        if (rc) return rc;
        if (n_secondary) n_secondary[i] = ns;
    }
    return 0;
}
