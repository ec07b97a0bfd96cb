/*
 * cigar_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of LandauVishkinWithCigar::computeEditDistance (SNAPLib/LandauVishkin.cpp:141-505): the Landau-Vishkin
 * variant SAMFormat::computeCigar runs once per written read to turn (read, location) into a CIGAR string -- first step of
 * SURVEY.md section 8(f) rank 1 (result -> SAM record on the device).  It is NOT the scoring LV of the hot path
 * (snap_oracle.c: oracle_lv): different diagonal order (0, -1, +1, -2, ...), a least-total-indels tie rule through the
 * totalIndels[][] array, and a "no indels if e straight mismatches explain it" shortcut.
 *
 * Pinned by tests/test_oracle.py on the reference's own known answers (tests/LandauVishkinTest.cpp:34-121) and live against
 * the compiled reference on fuzz.  Output format: the reference's COMPACT_CIGAR_STRING ("%d%c" per run).
 *
 * The reference compares 8 bytes at a time and may look at bytes before / after both strings; what it sees there can change the
 * answer when the text is shorter than the pattern (LandauVishkin.cpp:237-262).  The caller therefore says how many bytes are
 * readable around each string; anything outside compares unequal.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "snap_oracle.h"

#define MAXK 127                                    /* MAX_K, LandauVishkin.h:11 */

typedef struct { const char *p; int lo, hi; } mem_t;   /* p[i] readable for lo <= i < hi */

static int byte_eq(const mem_t *a, int i, const mem_t *b, int j) {
    if (i < a->lo || i >= a->hi || j < b->lo || j >= b->hi) return 0;
    return a->p[i] == b->p[j];
}
static int eq_run(const mem_t *pat, int pi, const mem_t *txt, int ti) {
    int n = 0;
    while (byte_eq(pat, pi + n, txt, ti + n)) n++;
    return n;
}

typedef struct { char *buf; int cap, used; uint32_t *ops; int ops_cap, n_ops; } out_t;
static int bam_code(char c) { return c == 'M' ? 0 : c == 'I' ? 1 : c == 'D' ? 2 : c == '=' ? 7 : c == 'X' ? 8 : 15; }   /* BAMAlignment::CigarToCode */
static int write_cigar(out_t *o, int count, char code) {       /* writeCigar, COMPACT_CIGAR_STRING (:77-82, :100-113) / BAM_CIGAR_OPS (:124-131) */
    if (count <= 0) return 1;
    if (o->ops) {
        if (o->n_ops >= o->ops_cap || count >= (1 << 28)) return 0;
        o->ops[o->n_ops++] = ((uint32_t)count << 4) | (uint32_t)bam_code(code);
        return 1;
    }
    if (o->cap - o->used == 0) return 0;
    char tmp[32];
    int w = snprintf(tmp, sizeof(tmp), "%d%c", count, code);
    if (w > o->cap - o->used - 1) { o->buf[o->used] = '\0'; return 0; }
    memcpy(o->buf + o->used, tmp, (size_t)w + 1);
    o->used += w;
    return 1;
}

/*
 * Returns the edit distance, -1 (ScoreAboveLimit) or -2 (cigar buffer too small).
 * text_lo / text_hi: readable range around text (text_lo <= 0, text_hi >= text_len); likewise for the pattern.
 */
static int lv_cigar_core(const char *text, int text_len, int text_lo, int text_hi, const char *pattern, int pattern_len, int pat_lo, int pat_hi,
                         int k, int use_m, out_t *outp, int *o_text_used, int *o_net_indel);

int oracle_lv_cigar(const char *text, int text_len, int text_lo, int text_hi, const char *pattern, int pattern_len, int pat_lo, int pat_hi,
                    int k, int use_m, char *cigar, int cigar_cap, int *o_text_used, int *o_net_indel)
{
    out_t out = { cigar, cigar_cap, 0, NULL, 0, 0 };
    if (cigar_cap > 0) cigar[0] = '\0';
    return lv_cigar_core(text, text_len, text_lo, text_hi, pattern, pattern_len, pat_lo, pat_hi, k, use_m, &out, o_text_used, o_net_indel);
}

static int lv_cigar_core(const char *text, int text_len, int text_lo, int text_hi, const char *pattern, int pattern_len, int pat_lo, int pat_hi,
                         int k, int use_m, out_t *outp, int *o_text_used, int *o_net_indel)
{
    static const int PrevDelta[3][3] = { {0, +1, -1}, {0, +1, -1}, {0, -1, +1} };    /* least absolute indels, :66-69 */
    const mem_t txt = { text, text_lo, text_hi }, pat = { pattern, pat_lo, pat_hi };
#define out (*outp)
    int net_indel = 0;
    if (o_net_indel) *o_net_indel = 0;
    if (text == NULL) return -1;
    if (k >= MAXK) k = MAXK - 1;

    const int W = 2 * MAXK + 1;
    int *L = (int *)malloc(sizeof(int) * (size_t)(MAXK + 1) * W);
    int *TI = (int *)calloc((size_t)(MAXK + 1) * W, sizeof(int));
    char *Act = (char *)calloc((size_t)(MAXK + 1) * W, 1);
    for (int i = 0; i < (MAXK + 1) * W; i++) L[i] = -2;                                /* the constructor, :14-22 */
#define LL(e, d) L[(e) * W + MAXK + (d)]
#define TT(e, d) TI[(e) * W + MAXK + (d)]
#define AA(e, d) Act[(e) * W + MAXK + (d)]
    int rc = -1;

    int end = pattern_len < text_len ? pattern_len : text_len;
    {   /* L[0][0]: the exact-match run, :170-186 */
        int r = eq_run(&pat, 0, &txt, 0);
        LL(0, 0) = r < end ? r : end;
    }
    if (LL(0, 0) == end) {                                                             /* :187-213 */
        int ok;
        if (use_m) ok = write_cigar(&out, pattern_len, 'M');
        else {
            ok = write_cigar(&out, end, '=');
            if (ok && pattern_len > end) ok = write_cigar(&out, pattern_len - end, 'X');
        }
        if (!ok) { rc = -2; goto done; }
        if (o_text_used) *o_text_used = end;
        rc = 0; goto done;
    }

    int e, last_best_indels = MAXK + 1, last_best_d = MAXK + 1, last_best_best = 0;
    for (e = 1; e <= k; e++) {
        for (int d = 0; d != -(e + 1); d = (d >= 0 ? -(d + 1) : -d)) {                 /* 0, -1, 1, -2, 2, ... :222 */
            int bestdelta = 0, bestbest = -1, best_best_indels = MAXK + 1;
            const int dy = (d >= 0) + (d > 0);
            for (int dx = 0; dx < 3; dx++) {
                const int delta = PrevDelta[dy][dx];
                if (d + delta < -MAXK || d + delta > MAXK) continue;                   /* (outside the arrays: never reachable cells) */
                int best = LL(e - 1, d + delta) + (delta >= 0);
                const int best_indels = TT(e - 1, d + delta) + (delta != 0);
                if (best < 0) continue;
                if (byte_eq(&pat, best, &txt, d + best)) {                             /* :239-262 */
                    const int e2 = pattern_len < text_len - d ? pattern_len : text_len - d;
                    int reach = best + eq_run(&pat, best, &txt, d + best);
                    best = reach < e2 ? reach : e2;
                }
                if (best > bestbest || (best == bestbest && best_indels < best_best_indels)) {
                    bestbest = best; bestdelta = delta; best_best_indels = best_indels;
                }
            }
            AA(e, d) = "DXI"[bestdelta + 1];
            LL(e, d) = bestbest;
            TT(e, d) = best_best_indels;
            if (bestbest == pattern_len) {                                             /* :276-292 */
                if (best_best_indels == 0) { last_best_indels = 0; last_best_d = d; last_best_best = bestbest; goto got_answer; }
                if (abs(last_best_indels) > best_best_indels) { last_best_indels = best_best_indels; last_best_d = d; last_best_best = bestbest; }
            }
        }
        if (last_best_d != MAXK + 1) goto got_answer;
    }
    rc = -1; goto done;                                                                /* more than k edits */

got_answer:
    {
        int straight = 0;                                                              /* :305-312 */
        for (int i = 0; i < end; i++) straight += pattern[i] != text[i];
        straight += pattern_len - end;
        if (straight == e) {                                                           /* no indels needed, :313-368 */
            int ok = 1;
            if (use_m) ok = write_cigar(&out, pattern_len, 'M');
            else {
                int streak_start = 0, matching = pattern[0] == text[0];
                for (int i = 0; i < end && ok; i++) {
                    int nm = pattern[i] == text[i];
                    if (nm != matching) {
                        ok = write_cigar(&out, i - streak_start, matching ? '=' : 'X');
                        matching = nm; streak_start = i;
                    }
                }
                if (ok && pattern_len > streak_start) {
                    if (!matching) ok = write_cigar(&out, pattern_len - streak_start, 'X');
                    else {
                        ok = write_cigar(&out, end - streak_start, '=');
                        if (ok && pattern_len > end) ok = write_cigar(&out, pattern_len - end, 'X');
                    }
                }
            }
            if (!ok) { rc = -2; goto done; }
            if (o_text_used) *o_text_used = end;
            rc = e; goto done;
        }
    }
    {   /* trace back, then emit forwards, :394-497 */
        char bt_action[MAXK + 2]; int bt_matched[MAXK + 2], bt_d[MAXK + 2];
        int cur_d = last_best_d;
        for (int ce = e; ce >= 1; ce--) {
            bt_action[ce] = AA(ce, cur_d);
            if (bt_action[ce] == 'I') { bt_d[ce] = cur_d + 1; bt_matched[ce] = LL(ce, cur_d) - LL(ce - 1, cur_d + 1) - 1; }
            else if (bt_action[ce] == 'D') { bt_d[ce] = cur_d - 1; bt_matched[ce] = LL(ce, cur_d) - LL(ce - 1, cur_d - 1); }
            else { bt_d[ce] = cur_d; bt_matched[ce] = LL(ce, cur_d) - LL(ce - 1, cur_d) - 1; }
            cur_d = bt_d[ce];
        }
        int acc_m = 0, ok = 1;
        if (use_m) acc_m = LL(0, 0);
        else if (LL(0, 0) > 0) ok = write_cigar(&out, LL(0, 0), '=');
        int ce = 1;
        while (ce <= e && ok) {
            const char action = bt_action[ce];
            int count = 1;
            while (ce + 1 <= e && bt_matched[ce] == 0 && bt_action[ce + 1] == action) { count++; ce++; }
            if (action == 'I') net_indel -= count; else if (action == 'D') net_indel += count;
            if (use_m) {
                if (action == '=' || action == 'X') acc_m += count;
                else {
                    if (acc_m != 0) { ok = write_cigar(&out, acc_m, 'M'); acc_m = 0; }
                    if (ok) ok = write_cigar(&out, count, action);
                }
            } else {
                ok = write_cigar(&out, count, action);
            }
            if (ok && bt_matched[ce] > 0) {
                if (use_m) acc_m += bt_matched[ce];
                else ok = write_cigar(&out, bt_matched[ce], '=');
            }
            ce++;
        }
        if (ok && use_m && acc_m != 0) ok = write_cigar(&out, acc_m, 'M');
        if (!ok) { rc = -2; goto done; }
        if (o_text_used) { int tu = last_best_best + last_best_d; *o_text_used = text_len < tu ? text_len : tu; }
        if (o_net_indel) *o_net_indel = net_indel;
        rc = e;
    }
done:
    free(L); free(TI); free(Act);
    return rc;
#undef LL
#undef TT
#undef AA
#undef out
}

/*
 * LandauVishkinWithCigar::computeEditDistanceNormalized with BAM_CIGAR_OPS output (LandauVishkin.cpp:507-648): the edit
 * distance, the op list (count << 4 | BAM code), and the "leading indel" convention -- a leading D is reported as
 * *add_front_clipping = count with return value 0 and NO cigar (the caller moves the alignment and calls again, SAM.cpp:1660-1684);
 * a leading I as -count with the cigar still returned.
 */
int oracle_lv_cigar_normalized(const char *text, int text_len, int text_lo, int text_hi, const char *pattern, int pattern_len, int pat_lo, int pat_hi,
                               int k, int use_m, uint32_t *ops, int ops_cap, int *n_ops, int *add_front_clipping, int *o_net_indel)
{
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(ops_cap > 0 ? ops_cap : 1));
    out_t out = { NULL, 0, 0, tmp, ops_cap, 0 };
    int text_used = 0;
    *n_ops = 0; *add_front_clipping = 0;
    int score = lv_cigar_core(text, text_len, text_lo, text_hi, pattern, pattern_len, pat_lo, pat_hi, k, use_m, &out, &text_used, o_net_indel);
    if (score < 0) { free(tmp); return score; }
    if (out.n_ops > 0) {
        const uint32_t code = tmp[0] & 0xf, cnt = tmp[0] >> 4;
        if (code == 2) { *add_front_clipping = (int)cnt; if (cnt != 0) { free(tmp); return 0; } }      /* :611-616 */
        else if (code == 1) *add_front_clipping = -(int)cnt;                                            /* :617-618 */
    }
    memcpy(ops, tmp, sizeof(uint32_t) * (size_t)out.n_ops);
    *n_ops = out.n_ops;
    free(tmp);
    return score;
}

/*
 * SAMFormat::computeCigar, Landau-Vishkin variant (SAM.cpp:2354-2467): the reference window comes from the genome, a read that
 * hangs off the end of its contig is soft-clipped there (iterating when the alignment's net indel changes how much hangs off),
 * k = MAX_K - 1.  data = the clipped read in reference orientation.  *n_ops = -1 stands for the "*" cigar of :2399-2408.
 */
int oracle_compute_cigar_lv(const oracle_genome *g, const char *data, int64_t data_len, int64_t extra_clipped_before, int64_t genome_location,
                            int use_m, uint32_t *ops, int ops_cap, int *n_ops, int *edit_distance, int *add_front_clipping,
                            int64_t *extra_clipped_after)
{
    int net_indel = 0;
    *extra_clipped_after = 0; *n_ops = 0; *edit_distance = 0; *add_front_clipping = 0;
    genome_location += extra_clipped_before; data += extra_clipped_before; data_len -= extra_clipped_before;     /* :2381-2383 */
    /* getContigAtLocation (Genome.cpp:574) */
    int lo = 0, hi = (int)g->n_contigs - 1, c = -1;
    while (lo <= hi) { int mid = (lo + hi) >> 1; if ((int64_t)g->contig_begin[mid] <= genome_location) { c = mid; lo = mid + 1; } else hi = mid - 1; }
    if (c < 0) return -1;
    const int64_t cend = c == (int)g->n_contigs - 1 ? (int64_t)g->n_bases : (int64_t)g->contig_begin[c + 1];      /* beginningLocation + length */
    const int64_t real_end = cend - (int64_t)g->chromosome_padding;
    if (genome_location + data_len > real_end) *extra_clipped_after = genome_location + data_len - real_end;       /* :2387-2395 */
    {   /* getSubstring(genomeLocation, dataLength), Genome.h:339-367 */
        const int64_t nb = (int64_t)g->n_bases;
        int ok;
        if (genome_location > nb || genome_location + data_len > nb + 1000) ok = 0;
        else if (data_len <= (int64_t)g->chromosome_padding && g->genome[genome_location] != 'n') ok = 1;
        else if (data_len == 0) ok = 1;
        else ok = cend > genome_location + data_len;
        if (!ok) { *n_ops = -1; return 0; }                                                                        /* :2398-2408 */
    }
    const char *reference = (const char *)g->genome + genome_location;
    const int64_t pad = (int64_t)g->genome_pad;
    for (int64_t pass = 0; pass <= data_len; pass++) {                                                              /* first call + the loop of :2435-2460 */
        const int plen = (int)(data_len - *extra_clipped_after);
        const int tlen = plen + MAXK;
        *edit_distance = oracle_lv_cigar_normalized(reference, tlen, (int)(-genome_location - pad), (int)((int64_t)g->n_bases + pad - genome_location),
                                                    data, plen, 0, plen, MAXK - 1, use_m, ops, ops_cap, n_ops, add_front_clipping, &net_indel);
        if (pass == 0 && *add_front_clipping != 0) return 0;                                                        /* :2425-2431 */
        int64_t nw = genome_location + data_len + net_indel - real_end; if (nw < 0) nw = 0;                         /* :2434 / :2459 */
        if (nw == *extra_clipped_after) return 0;
        *extra_clipped_after = nw;
    }
    return 0;
}
