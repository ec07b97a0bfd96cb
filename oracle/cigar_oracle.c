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

typedef struct { char *buf; int cap, used; } out_t;
static int write_cigar(out_t *o, int count, char code) {       /* writeCigar, COMPACT_CIGAR_STRING (:77-82, :100-113) */
    if (count <= 0) return 1;
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
int oracle_lv_cigar(const char *text, int text_len, int text_lo, int text_hi, const char *pattern, int pattern_len, int pat_lo, int pat_hi,
                    int k, int use_m, char *cigar, int cigar_cap, int *o_text_used, int *o_net_indel)
{
    static const int PrevDelta[3][3] = { {0, +1, -1}, {0, +1, -1}, {0, -1, +1} };    /* least absolute indels, :66-69 */
    const mem_t txt = { text, text_lo, text_hi }, pat = { pattern, pat_lo, pat_hi };
    out_t out = { cigar, cigar_cap, 0 };
    int net_indel = 0;
    if (o_net_indel) *o_net_indel = 0;
    if (cigar_cap > 0) cigar[0] = '\0';
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
}
