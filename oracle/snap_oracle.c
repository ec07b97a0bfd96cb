/*
 * snap_oracle.c -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or called from
 * the product (snap_amd/); only tests/, bench.py's cpu_baseline leg and
 * __graft_entry__.smoke() may use it, and only as the checker.
 *
 * A plain-C restatement (written for this repo, not copied) of the primitives on SNAP's
 * single-end hot path.  Each function cites the reference file:line it follows.  The
 * restatement is pinned two ways (tests/test_oracle.py):
 *   - against the reference's own known-answer tests: tests/LandauVishkinTest.cpp:11-32 and
 *     tests/AffineGapVectorizedTest.cpp:39-67 (vectors in tests/golden/reference_kats.json);
 *   - against the reference itself (oracle/_ref/libsnapref.so, the unmodified SNAP 2.0.5
 *     sources compiled by oracle/Makefile) on seeded fuzz inputs, bit-for-bit incl. FP64.
 * BaseAligner::AlignRead as a whole is NOT restated here: for that the oracle is the
 * compiled reference itself (oracle/_ref), see DESIGN.md "Oracle".
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 */
#include "snap_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_K 127                 /* LandauVishkin.h:11 */
#define MAX_READ_LENGTH 1000      /* Read.h:49 */
#define N_INDEL 10001             /* LandauVishkin.cpp:739 (maxIndels + 1) */

static double g_phred[256], g_indel[N_INDEL], g_perfect[MAX_READ_LENGTH + 1];
static unsigned g_wrapped[33][33];
static int g_inited = 0;

static int base_value(unsigned char c)           /* Tables.cpp:52-58: A0 G1 C2 T3 else 4 */
{
    switch (c) { case 'A': return 0; case 'G': return 1; case 'C': return 2; case 'T': return 3; default: return 4; }
}

void oracle_init(void)
{
    if (g_inited) return;
    /* LandauVishkin.cpp:734-760; constants BaseAligner.h:368-370 */
    const double SNP_PROB = 0.001, GAP_OPEN_PROB = 0.001, GAP_EXTEND_PROB = 0.5;
    g_indel[0] = 1.0;
    g_indel[1] = GAP_OPEN_PROB;
    for (int i = 2; i < N_INDEL; i++) g_indel[i] = g_indel[i - 1] * GAP_EXTEND_PROB;
    for (int i = 0; i < 33; i++) g_phred[i] = SNP_PROB;
    for (int i = 33; i <= 93 + 33; i++) g_phred[i] = 1.0 - (1.0 - pow(10.0, -1.0 * (i - 33.0) / 10.0)) * (1.0 - SNP_PROB);
    for (int i = 93 + 33 + 1; i < 256; i++) g_phred[i] = SNP_PROB;
    g_perfect[0] = 1.0;
    for (int i = 1; i <= MAX_READ_LENGTH; i++) g_perfect[i] = g_perfect[i - 1] * (1 - SNP_PROB);

    /* SeedSequencer.cpp:36-103: FIFO of intervals, the midpoint of each gets the next visit number */
    for (unsigned s = 2; s <= 32; s++) {
        unsigned lo[64], hi[64]; int head = 0, tail = 0;
        unsigned *off = g_wrapped[s];
        memset(off, 0, sizeof(g_wrapped[s]));
        lo[tail] = 1; hi[tail] = s - 1; tail++;
        unsigned filled = 1;
        while (head < tail) {
            unsigned l = lo[head % 64], h = hi[head % 64]; head++;
            unsigned sel = (l + h) / 2;
            off[sel] = filled++;
            if (h > sel) { lo[tail % 64] = sel + 1; hi[tail % 64] = h; tail++; }
            if (l < sel) { lo[tail % 64] = l; hi[tail % 64] = sel - 1; tail++; }
        }
    }
    g_inited = 1;
}

const double *oracle_phred_table(void)   { oracle_init(); return g_phred; }
const double *oracle_indel_table(void)   { oracle_init(); return g_indel; }
const double *oracle_perfect_table(void) { oracle_init(); return g_perfect; }

double oracle_seed_prob(int seed_len)
{
    /* BaseAligner.cpp:1314 `pow(1 - SNP_PROB, seedLen)` with int seedLen, compiled as C++98:
     * resolves to std::pow(double,int) == __builtin_powi, i.e. libgcc's square-and-multiply. */
    double x = 1 - 0.001; unsigned n = (unsigned)seed_len; double y = (n % 2) ? x : 1.0;
    while (n >>= 1) { x = x * x; if (n % 2) y *= x; }
    return y;
}

double oracle_seed_prob_pow(int seed_len)
{
    /* BaseAligner.cpp:907 (scoreLocationWithAffineGap, reached through alignAffineGap): the same expression, but seedLen is the
     * `unsigned` member there (BaseAligner.h:434) -- no local int shadows it as at :1141 -- so the call resolves to <cmath>'s promoting
     * template, i.e. libm's pow(double, double): one ulp below the powi value at seed 20. */
    return pow(1 - 0.001, (double)seed_len);
}

int oracle_compute_mapq(double p_all, double p_best, int score, int popular_seeds_skipped)   /* mapq.h:31-68 */
{
    (void)score;
    if (p_all < p_best) p_all = p_best;
    double correctness = p_best / p_all;
    int base;
    if (correctness >= 1) base = 70;
    else { base = (int)(-10 * log10(1 - correctness)); if (base > 70) base = 70; }
    int pen = popular_seeds_skipped - 10; if (pen < 0) pen = 0;
    base -= pen / 2;
    return base < 0 ? 0 : base;
}

unsigned oracle_wrapped_next_seed(unsigned seed_len, unsigned wrap_count)    /* SeedSequencer.h:40-43 */
{
    oracle_init();
    return g_wrapped[seed_len][wrap_count];
}

/* ------------------------------------------------------------------ seeds & index probe */

int oracle_pack_seed(const char *text, unsigned seed_len, uint64_t *bases, uint64_t *rc)   /* Seed.h:40-53, Seed.cpp:29 */
{
    uint64_t b = 0, r = 0;
    for (unsigned i = 0; i < seed_len; i++) {
        int e = base_value((unsigned char)text[i]);
        if (e > 3) return 0;
        b |= (uint64_t)e << ((seed_len - i - 1) * 2);
        r |= (uint64_t)(e ^ 3) << (i * 2);
    }
    *bases = b; *rc = r;
    return 1;
}

static uint64_t murmur_fin(uint64_t key)       /* HashTable.h:72-85 */
{
    key ^= key >> 33; key *= 0xff51afd7ed558ccdULL; key ^= key >> 33; key *= 0xc4ceb9fe1a85ec53ULL; key ^= key >> 33;
    return key;
}

/* SNAPHashTable::GetFirstValueForKey (HashTable.h:87-118): returns entry pointer or NULL */
static const uint8_t *probe(const oracle_index *ix, uint32_t table, uint64_t key, uint32_t *slots)
{
    const uint32_t vc = ix->large ? 2 : 1, entry = 4 * vc + ix->key_bytes;
    const uint64_t size = ix->table_size[table];
    const uint8_t *base = ix->hash_blob + ix->table_offset[table];
    uint64_t idx = murmur_fin(key) % size;
    uint64_t n_probes = 0;
    *slots = 0;
    for (;;) {
        const uint8_t *e = base + idx * entry;
        uint32_t v0; memcpy(&v0, e, 4);
        uint64_t k = 0; memcpy(&k, e + 4 * vc, ix->key_bytes);
        int key_eq = (k == key), invalid = (v0 == 0xffffffffu);
        (*slots)++;
        if (n_probes == 0) { if (key_eq && !invalid) return e; }
        else if (key_eq || invalid) return invalid ? NULL : e;
        n_probes++;
        if (n_probes > size + 5) return NULL;
        idx = (idx + (n_probes < 5 ? n_probes * n_probes : 1)) % size;
    }
}

static void fill_hits(const oracle_index *ix, uint32_t sub, int64_t *n, const uint32_t **hits, uint32_t *single)
{   /* GenomeIndex::fillInLookedUpResults32, GenomeIndex.cpp:2160-2202 */
    if ((uint64_t)sub < ix->n_bases) { *n = 1; *single = sub; *hits = single; }
    else if (sub == 0xfffffffeu) { *n = 0; *hits = NULL; }
    else { uint32_t o = sub - (uint32_t)ix->n_bases; *n = (int32_t)ix->overflow[o]; *hits = ix->overflow + o + 1; }
}

void oracle_lookup_seed(const oracle_index *ix, uint64_t bases, uint64_t rc, int64_t n_hits[2],
                        const uint32_t *hits[2], uint32_t singleton[2], uint32_t slots[2])
{   /* GenomeIndex::lookupSeed32, GenomeIndex.cpp:2096-2157 */
    const unsigned kb = ix->key_bytes * 8;
    const uint64_t mask = kb >= 64 ? ~0ULL : ((1ULL << kb) - 1);
    n_hits[0] = n_hits[1] = 0; hits[0] = hits[1] = NULL; slots[0] = slots[1] = 0;
    if (ix->large) {
        int comp = bases > rc;
        uint64_t s = comp ? rc : bases;
        const uint8_t *e = probe(ix, kb >= 64 ? 0 : (uint32_t)(s >> kb), s & mask, &slots[0]);
        if (!e) return;
        uint32_t v[2]; memcpy(v, e, 8);
        fill_hits(ix, comp ? v[1] : v[0], &n_hits[0], &hits[0], &singleton[0]);
        if (bases == rc) { n_hits[1] = n_hits[0]; hits[1] = hits[0]; singleton[1] = singleton[0]; if (n_hits[0] == 1) hits[1] = &singleton[1]; }
        else fill_hits(ix, comp ? v[0] : v[1], &n_hits[1], &hits[1], &singleton[1]);
    } else {
        uint64_t s[2] = {bases, rc};
        for (int d = 0; d < 2; d++) {
            const uint8_t *e = probe(ix, kb >= 64 ? 0 : (uint32_t)(s[d] >> kb), s[d] & mask, &slots[d]);
            if (e) { uint32_t v; memcpy(&v, e, 4); fill_hits(ix, v, &n_hits[d], &hits[d], &singleton[d]); }
        }
    }
}

/* ------------------------------------------------------------------ Landau-Vishkin */

/* LandauVishkin<dir>::computeEditDistance, LandauVishkin.h:100-351 (tie-breaking: SURVEY.md A.3).
 * The reference's countPerfectMatch (:377-407) is an 8-byte XOR/ctz run counter; its result is
 * the length of the matching run capped at the available bytes, which is what run() returns. */
#define PAT(i)  ((unsigned char)pattern[(i)])
#define TXT(j)  ((unsigned char)tx[(long)(j) * dir])

static int lv_run(const char *pattern, const char *tx, int dir, int p, int tpos, int end)
{
    int n = 0;
    while (p + n < end && PAT(p + n) == TXT(tpos + n)) n++;
    return n;
}

int oracle_lv(int dir, const char *text, int text_len, const char *pattern, const char *quality,
              int pattern_len, int k, double *match_probability, int *net_indel, int *total_indels, int *text_span)
{
    oracle_init();
    double local_p; int l1, l2, l3;
    if (!match_probability) match_probability = &local_p;
    if (!net_indel) net_indel = &l1;
    if (!total_indels) total_indels = &l2;
    if (!text_span) text_span = &l3;
    *net_indel = 0; *total_indels = 0; *text_span = 0; *match_probability = 0.0;
    if (k < 0) return -1;                                           /* :117 */
    if (k > MAX_K - 1) k = MAX_K - 1;                               /* :142 */
    *match_probability = 1.0;
    const char *tx = dir == -1 ? text - 1 : text;                   /* :159-161 */

    static __thread int16_t L[MAX_K + 1][2 * MAX_K + 1];
    static __thread char A[MAX_K + 1][2 * MAX_K + 1];
#define LL(e, d) L[(e)][(d) + MAX_K]
#define AA(e, d) A[(e)][(d) + MAX_K]
#define LGET(e, d) ((abs(d) <= (e)) ? (int)LL(e, d) : -2)

    int end0 = pattern_len < text_len ? pattern_len : text_len;
    int l00 = lv_run(pattern, tx, dir, 0, 0, end0);
    LL(0, 0) = (int16_t)l00;
    if (l00 == end0) {                                              /* :170-185 */
        int result = pattern_len > end0 ? pattern_len - end0 : 0;
        *match_probability = g_perfect[pattern_len];
        if (result > k) return -1;
        *text_span += pattern_len;
        return result;
    }
    int last_best_d = MAX_K + 1, e, found_x = 0;
    for (e = 1; e <= k; e++) {
        int d = 0;
        for (int it = 0; it < 2 * e + 1; it++, d = (d > 0 ? -d : -d + 1)) {   /* 0, 1, -1, 2, -2, ... (:64-66, :194) */
            int end = pattern_len < text_len - d ? pattern_len : text_len - d;
            int best = LGET(e - 1, d) + 1; char act = 'X';
            if (best >= 0) best += lv_run(pattern, tx, dir, best, d + best, end);
            int left = LGET(e - 1, d - 1);
            if (left >= 0) left += lv_run(pattern, tx, dir, left, d + left, end);
            if (left > best) { best = left; act = 'D'; }
            int right = LGET(e - 1, d + 1) + 1;
            if (right >= 0) right += lv_run(pattern, tx, dir, right, d + right, end);
            if (right > best) { best = right; act = 'I'; }
            AA(e, d) = act;
            LL(e, d) = (int16_t)best;
            if (best == pattern_len) {
                if (act == 'X') { last_best_d = d; found_x = 1; break; }      /* :243-248 */
                if (abs(d) < abs(last_best_d)) last_best_d = d;               /* :253-255 */
            }
        }
        if (found_x || last_best_d != MAX_K + 1) break;
    }
    if (last_best_d == MAX_K + 1) return -1;                        /* :267-269 */

    /* backtrace (:286-304) then forward pass (:306-342) */
    char bt_act[MAX_K + 1]; int bt_matched[MAX_K + 1];
    int cur_d = last_best_d;
    for (int ce = e; ce >= 1; ce--) {
        char a = AA(ce, cur_d); int pd;
        if (a == 'I') { pd = cur_d + 1; bt_matched[ce] = LL(ce, cur_d) - LGET(ce - 1, pd) - 1; }
        else if (a == 'D') { pd = cur_d - 1; bt_matched[ce] = LL(ce, cur_d) - LGET(ce - 1, pd); }
        else { pd = cur_d; bt_matched[ce] = LL(ce, cur_d) - LGET(ce - 1, pd) - 1; }
        bt_act[ce] = a; cur_d = pd;
    }
    int ce = 1, offset = l00;
    while (ce <= e) {
        char action = bt_act[ce]; int count = 1;
        while (ce + 1 <= e && bt_matched[ce] == 0 && bt_act[ce + 1] == action) { count++; ce++; }
        if (action == 'I') { *match_probability *= g_indel[count]; offset += count; *net_indel += count; *total_indels += count; }
        else if (action == 'D') { *match_probability *= g_indel[count]; offset -= count; *net_indel -= count; *total_indels += count; *text_span += count; }
        else {
            for (int i = 0; i < count; i++) {
                int qi = offset < 0 ? 0 : offset; if (qi > pattern_len - 1) qi = pattern_len - 1;
                *match_probability *= g_phred[(unsigned char)quality[qi]];
                offset++;
            }
        }
        offset += bt_matched[ce];
        ce++;
    }
    *match_probability *= g_perfect[pattern_len - e];
    *text_span += pattern_len;
    return e;
}

/* ------------------------------------------------------------------ affine gap */

/* Literal emulation of the SSE2 code (8 lanes of int16 per vector) of
 * AffineGapVectorized<dir>::computeScore (AffineGapVectorized.h:821-1339) and
 * computeScoreBanded (:256-819); semantics summarised in SURVEY.md A.4.  The full variant is
 * the banded one with a single segment and no band limits, except for three details that are
 * kept apart below: the lazy-F loop runs 8 (full) vs 7 (banded) rounds, the banded variant
 * carries F/H across segments, and the banded variant only touches vectors inside the band. */
typedef struct { int16_t v[8]; } vec8;

/* One reference AffineGapVectorized object keeps its traceback array between calls (backtraceAction, AffineGapVectorized.h:1374), and
 * the banded traceback can step onto cells of an earlier call (:740-788).  oracle_ag_bind_objects gives the calling thread the images
 * of the two objects of ONE aligner (affineGap: dir 1, reverseAffineGap: dir -1), zero-filled by the caller when "the aligner is
 * constructed": oracle_ag then writes and reads them with the reference's flat addressing, so its answers are those of a newly
 * constructed reference aligner scoring the same sequence of calls.  Unbound (NULL): every call is its own zero-filled object. */
static __thread uint8_t *g_ag_object[2] = {NULL, NULL};
static __thread size_t g_ag_object_cap = 0;
void oracle_ag_bind_objects(uint8_t *fwd, uint8_t *bwd, size_t cap_each) { g_ag_object[0] = fwd; g_ag_object[1] = bwd; g_ag_object_cap = cap_each; }
void oracle_ag_bound_objects(uint8_t **fwd, uint8_t **bwd, size_t *cap_each) { *fwd = g_ag_object[0]; *bwd = g_ag_object[1]; *cap_each = g_ag_object_cap; }

static int16_t sat16(int x) { return (int16_t)(x > 32767 ? 32767 : x < -32768 ? -32768 : x); }

int oracle_ag(int dir, int banded, const oracle_ag_params *prm, const char *text, int text_len,
              const char *pattern, const char *quality, int pattern_len, int w, int score_init,
              int is_rc, int use_clipping, int *text_offset, int *pattern_offset, int *n_edits,
              double *match_probability, int *stale_reads)
{
    oracle_init();
    int lto, lpo, lne; double lmp;
    if (!text_offset) text_offset = &lto;
    if (!pattern_offset) pattern_offset = &lpo;
    if (!n_edits) n_edits = &lne;
    if (!match_probability) match_probability = &lmp;
    if (stale_reads) *stale_reads = 0;
    if (w > MAX_K - 1) w = MAX_K - 1;
    if (w < 0) { *n_edits = -1; return -1; }                        /* :325 / :890 */
    *match_probability = 1.0;
    const char *tx = dir == -1 ? text - 1 : text;

    const int match = prm->match_reward, sub = -prm->sub_penalty;   /* init(), :105-133 */
    const int gap_open = prm->gap_open + prm->gap_extend, gap_ext = prm->gap_extend;

    int num_vec, seg_len, num_seg;
    if (banded) {
        int band_width = (2 * w + 1) < pattern_len ? (2 * w + 1) : pattern_len;   /* :339-342 */
        num_vec = (band_width + 7) / 8; seg_len = num_vec * 8; num_seg = (pattern_len + seg_len - 1) / seg_len;
    } else {
        num_vec = (pattern_len + 7) / 8; seg_len = num_vec * 8; num_seg = 1;      /* :914-915 */
    }
    const int nv_tot = num_vec * num_seg;
    /* pattern index held by (vector index vi in [0,nv_tot), lane l): segment s = vi / num_vec, k = vi % num_vec */
#define PIDX(vi, l) (((vi) / num_vec) * seg_len + (l) * num_vec + ((vi) % num_vec))

    int end_bonus;                                                  /* :380-394 / :950-966 */
    if (!is_rc) end_bonus = dir == -1 ? prm->five_bonus : prm->three_bonus;
    else        end_bonus = dir == -1 ? prm->three_bonus : prm->five_bonus;

    vec8 *H = calloc(nv_tot, sizeof(vec8)), *Hm1 = calloc(nv_tot, sizeof(vec8)), *E = calloc(nv_tot, sizeof(vec8));
    size_t bt_cells = (size_t)text_len * nv_tot * 8;
    uint8_t *bound = g_ag_object[dir == -1 ? 1 : 0];
    if (bound && bt_cells > g_ag_object_cap) { fprintf(stderr, "oracle_ag: bound traceback object too small (%zu > %zu)\n", bt_cells, g_ag_object_cap); abort(); }
    uint8_t *BT = bound ? bound : calloc(bt_cells ? bt_cells : 1, 1), *BTw = calloc(bt_cells ? bt_cells : 1, 1);

    /* first row (:399-414 / :971-983): note scoreFirstRow[] keeps stale lane values for padding lanes */
    {
        uint16_t first[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int vi = 0; vi < nv_tot; vi++) {
            for (int l = 0; l < 8; l++) {
                int pi = PIDX(vi, l);
                if (pi < pattern_len) { int v = score_init - gap_open - pi * gap_ext; first[l] = (uint16_t)(v > 0 ? v : 0); }
            }
            for (int l = 0; l < 8; l++) { H[vi].v[l] = (int16_t)first[l]; Hm1[vi].v[l] = 0; E[vi].v[l] = 0; }
        }
    }

    int score = -1;
    int best_global = -1, best_global_text = -1, best_local = -1, best_local_text = -1, best_local_pat = -1;
    *text_offset = -1; *pattern_offset = -1; *n_edits = -1;
    vec8 *Hp = H, *Hm = Hm1;

    for (int i = 0; i < text_len; i++) {
        const int tb = base_value((unsigned char)tx[(long)i * dir]);
        vec8 f, max, X; memset(&f, 0, sizeof f); memset(&max, 0, sizeof max); memset(&X, 0, sizeof X);
        int band_beg = 0, band_end = pattern_len - 1, seg_beg = 0, seg_end = 0;
        if (banded) {
            band_beg = i - w > 0 ? i - w : 0;
            band_end = i + w < pattern_len - 1 ? i + w : pattern_len - 1;
            seg_beg = band_beg / seg_len; seg_end = band_end / seg_len;
        }
        for (int j = seg_beg; j <= seg_end; j++) {
            vec8 h = Hp[j * num_vec + num_vec - 1];
            for (int l = 7; l > 0; l--) h.v[l] = h.v[l - 1];          /* _mm_slli_si128(h, 2) */
            int h_init;
            if (j == 0) {
                h_init = score_init;
                if (i > 0) { int v = score_init - gap_open - (i - 1) * gap_ext; h_init = v > 0 ? v : 0; }
            } else if (band_beg > j * seg_len) {
                h_init = 0;
            } else {
                h_init = Hp[j * num_vec - 1].v[7];                    /* _mm_srli_si128(.., 14) */
            }
            h.v[0] = (int16_t)h_init;
            uint8_t *bt_row = BT + ((size_t)i * nv_tot + (size_t)j * num_vec) * 8;
            uint8_t *btw_row = BTw + ((size_t)i * nv_tot + (size_t)j * num_vec) * 8;
            int nk = 0;
            for (int k = 0; k < num_vec && (!banded || (j * seg_len + k) <= band_end); k++, nk++) {
                const int vi = j * num_vec + k;
                vec8 hn = Hp[vi];                                     /* "Load the next score vector" (read before Hm is written: distinct buffers) */
                for (int l = 0; l < 8; l++) {
                    int pi = PIDX(vi, l);
                    int prof;
                    if (pi < pattern_len) {
                        int pb = base_value((unsigned char)pattern[pi]);
                        prof = (tb > 3 || pb > 3) ? -1 : (tb == pb ? match : sub);
                    } else prof = -32768;
                    int hv = h.v[l];
                    int m = hv > 0 ? sat16(hv + prof) : 0;            /* adds_epi16 then mask(h > 0) */
                    int e = E[vi].v[l];
                    int bt = e > m ? 1 : 0;
                    int hh = m > e ? m : e;
                    if (f.v[l] > hh) bt |= 2;
                    if (f.v[l] > hh) hh = f.v[l];
                    if (hh > max.v[l]) max.v[l] = (int16_t)hh;
                    Hm[vi].v[l] = (int16_t)hh;
                    int e2 = sat16(e - gap_ext);
                    int tmp = sat16(m - gap_open); if (tmp < 0) tmp = 0;
                    if (e2 > tmp) bt |= 4;
                    E[vi].v[l] = (int16_t)(e2 > tmp ? e2 : tmp);
                    int f2 = sat16(f.v[l] - gap_ext);
                    if (f2 > tmp) bt |= 32;
                    f.v[l] = (int16_t)(f2 > tmp ? f2 : tmp);
                    bt_row[k * 8 + l] = (uint8_t)bt; btw_row[k * 8 + l] = 1;
                }
                h = hn;
            }
            /* lazy F (:1080-1112 full, 8 rounds; :534-569 banded, 7 rounds with the segment carry X) */
            int rounds = banded ? 7 : 8, converged = 0;
            for (int r = 0; r < rounds && !converged; r++) {
                if (banded && f.v[7] > X.v[0]) X.v[0] = f.v[7];       /* X = max(X, f >> 14 bytes) */
                for (int l = 7; l > 0; l--) f.v[l] = f.v[l - 1];
                f.v[0] = 0;
                for (int v = 0; v < nk; v++) {
                    const int vi = j * num_vec + v;
                    int any = 0;
                    for (int l = 0; l < 8; l++) {
                        int hv = Hm[vi].v[l], fv = f.v[l];
                        int bt = bt_row[v * 8 + l];
                        if (fv > hv) { bt |= 2; hv = fv; }
                        Hm[vi].v[l] = (int16_t)hv;
                        if (hv > max.v[l]) max.v[l] = (int16_t)hv;
                        int tmp = (uint16_t)hv > (uint16_t)gap_open ? (uint16_t)hv - gap_open : 0;   /* subs_epu16 */
                        int f2 = (uint16_t)fv > (uint16_t)gap_ext ? (uint16_t)fv - gap_ext : 0;
                        f2 = (int16_t)f2; tmp = (int16_t)tmp;
                        if (f2 > tmp) { bt |= 32; any = 1; }
                        f.v[l] = (int16_t)f2;
                        bt_row[v * 8 + l] = (uint8_t)bt;
                    }
                    if (!any) { converged = 1; break; }
                }
            }
            if (banded) f = X;                                        /* :571-572 */
        }
        int max_row = 0;
        for (int l = 0; l < 8; l++) if (max.v[l] > max_row) max_row = max.v[l];

        if (!banded || band_end == pattern_len - 1) {                 /* :593-606 / :1125-1131 */
            int pe = pattern_len - 1, vi, li;
            if (banded) { vi = (pe / seg_len) * num_vec + (pe % seg_len) % num_vec; li = (pe % seg_len) / num_vec; }
            else { vi = pe % num_vec; li = pe / num_vec; }
            int g = Hm[vi].v[li];
            if (g >= best_global) { best_global = g; best_global_text = i; }
        }
        if (max_row == 0) break;
        if (max_row > best_local) {
            int local_off = -1;
            for (int j = seg_beg; j <= seg_end; j++)
                for (int k = 0; k < num_vec && (!banded || (j * seg_len + k) <= band_end); k++) {
                    const int vi = j * num_vec + k;
                    int top = -1;
                    for (int l = 0; l < 8; l++) if (Hm[vi].v[l] == (int16_t)max_row) top = l;   /* highest set bit of the lane mask */
                    if (top >= 0) { int po = j * seg_len + top * num_vec + k; if (po > local_off) local_off = po; }
                }
            best_local = max_row; best_local_text = i; best_local_pat = local_off;
        }
        vec8 *t = Hm; Hm = Hp; Hp = t;
    }

    /* local vs global (:643-730 / :1163-1251) */
    if (best_local != best_global && best_local >= best_global + end_bonus) {
        *pattern_offset = best_local_pat; *text_offset = best_local_text; score = best_local;
        if (use_clipping) {
            int pa = *pattern_offset - 1, ta = *text_offset, cnt = 0;
            while (pa + 1 != pattern_len && pattern[pa + 1] == tx[(long)(ta + 1) * dir]) { cnt++; pa++; ta++; }
            if (cnt >= 3) { *pattern_offset = pa; *text_offset = ta; }
            else {
                pa = *pattern_offset + 1; ta = *text_offset; cnt = 0;
                while (pa < pattern_len && pattern[pa] == tx[(long)ta * dir]) { cnt++; pa++; ta++; }
                if (cnt >= 3) { *pattern_offset = pa - 1; *text_offset = ta - 1; }
            }
            if (use_clipping != 2 && *pattern_offset == best_local_pat && *text_offset == best_local_text) {   /* 2 = useAltLiftover, :1212 */
                pa = *pattern_offset;
                while (pa != pattern_len - 1 && quality[pa] >= 65 && quality[pa + 1] >= 65) pa++;
                if (pa == pattern_len - 1) *pattern_offset = pa;
                else if (pa >= *pattern_offset + 2) {
                    int tmp_off = pa + 1, cnt_hq = 0, rem = pattern_len - tmp_off;
                    while (tmp_off != pattern_len - 1) { if (quality[tmp_off] >= 65) cnt_hq++; tmp_off++; }
                    if (((float)cnt_hq) / rem < 0.1) *pattern_offset = pa;
                }
            }
        }
    } else {
        *pattern_offset = pattern_len - 1; *text_offset = best_global_text; score = best_global;
    }

    int ret = -1;
    if (score > score_init) {                                       /* traceback, :732-815 / :1253-1335 */
        int row = *text_offset, col = *pattern_offset;
        int action = 0, prev_action = 0, action_count = 1, n_matches = 0, n_mismatches = 0, n_gaps = 0;
        while (row >= 0 && col >= 0) {
            int vi, li;
            if (banded) { vi = (col / seg_len) * num_vec + (col % seg_len) % num_vec; li = (col % seg_len) / num_vec; }
            else { vi = col % num_vec; li = col / num_vec; }
            size_t cell = ((size_t)row * nv_tot + vi) * 8 + li;
            if (!BTw[cell] && stale_reads) (*stale_reads)++;
            int bits = BT[cell];
            action = (bits >> (action << 1)) & 3;
            if (action == 0) {
                if (pattern[col] != tx[(long)row * dir]) { *match_probability *= g_phred[(unsigned char)quality[col]]; n_mismatches++; }
                else n_matches++;
                row--; col--;
            } else if (action == 1) row--;
            else { col--; action = 2; }
            if (prev_action != 0) {
                if (prev_action == action) action_count++;
                else { n_gaps += action_count; *match_probability *= g_indel[action_count]; action_count = 1; }
            }
            prev_action = action;
        }
        if (row >= 0) { action_count = row + 1; n_gaps += action_count; *match_probability *= g_indel[action_count]; }
        if (col >= 0) { action_count = col + 1; n_gaps += action_count; *match_probability *= g_indel[action_count]; }
        *n_edits = n_mismatches + n_gaps;
        *match_probability *= g_perfect[n_matches];
        *text_offset += 1; *pattern_offset += 1;
        *text_offset = pattern_len - *text_offset;
        *pattern_offset = pattern_len - *pattern_offset;
        *match_probability *= g_indel[*pattern_offset];
        ret = score;
    }
    free(H); free(Hm1); free(E); if (!bound) free(BT); free(BTw);
    return ret;
}
