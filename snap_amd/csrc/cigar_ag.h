// cigar_ag.h -- the CIGAR of a read that was scored with affine gap, one wavefront per read.
//
// Restates SAMFormat::computeCigar, affine-gap variant (SNAPLib/SAM.cpp:2470-2588), over
// AffineGapVectorizedWithCigar::computeGlobalScoreNormalized (SNAPLib/AffineGapVectorized.cpp:1043-1128),
// computeGlobalScoreBanded (:520-938), computeGlobalScore (:159-518) and computeFinalCigarString (:940-1041): what
// SAMFormat::writePairs / writeReads run for a read with usedAffineGapScoring or score > 0 (SAM.cpp:1653, :2200).
//
// The reference's answer is defined by its striped SSE2 evaluation order (Farrar): which lazy-F rounds run decides which
// traceback bits are set, and the banded form leaves cells outside the band untouched.  This first version therefore keeps the
// reference's layout literally: lane e < 8 is SSE element e, vectors are visited one after the other, H / H-1 / E rows live in
// LDS as int16[vector][8], one traceback byte per cell goes to a per-wave slab of HBM scratch.  8 of 64 lanes compute; the
// banded form -- the common case, taken when patternLen >= 3 (2k + 1) -- touches about two vectors per text row, so a 150 bp
// read costs ~10^4 wave instructions, against ~10^5 for its alignment.  (Packing 8 reads into a wave is the obvious next step.)
// Round 4: the row loops keep the row's traceback bytes in LDS while the row is worked on (first pass, then lazy F's read-modify-write of
// the same cells) and hand the finished row to the HBM slab with one store -- the slab used to be read and written cell by cell INSIDE the
// lazy-F loop, a ~1 us HBM round trip per vector and round, which is what a read cost (5.7 M reads/s for the whole chip, three times the
// wave cycles of the alignment itself: profiles/r04n, r04zy).  The traceback fetches 64 cells up the diagonal per load instead of one
// cell per step.
//
// Reference nondeterminism: the banded traceback may step to a cell the call did not evaluate and then reads what an EARLIER call
// left in the object's backtraceAction array (:811-824 over :1441).  Here such a cell reads 0 and the item is flagged `stale`.
#pragma once
#include "dev_common.h"
#include "cigar_lv.h"

#define AGC_MAX_READ_LENGTH 1000             // MAX_READ_LENGTH (Read.h:49): scoreInit of the banded call (:1070)
#define AGC_ACT_M 0
#define AGC_ACT_D 1
#define AGC_ACT_I 2
#define AGC_ACT_X 3

struct AGCParams { int match, sub, gap_open, gap_ext; };                 // constructor (:15-40): sub = -subPenalty, gap_open = open + extend

static __host__ __device__ __forceinline__ uint32_t agc_positions(uint32_t RL) { return RL + RL / 3 + 24; }               // striped positions incl. padding, both forms
static __host__ __device__ __forceinline__ uint32_t agc_rows(uint32_t RL) { return RL + LVC_MAX_K; }
static __host__ __device__ __forceinline__ size_t agc_scratch_bytes(uint32_t RL) {
    return (size_t)agc_rows(RL) * agc_positions(RL) + (size_t)(2 * RL + LVC_MAX_K + 8) * 4;                               // traceback bytes + (action, count) list
}
static __host__ __device__ __forceinline__ uint32_t agc_lds_bytes(uint32_t RL) {
    const uint32_t a = (RL + 15) & ~15u, t = (RL + LVC_MAX_K + 15) & ~15u, h = (agc_positions(RL) * 2 + 15) & ~15u;
    return 2 * a + t + 3 * h + ((agc_positions(RL) + 15) & ~15u);                                                         // pattern, quality, text, H, H-1, E, the row's traceback bytes
}

static __device__ __forceinline__ int agc_sat(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }             // _mm_adds/_mm_subs_epi16

struct AGCState {
    const uint8_t *pat, *qual, *txt;         // LDS
    int16_t *H, *Hm1, *E;                    // LDS, [vector][8]
    uint8_t *bt;                             // HBM scratch: [row][vector][8]
    uint8_t *bt_row;                         // LDS: the bytes of the row being computed, [vector][8]
    uint32_t *res;                           // HBM scratch: action | count << 2
    int bt_stride;                           // bytes per row
    AGCParams prm;
};

// ntTransitionMatrix (:24-36) through the query profile (:193-207): score of text base tb against pattern position p
static __device__ __forceinline__ int agc_profile(const AGCState &s, int tb, int p, int plen) {
    if (p >= plen) return -32768;
    const int pb = (int)base_value(s.pat[p]);
    if (tb > 3 || pb > 3) return -1;
    return tb == pb ? s.prm.match : s.prm.sub;
}

struct AGCOut { int n_edits, net_del, tail_ins, n_ops; bool stale; };

// What both forms share after the row loop: trace back from (text_used, plen - 1) (:374-455 / :802-875), the two "flip" passes
// (:457-507), computeFinalCigarString (:940-1041).  CELL(row, col, &evaluated) returns the traceback byte of a cell.
template <typename CellFn>
static __device__ __forceinline__ AGCOut agc_traceback_and_emit(const AGCState &s, int plen, int text_used, bool use_m, CellFn cell,
                                                                uint32_t *ops, int ops_cap)
{
    const int lane = lane_id();
    AGCOut o; o.n_edits = -1; o.net_del = 0; o.tail_ins = 0; o.n_ops = 0; o.stale = false;
    int n_res = 0;
    {
        int row = text_used, col = plen - 1;
        int action = AGC_ACT_M, prev = AGC_ACT_X, count = 1;
        while (row >= 0 && col >= 0) {
            // one load fetches the next 64 cells up the diagonal; they are consumed for as long as the path keeps stepping diagonally
            EMU_STAT(41, 1);
            const int rt = row - lane, ct = col - lane;
            bool evaluated = true;
            const int bl = (rt >= 0 && ct >= 0) ? cell(rt, ct, &evaluated) : 0;
            const int info = (bl & 0xff) | (evaluated ? 0 : 0x100);
            for (int t = 0; t < WAVE && row >= 0 && col >= 0; t++) {
                const int b = __builtin_amdgcn_readlane(info, t);
                if (b & 0x100) o.stale = true;
                action = ((b & 0xff) >> (action << 1)) & 3;
                bool left_diagonal = false;
                if (action == AGC_ACT_M) { row--; col--; }
                else if (action == AGC_ACT_D) { row--; left_diagonal = true; }
                else { col--; action = AGC_ACT_I; left_diagonal = true; }
                if (prev == action) count++;
                else if (prev != AGC_ACT_X) {
                    if (lane == 0) s.res[n_res] = (uint32_t)prev | ((uint32_t)count << 2);
                    n_res++; count = 1;
                }
                prev = action;
                if (left_diagonal) break;
            }
        }
        if (prev == action) { if (lane == 0) s.res[n_res] = (uint32_t)prev | ((uint32_t)count << 2); n_res++; }                 // :433-437
        if (row >= 0) { if (lane == 0) s.res[n_res] = (uint32_t)AGC_ACT_D | ((uint32_t)(row + 1) << 2); n_res++; }              // :439-444
        if (col >= 0) { if (lane == 0) s.res[n_res] = (uint32_t)AGC_ACT_I | ((uint32_t)(col + 1) << 2); n_res++; o.tail_ins = col + 1; }
        WAVE_SYNC();
    }
    auto R = [&](int i) -> uint32_t { return first_u32(s.res[i]); };
    auto ACT = [&](int i) -> int { return (int)(R(i) & 3u); };
    auto CNT = [&](int i) -> int { return (int)(R(i) >> 2); };
    auto SETCNT = [&](int i, int c) { const uint32_t a = R(i) & 3u; WAVE_SYNC(); if (lane == 0) s.res[i] = a | ((uint32_t)c << 2); WAVE_SYNC(); };
    int min_i = 0;
    if (ACT(0) == AGC_ACT_I) { min_i = 1; o.tail_ins = CNT(0); }                                                                // :464-468
    // "flip order of insertions followed by substitutions" (:470-493)
    {
        int row = 0, col = 0;
        for (int i = n_res - 1; i >= min_i; --i) {
            const int a = ACT(i), c = CNT(i);
            if (a == AGC_ACT_M) { row += c; col += c; }
            else if (a == AGC_ACT_D) row += c;
            else {
                if (i > 0 && row < text_used && col < plen - 1) {
                    if (s.pat[col + 1] == s.pat[col] && s.pat[col + 1] != s.txt[row] && s.qual[col] < 65) {
                        const int cm1 = CNT(i - 1);
                        if (i + 1 <= n_res - 1 && ACT(i + 1) == AGC_ACT_M && cm1 > 1) { SETCNT(i + 1, CNT(i + 1) + 1); row++; col++; }
                        if (ACT(i - 1) == AGC_ACT_M && cm1 > 1) SETCNT(i - 1, cm1 - 1);
                    }
                }
                col += c;
            }
        }
    }
    // "flip order of insertions and substitution with match in between" (:495-518)
    {
        int row = 0, col = 0;
        for (int i = n_res - 1; i >= min_i; --i) {
            const int a = ACT(i), c = CNT(i);
            if (a == AGC_ACT_M) { row += c; col += c; }
            else if (a == AGC_ACT_D) row += c;
            else {
                if (i > 0 && row + 1 < text_used && col + c < plen - 1) {
                    if (s.pat[col + c] == s.pat[col] && s.pat[col + c + 1] != s.txt[row + 1] && s.qual[col] < 65) {
                        const int cm1 = CNT(i - 1);
                        if (i + 1 <= n_res - 1 && ACT(i + 1) == AGC_ACT_M && cm1 > 2) { SETCNT(i + 1, CNT(i + 1) + 2); row += 2; col += 2; }
                        if (ACT(i - 1) == AGC_ACT_M && cm1 > 2) SETCNT(i - 1, cm1 - 2);
                    }
                }
                col += c;
            }
        }
    }
    // computeFinalCigarString (:940-1041)
    LVCOut out; out.ops = ops; out.cap = ops_cap; out.n = 0; out.ok = true;
    int n_edits = 0, row = 0, col = 0;
    for (int i = n_res - 1; i >= min_i && out.ok; --i) {
        const int a = ACT(i), c = CNT(i);
        if (a == AGC_ACT_D) { row += c; o.net_del += c; n_edits += c; out.put(c, LVC_OP_D, lane); }
        else if (a == AGC_ACT_I) { col += c; n_edits += c; out.put(c, LVC_OP_I, lane); }
        else if (a == AGC_ACT_M) {
            if (use_m) {
                for (int j0 = 0; j0 < c; j0 += WAVE) {
                    const int j = j0 + lane;
                    n_edits += __popcll(BALLOT(j < c && s.txt[row + j] != s.pat[col + j]));
                }
                out.put(c, LVC_OP_M, lane);
            } else {
                // maximal runs of "differs" / "equal" over the c bases (:984-1016)
                bool is_x = s.txt[row] != s.pat[col];
                int run_start = 0;
                for (int j0 = 0; j0 < c && out.ok; j0 += WAVE) {
                    const int j = j0 + lane;
                    const unsigned long long xm = BALLOT(j < c && s.txt[row + j] != s.pat[col + j]);
                    n_edits += __popcll(xm);
                    const int nb = c - j0 < WAVE ? c - j0 : WAVE;
                    int b = 0;
                    while (b < nb) {
                        unsigned long long want = is_x ? ~xm : xm;                 // positions whose state differs from the current run's
                        want = (want >> b) << b;
                        if (nb < 64) want &= (1ull << nb) - 1ull;
                        if (!want) break;
                        const int jc = j0 + (__ffsll((long long)want) - 1);
                        out.put(jc - run_start, is_x ? LVC_OP_X : LVC_OP_EQ, lane);
                        is_x = !is_x; run_start = jc; b = jc - j0;
                    }
                }
                out.put(c - run_start, is_x ? LVC_OP_X : LVC_OP_EQ, lane);
            }
            row += c; col += c;
        }
    }
    o.n_edits = out.ok ? n_edits : -2;
    o.n_ops = out.n;
    return o;
}

// computeGlobalScore (:159-518)
static __device__ __forceinline__ AGCOut agc_full(const AGCState &s, int plen, int tlen, bool use_m, uint32_t *ops, int ops_cap)
{
    const int lane = lane_id(), el = lane & 7;
    const bool st = lane < 8;                                               // the lanes that store
    const int open = s.prm.gap_open, ext = s.prm.gap_ext;
    const int num_vec = (plen + 7) / 8;
    int16_t *Hp = s.H, *Hm = s.Hm1;
    for (int j = 0; j < num_vec; j++) {                                     // first row (:224-239)
        const int p = el * num_vec + j;
        if (st) { Hp[j * 8 + el] = (int16_t)(p < plen ? -(open + p * ext) : -32768); s.E[j * 8 + el] = (int16_t)-32768; }
    }
    WAVE_SYNC();
    int score = -32768, text_used = -1;
    EMU_STAT(34, 1);
    for (int i = 0; i < tlen; i++) {
        EMU_STAT(36, 1); EMU_STAT(39, num_vec);
        const int tb = (int)base_value(s.txt[i]);
        uint8_t *btrow = s.bt_row;                                           // (LDS; goes to s.bt + i * s.bt_stride when the row is done)
        int f = -32768;
        int h = __shfl_up((int)Hp[(num_vec - 1) * 8 + el], 1);
        const int h_init = i > 0 ? -(open + (i - 1) * ext) : 0;
        if (el == 0) h = h_init;
        for (int j = 0; j < num_vec; j++) {
            const int m = agc_sat(h + agc_profile(s, tb, el * num_vec + j, plen));
            int e = (int)s.E[j * 8 + el];
            int bt = e > m ? 1 : 0;
            h = m > e ? m : e;
            { const int t = f > h ? 2 : 0; bt = t | (bt & ~t); }
            h = h > f ? h : f;
            const int hnext = (int)Hp[j * 8 + el];
            if (st) Hm[j * 8 + el] = (int16_t)h;
            e = agc_sat(e - ext);
            const int temp = agc_sat(m - open);
            if (e > temp) bt |= 4;
            e = e > temp ? e : temp;
            if (st) s.E[j * 8 + el] = (int16_t)e;
            f = agc_sat(f - ext);
            if (f > temp) bt |= 32;
            f = f > temp ? f : temp;
            if (st) btrow[j * 8 + el] = (uint8_t)bt;
            h = hnext;
        }
        WAVE_SYNC();
        // lazy F (:305-337)
        bool done = false;
        for (int k = 0; k < 7 && !done; k++) {
            f = __shfl_up(f, 1);
            if (el == 0) f = -32768;
            for (int j = 0; j < num_vec; j++) {
                EMU_STAT(40, 1);
                int hh = (int)Hm[j * 8 + el];
                int bt = (int)btrow[j * 8 + el];
                { const int t = f > hh ? 2 : 0; bt = t | (bt & ~t); }
                hh = hh > f ? hh : f;
                const int temp = agc_sat(hh - open);
                f = agc_sat(f - ext);
                if (f > temp) bt |= 32;
                if (st) { Hm[j * 8 + el] = (int16_t)hh; btrow[j * 8 + el] = (uint8_t)bt; }
                if ((BALLOT(f > temp) & 0xffull) == 0) { done = true; break; }
            }
        }
        WAVE_SYNC();
        for (int x = lane; x < num_vec * 8; x += WAVE) s.bt[(size_t)i * s.bt_stride + x] = btrow[x];                     // the finished row
        WAVE_SYNC();
        const int g = (int)first_u32((uint32_t)(int)Hm[((plen - 1) % num_vec) * 8 + (plen - 1) / num_vec]);              // :341-346
        if (g >= score) { score = g; text_used = i; }
        { int16_t *t = Hm; Hm = Hp; Hp = t; }
    }
    WAVE_SYNC();
    if (!(score > -32768)) { AGCOut o; o.n_edits = -1; o.net_del = 0; o.tail_ins = 0; o.n_ops = 0; o.stale = false; return o; }
    auto cell = [&](int row, int col, bool *evaluated) -> int {                 // (per lane)
        *evaluated = true;
        return (int)s.bt[(size_t)row * s.bt_stride + (col % num_vec) * 8 + col / num_vec];
    };
    return agc_traceback_and_emit(s, plen, text_used, use_m, cell, ops, ops_cap);
}

// computeGlobalScoreBanded (:520-938)
static __device__ __forceinline__ AGCOut agc_banded(const AGCState &s, int plen, int tlen, int w, int score_init, bool use_m,
                                                    uint32_t *ops, int ops_cap)
{
    const int lane = lane_id(), el = lane & 7;
    const bool st = lane < 8;
    const int open = s.prm.gap_open, ext = s.prm.gap_ext;
    if (w > LVC_MAX_K - 1) w = LVC_MAX_K - 1;
    const int bw = 2 * w + 1 < plen ? 2 * w + 1 : plen;
    const int num_vec = (bw + 7) / 8, seg_len = num_vec * 8, num_seg = (plen + seg_len - 1) / seg_len;
    int16_t *Hp = s.H, *Hm = s.Hm1;
    {   // first row (:611-628); a lane's scoreFirstRow element keeps its last value where the position is past the pattern
        int sfr = 0;
        for (int sg = 0; sg < num_seg; sg++)
            for (int v = 0; v < num_vec; v++) {
                const int p = sg * seg_len + el * num_vec + v;
                if (p < plen) { const int x = score_init - (open + p * ext); sfr = x > 0 ? x : 0; }
                if (st) { Hp[(sg * num_vec + v) * 8 + el] = (int16_t)sfr; Hm[(sg * num_vec + v) * 8 + el] = 0; s.E[(sg * num_vec + v) * 8 + el] = 0; }
            }
    }
    WAVE_SYNC();
    int score = score_init, text_used = -1;
    int idle_rows = 0;                                                      // rows since the band left the last segment (see the end of the loop)
    EMU_STAT(33, 1);
    for (int i = 0; i < tlen; i++) {
        EMU_STAT(35, 1);
        const int tb = (int)base_value(s.txt[i]);
        uint8_t *btrow = s.bt_row;                                           // (LDS; the evaluated vectors go to s.bt + i * s.bt_stride when the row is done)
        int f = 0, X = 0;
        const int band_beg = i - w > 0 ? i - w : 0;
        const int band_end = i + w < plen - 1 ? i + w : plen - 1;
        const int seg_beg = band_beg / seg_len, seg_end = band_end / seg_len;
        for (int j = seg_beg; j <= seg_end; j++) {
            int h = __shfl_up((int)Hp[(j * num_vec + num_vec - 1) * 8 + el], 1);
            int h_init;
            if (j == 0) h_init = (int)(int16_t)(i > 0 ? score_init - (open + (i - 1) * ext) : score_init);
            else if (band_beg > j * seg_len) h_init = 0;
            else h_init = (int)first_u32((uint32_t)(int)Hp[(j * num_vec - 1) * 8 + 7]);
            if (el == 0) h = h_init;
            for (int k = 0; k < num_vec && j * seg_len + k <= band_end; k++) {
                const int vi = (j * num_vec + k) * 8 + el;
                EMU_STAT(37, 1);
                const int m = agc_sat(h + agc_profile(s, tb, j * seg_len + el * num_vec + k, plen));
                int e = (int)s.E[vi];
                int bt = e > m ? 1 : 0;
                h = m > e ? m : e;
                { const int t = f > h ? 2 : 0; bt = t | (bt & ~t); }
                h = h > f ? h : f;
                const int hnext = (int)Hp[vi];
                if (st) Hm[vi] = (int16_t)h;
                e = agc_sat(e - ext);
                const int temp = agc_sat(m - open);
                if (e > temp) bt |= 4;
                e = e > temp ? e : temp;
                if (st) s.E[vi] = (int16_t)e;
                f = agc_sat(f - ext);
                if (f > temp) bt |= 32;
                f = f > temp ? f : temp;
                if (st) btrow[vi] = (uint8_t)bt;
                h = hnext;
            }
            WAVE_SYNC();
            bool done = false;
            for (int kk = 0; kk < 7 && !done; kk++) {
                {   // X = max(X, f >> 14 bytes): element 0 takes element 7 (:745)
                    const int f7 = __builtin_amdgcn_readlane(f, 7);
                    if (el == 0) X = X > f7 ? X : f7;
                }
                f = __shfl_up(f, 1);
                if (el == 0) f = 0;
                for (int v = 0; v < num_vec && j * seg_len + v <= band_end; v++) {
                    const int vi = (j * num_vec + v) * 8 + el;
                    EMU_STAT(38, 1);
                    int hh = (int)Hm[vi];
                    int bt = (int)btrow[vi];
                    { const int t = f > hh ? 2 : 0; bt = t | (bt & ~t); }
                    hh = hh > f ? hh : f;
                    const int temp = agc_sat(hh - open);
                    f = agc_sat(f - ext);
                    if (f > temp) bt |= 32;
                    if (st) { Hm[vi] = (int16_t)hh; btrow[vi] = (uint8_t)bt; }
                    if ((BALLOT(f > temp) & 0xffull) == 0) { done = true; break; }
                }
            }
            f = el == 0 ? X : 0;                                            // :783
        }
        WAVE_SYNC();
        {   // the finished row: the vectors this row evaluated are a contiguous range (every vector of segments seg_beg .. seg_end - 1, and
            // of the last segment those that start inside the band); nothing else of the slab's row is touched
            const int k_last = band_end - seg_end * seg_len < num_vec - 1 ? band_end - seg_end * seg_len : num_vec - 1;
            const int x0 = seg_beg * num_vec * 8, x1 = (seg_end * num_vec + k_last + 1) * 8;
            for (int x = x0 + lane; x < x1; x += WAVE) s.bt[(size_t)i * s.bt_stride + x] = btrow[x];
            WAVE_SYNC();
        }
        if (band_end == plen - 1) {                                         // :803-815
            const int vec = (band_end / seg_len) * num_vec + (band_end % seg_len) % num_vec, e_i = (band_end % seg_len) / num_vec;
            const int g = (int)first_u32((uint32_t)(int)Hm[vec * 8 + e_i]);
            if (g > score) { score = g; text_used = i; }
        }
        { int16_t *t = Hm; Hm = Hp; Hp = t; }
        // Once the band's start has left the last segment a row evaluates nothing: all it does is look at the pattern-end cell of the
        // buffer the swap hands it -- the rows of two and of one row ago, in turn.  After two such rows both have been looked at and the
        // remaining ~100 rows of the text (plen + LVC_MAX_K of them) cannot change score or text_used: stop.
        if (seg_beg > seg_end && ++idle_rows == 2) break;
    }
    WAVE_SYNC();
    // score >= scoreInit > 0 always (:802).  With textUsed == -1 (no row beat scoreInit) the walk below does nothing and the result
    // is "the whole pattern is a tail insertion", which computeGlobalScoreNormalized takes as a failed band (:1074).
    auto cell = [&](int row, int col, bool *evaluated) -> int {                 // (per lane)
        const int bb = row - w > 0 ? row - w : 0, be = row + w < plen - 1 ? row + w : plen - 1;
        const int sg = col / seg_len, v = (col % seg_len) % num_vec;
        *evaluated = sg >= bb / seg_len && sg <= be / seg_len && sg * seg_len + v <= be;
        if (!*evaluated) return 0;
        return (int)s.bt[(size_t)row * s.bt_stride + (sg * num_vec + v) * 8 + (col % seg_len) / num_vec];
    };
    return agc_traceback_and_emit(s, plen, text_used, use_m, cell, ops, ops_cap);
}

// computeGlobalScoreBanded once more, with the vectors of a segment side by side: lane = vector k * 8 + SSE element el (num_vec <= 8).
// The reference visits a segment's vectors one after the other, and so did agc_banded above -- 8 lanes, 4.2 lazy-F steps per first-pass
// vector, three LDS round trips per step.  But within a row only F runs along the vectors: the diagonal input of vector k is the
// PREVIOUS row's H of vector k - 1, E is the cell's own, so the first pass of all vectors is one step plus a (K - 1)-step hand-over of F
// from lane group to lane group; and in a lazy-F round what reaches vector v is the round's incoming F minus v * ext (the round never
// raises F), so a round is one step for all vectors, the reference's "stop at the first vector in which no element's F goes on" is a
// ballot, and every cell's H / E / traceback byte stays in ITS lane's registers from the first pass to the end of the lazy rounds.
// Same operations on the same values in the reference's order where order matters; results identical to agc_banded (tests: the
// reference's fixtures, tests/golden/cigar_ag.npz, sam_fields*.npz, the FASTQ -> SAM identity tests).
static __device__ __forceinline__ AGCOut agc_banded_par(const AGCState &s, int plen, int tlen, int w, int score_init, bool use_m,
                                                        uint32_t *ops, int ops_cap)
{
    const int lane = lane_id(), el = lane & 7, k = lane >> 3;
    const int open = s.prm.gap_open, ext = s.prm.gap_ext;
    if (w > LVC_MAX_K - 1) w = LVC_MAX_K - 1;
    const int bw = 2 * w + 1 < plen ? 2 * w + 1 : plen;
    const int num_vec = (bw + 7) / 8, seg_len = num_vec * 8, num_seg = (plen + seg_len - 1) / seg_len;      // (the caller guarantees num_vec <= 8)
    int16_t *Hp = s.H, *Hm = s.Hm1;
    {   // first row (:611-628), as in agc_banded
        const bool st = lane < 8;
        int sfr = 0;
        for (int sg = 0; sg < num_seg; sg++)
            for (int v = 0; v < num_vec; v++) {
                const int p = sg * seg_len + el * num_vec + v;
                if (p < plen) { const int x = score_init - (open + p * ext); sfr = x > 0 ? x : 0; }
                if (st) { Hp[(sg * num_vec + v) * 8 + el] = (int16_t)sfr; Hm[(sg * num_vec + v) * 8 + el] = 0; s.E[(sg * num_vec + v) * 8 + el] = 0; }
            }
    }
    WAVE_SYNC();
    int score = score_init, text_used = -1;
    int idle_rows = 0;
    EMU_STAT(33, 1);
    for (int i = 0; i < tlen; i++) {
        EMU_STAT(35, 1);
        const int tb = (int)base_value(s.txt[i]);
        int fcar = 0, X = 0;                                                // F entering the segment (element 0 only: X of the segment before)
        const int band_beg = i - w > 0 ? i - w : 0;
        const int band_end = i + w < plen - 1 ? i + w : plen - 1;
        const int seg_beg = band_beg / seg_len, seg_end = band_end / seg_len;
        for (int j = seg_beg; j <= seg_end; j++) {
            const int K = band_end - j * seg_len + 1 < num_vec ? band_end - j * seg_len + 1 : num_vec;     // vectors this row evaluates in the segment
            const bool act = k < K;
            const int kk_ = k < num_vec ? k : num_vec - 1;                  // (lanes beyond the segment's vectors compute on vector num_vec - 1's data and store nothing)
            const int vi = (j * num_vec + kk_) * 8 + el;
            EMU_STAT(37, 1);
            int h_init;
            if (j == 0) h_init = (int)(int16_t)(i > 0 ? score_init - (open + (i - 1) * ext) : score_init);
            else if (band_beg > j * seg_len) h_init = 0;
            else h_init = (int)first_u32((uint32_t)(int)Hp[(j * num_vec - 1) * 8 + 7]);
            // diagonal input: vector 0 takes the segment's LAST vector of the previous row one element down (element 0: h_init), vector k > 0 the previous row's vector k - 1
            const int hsrc = kk_ == 0 ? (j * num_vec + num_vec - 1) * 8 + (el > 0 ? el - 1 : 0) : vi - 8;
            int h = (int)Hp[hsrc];
            if (kk_ == 0 && el == 0) h = h_init;
            const int m = agc_sat(h + agc_profile(s, tb, j * seg_len + el * num_vec + kk_, plen));
            const int e = (int)s.E[vi];
            const int me = m > e ? m : e;
            int bt = e > m ? 1 : 0;
            const int e2 = agc_sat(e - ext);
            const int temp = agc_sat(m - open);
            if (e2 > temp) bt |= 4;
            const int e_new = e2 > temp ? e2 : temp;
            // F along the vectors: f_0 = what the segment before left (element 0) or 0; f_{k+1} = max(sat(f_k - ext), temp_k)
            int fk = fcar;
            for (int st_ = 1; st_ < K; st_++) {
                const int fo = agc_sat(fk - ext);
                const int fn = fo > temp ? fo : temp;                       // (valid in the lanes of vector st_ - 1)
                const int up = __shfl_up(fn, 8);
                if (k >= st_) fk = up;
            }
            if (fk > me) bt |= 2;
            int hv = me > fk ? me : fk;
            const int f2 = agc_sat(fk - ext);
            if (f2 > temp) bt |= 32;
            const int fout = f2 > temp ? f2 : temp;
            // lazy F (:737-781): the F that left the segment's last evaluated vector, element by element
            int f = __shfl(fout, (K - 1) * 8 + el);
            for (int kk = 0; kk < 7; kk++) {
                EMU_STAT(38, 1);
                {   // X = max(X, f >> 14 bytes): element 0 takes element 7 (:745)
                    const int f7 = __builtin_amdgcn_readlane(f, 7);
                    X = X > f7 ? X : f7;
                }
                int fin = __shfl_up(f, 1);
                if (el == 0) fin = 0;
                int fv = fin - kk_ * ext; if (fv < -32768) fv = -32768;     // kk_ saturating steps down from the round's incoming F
                const bool f_wins = fv > hv;
                const int hh = f_wins ? fv : hv;
                const int temp2 = agc_sat(hh - open);
                const int fnx = agc_sat(fv - ext);
                const bool cont = fnx > temp2;
                // the round stops after the first vector in which no element's F goes on; that vector is still updated
                const unsigned long long cm = BALLOT(cont && act);
                int vstop = K;                                              // K: no such vector, the round is complete
                for (int v = 0; v < K; v++) if (((cm >> (8 * v)) & 0xffull) == 0ull) { vstop = v; break; }
                if (act && k <= vstop) {
                    if (f_wins) bt |= 2;
                    hv = hh;
                    if (cont) bt |= 32;
                }
                if (vstop < K) break;
                int fe = fin - K * ext; if (fe < -32768) fe = -32768;       // F after the round's K vectors
                f = fe;
            }
            fcar = el == 0 ? X : 0;                                         // :783
            if (act) { Hm[vi] = (int16_t)hv; s.E[vi] = (int16_t)e_new; s.bt[(size_t)i * s.bt_stride + vi] = (uint8_t)bt; }
            WAVE_SYNC();
        }
        if (band_end == plen - 1) {                                         // :803-815
            const int vec = (band_end / seg_len) * num_vec + (band_end % seg_len) % num_vec, e_i = (band_end % seg_len) / num_vec;
            const int g = (int)first_u32((uint32_t)(int)Hm[vec * 8 + e_i]);
            if (g > score) { score = g; text_used = i; }
        }
        { int16_t *t = Hm; Hm = Hp; Hp = t; }
        if (seg_beg > seg_end && ++idle_rows == 2) break;                   // (see agc_banded)
    }
    WAVE_SYNC();
    auto cell = [&](int row, int col, bool *evaluated) -> int {                 // (per lane)
        const int bb = row - w > 0 ? row - w : 0, be = row + w < plen - 1 ? row + w : plen - 1;
        const int sg = col / seg_len, v = (col % seg_len) % num_vec;
        *evaluated = sg >= bb / seg_len && sg <= be / seg_len && sg * seg_len + v <= be;
        if (!*evaluated) return 0;
        return (int)s.bt[(size_t)row * s.bt_stride + (sg * num_vec + v) * 8 + (col % seg_len) / num_vec];
    };
    return agc_traceback_and_emit(s, plen, text_used, use_m, cell, ops, ops_cap);
}

// ---- the banded row loop done AHEAD of the record, eight reads to a wavefront (cigar_k.hip: k_samf_dp8) ------------------------------------
// The row loop is ~90 % of a record's instructions, and for the reads a run mostly consists of -- edit distance k <= 3, i.e. a band of at
// most 7 cells, ONE SSE vector per segment -- it keeps 8 of 64 lanes busy.  k_samf_dp8 runs exactly that case (num_vec == 1, the first
// attempt of a read that lies inside its contig) for eight reads side by side, one read per 8-lane group, and leaves per read what the rest
// of the record needs from the loop: textUsed and the traceback bytes of the cells each row evaluated (at most two segments: 16 bytes a row).
// cigar_ag_item takes them instead of running agc_banded_par when the call it is about to make IS that call (same pattern length, limit,
// location, orientation and clipping); anything else -- wider bands, a retry after a leading indel, a read at a contig's end, a failed band --
// is computed here as before.  The loop below is agc_banded with num_vec == 1, value for value.
struct SamfPre { int32_t valid, plen, w, bcb; int64_t loc; int32_t score, text_used, dir, rows, pad0, pad1; };
static_assert(sizeof(SamfPre) == 48, "SamfPre layout");
#define SAMF_PRE_ROW 32                      // traceback bytes a row can evaluate: two segments x two vectors x 8
static __host__ __device__ __forceinline__ uint32_t samf_pre_rows(uint32_t RL) { return RL + 16; }
static __host__ __device__ __forceinline__ size_t samf_pre_stride(uint32_t RL) { return (sizeof(SamfPre) + (size_t)samf_pre_rows(RL) * SAMF_PRE_ROW + 63) & ~(size_t)63; }
#define SAMF_PRE_MAX_W 7                     // 2 w + 1 <= 16: at most two vectors per segment

// the rest of computeGlobalScoreBanded for a call whose row loop k_samf_dp8 has run
static __device__ __forceinline__ AGCOut agc_from_pre(const AGCState &s, int plen, int w, const SamfPre *pre, bool use_m, uint32_t *ops, int ops_cap)
{
    const int text_used = (int)first_u32((uint32_t)pre->text_used);
    const uint8_t *bt = (const uint8_t *)(pre + 1);
    const int bw = 2 * w + 1 < plen ? 2 * w + 1 : plen, nv = (bw + 7) >> 3, seg_len = nv * 8;
    EMU_STAT(42, 1);
    auto cell = [&](int row, int col, bool *evaluated) -> int {                 // (per lane)
        const int bb = row - w > 0 ? row - w : 0, be = row + w < plen - 1 ? row + w : plen - 1;
        const int sg = col / seg_len, v = (col % seg_len) % nv;
        *evaluated = sg >= bb / seg_len && sg <= be / seg_len && sg * seg_len + v <= be;
        if (!*evaluated) return 0;
        return (int)bt[(size_t)row * SAMF_PRE_ROW + (size_t)((sg - bb / seg_len) * seg_len + v * 8 + (col % seg_len) / nv)];
    };
    return agc_traceback_and_emit(s, plen, text_used, use_m, cell, ops, ops_cap);
}

// computeGlobalScoreNormalized (:1043-1128) + SAMFormat::computeCigar, affine-gap variant (SAM.cpp:2470-2588)
struct CigarAGItemOut { int n_ops, edit_distance, add_front_clipping, tail_ins, stale; long long extra_after; };

static __device__ __forceinline__ CigarAGItemOut cigar_ag_item(const DevIndex &ix, const AGCParams &prm, const uint8_t *data, const uint8_t *quality,
                                                               long long data_len, int k, long long extra_before, long long loc, bool use_m,
                                                               uint8_t *lds, uint32_t RL, uint8_t *scratch, uint32_t *ops, int ops_cap,
                                                               const SamfPre *pre = nullptr)
{
    const int lane = lane_id();
    CigarAGItemOut o; o.n_ops = 0; o.edit_distance = 0; o.add_front_clipping = 0; o.tail_ins = 0; o.stale = 0; o.extra_after = 0;
    loc += extra_before; data += extra_before; data_len -= extra_before;        // SAM.cpp:2500-2502 (`quality` is NOT advanced there: the reference
                                                                                //  keeps indexing it from the clipped read's first base)
    int lo = 0, hi = (int)ix.n_contigs - 1, c = -1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if ((long long)first_u64(ix.contig_begin[mid]) <= loc) { c = mid; lo = mid + 1; } else hi = mid - 1;
    }
    if (c < 0 || data_len < 0) { o.n_ops = -1; return o; }
    const long long nb = (long long)ix.n_bases;
    const long long cend = c == (int)ix.n_contigs - 1 ? nb : (long long)first_u64(ix.contig_begin[c + 1]);
    const long long real_end = cend - (long long)ix.chromosome_padding;
    if (loc + data_len > real_end) o.extra_after = loc + data_len - real_end;
    {
        bool ok;
        if (loc > nb || loc + data_len > nb + 1000) ok = false;
        else if (data_len <= (long long)ix.chromosome_padding && first_u32(ix.genome[loc]) != 'n') ok = true;
        else if (data_len == 0) ok = true;
        else ok = cend > loc + data_len;
        if (!ok) { o.n_ops = -1; return o; }
    }
    EMU_STAT(32, 1);
    AGCState s;
    const uint32_t a = (RL + 15) & ~15u, t = (RL + LVC_MAX_K + 15) & ~15u, h = (agc_positions(RL) * 2 + 15) & ~15u;
    uint8_t *lp = lds, *lq = lds + a, *lt = lds + 2 * a;
    s.pat = lp; s.qual = lq; s.txt = lt;
    s.H = (int16_t *)(lds + 2 * a + t); s.Hm1 = (int16_t *)(lds + 2 * a + t + h); s.E = (int16_t *)(lds + 2 * a + t + 2 * h);
    s.bt = scratch; s.bt_stride = (int)agc_positions(RL);
    s.bt_row = lds + 2 * a + t + 3 * h;
    s.res = (uint32_t *)(scratch + (size_t)agc_rows(RL) * agc_positions(RL));
    s.prm = prm;
    const long long readable = nb + (long long)ix.genome_pad - loc;
    const int full_t = (int)data_len + LVC_MAX_K;
    for (int i = lane; i < (int)data_len; i += WAVE) { lp[i] = data[i]; lq[i] = quality[i]; }
    for (int i = lane; i < full_t; i += WAVE) lt[i] = i < readable ? ix.genome[loc + i] : (uint8_t)0;
    WAVE_SYNC();
    if (k > LVC_MAX_K - 1) k = LVC_MAX_K - 1;
    for (long long pass = 0; pass <= data_len; pass++) {
        const int plen = (int)(data_len - o.extra_after), tlen = plen + LVC_MAX_K;
        AGCOut r;
        if (plen >= 3 * (2 * k + 1)) {                                                                   // AffineGapVectorized.cpp:1068-1079
            // (2k + 1 <= 64 positions per segment: the vectors of a segment side by side; wider bands -- k >= 32 -- one vector at a time)
            const bool have_pre = pre != nullptr && pass == 0 && extra_before == 0 && k <= SAMF_PRE_MAX_W && first_u32((uint32_t)pre->valid) == 1u &&
                                  (int)first_u32((uint32_t)pre->plen) == plen && (int)first_u32((uint32_t)pre->w) == k && (long long)first_u64((uint64_t)pre->loc) == loc;
            r = have_pre ? agc_from_pre(s, plen, k, pre, use_m, ops, ops_cap)
              : k <= 31 ? agc_banded_par(s, plen, tlen, k, AGC_MAX_READ_LENGTH, use_m, ops, ops_cap)
                        : agc_banded(s, plen, tlen, k, AGC_MAX_READ_LENGTH, use_m, ops, ops_cap);
            WAVE_SYNC();
            if (r.n_edits < 0 || r.n_edits > k || r.tail_ins >= plen) r = agc_full(s, plen, tlen, use_m, ops, ops_cap);     // "failed band"
        } else r = agc_full(s, plen, tlen, use_m, ops, ops_cap);
        WAVE_SYNC();
        if (r.stale) o.stale = 1;
        o.edit_distance = r.n_edits; o.n_ops = r.n_edits < 0 ? 0 : r.n_ops; o.tail_ins = r.tail_ins; o.add_front_clipping = 0;
        if (r.n_edits >= 0 && r.n_ops > 0) {                                                              // :1089-1101
            const uint32_t op0 = first_u32(ops[0]);
            if ((op0 & 0xfu) == LVC_OP_D) {
                o.add_front_clipping = (int)(op0 >> 4);
                if (o.add_front_clipping != 0) { o.edit_distance = 0; o.n_ops = 0; }
            } else if ((op0 & 0xfu) == LVC_OP_I) o.add_front_clipping = -(int)(op0 >> 4);
        }
        const int net_indel = r.n_edits >= 0 ? r.net_del : 0;                                             // (*o_netDel; zeroed at the start of either form)
        if (pass == 0 && o.add_front_clipping != 0) return o;                                             // SAM.cpp:2546-2552
        long long nw = loc + data_len + net_indel - real_end; if (nw < 0) nw = 0;
        if (nw <= o.extra_after) return o;                                                                // :2562 (<=, unlike the LV variant)
        o.extra_after = nw;
    }
    return o;
}
