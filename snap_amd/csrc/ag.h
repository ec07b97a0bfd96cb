// ag.h -- affine-gap scoring, one problem per wavefront.
//
// Restates AffineGapVectorized<TEXT_DIRECTION>::computeScore (SNAPLib/AffineGapVectorized.h:821-1339)
// and computeScoreBanded (:256-819); semantics collected in SURVEY.md Appendix A.4.
//
// The reference is Farrar's striped Smith-Waterman on 8 x int16 SSE2 lanes with a per-cell
// 6-bit traceback word, a speculative first pass that ignores F across stripe boundaries and
// a "lazy F" fix-up with an all-lanes early exit.  Which traceback bits end up set (and hence
// nEdits, clipping and matchProbability) is *defined* by that procedure, so this is an
// emulation of the striped algorithm, not a textbook Gotoh DP.  GPU mapping:
//   * a row's first pass is evaluated for 8 vectors x 8 SSE lanes = 64 cells per step, one
//     cell per wavefront lane (lane = 8*vector + sselane), so H/E state is read from LDS at
//     consecutive addresses;
//   * the only sequential dependence of the first pass, F along a stripe
//     (F[k+1] = max(F[k] - ext, max(m[k] - open, 0))), is a max-plus recurrence and is
//     evaluated in closed form with a segmented prefix-max across the 8 vectors of a step
//     (three __shfl_up rounds) plus a running carry; no saturation can occur on it because
//     F >= 0 and scores stay far below 32767;
//   * the lazy-F loop is executed literally on lanes 0-7 (it almost always exits at its
//     first vector), touching the traceback bytes only when it changes them;
//   * traceback bytes (one per cell) stream to a per-wave slab of HBM scratch with 64-byte
//     coalesced stores; H/H-1/E rows live in LDS.
// Like the reference, the banded variant can trace back through cells outside what this call
// computed; the reference then reads whatever an earlier call left in its object, i.e. its
// result is not a function of its inputs.  (Measured with oracle/snap_oracle.c: only for
// w <= 1-2 with many edits, outside what AlignRead asks for.)  Here such cells read as 0, so
// results never depend on what a wave processed before.
#pragma once
#include "dev_common.h"

struct AGParams { int match_reward, sub_penalty, gap_open, gap_extend, five_bonus, three_bonus; };

struct AGResult {
    int    ag_score;         // -1 when no alignment scores above score_init
    int    text_offset;
    int    pattern_offset;
    int    n_edits;
    double match_probability;
    int    stale_reads;      // traceback steps through cells this call never computed (reference result undefined)
};

// cells per row rounded up: a (possibly banded) row has at most pattern_len + seg_len cells
static __host__ __device__ __forceinline__ uint32_t ag_row_cells(uint32_t RL) { return ((RL + 7) / 8) * 8 + 512; }
// LDS: H, H-1, E rows of int16
static __host__ __device__ __forceinline__ uint32_t ag_lds_bytes(uint32_t RL) { return 3 * ag_row_cells(RL) * 2; }
// The register forms (AGC chunks of 64 striped positions, ag_reg.h / ag_win.h) keep their rows in VGPRs; their LDS is tables only: the
// windowed form's first-row values (2 B), pattern codes (1 B) per striped position and the text's codes, or the register form's stripe-end
// table, vector tags (4 B per vector) and text codes -- well under a quarter of the three int16 rows of the LDS form.
static __host__ __device__ __forceinline__ uint32_t ag_lds_bytes_reg(uint32_t RL, uint32_t agc) {
    const uint32_t tot = 64 * agc, text = RL + 128;
    const uint32_t win = 16 + 2 * tot + ((tot + 15) & ~15u) + text;            // ag_banded_win: fr16, pcode, tcode
    const uint32_t reg = 64 + 4 * (tot / 8 + 8) + text;                        // ag_compute_reg: lds_end, lds_flag[num_vec], tcode
    return ((win > reg ? win : reg) + 63) & ~63u;
}
// HBM scratch: one byte per cell, (RL + MAX_K) rows
static __host__ __device__ __forceinline__ size_t ag_scratch_bytes(uint32_t RL) {
    return (size_t)(RL + 128) * ag_row_cells(RL);
}

static __device__ __forceinline__ int ag_sat16(int x) { return x > 32767 ? 32767 : (x < -32768 ? -32768 : x); }

static __device__ __forceinline__ int wave_max_i32(int v) {
    for (int o = 32; o >= 1; o >>= 1) { int t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}

// EXACT: the replay mode of flagged reads (DESIGN.md "Reference nondeterminism"): bt_scratch is then the wave's image of ONE reference
// object's backtraceAction array (AffineGapVectorized.h:1374) -- same flat addressing (row * numVec * numSeg + vector, SSE element), zeroed
// when the read starts and kept across the calls of the read -- and the traceback reads whatever that array holds, so a step outside the
// band of this call sees what an EARLIER call for the same read left there, exactly as a newly constructed reference aligner does.
template <bool EXACT = false, typename PSeq, typename TSeq, typename QSeq>
static __device__ __forceinline__ AGResult ag_compute(
    bool banded, int dir, const AGParams &prm, const PSeq &P, const QSeq &Q, int pattern_len,
    const TSeq &T, int text_len, int w, int score_init, bool is_rc, int use_clipping,
    int16_t *lds_rows, uint8_t *bt_scratch, uint32_t RL, const DevTables *tab, uint32_t bt_tag = 0)
{
    const int lane = lane_id();
    const int sl = lane & 7;            // SSE lane emulated by this wavefront lane
    const int kk = lane >> 3;           // vector within the current step of 8 vectors
    AGResult res; res.ag_score = -1; res.text_offset = -1; res.pattern_offset = -1; res.n_edits = -1; res.match_probability = 0.0;
    res.stale_reads = 0;
    if (w > 126) w = 126;
    if (w < 0) return res;                                             // :325 / :890
    res.match_probability = 1.0;

    const int match = prm.match_reward, sub = -prm.sub_penalty;        // init(), :105-133
    const int gap_open = prm.gap_open + prm.gap_extend, gap_ext = prm.gap_extend;
    int num_vec, seg_len, num_seg;
    if (banded) {
        int bw = (2 * w + 1) < pattern_len ? (2 * w + 1) : pattern_len;   // :339-342
        num_vec = (bw + 7) >> 3; seg_len = num_vec * 8; num_seg = (pattern_len + seg_len - 1) / seg_len;
    } else {
        num_vec = (pattern_len + 7) >> 3; seg_len = num_vec * 8; num_seg = 1;   // :914-915
    }
    const int nv_tot = num_vec * num_seg;
    const int row_cells = nv_tot * 8;
    const int cells_cap = (int)ag_row_cells(RL);
    if (row_cells > cells_cap || text_len > (int)RL + 128) return res;   // (cannot happen for pattern_len <= RL)
    int16_t *Hp = lds_rows, *Hm = lds_rows + cells_cap, *E = lds_rows + 2 * cells_cap;

    int end_bonus;                                                     // :380-394 / :950-966
    if (!is_rc) end_bonus = dir == -1 ? prm.five_bonus : prm.three_bonus;
    else        end_bonus = dir == -1 ? prm.three_bonus : prm.five_bonus;

    // ---- first row (:399-414 / :971-983).  scoreFirstRow[] is not reset between vectors, so a
    // padding lane inherits the value its SSE lane had in the last vector that was in range.
    for (int c0 = 0; c0 < row_cells; c0 += WAVE) {
        int c = c0 + lane;
        if (c < row_cells) {
            int vi = c >> 3;
            // walk back to the most recent vector whose cell for this SSE lane is inside the pattern
            int val = 0;
            for (int v = vi; v >= 0; v--) {
                int pi = (v / num_vec) * seg_len + sl * num_vec + (v % num_vec);
                if (pi < pattern_len) { int x = score_init - gap_open - pi * gap_ext; val = x > 0 ? x : 0; break; }
            }
            Hp[c] = (int16_t)val; Hm[c] = 0; E[c] = 0;
        }
    }
    WAVE_SYNC();

    int best_global = -1, best_global_text = -1, best_local = -1, best_local_text = -1, best_local_pat = -1;

    for (int i = 0; i < text_len; i++) {
        const int tb = (int)base_value(T(i));
        int band_beg = 0, band_end = pattern_len - 1, seg_beg = 0, seg_end = 0;
        if (banded) {
            band_beg = i - w > 0 ? i - w : 0;
            band_end = i + w < pattern_len - 1 ? i + w : pattern_len - 1;
            seg_beg = band_beg / seg_len; seg_end = band_end / seg_len;
        }
        uint8_t *bt_row = bt_scratch + (size_t)i * row_cells;
        int mx = 0;                       // per-lane running max of h (row max after a wave reduction)
        int X0 = 0;                       // lane 0 of the reference's X vector (F carried into the next segment)
        int fin = 0;                      // F entering the segment, per SSE lane (only lane 0 can be non-zero)

        for (int j = seg_beg; j <= seg_end; j++) {
            int nk = num_vec;
            if (banded) { int lim = band_end - j * seg_len + 1; if (lim < nk) nk = lim; }
            int h_init;
            if (j == 0) {
                h_init = score_init;
                if (i > 0) { int v = score_init - gap_open - (i - 1) * gap_ext; h_init = v > 0 ? v : 0; }
            } else if (band_beg > j * seg_len) {
                h_init = 0;
            } else {
                h_init = Hp[(j * num_vec - 1) * 8 + 7];
            }
            const int vbase = j * num_vec;
            int carry = -(1 << 28);          // prefix max of g over vectors already processed (per SSE lane)
            int f_last = sl == 0 ? fin : 0;  // becomes F after the last vector of the segment
            const int fin_l = sl == 0 ? fin : 0;

            // ---- first pass, 8 vectors x 8 SSE lanes per step
            for (int q0 = 0; q0 < nk; q0 += 8) {
                const int k = q0 + kk;
                const bool live = k < nk;
                const int vi = vbase + k;
                const int cell = vi * 8 + sl;
                int hp = 0, tmp = 0, bt = 0, g = -(1 << 28);
                if (live) {
                    int h_in;
                    if (k == 0) h_in = (sl == 0) ? h_init : (int)Hp[(vbase + num_vec - 1) * 8 + sl - 1];
                    else        h_in = (int)Hp[cell - 8];
                    int pi = j * seg_len + sl * num_vec + k;
                    int prof;
                    if (pi < pattern_len) {
                        int pb = (int)base_value(P(pi));
                        prof = (tb > 3 || pb > 3) ? -1 : (tb == pb ? match : sub);
                    } else prof = -32768;
                    int m = h_in > 0 ? ag_sat16(h_in + prof) : 0;
                    int e = (int)E[cell];
                    bt = e > m ? 1 : 0;
                    hp = m > e ? m : e;
                    int e2 = ag_sat16(e - gap_ext);
                    tmp = ag_sat16(m - gap_open); if (tmp < 0) tmp = 0;
                    if (e2 > tmp) bt |= 4;
                    E[cell] = (int16_t)(e2 > tmp ? e2 : tmp);
                    g = tmp + k * gap_ext;
                }
                // inclusive prefix max of g over the vectors of this step (same SSE lane = lanes 8 apart)
                int inc = g, t;
                t = __shfl_up(inc, 8);  if (kk >= 1) inc = t > inc ? t : inc;
                t = __shfl_up(inc, 16); if (kk >= 2) inc = t > inc ? t : inc;
                t = __shfl_up(inc, 32); if (kk >= 4) inc = t > inc ? t : inc;
                int exc = __shfl_up(inc, 8); if (kk == 0) exc = -(1 << 28);
                int pm = exc > carry ? exc : carry;                     // max_{i<k} g_i
                if (live) {
                    int fk = fin_l - k * gap_ext;
                    if (k >= 1) { int a = pm - (k - 1) * gap_ext; fk = a > fk ? a : fk; }
                    if (fk > hp) { bt |= 2; hp = fk; }
                    Hm[cell] = (int16_t)hp;
                    mx = hp > mx ? hp : mx;
                    int f2 = ag_sat16(fk - gap_ext);
                    if (f2 > tmp) bt |= 32;
                    bt_row[cell] = (uint8_t)(bt | (EXACT ? (int)bt_tag : 0));
                }
                // carry the prefix max past this step: take it from the last vector row of the step
                int last_inc = __shfl(inc, 56 + sl);
                carry = last_inc > carry ? last_inc : carry;
            }
            // F after the last computed vector of the segment
            {
                int a = carry - (nk - 1) * gap_ext, b = fin_l - nk * gap_ext;
                f_last = a > b ? a : b;
                if (nk == 0) f_last = fin_l;
            }
            WAVE_SYNC();

            // ---- lazy F (:1080-1112 full: 8 rounds; :534-569 banded: 7 rounds + segment carry X), lanes 0-7
            {
                int f = f_last;                                   // meaningful on lanes 0..7 (sl == lane there)
                const int rounds = banded ? 7 : 8;
                bool converged = false;
                for (int r = 0; r < rounds && !converged; r++) {
                    if (banded) { int f7 = __shfl(f, 7); if (f7 > X0) X0 = f7; }
                    int up = __shfl_up(f, 1);
                    f = lane == 0 ? 0 : up;
                    for (int v = 0; v < nk; v++) {
                        const int cell = (vbase + v) * 8 + lane;
                        int add = 0, tmp = 0, f2 = 0;
                        bool any = false;
                        if (lane < 8) {
                            int hv = (int)Hm[cell];
                            if (f > hv) { add |= 2; hv = f; Hm[cell] = (int16_t)hv; }
                            mx = hv > mx ? hv : mx;
                            tmp = hv > gap_open ? hv - gap_open : 0;       // _mm_subs_epu16 on non-negative values
                            f2 = f > gap_ext ? f - gap_ext : 0;
                            if (f2 > tmp) { add |= 32; any = true; }
                            f = f2;
                        }
                        uint64_t chg = BALLOT(add != 0);
                        if (chg) {
                            if (add != 0) bt_row[cell] = (uint8_t)(bt_row[cell] | add);
                            WAVE_SYNC();
                        }
                        if (!BALLOT(any)) { converged = true; break; }
                    }
                }
                WAVE_SYNC();
            }
            fin = banded ? X0 : 0;                                // "f = X" (:571-572)
        }

        const int max_row = wave_max_i32(mx);
        if (!banded || band_end == pattern_len - 1) {                 // :593-606 / :1125-1131
            int pe = pattern_len - 1, vi, li;
            if (banded) { vi = (pe / seg_len) * num_vec + (pe % seg_len) % num_vec; li = (pe % seg_len) / num_vec; }
            else { vi = pe % num_vec; li = pe / num_vec; }
            int gscore = (int)Hm[vi * 8 + li];
            if (gscore >= best_global) { best_global = gscore; best_global_text = i; }
        }
        if (max_row == 0) break;
        if (max_row > best_local) {
            int off = -1;
            for (int j = seg_beg; j <= seg_end; j++) {
                int nk = num_vec;
                if (banded) { int lim = band_end - j * seg_len + 1; if (lim < nk) nk = lim; }
                for (int q0 = 0; q0 < nk; q0 += 8) {
                    int k = q0 + kk;
                    if (k < nk) {
                        int cell = (j * num_vec + k) * 8 + sl;
                        if ((int)Hm[cell] == max_row) { int po = j * seg_len + sl * num_vec + k; off = po > off ? po : off; }
                    }
                }
            }
            best_local_pat = wave_max_i32(off);
            best_local = max_row; best_local_text = i;
        }
        int16_t *tsw = Hm; Hm = Hp; Hp = tsw;
        WAVE_SYNC();
    }

    // ---- local vs global (:643-730 / :1163-1251); everything below is wave-uniform
    int score, pat_off, text_off;
    if (best_local != best_global && best_local >= best_global + end_bonus) {
        pat_off = best_local_pat; text_off = best_local_text; score = best_local;
        if (use_clipping) {
            int pa = pat_off - 1, ta = text_off, cnt = 0;
            while (pa + 1 != pattern_len && P(pa + 1) == T(ta + 1)) { cnt++; pa++; ta++; }
            if (cnt >= 3) { pat_off = pa; text_off = ta; }
            else {
                pa = pat_off + 1; ta = text_off; cnt = 0;
                while (pa < pattern_len && P(pa) == T(ta)) { cnt++; pa++; ta++; }
                if (cnt >= 3) { pat_off = pa - 1; text_off = ta - 1; }
            }
            if (use_clipping != 2 && pat_off == best_local_pat && text_off == best_local_text) {   // 2 = useAltLiftover: no quality-aware step (:1212)
                pa = pat_off;
                while (pa != pattern_len - 1 && Q(pa) >= 65 && Q(pa + 1) >= 65) pa++;
                if (pa == pattern_len - 1) pat_off = pa;
                else if (pa >= pat_off + 2) {
                    int tmp_off = pa + 1, cnt_hq = 0, rem = pattern_len - tmp_off;
                    while (tmp_off != pattern_len - 1) { if (Q(tmp_off) >= 65) cnt_hq++; tmp_off++; }
                    if (((float)cnt_hq) / (float)rem < 0.1f) pat_off = pa;
                }
            }
        }
    } else {
        pat_off = pattern_len - 1; text_off = best_global_text; score = best_global;
    }
    res.text_offset = text_off; res.pattern_offset = pat_off;

    if (score > score_init) {                                          // traceback, :732-815 / :1253-1335
        double prob = 1.0;
        int row = text_off, col = pat_off;
        int action = 0, prev_action = 0, action_count = 1, n_matches = 0, n_mismatches = 0, n_gaps = 0;
        while (row >= 0 && col >= 0) {
            int vi, li;
            if (banded) { vi = (col / seg_len) * num_vec + (col % seg_len) % num_vec; li = (col % seg_len) / num_vec; }
            else { vi = col % num_vec; li = col / num_vec; }
            bool computed = true;
            if (banded) {   // was this cell inside what row `row` evaluated?  (:447-452, :483)
                int bb = row - w > 0 ? row - w : 0, be = row + w < pattern_len - 1 ? row + w : pattern_len - 1;
                int cj = col / seg_len, ck = (col % seg_len) % num_vec;
                computed = cj >= bb / seg_len && cj <= be / seg_len && cj * seg_len + ck <= be;
            }
            int bits = (EXACT || computed) ? (int)first_u32(bt_scratch[(size_t)row * row_cells + vi * 8 + li]) : 0;
            if constexpr (EXACT) bits = bt_cell(bits, bt_tag);             // (a cell of an earlier read is a zeroed cell: dev_common.h)
            if (!computed) res.stale_reads++;
            action = (bits >> (action << 1)) & 3;
            if (action == 0) {
                if (P(col) != T(row)) { prob *= tab->phred[Q(col)]; n_mismatches++; }
                else n_matches++;
                row--; col--;
            } else if (action == 1) {
                row--;
            } else {
                col--; action = 2;
            }
            if (prev_action != 0) {
                if (prev_action == action) action_count++;
                else { n_gaps += action_count; prob *= tab->indel[action_count]; action_count = 1; }
            }
            prev_action = action;
        }
        if (row >= 0) { action_count = row + 1; n_gaps += action_count; prob *= tab->indel[action_count]; }
        if (col >= 0) { action_count = col + 1; n_gaps += action_count; prob *= tab->indel[action_count]; }
        res.n_edits = n_mismatches + n_gaps;
        prob *= tab->perfect[n_matches];
        text_off += 1; pat_off += 1;
        res.text_offset = pattern_len - text_off;
        res.pattern_offset = pattern_len - pat_off;
        prob *= tab->indel[res.pattern_offset];
        res.match_probability = prob;
        res.ag_score = score;
    }
    return res;
}
