// ag_reg.h -- register/DPP formulation of the affine-gap emulation (same results as ag.h).
//
// ag.h keeps the DP rows in LDS in the reference's striped order and moves data between lanes
// through the LDS crossbar (ds_bpermute); a wave then spends its time waiting on ~100-cycle
// cross-lane round trips, ~3 ms per call.  Here the same arithmetic is laid out in *pattern
// order*: wavefront lane L of chunk c owns pattern position p = 64c + L for the whole call.
//   * In striped terms position p is SSE lane (p % segLen) / numVec, vector (p % segLen) % numVec,
//     and the striped code's "previous vector, same SSE lane" / "shifted last vector" inputs are
//     always simply position p-1.  H(i-1, p-1) therefore arrives with one DPP wave_shr:1 of the
//     register that holds the previous row -- no LDS, no bpermute.  H, H-1 and E live in VGPRs
//     (AG_MAXC chunks of 64 positions), so stale out-of-band cells and the reference's H/H-1
//     pointer swap are reproduced for free.
//   * The first pass's F chain restarts at every stripe (sub-segment of numVec positions).  With
//     g(p) = max(m-open,0) + p*ext + BIG*stripe(p), F is a plain 64-lane prefix max of g done with
//     six DPP steps (row_shr 1/2/4/8, row_bcast 15/31): the BIG term makes earlier stripes lose.
//   * Lazy F: the eight stripe-end F values sit in lanes 0-7 of one register; a round evaluates
//     every position at once (each lane knows its vector index, hence how far F has decayed when
//     the reference's walk reaches it).  The reference's "stop at the first vector where no SSE
//     lane still has F > H - open" becomes: stop vector = first vector index with no such lane,
//     found with a ballot (common case: vector 0) or a 64-bit LDS bitmap.
//   * Traceback bytes are written once per cell, after lazy F, 64 consecutive bytes per store.
// The choice between local and global alignment, the clipping heuristics and the traceback walk
// are the same scalar code as ag.h.  Patterns longer than 64*AG_MAXC - stripe slack use ag.h.
#pragma once
#include "ag.h"

#define AG_NEG (-(1 << 29))
#define AG_BIG (1 << 17)

template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ int ag_dpp_max(int v) {
    int t = __builtin_amdgcn_update_dpp(AG_NEG, v, CTRL, ROW_MASK, 0xF, false);
    return t > v ? t : v;
}
// inclusive prefix max over the 64 lanes (GFX9 DPP scan)
static __device__ __forceinline__ int ag_prefix_max(int v) {
    v = ag_dpp_max<0x111, 0xF>(v);      // row_shr:1
    v = ag_dpp_max<0x112, 0xF>(v);      // row_shr:2
    v = ag_dpp_max<0x114, 0xF>(v);      // row_shr:4
    v = ag_dpp_max<0x118, 0xF>(v);      // row_shr:8
    v = ag_dpp_max<0x142, 0xA>(v);      // row_bcast:15 -> rows 1,3
    v = ag_dpp_max<0x143, 0xC>(v);      // row_bcast:31 -> rows 2,3
    return v;
}
// lane L gets src[L-1]; lane 0 gets `lane0`
static __device__ __forceinline__ int ag_shr1(int lane0, int src) {
    return __builtin_amdgcn_update_dpp(lane0, src, 0x138, 0xF, 0xF, false);   // wave_shr:1
}

// packed per-position constants: k (vector index in stripe) | stripe-in-segment l | segment j | pattern base | valid
struct AGPos {
    uint32_t w;
    __device__ __forceinline__ int k() const { return (int)(w & 0x3FF); }
    __device__ __forceinline__ int l() const { return (int)((w >> 10) & 7); }
    __device__ __forceinline__ int j() const { return (int)((w >> 13) & 0xFF); }
    __device__ __forceinline__ int pb() const { return (int)((w >> 21) & 7); }      // 0..3 base, 4 N, 5 padding
    __device__ __forceinline__ bool valid() const { return (w >> 24) & 1; }
};

template <int AG_MAXC, typename PSeq, typename TSeq, typename QSeq>
static __device__ __forceinline__ AGResult ag_compute_reg(
    bool banded, int dir, const AGParams &prm, const PSeq &P, const QSeq &Q, int pattern_len,
    const TSeq &T, int text_len, int w, int score_init, bool is_rc, bool use_clipping,
    int16_t *lds_rows, uint8_t *bt_scratch, const DevTables *tab,
    int num_vec, int seg_len, int num_seg)
{
    const int lane = lane_id();
    AGResult res; res.ag_score = -1; res.text_offset = -1; res.pattern_offset = -1; res.n_edits = -1;
    res.match_probability = 1.0; res.stale_reads = 0;
    const int match = prm.match_reward, sub = -prm.sub_penalty;
    const int gap_open = prm.gap_open + prm.gap_extend, gap_ext = prm.gap_extend;
    const int tot = num_seg * seg_len;                  // positions that exist in the striped layout
    const int nch = (tot + 63) >> 6;
    const int row_stride = nch * 64;
    int *lds_f = (int *)lds_rows;                       // [8] stripe-entry F of the current lazy round
    unsigned long long *lds_bits = (unsigned long long *)(lds_f + 8);

    int end_bonus;
    if (!is_rc) end_bonus = dir == -1 ? prm.five_bonus : prm.three_bonus;
    else        end_bonus = dir == -1 ? prm.three_bonus : prm.five_bonus;

    AGPos pos[AG_MAXC];
    int Hp[AG_MAXC], Hm[AG_MAXC], E[AG_MAXC];
#pragma unroll
    for (int c = 0; c < AG_MAXC; c++) {
        const int p = c * 64 + lane;
        uint32_t wd = 0; int hv = 0;
        if (c < nch && p < tot) {
            int j = p / seg_len, r = p - j * seg_len, l = r / num_vec, k = r - l * num_vec;
            int pb = p < pattern_len ? (int)base_value(P(p)) : 5;
            wd = (uint32_t)k | ((uint32_t)l << 10) | ((uint32_t)j << 13) | ((uint32_t)pb << 21) | (1u << 24);
            // first row (:399-414 / :971-983) incl. the stale scoreFirstRow[] inheritance of padding lanes
            int vi = j * num_vec + k;
            for (int v = vi; v >= 0; v--) {
                int pi = (v / num_vec) * seg_len + l * num_vec + (v % num_vec);
                if (pi < pattern_len) { int x = score_init - gap_open - pi * gap_ext; hv = x > 0 ? x : 0; break; }
            }
        }
        pos[c].w = wd; Hp[c] = hv; Hm[c] = 0; E[c] = 0;
    }

    int best_global = -1, best_global_text = -1, best_local = -1, best_local_text = -1, best_local_pat = -1;

    for (int i = 0; i < text_len; i++) {
        const int tb = (int)base_value(T(i));
        int band_beg = 0, band_end = pattern_len - 1, seg_beg = 0, seg_end = 0;
        if (banded) {
            band_beg = i - w > 0 ? i - w : 0;
            band_end = i + w < pattern_len - 1 ? i + w : pattern_len - 1;
            seg_beg = band_beg / seg_len; seg_end = band_end / seg_len;
        }
        int h_init0 = score_init;
        if (i > 0) { int v = score_init - gap_open - (i - 1) * gap_ext; h_init0 = v > 0 ? v : 0; }
        int mxv = 0, X0 = 0, fin = 0;
        int btr[AG_MAXC], fo[AG_MAXC];
        bool did[AG_MAXC];
#pragma unroll
        for (int c = 0; c < AG_MAXC; c++) { btr[c] = 0; fo[c] = 0; did[c] = false; }

        for (int j = seg_beg; j <= seg_end; j++) {
            int nk = num_vec;
            if (banded) { int lim = band_end - j * seg_len + 1; if (lim < nk) nk = lim; }
            const int c_lo = (j * seg_len) >> 6;
            int c_hi = ((j + 1) * seg_len - 1) >> 6; if (c_hi > nch - 1) c_hi = nch - 1;
            const bool zero_seg_start = banded && j > 0 && band_beg > j * seg_len;

            // ---------------- first pass
            int carry = AG_NEG;
#pragma unroll
            for (int c = 0; c < AG_MAXC; c++) {
                if (c >= c_lo && c <= c_hi) {
                    const int p = c * 64 + lane;
                    const AGPos ps = pos[c];
                    const bool inseg = ps.valid() && ps.j() == j && ps.k() < nk;
                    int lane0_in = h_init0;
                    if (c > 0) lane0_in = __builtin_amdgcn_readlane(Hp[c > 0 ? c - 1 : 0], 63);
                    int h_in = ag_shr1(lane0_in, Hp[c]);
                    if (zero_seg_start && p == j * seg_len) h_in = 0;
                    const int pb = ps.pb();
                    int prof = pb == 5 ? -32768 : ((tb > 3 || pb > 3) ? -1 : (tb == pb ? match : sub));
                    int m = h_in > 0 ? ag_sat16(h_in + prof) : 0;
                    int e = E[c];
                    int bt = e > m ? 1 : 0;
                    int hp = m > e ? m : e;
                    int e2 = ag_sat16(e - gap_ext);
                    int tmp = ag_sat16(m - gap_open); if (tmp < 0) tmp = 0;
                    if (e2 > tmp) bt |= 4;
                    const int stripe = ps.j() * 8 + ps.l();
                    int g = inseg ? tmp + p * gap_ext + AG_BIG * stripe : AG_NEG;
                    int inc = ag_prefix_max(g);
                    int exc = ag_shr1(carry, inc);
                    int pm = exc > carry ? exc : carry;
                    const int k = ps.k();
                    const int fin_cell = ps.l() == 0 ? fin : 0;
                    int fk = fin_cell - k * gap_ext;
                    if (k >= 1) { int a = pm - AG_BIG * stripe - (p - 1) * gap_ext; fk = a > fk ? a : fk; }
                    if (inseg) {
                        if (fk > hp) { bt |= 2; hp = fk; }
                        Hm[c] = hp;
                        E[c] = e2 > tmp ? e2 : tmp;
                        mxv = hp > mxv ? hp : mxv;
                        int f2 = ag_sat16(fk - gap_ext);
                        if (f2 > tmp) bt |= 32;
                        fo[c] = f2 > tmp ? f2 : tmp;
                        btr[c] = bt; did[c] = true;
                    }
                    int last_inc = __builtin_amdgcn_readlane(inc, 63);
                    carry = last_inc > carry ? last_inc : carry;
                }
            }

            // ---------------- stripe-end F values -> lanes 0..7
            int fvec = 0;
#pragma unroll
            for (int l = 0; l < 8; l++) {
                const int pe = j * seg_len + l * num_vec + nk - 1;
                const int pc = pe >> 6, pl = pe & 63;
                int v = 0;
#pragma unroll
                for (int c = 0; c < AG_MAXC; c++) if (c == pc) v = __builtin_amdgcn_readlane(fo[c], pl);
                if (lane == l) fvec = v;
            }

            // ---------------- lazy F
            const int rounds = banded ? 7 : 8;
            for (int r = 0; r < rounds; r++) {
                if (banded) { int f7 = __builtin_amdgcn_readlane(fvec, 7); if (f7 > X0) X0 = f7; }
                fvec = ag_shr1(0, fvec);
                if (lane < 8) lds_f[lane] = fvec;
                if (lane == 8) *lds_bits = 0ull;
                WAVE_SYNC();
                int fj[AG_MAXC]; bool cont[AG_MAXC]; bool ins[AG_MAXC];
                bool any_cont = false;
#pragma unroll
                for (int c = 0; c < AG_MAXC; c++) {
                    fj[c] = 0; cont[c] = false; ins[c] = false;
                    if (c >= c_lo && c <= c_hi) {
                        const AGPos ps = pos[c];
                        ins[c] = ps.valid() && ps.j() == j && ps.k() < nk;
                        int fv = lds_f[ps.l()];
                        int f = fv - ps.k() * gap_ext; if (f < 0) f = 0;
                        int hn = Hm[c] > f ? Hm[c] : f;
                        int t2 = hn > gap_open ? hn - gap_open : 0;
                        int f2 = f > gap_ext ? f - gap_ext : 0;
                        fj[c] = f;
                        cont[c] = ins[c] && (f2 > t2);
                        if (__ballot(cont[c])) any_cont = true;
                    }
                }
                int jstar = 0;
                if (any_cont) {
#pragma unroll
                    for (int c = 0; c < AG_MAXC; c++)
                        if (c >= c_lo && c <= c_hi && cont[c]) atomicOr(lds_bits, 1ull << pos[c].k());
                    WAVE_SYNC();
                    unsigned long long bits = *lds_bits;
                    bits = first_u64(bits);
                    jstar = (~bits == 0ull) ? 64 : (__ffsll((long long)~bits) - 1);
                }
                const bool round_complete = jstar >= nk;        // never converged in this round
                const int jlim = round_complete ? nk - 1 : jstar;
#pragma unroll
                for (int c = 0; c < AG_MAXC; c++) {
                    if (c >= c_lo && c <= c_hi) {
                        if (ins[c] && pos[c].k() <= jlim) {
                            if (fj[c] > Hm[c]) { btr[c] |= 2; Hm[c] = fj[c]; }
                            mxv = Hm[c] > mxv ? Hm[c] : mxv;
                            if (cont[c]) btr[c] |= 32;
                        }
                    }
                }
                WAVE_SYNC();
                if (!round_complete) break;
                // F after walking all nk vectors of the stripe
                int dec = fvec - nk * gap_ext; fvec = dec > 0 ? dec : 0;
            }
            fin = banded ? X0 : 0;
        }

        // ---------------- traceback bytes, row max, bookkeeping
        uint8_t *bt_row = bt_scratch + (size_t)i * row_stride;
#pragma unroll
        for (int c = 0; c < AG_MAXC; c++) if (c < nch && did[c]) bt_row[c * 64 + lane] = (uint8_t)btr[c];
        const int max_row = __builtin_amdgcn_readlane(ag_prefix_max(mxv), 63);

        if (!banded || band_end == pattern_len - 1) {
            const int pe = pattern_len - 1, pc = pe >> 6, pl = pe & 63;
            int gscore = 0;
#pragma unroll
            for (int c = 0; c < AG_MAXC; c++) if (c == pc) gscore = __builtin_amdgcn_readlane(Hm[c], pl);
            if (gscore >= best_global) { best_global = gscore; best_global_text = i; }
        }
        if (max_row == 0) break;
        if (max_row > best_local) {
            int off = -1;
#pragma unroll
            for (int c = AG_MAXC - 1; c >= 0; c--) {
                if (c < nch && off < 0) {
                    unsigned long long mk = __ballot(did[c] && Hm[c] == max_row);
                    if (mk) off = c * 64 + 63 - __clzll((long long)mk);
                }
            }
            best_local_pat = off; best_local = max_row; best_local_text = i;
        }
#pragma unroll
        for (int c = 0; c < AG_MAXC; c++) { int t = Hm[c]; Hm[c] = Hp[c]; Hp[c] = t; }
    }
    WAVE_SYNC();

    // ---------------- local vs global (:643-730 / :1163-1251)
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
            if (pat_off == best_local_pat && text_off == best_local_text) {
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
            bool computed = true;
            if (banded) {
                int bb = row - w > 0 ? row - w : 0, be = row + w < pattern_len - 1 ? row + w : pattern_len - 1;
                int cj = col / seg_len, ck = (col % seg_len) % num_vec;
                computed = cj >= bb / seg_len && cj <= be / seg_len && cj * seg_len + ck <= be;
            }
            int bits = computed ? (int)first_u32(bt_scratch[(size_t)row * row_stride + col]) : 0;
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

// Striped-layout dimensions of one problem (AffineGapVectorized.h:339-342 / :914-915).
static __host__ __device__ __forceinline__ void ag_dims(bool banded, int pattern_len, int w, int *num_vec, int *seg_len, int *num_seg) {
    if (banded) {
        int bw = (2 * w + 1) < pattern_len ? (2 * w + 1) : pattern_len;
        *num_vec = (bw + 7) >> 3; *seg_len = *num_vec * 8; *num_seg = (pattern_len + *seg_len - 1) / *seg_len;
    } else {
        *num_vec = (pattern_len + 7) >> 3; *seg_len = *num_vec * 8; *num_seg = 1;
    }
}

// Largest number of striped positions any call can need when patterns are at most max_pattern
// long and limits at most max_w (banded is only chosen when pattern_len >= 3*(2w+1),
// BaseAligner.cpp:821,1213).  The host picks the kernel variant (chunks of 64 positions) from it.
static __host__ __forceinline__ int ag_max_positions(int max_pattern, int max_w) {
    int best = 0;
    for (int pl = 1; pl <= max_pattern; pl++) {
        int nv, sl, ns;
        ag_dims(false, pl, 0, &nv, &sl, &ns);
        if (ns * sl > best) best = ns * sl;
        for (int w = 0; w <= max_w; w++) {
            if (pl >= 3 * (2 * w + 1)) { ag_dims(true, pl, w, &nv, &sl, &ns); if (ns * sl > best) best = ns * sl; }
        }
    }
    return best;
}

// AGC > 0: register formulation with AGC chunks of 64 positions (the host guarantees it fits);
// AGC == 0: the LDS formulation of ag.h (any pattern length up to RL).
template <int AGC, typename PSeq, typename TSeq, typename QSeq>
static __device__ __forceinline__ AGResult ag_dispatch(
    bool banded, int dir, const AGParams &prm, const PSeq &P, const QSeq &Q, int pattern_len,
    const TSeq &T, int text_len, int w, int score_init, bool is_rc, bool use_clipping,
    int16_t *lds_rows, uint8_t *bt_scratch, uint32_t RL, const DevTables *tab)
{
    if constexpr (AGC > 0) {
        AGResult res; res.ag_score = -1; res.text_offset = -1; res.pattern_offset = -1; res.n_edits = -1;
        res.match_probability = 0.0; res.stale_reads = 0;
        int ww = w > 126 ? 126 : w;
        if (ww < 0) return res;                                           // :325 / :890
        int num_vec, seg_len, num_seg;
        ag_dims(banded, pattern_len, ww, &num_vec, &seg_len, &num_seg);
        if (num_seg * seg_len > 64 * AGC || num_vec > 1023 || num_seg > 255 ||
            (size_t)text_len * (size_t)(((num_seg * seg_len + 63) >> 6) * 64) > ag_scratch_bytes(RL)) {
            __builtin_trap();                                             // host sizing bug: fail loudly
        }
        return ag_compute_reg<AGC>(banded, dir, prm, P, Q, pattern_len, T, text_len, ww, score_init, is_rc, use_clipping,
                                   lds_rows, bt_scratch, tab, num_vec, seg_len, num_seg);
    } else {
        return ag_compute(banded, dir, prm, P, Q, pattern_len, T, text_len, w, score_init, is_rc, use_clipping,
                          lds_rows, bt_scratch, RL, tab);
    }
}
