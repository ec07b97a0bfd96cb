// ag_reg.h -- register/DPP formulation of the affine-gap emulation (same results as ag.h).
//
// ag.h keeps the DP rows in LDS in the reference's striped order and moves data between lanes
// through the LDS crossbar (ds_bpermute); a wave then spends its time waiting on ~100-cycle
// cross-lane round trips.  Here the same arithmetic is laid out in *pattern order*: wavefront
// lane L of chunk c owns pattern position p = 64c + L for the whole call.
//   * In striped terms position p is SSE lane (p % segLen) / numVec, vector (p % segLen) % numVec,
//     and the striped code's "previous vector, same SSE lane" / "shifted last vector" inputs are
//     always simply position p-1.  H(i-1, p-1) therefore arrives with one DPP wave_shr:1 of the
//     register that holds the previous row -- no LDS, no bpermute.  H, H-1 and E live in VGPRs
//     (AGC chunks of 64 positions), so stale out-of-band cells and the reference's H/H-1 pointer
//     swap are reproduced for free.
//   * The first pass's F chain restarts at every stripe (numVec consecutive positions).  With
//     g(p) = max(m-open,0) + p*ext + BIG*stripe(p), F is a plain 64-lane prefix max of g done with
//     six DPP steps (row_shr 1/2/4/8, row_bcast 15/31): the BIG term makes earlier stripes lose.
//   * Lazy F: "the F that leaves stripe l enters stripe l+1" is the same kind of tagged prefix
//     max (stripe-end cells publish value + BIG*(l+1), every lane picks up the entry tagged with
//     its own stripe).  A round evaluates every position at once -- each lane knows its vector
//     index, hence how far F has decayed when the reference's walk reaches it -- and the
//     reference's "stop at the first vector where no SSE lane still has F > H - open" becomes
//     "first vector index with no such lane": a ballot in the common case, a 64-bit LDS bitmap
//     otherwise.
//   * Traceback bytes are written once per cell, after lazy F, 64 consecutive bytes per store,
//     and read back 64 diagonal steps per load (one round trip per gap-free stretch).
// The choice between local and global alignment and the clipping heuristics are the scalar code
// of ag.h.
#pragma once
#include "ag.h"

#define AG_NEG (-(1 << 29))
#define AG_BIG (1 << 17)
#define AG_HUGE (1 << 28)
// lane L gets src[L-1]; lane 0 gets 0 (bound_ctrl): foldable into the consumer, no `old` register to set up
static __device__ __forceinline__ int ag_shr1z(int src) {
    return __builtin_amdgcn_update_dpp(0, src, 0x138, 0xF, 0xF, true);   // wave_shr:1 bound_ctrl:1
}

template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ int ag_dpp_max(int v) {
    // old == INT_MIN, the identity of max: lanes without a source (or rows the mask leaves out) get it, so the result there is v -- and
    // this is the shape LLVM's DPP combiner folds into ONE v_max_i32_dpp.  (With old == v, as this read until round 4, every step came out
    // as v_mov + v_mov_dpp + v_max: 18 VALU instructions for the row maximum instead of 6.)
    int t = __builtin_amdgcn_update_dpp((int)0x80000000, v, CTRL, ROW_MASK, 0xF, false);
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

// EXACT: see ag_win.h / ag.h -- cells at the reference's own byte addresses in a traceback array that persists over the calls of a read.
template <int AGC, bool BANDED, bool EXACT = false, typename PSeq, typename TSeq, typename QSeq>
static __device__ __forceinline__ AGResult ag_compute_reg(
    int dir, const AGParams &prm, const PSeq &P, const QSeq &Q, int pattern_len,
    const TSeq &T, int text_len, int w, int score_init, bool is_rc, int use_clipping,
    int16_t *lds_rows, uint8_t *bt_scratch, const DevTables *tab,
    int num_vec, int seg_len, int num_seg, uint32_t bt_bytes, uint32_t bt_tag = 0)
{
    const int lane = lane_id();
    AGResult res; res.ag_score = -1; res.text_offset = -1; res.pattern_offset = -1; res.n_edits = -1;
    res.match_probability = 1.0; res.stale_reads = 0;
    const int match = prm.match_reward, sub = -prm.sub_penalty;
    const int gap_open = prm.gap_open + prm.gap_extend, gap_ext = prm.gap_extend;
    const int tot = num_seg * seg_len;                  // positions that exist in the striped layout
    const int nch = (tot + 63) >> 6;
    const int row_stride = EXACT ? tot : nch * 64;      // EXACT: numVec * numSeg * 8 bytes per row, as in the reference
    const BtSink sink = bt_sink(bt_scratch, bt_bytes);
    // LDS scratch of the lazy-F rounds: F leaving each of the 8 stripes in the first pass, and one "some lane of vector k
    // continues" tag per vector (tags instead of a bitmap: no clearing, no atomics)
    LDS_AS int *lds_end = (LDS_AS int *)lds_rows + 2;                  // [8]
    LDS_AS uint32_t *lds_flag = (LDS_AS uint32_t *)lds_rows + 16;      // [num_vec <= 1023]
    uint32_t flag_tag = 0;
    for (int kz = lane_id(); kz < num_vec; kz += WAVE) lds_flag[kz] = 0;     // tags of an earlier call must not look current
    // base codes of the text, staged once: a per-row read of T(i) is a FLAT load of a generic pointer, and its wait (vmcnt) is also a
    // wait for the previous row's traceback store.  (64 + 4 * num_vec + text_len <= 384 + RL < ag_lds_bytes(RL) for every AGC > 0.)
    LDS_AS uint8_t *tcode = (LDS_AS uint8_t *)(lds_flag + num_vec);
    for (int i0 = 0; i0 < text_len; i0 += WAVE) {
        const int i = i0 + lane;
        if (i < text_len) tcode[i] = (uint8_t)base_value(T(i));
    }
    WAVE_SYNC();

    int end_bonus;
    if (!is_rc) end_bonus = dir == -1 ? prm.five_bonus : prm.three_bonus;
    else        end_bonus = dir == -1 ? prm.three_bonus : prm.five_bonus;

    AGPos pos[AGC];
    int Hp[AGC], Hm[AGC], E[AGC];
#pragma unroll
    for (int c = 0; c < AGC; c++) {
        const int p = c * 64 + lane;
        uint32_t wd = 0; int hv = 0;
        if (c < nch && p < tot) {
            int j = p / seg_len, r = p - j * seg_len, l = r / num_vec, k = r - l * num_vec;
            int pb = p < pattern_len ? (int)base_value(P(p)) : 5;
            wd = (uint32_t)k | ((uint32_t)l << 10) | ((uint32_t)j << 13) | ((uint32_t)pb << 21) | (1u << 24);
            // first row (:399-414 / :971-983) incl. the stale scoreFirstRow[] inheritance of padding lanes
            if (p < pattern_len) {
                int x = score_init - gap_open - p * gap_ext; hv = x > 0 ? x : 0;
            } else {
                int vi = j * num_vec + k;
                for (int v = vi - 1; v >= 0; v--) {
                    int pi = (v / num_vec) * seg_len + l * num_vec + (v % num_vec);
                    if (pi < pattern_len) { int x = score_init - gap_open - pi * gap_ext; hv = x > 0 ? x : 0; break; }
                }
            }
        }
        pos[c].w = wd; Hp[c] = hv; Hm[c] = 0; E[c] = 0;
    }

    int best_global = -1, best_global_text = -1, best_local = -1, best_local_text = -1, best_local_pat = -1;
    // band bookkeeping without per-row divisions: segment of band_beg / band_end advances monotonically
    int seg_beg = 0, seg_end = 0;
    const int pe_glob = pattern_len - 1, pe_glob_c = pe_glob >> 6, pe_glob_l = pe_glob & 63;

    for (int i = 0; i < text_len; i++) {
#if defined(SNAPGPU_WAVE_EMU) && defined(SNAPGPU_AG_WIN_STATS)
        { extern unsigned long long g_agwin_stats[64]; if (lane == 0) { __atomic_fetch_add(&g_agwin_stats[32 + (BANDED ? 1 : 0) * 4 + (AGC > 3 ? 3 : AGC)], 1, __ATOMIC_RELAXED); __atomic_fetch_add(&g_agwin_stats[40 + (BANDED ? 1 : 0)], (unsigned long long)(num_seg * seg_len), __ATOMIC_RELAXED); } }
#endif
        const int tb = (int)first_u32(tcode[i]);
        int band_beg = 0, band_end = pattern_len - 1;
        if (BANDED) {
            band_beg = i - w > 0 ? i - w : 0;
            band_end = i + w < pattern_len - 1 ? i + w : pattern_len - 1;
            while ((seg_beg + 1) * seg_len <= band_beg) seg_beg++;
            while ((seg_end + 1) * seg_len <= band_end) seg_end++;
        }
        int h_init0 = score_init;
        if (i > 0) { int v = score_init - gap_open - (i - 1) * gap_ext; h_init0 = v > 0 ? v : 0; }
        int mxv = 0, X0 = 0, fin = 0;
        int btr[AGC];
        bool did[AGC];
#pragma unroll
        for (int c = 0; c < AGC; c++) { btr[c] = 0; did[c] = false; }
        int row_c_lo = AGC, row_c_hi = -1;

        for (int j = seg_beg; j <= seg_end; j++) {
            const int seg_start = j * seg_len;
            int nk = num_vec;
            if (BANDED) { int lim = band_end - seg_start + 1; if (lim < nk) nk = lim; }
            const int c_lo = seg_start >> 6;
            int c_hi = (seg_start + seg_len - 1) >> 6; if (c_hi > nch - 1) c_hi = nch - 1;
            if (c_lo < row_c_lo) row_c_lo = c_lo;
            if (c_hi > row_c_hi) row_c_hi = c_hi;
            const bool zero_seg_start = BANDED && j > 0 && band_beg > seg_start;

            // ---------------- first pass
            int endv[AGC];                       // F leaving each cell (used at stripe-end cells)
            bool ins[AGC], isend[AGC];
            int carry = AG_NEG;
#pragma unroll
            for (int c = 0; c < AGC; c++) {
                endv[c] = 0; ins[c] = false; isend[c] = false;
                if (c >= c_lo && c <= c_hi) {
                    const int p = c * 64 + lane;
                    const AGPos ps = pos[c];
                    const int k = ps.k(), l = ps.l();
                    const bool inseg = ps.valid() && ps.j() == j && k < nk;
                    ins[c] = inseg; isend[c] = inseg && k == nk - 1;
                    int lane0_in = h_init0;
                    if (c > 0) lane0_in = __builtin_amdgcn_readlane(Hp[c > 0 ? c - 1 : 0], 63);
                    int h_in = ag_shr1(lane0_in, Hp[c]);
                    if (zero_seg_start && p == seg_start) h_in = 0;
                    const int pb = ps.pb();
                    int prof = pb == 5 ? -32768 : ((tb > 3 || pb > 3) ? -1 : (tb == pb ? match : sub));
                    int m = h_in > 0 ? ag_sat16(h_in + prof) : 0;
                    int e = E[c];
                    int bt = e > m ? 1 : 0;
                    int hp = m > e ? m : e;
                    int e2 = e - gap_ext;
                    int tmp = m - gap_open; if (tmp < 0) tmp = 0;
                    if (e2 > tmp) bt |= 4;
                    const int tag = AG_BIG * (j * 8 + l);
                    int g = inseg ? tmp + p * gap_ext + tag : AG_NEG;
                    int inc = ag_prefix_max(g);
                    int exc = ag_shr1(carry, inc);
                    int pm = exc > carry ? exc : carry;
                    const int fin_cell = l == 0 ? fin : 0;
                    int fk = fin_cell - k * gap_ext;
                    if (k >= 1) { int a = pm - tag - (p - 1) * gap_ext; fk = a > fk ? a : fk; }
                    if (inseg) {
                        if (fk > hp) { bt |= 2; hp = fk; }
                        Hm[c] = hp;
                        E[c] = e2 > tmp ? e2 : tmp;
                        mxv = hp > mxv ? hp : mxv;
                        int f2 = fk - gap_ext;
                        if (f2 > tmp) bt |= 32;
                        endv[c] = f2 > tmp ? f2 : tmp;
                        btr[c] = bt; did[c] = true;
                    }
                    int last_inc = __builtin_amdgcn_readlane(inc, 63);
                    carry = last_inc > carry ? last_inc : carry;
                }
            }

            // ---------------- lazy F (:1080-1112 full: 8 rounds; :534-569 banded: 7 rounds + segment carry X)
            // Round r brings each stripe the F that left the stripe r+1 to its left in the first pass, decayed by r whole
            // stripes (the reference's per-round  vF = max(vF - nk*ext, 0)  composes to exactly that), so the eight
            // stripe-end values go to LDS once and every round is a table read.
#pragma unroll
            for (int c = 0; c < AGC; c++)
                if (c >= c_lo && c <= c_hi && isend[c]) lds_end[pos[c].l()] = endv[c];
            WAVE_SYNC();
            const int rounds = BANDED ? 7 : 8;
            for (int r = 0; r < rounds; r++) {
                const int decay = r * nk * gap_ext;
                if (BANDED) {
                    int f7 = (int)first_u32((uint32_t)lds_end[7 - r]) - decay;
                    if (f7 > X0) X0 = f7;
                }
                int fj[AGC]; bool cont[AGC];
                unsigned long long any_cont = 0;
#pragma unroll
                for (int c = 0; c < AGC; c++) {
                    fj[c] = 0; cont[c] = false;
                    if (c >= c_lo && c <= c_hi) {
                        const AGPos ps = pos[c];
                        const int k = ps.k(), ls = ps.l() - 1 - r;
                        int f_in = lds_end[ls < 0 ? 0 : ls] - decay;
                        if (ls < 0 || f_in < 0) f_in = 0;
                        int f = f_in - k * gap_ext; if (f < 0) f = 0;
                        int hn = Hm[c] > f ? Hm[c] : f;
                        int t2 = hn > gap_open ? hn - gap_open : 0;
                        int f2 = f > gap_ext ? f - gap_ext : 0;
                        fj[c] = f;
                        cont[c] = ins[c] && (f2 > t2);
                        any_cont |= BALLOT(cont[c]);
                    }
                }
                // the reference stops the round at the first vector in which no SSE lane continues (:560 / :1104)
                int jstar = 0;
                if (any_cont) {
                    flag_tag++;
#pragma unroll
                    for (int c = 0; c < AGC; c++)
                        if (c >= c_lo && c <= c_hi && cont[c]) lds_flag[pos[c].k()] = flag_tag;
                    WAVE_SYNC();
                    jstar = 64 * 16;
                    for (int k0 = 0; k0 < nk; k0 += WAVE) {
                        const int kq = k0 + lane;
                        unsigned long long none = BALLOT(kq < nk && lds_flag[kq] != flag_tag);
                        if (none) { jstar = k0 + __ffsll((long long)none) - 1; break; }
                    }
                }
                const bool round_complete = jstar >= nk;        // never converged in this round
                const int jlim = round_complete ? nk - 1 : jstar;
#pragma unroll
                for (int c = 0; c < AGC; c++) {
                    if (c >= c_lo && c <= c_hi) {
                        const bool upd = ins[c] && pos[c].k() <= jlim;
                        btr[c] |= (upd && fj[c] > Hm[c]) ? 2 : 0;
                        Hm[c] = (upd && fj[c] > Hm[c]) ? fj[c] : Hm[c];
                        mxv = (upd && Hm[c] > mxv) ? Hm[c] : mxv;
                        btr[c] |= (upd && cont[c]) ? 32 : 0;
                    }
                }
#if defined(SNAPGPU_WAVE_EMU) && defined(SNAPGPU_AG_WIN_STATS)
                { extern unsigned long long g_agwin_stats[64]; if (lane == 0) __atomic_fetch_add(&g_agwin_stats[48 + (BANDED ? 8 : 0) + r], 1, __ATOMIC_RELAXED); }
#endif
                if (!round_complete) break;
            }
            fin = BANDED ? X0 : 0;
        }

        // ---------------- traceback bytes, row max, bookkeeping

#pragma unroll
        for (int c = 0; c < AGC; c++) if (c >= row_c_lo && c <= row_c_hi && did[c]) {
            if constexpr (EXACT) bt_store(sink, (uint32_t)(i * row_stride), (uint32_t)((pos[c].j() * num_vec + pos[c].k()) * 8 + pos[c].l()), (uint32_t)btr[c] | bt_tag);
            else bt_store(sink, (uint32_t)(i * row_stride), (uint32_t)(c * 64 + lane), (uint32_t)btr[c]);
        }
        const int max_row = __builtin_amdgcn_readlane(ag_prefix_max(mxv), 63);

        if (!BANDED || band_end == pattern_len - 1) {
            int gscore = 0;
#pragma unroll
            for (int c = 0; c < AGC; c++) if (c == pe_glob_c) gscore = __builtin_amdgcn_readlane(Hm[c], pe_glob_l);
            if (gscore >= best_global) { best_global = gscore; best_global_text = i; }
        }
        if (max_row == 0) break;
        if (max_row > best_local) {
            int off = -1;
#pragma unroll
            for (int c = AGC - 1; c >= 0; c--) {
                if (c >= row_c_lo && c <= row_c_hi && off < 0) {
                    unsigned long long mk = BALLOT(did[c] && Hm[c] == max_row);
                    if (mk) off = c * 64 + 63 - __clzll((long long)mk);
                }
            }
            best_local_pat = off; best_local = max_row; best_local_text = i;
        }
#pragma unroll
        for (int c = 0; c < AGC; c++) { int t = Hm[c]; Hm[c] = Hp[c]; Hp[c] = t; }
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
            // One load fetches the next 64 cells up the diagonal; they are consumed for as long as
            // the path keeps stepping diagonally (a gap step leaves the diagonal -> refetch).
            const int rt = row - lane, ct = col - lane;
            const bool ok = rt >= 0 && ct >= 0;
            bool computed = ok;
            if (BANDED && ok) {
                int bb = rt - w > 0 ? rt - w : 0, be = rt + w < pattern_len - 1 ? rt + w : pattern_len - 1;
                int cj = ct / seg_len, ck = (ct - cj * seg_len) % num_vec;
                computed = cj >= bb / seg_len && cj <= be / seg_len && cj * seg_len + ck <= be;
            }
            int cell;
            if constexpr (EXACT) {
                int vi = 0, li = 0;
                if (ok) { const int cj = ct / seg_len, cr = ct - cj * seg_len; vi = cj * num_vec + cr % num_vec; li = cr / num_vec; }
                const uint32_t at = (uint32_t)rt * (uint32_t)row_stride + (uint32_t)(vi * 8 + li);
                cell = (ok && at < bt_bytes) ? bt_cell((int)bt_scratch[at], bt_tag) : 0;
            } else {
                cell = computed ? (int)bt_scratch[(size_t)rt * row_stride + ct] : 0;
            }
            int pbyte = ok ? (int)P(ct) : 0, tbyte = ok ? (int)T(rt) : 0, qbyte = ok ? (int)Q(ct) : 0;
            int info = cell | ((ok && !computed) ? 0x100 : 0) | ((pbyte != tbyte) ? 0x200 : 0) | (qbyte << 16);
            int t = 0;
            for (; t < WAVE && row >= 0 && col >= 0; t++) {
                const int inf = __builtin_amdgcn_readlane(info, t);
                if (inf & 0x100) res.stale_reads++;
                action = ((inf & 0xff) >> (action << 1)) & 3;
                bool left_diagonal = false;
                if (action == 0) {
                    if (inf & 0x200) { prob *= tab->phred[(inf >> 16) & 0xff]; n_mismatches++; }
                    else n_matches++;
                    row--; col--;
                } else if (action == 1) {
                    row--; left_diagonal = true;
                } else {
                    col--; action = 2; left_diagonal = true;
                }
                if (prev_action != 0) {
                    if (prev_action == action) action_count++;
                    else { n_gaps += action_count; prob *= tab->indel[action_count]; action_count = 1; }
                }
                prev_action = action;
                if (left_diagonal) break;
            }
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

