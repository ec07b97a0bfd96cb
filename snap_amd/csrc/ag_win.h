// ag_win.h -- banded affine gap with a sliding 64-position window (same results as ag.h / ag_reg.h).
//
// computeScoreBanded (AffineGapVectorized.h:256-819) only ever touches the segments that
// intersect the band [i-w, i+w]: at most two segments of segLen = 8*ceil((2w+1)/8) positions per
// row.  When 2*segLen <= 64 (w <= 15: every call AlignRead makes with -d <= 14) those two
// segments fit one wavefront, so instead of keeping all pattern positions resident (ag_reg.h) the
// lanes follow the band: lane L holds position wbase + L, wbase = segLen * (band_beg / segLen).
//   * a row is two masked passes (segment j, then j+1, because the reference carries F from the
//     first segment's lazy-F loop into the second) over ONE register each of H, H-1, E -- no
//     chunk loops, no carries between chunks;
//   * when the band start crosses into the next segment the three registers slide down by segLen
//     lanes (ds_bpermute, once every segLen rows) and the lanes that enter are initialised to what
//     the reference's never-touched H / H-1 / E hold for those positions (row parity decides
//     which of the two H buffers is being read);
//   * positions behind the window are never read again by the reference (the band only moves
//     forward) except H(i-1, wbase-1) on the row of the slide, which is kept in a scalar;
//   * traceback bytes are 64 per row; the row's wbase goes to an LDS table for the traceback.
#pragma once
#include <type_traits>
#include "ag_reg.h"
#ifndef SNAPGPU_AG_DUP
#define SNAPGPU_AG_DUP 0         // measurement builds only (scripts/ab_bench.py): 1 / 2 / 3 run the prologue / the row loop / the traceback of the window form twice
#endif

// The row loop of rounds 1-3: the statement of what the rewritten loop must equal, and the loop a ZERO gap-open penalty still takes (the
// rewritten loop's lazy-F rules assume open > extend: ag_dispatch).
// EXACT (replay of flagged reads, ag.h): bt_scratch_in is the wave's image of one reference object's traceback array; cells go where the
// reference puts them -- byte (row * numVec * numSeg + vector) * 8 + SSE element -- only evaluated cells are written, and the traceback
// reads whatever the array holds.
template <bool EXACT = false, typename PSeq, typename TSeq, typename QSeq>
static __device__ __forceinline__ AGResult ag_banded_win_v1(
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
    const int tot = num_seg * seg_len;
    // LDS tables (the H/E rows of the LDS formulation are not used here): what the row loop would otherwise recompute or
    // reload -- first-row H per position, base codes of pattern and text.  (The window base of row i needs no table: the
    // window slides whenever the band start enters the next segment, so it is seg_len * (max(i - w, 0) / seg_len).)
    LDS_AS int16_t  *fr16 = (LDS_AS int16_t *)(lds_rows + 8);               // [tot]   first_row(p)
    LDS_AS uint8_t  *pcode = (LDS_AS uint8_t *)(fr16 + ((tot + 7) & ~7));  // [tot]   base_value(P(p)), 5 beyond the pattern
    LDS_AS uint8_t  *tcode = pcode + ((tot + 15) & ~15);                   // [text_len] base_value(T(i))

    int end_bonus;
    if (!is_rc) end_bonus = dir == -1 ? prm.five_bonus : prm.three_bonus;
    else        end_bonus = dir == -1 ? prm.three_bonus : prm.five_bonus;

    // lane constants: which of the (up to) two window segments, stripe l and vector k inside it
    const int segsel = lane / seg_len;                          // 0, 1 (>= 2: lane beyond the two segments)
    const int rr = lane - segsel * seg_len;
    const int l = rr / num_vec, k = rr - l * num_vec;
    const bool is_x_lane = segsel == 1 && l == 0;                // stripe 0 of the window's second segment: where the F carried over enters
    const BtSink sink = bt_sink(bt_scratch, bt_bytes);
    const int nv_tot8 = num_vec * num_seg * 8;                   // EXACT: bytes per row of the reference's array
    const int flat_lane = segsel * seg_len + k * 8 + l;          // EXACT: byte of this lane's cell inside the window's two segments

    // value of the reference's first-row H at position p, incl. the stale scoreFirstRow[] inheritance
    auto first_row = [&](int p) -> int {
        if (p >= tot) return 0;
        int pi = p;
        if (p >= pattern_len) {
            int j2 = p / seg_len, r2 = p - j2 * seg_len, l2 = r2 / num_vec, k2 = r2 - l2 * num_vec;
            pi = -1;
            for (int v = j2 * num_vec + k2 - 1; v >= 0; v--) {
                int q = (v / num_vec) * seg_len + l2 * num_vec + (v % num_vec);
                if (q < pattern_len) { pi = q; break; }
            }
            if (pi < 0) return 0;
        }
        int x = score_init - gap_open - pi * gap_ext;
        return x > 0 ? x : 0;
    };
    for (int p0 = 0; p0 < tot; p0 += WAVE) {
        const int p = p0 + lane;
        if (p < tot) {
            fr16[p] = (int16_t)first_row(p);
            pcode[p] = (uint8_t)(p < pattern_len ? base_value(P(p)) : 5u);
        }
    }
    for (int i0 = 0; i0 < text_len; i0 += WAVE) {
        const int i = i0 + lane;
        if (i < text_len) tcode[i] = (uint8_t)base_value(T(i));
    }
    WAVE_SYNC();

    int wbase = 0, jbase = 0;
    int Hp = lane < tot ? (int)fr16[lane] : 0, Hm = 0, E = 0;
    int pbv = lane < tot ? (int)pcode[lane] : 5;
    int left_h = 0;
    // H / H-1 of the global-alignment cell (position pattern_len-1) once it has left the window: the
    // reference keeps reading its stale value on the row(s) after the band has passed the pattern end
    int gl_p = 0, gl_m = 0;

    int best_global = -1, best_global_text = -1, best_local = -1, best_local_text = -1, best_local_pat = -1;

    for (int i = 0; i < text_len; i++) {
        const int tb = (int)first_u32(tcode[i]);
        const int band_beg = i - w > 0 ? i - w : 0;
        const int band_end = i + w < pattern_len - 1 ? i + w : pattern_len - 1;
        if ((jbase + 1) * seg_len <= band_beg) {                // slide the window by one segment
            left_h = __builtin_amdgcn_readlane(Hp, seg_len - 1);
            if (pattern_len - 1 >= wbase && pattern_len - 1 < wbase + seg_len) {
                gl_p = __builtin_amdgcn_readlane(Hp, pattern_len - 1 - wbase);
                gl_m = __builtin_amdgcn_readlane(Hm, pattern_len - 1 - wbase);
            }
            int nHp = __shfl_down(Hp, seg_len), nHm = __shfl_down(Hm, seg_len), nE = __shfl_down(E, seg_len);
            int npb = __shfl_down(pbv, seg_len);
            wbase += seg_len; jbase++;
            if (lane >= WAVE - seg_len) {                       // positions entering the window
                const int p = wbase + lane;
                const int fr = p < tot ? (int)fr16[p] : 0;
                nHp = (i & 1) ? 0 : fr;                         // Hptr is H on even rows, Hminus1 on odd rows
                nHm = (i & 1) ? fr : 0;
                nE = 0;
                npb = p < tot ? (int)pcode[p] : 5;
            }
            Hp = nHp; Hm = nHm; E = nE; pbv = npb;
        }
        const int p = wbase + lane;
        const bool valid = p < tot;
        int h_init0 = score_init;
        if (i > 0) { int v = score_init - gap_open - (i - 1) * gap_ext; h_init0 = v > 0 ? v : 0; }
        int mxv = 0, X0 = 0, btr = 0;
        const int prof = pbv == 5 ? -32768 : ((tb > 3 || pbv > 3) ? -1 : (tb == pbv ? match : sub));
        // The reference finishes segment j (first pass, then lazy F) before it starts segment j + 1, and the only thing that crosses
        // over is X, the F that left segment j's last stripe, which enters stripe 0 of segment j + 1.  Everything else of the two
        // first passes is independent, so both segments go through ONE first pass (each lane knows its segment), the few stripe-0
        // lanes of the second segment take X in afterwards (F only ever raises the values derived from it), and the first lazy-F round
        // -- the only one in all but a few rows -- runs for both segments at once.
        const int two = ((jbase + 1) * seg_len <= band_end) ? 1 : 0;
        int nk0 = band_end - wbase + 1; if (nk0 > num_vec) nk0 = num_vec;
        int nk1 = 0;
        if (two) { nk1 = band_end - (wbase + seg_len) + 1; if (nk1 > num_vec) nk1 = num_vec; }
        const int nkl = segsel == 0 ? nk0 : nk1;
        const bool inseg = valid && segsel <= two && k < nkl;
        const unsigned long long inseg_mask = BALLOT(inseg);

        // ---------------- first pass, both segments (:483-531)
        int lane0_in;                                              // H(i-1, p-1) of window lane 0: the reference's segment-start rule (:461-476)
        if (wbase == 0) lane0_in = h_init0;
        else lane0_in = (band_beg > wbase) ? 0 : left_h;
        const int h_in = ag_shr1(lane0_in, Hp);
        const int m = h_in > 0 ? ag_sat16(h_in + prof) : 0;
        const int e = E;
        const int bt_e0 = e > m ? 1 : 0;
        const int hpp = m > e ? m : e;
        const int e2 = e - gap_ext;
        int tmp = m - gap_open; if (tmp < 0) tmp = 0;
        const int bt_e = bt_e0 | (e2 > tmp ? 4 : 0);
        // first-pass F along the (at most 4) vectors of a stripe: F(k) = max_{j<k} (tmp_j - (k-1-j)*ext), a prefix max of
        // tmp_j + p_j*ext over the lanes to the left that belong to the same stripe -- two shift-and-max steps
        int g = inseg ? tmp + p * gap_ext : AG_NEG;
        { int a = ag_shr1(AG_NEG, g); if (k >= 1) g = a > g ? a : g; }
        { int b = ag_shr1(AG_NEG, ag_shr1(AG_NEG, g)); if (k >= 2) g = b > g ? b : g; }
        const int pm = ag_shr1(AG_NEG, g);
        int fk = -k * gap_ext;                                      // F entering the segment is 0 for now
        if (k >= 1) { int a = pm - (p - 1) * gap_ext; fk = a > fk ? a : fk; }
        if (two) {
            // X after the first segment's lazy F, assuming -- as in all but a few rows -- that its first round is also its last:
            // the F that left stripe 7 in the first pass (:538).  It enters stripe 0 of the second segment (f = X, :571).
            const int f2p0 = fk - gap_ext;
            const int endv_a = f2p0 > tmp ? f2p0 : tmp;
            const int f7 = __builtin_amdgcn_readlane(endv_a, nk0 - 1 + 7 * num_vec);
            X0 = f7 > 0 ? f7 : 0;
            const int fkp = X0 - k * gap_ext;
            fk = (is_x_lane && fkp > fk) ? fkp : fk;
        }
        int bt = bt_e | (fk > hpp ? 2 : 0);
        int hp = fk > hpp ? fk : hpp;
        int f2p = fk - gap_ext;
        bt |= f2p > tmp ? 32 : 0;
        int endv = inseg ? (f2p > tmp ? f2p : tmp) : 0;
        Hm = inseg ? hp : Hm;
        E = inseg ? (e2 > tmp ? e2 : tmp) : E;
        mxv = inseg ? hp : 0;
        btr = inseg ? bt : 0;
        const bool did = inseg;

        // ---------------- lazy F (:534-569): up to 7 rounds per segment; round r brings each stripe the F that left the stripe r+1 to
        // its left in the first pass, decayed by r whole stripes (the reference's per-round  vF = max(vF - nk*ext, 0)  composes to
        // that), so every round is one gather from the stripe-end lanes.  The reference leaves the loop at the first vector in which no
        // SSE lane continues (:560): lanes of one vector index sit num_vec apart, so OR-folding the continuation mask over the 8
        // stripes leaves "some lane of vector kk continues" in bit kk -- scalar work only.
        auto fold = [&](unsigned long long cm, int nk, int *jlim) -> bool {          // returns round_complete
            cm |= cm >> num_vec; cm |= cm >> (2 * num_vec); cm |= cm >> (4 * num_vec);
            const uint32_t full = (1u << nk) - 1u;
            const uint32_t low = (uint32_t)cm & full;
            const int jstar = low == full ? 64 : (int)__builtin_ctz(~low);
            const bool complete = jstar >= nk;
            *jlim = complete ? nk - 1 : jstar;
            return complete;
        };
        // rounds r0 .. 6 of segment s alone (the rare continuation): endv still holds the first-pass values
        auto more_rounds = [&](int s, int nk, int r0) {
            const bool ins = inseg && segsel == s;
            const unsigned long long ins_mask = BALLOT(ins);
            const int src_end = s * seg_len + nk - 1;                 // lane of stripe 0's last vector
            const int decay_step = nk * gap_ext;
            int decay = r0 * decay_step;
            int src7 = src_end + (7 - r0) * num_vec;                  // stripe 7 - r's last vector
            int src_addr = (src_end + (l - 1 - r0) * num_vec) * 4;    // byte address for ds_bpermute: the stripe r+1 to the left
            int ls = l - 1 - r0;
            for (int r = r0; r < 7; r++, decay += decay_step, src7 -= num_vec, src_addr -= num_vec * 4, ls--) {
                if (s == 0) {
                    int f7 = __builtin_amdgcn_readlane(endv, src7) - decay;
                    if (f7 > X0) X0 = f7;
                }
                int f_in = __builtin_amdgcn_ds_bpermute(src_addr, endv) - decay;
                if (ls < 0 || f_in < 0) f_in = 0;
                int f = f_in - k * gap_ext; if (f < 0) f = 0;
                int hn = Hm > f ? Hm : f;
                int t2 = hn > gap_open ? hn - gap_open : 0;
                int f2 = f > gap_ext ? f - gap_ext : 0;
                const bool cont = ins && (f2 > t2);
                int jlim;
                const bool complete = fold((__builtin_amdgcn_ballot_w64(f2 > t2) & ins_mask) >> (s * seg_len), nk, &jlim);
                const bool upd = ins && k <= jlim;
                btr |= (upd && f > Hm) ? 2 : 0;
                Hm = (upd && f > Hm) ? f : Hm;
                mxv = (upd && Hm > mxv) ? Hm : mxv;
                btr |= (upd && cont) ? 32 : 0;
                if (!complete) break;
            }
        };
        {   // round 0 of both segments
            const int src_addr = (segsel * seg_len + nkl - 1 + (l - 1) * num_vec) * 4;
            int f_in = __builtin_amdgcn_ds_bpermute(src_addr, endv);
            if (l == 0 || f_in < 0) f_in = 0;
            int f = f_in - k * gap_ext; if (f < 0) f = 0;
            int hn = Hm > f ? Hm : f;
            int t2 = hn > gap_open ? hn - gap_open : 0;
            int f2 = f > gap_ext ? f - gap_ext : 0;
            const bool cont = inseg && (f2 > t2);
            const unsigned long long cb = __builtin_amdgcn_ballot_w64(f2 > t2) & inseg_mask;
            int jlim0 = 0, jlim1 = 0;
            const bool complete0 = fold(cb, nk0, &jlim0);
            const bool complete1 = two ? fold(cb >> seg_len, nk1, &jlim1) : false;
            if (!complete0) {
                const int jl = segsel == 0 ? jlim0 : jlim1;
                const bool upd = inseg && k <= jl;
                btr |= (upd && f > Hm) ? 2 : 0;
                Hm = (upd && f > Hm) ? f : Hm;
                mxv = (upd && Hm > mxv) ? Hm : mxv;
                btr |= (upd && cont) ? 32 : 0;
                if (complete1) more_rounds(1, nk1, 1);
            } else {
                // the first segment's lazy F goes on: X can still grow, so the second segment waits for it
                const bool upd = inseg && segsel == 0;              // (a complete round updates every vector of the segment)
                btr |= (upd && f > Hm) ? 2 : 0;
                Hm = (upd && f > Hm) ? f : Hm;
                mxv = (upd && Hm > mxv) ? Hm : mxv;
                btr |= (upd && cont) ? 32 : 0;
                more_rounds(0, nk0, 1);
                if (two) {
                    // stripe 0 of the second segment again, with the final X (values derived from F only go up with it)
                    const bool xl = is_x_lane && inseg;
                    const int fkp = X0 - k * gap_ext;
                    const int fkx = fkp > fk ? fkp : fk;
                    const int btx = bt_e | (fkx > hpp ? 2 : 0) | ((fkx - gap_ext) > tmp ? 32 : 0);
                    const int hpx = fkx > hpp ? fkx : hpp;
                    const int evx = (fkx - gap_ext) > tmp ? (fkx - gap_ext) : tmp;
                    btr = xl ? btx : btr;
                    Hm = xl ? hpx : Hm;
                    mxv = (xl && hpx > mxv) ? hpx : mxv;
                    endv = xl ? evx : endv;
                    more_rounds(1, nk1, 0);
                }
            }
        }

        // (uniform row pointer + zero-extended 32-bit lane offset: the form that selects the SGPR-base store; with a sign-extended
        //  lane the address lives in a VGPR pair, which the 80-VGPR build spills and reloads -- with a vmcnt(0) wait -- every row)
        if constexpr (EXACT) {
            if (did) bt_store(sink, (uint32_t)(i * nv_tot8 + jbase * seg_len), (uint32_t)flat_lane, (uint32_t)btr | bt_tag);
        } else {
            bt_store(sink, (uint32_t)i * 64u, (uint32_t)lane, (uint32_t)btr);       // (lanes outside the band write 0; the traceback never reads them)
        }
        const int max_row = __builtin_amdgcn_readlane(ag_prefix_max(mxv), 63);
        if (band_end == pattern_len - 1) {
            int gscore = pattern_len - 1 >= wbase ? __builtin_amdgcn_readlane(Hm, pattern_len - 1 >= wbase ? pattern_len - 1 - wbase : 0) : gl_m;
            if (gscore >= best_global) { best_global = gscore; best_global_text = i; }
        }
        if (max_row == 0) break;
        if (max_row > best_local) {
            unsigned long long mk = BALLOT(did && Hm == max_row);
            best_local_pat = mk ? wbase + 63 - __clzll((long long)mk) : -1;
            best_local = max_row; best_local_text = i;
        }
        { int t = Hm; Hm = Hp; Hp = t; }
        { int t = gl_m; gl_m = gl_p; gl_p = t; }
    }
    WAVE_SYNC();

    // ---------------- local vs global (:643-730)
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

    if (score > score_init) {                                          // traceback, :732-815
        double prob = 1.0;
        int row = text_off, col = pat_off;
        int action = 0, prev_action = 0, action_count = 1, n_matches = 0, n_mismatches = 0, n_gaps = 0;
        while (row >= 0 && col >= 0) {
            const int rt = row - lane, ct = col - lane;
            const bool ok = rt >= 0 && ct >= 0;
            bool computed = false;
            int wb = 0;
            if (ok) {
                int bb = rt - w > 0 ? rt - w : 0, be = rt + w < pattern_len - 1 ? rt + w : pattern_len - 1;
                int cj = ct / seg_len, ck = (ct - cj * seg_len) % num_vec;
                computed = cj >= bb / seg_len && cj <= be / seg_len && cj * seg_len + ck <= be;
                wb = (bb / seg_len) * seg_len;
            }
            int cell;
            if constexpr (EXACT) {
                int vi = 0, li = 0;
                if (ok) { const int cj = ct / seg_len, cr = ct - cj * seg_len; vi = cj * num_vec + cr % num_vec; li = cr / num_vec; }
                const uint32_t at = (uint32_t)rt * (uint32_t)nv_tot8 + (uint32_t)(vi * 8 + li);
                cell = (ok && at < bt_bytes) ? bt_cell((int)bt_scratch[at], bt_tag) : 0;
            } else {
                cell = computed ? (int)bt_scratch[(size_t)rt * 64 + (ct - wb)] : 0;
            }
            int pbyte = ok ? (int)P(ct) : 0, tbyte = ok ? (int)T(rt) : 0, qbyte = ok ? (int)Q(ct) : 0;
            int info = cell | ((ok && !computed) ? 0x100 : 0) | ((pbyte != tbyte) ? 0x200 : 0) | (qbyte << 16);
            for (int t = 0; t < WAVE && row >= 0 && col >= 0; t++) {
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

// EXACT (replay of flagged reads, ag.h): bt_scratch_in is the wave's image of one reference object's traceback array; cells go where the
// reference puts them -- byte (row * numVec * numSeg + vector) * 8 + SSE element -- only evaluated cells are written, and the traceback
// reads whatever the array holds.
// FULL: the UNBANDED computation (AffineGapVectorized.h:914-1112) of a pattern whose striped layout is one segment of at most 64 positions
// (num_vec <= 8) -- the head of a read before an early seed, a third of all rows on 150-bp reads.  It is the banded row with a band that
// covers everything and a window that never slides: one segment, nk = num_vec on every row, no X; the reference's eighth lazy-F round
// has no stripe left to bring anything from.  (Until round 4 these calls went through ag_compute_reg<1, false>, whose rounds cost four
// times as much.)
template <bool EXACT = false, bool FULL = false, typename PSeq, typename TSeq, typename QSeq>
static __device__ __forceinline__ AGResult ag_banded_win(
    int dir, const AGParams &prm, const PSeq &P, const QSeq &Q, int pattern_len,
    const TSeq &T, int text_len, int w_in, int score_init, bool is_rc, int use_clipping,
    int16_t *lds_rows, uint8_t *bt_scratch, const DevTables *tab,
    int num_vec, int seg_len, int num_seg, uint32_t bt_bytes, uint32_t bt_tag = 0)
{
    const int lane = lane_id();
    const int w = FULL ? (1 << 20) : w_in;                       // (FULL: the band is everything, on every row and in the traceback's "was this cell computed")
    AGResult res; res.ag_score = -1; res.text_offset = -1; res.pattern_offset = -1; res.n_edits = -1;
    res.match_probability = 1.0; res.stale_reads = 0;
    const int match = prm.match_reward, sub = -prm.sub_penalty;
    const int gap_open = prm.gap_open + prm.gap_extend, gap_ext = prm.gap_extend;
    const int tot = num_seg * seg_len;
    // LDS tables (the H/E rows of the LDS formulation are not used here): what the row loop would otherwise recompute or
    // reload -- first-row H per position, base codes of pattern and text.  (The window base of row i needs no table: the
    // window slides whenever the band start enters the next segment, so it is seg_len * (max(i - w, 0) / seg_len).)
    LDS_AS int16_t  *fr16 = (LDS_AS int16_t *)(lds_rows + 8);               // [tot]   first_row(p)
    LDS_AS uint8_t  *pcode = (LDS_AS uint8_t *)(fr16 + ((tot + 7) & ~7));  // [tot]   base_value(P(p)), 5 beyond the pattern
    LDS_AS uint8_t  *tcode = pcode + ((tot + 15) & ~15);                   // [text_len] base_value(T(i))

    int end_bonus;
    if (!is_rc) end_bonus = dir == -1 ? prm.five_bonus : prm.three_bonus;
    else        end_bonus = dir == -1 ? prm.three_bonus : prm.five_bonus;

    // lane constants: which of the (up to) two window segments, stripe l and vector k inside it
    const int segsel = lane / seg_len;                          // 0, 1 (>= 2: lane beyond the two segments)
    const int rr = lane - segsel * seg_len;
    const int l = rr / num_vec, k = rr - l * num_vec;
    const bool is_x_lane = segsel == 1 && l == 0;                // stripe 0 of the window's second segment: where the F carried over enters
    const BtSink sink = bt_sink(bt_scratch, bt_bytes);
    const int nv_tot8 = num_vec * num_seg * 8;                   // EXACT: bytes per row of the reference's array
    const int flat_lane = segsel * seg_len + k * 8 + l;          // EXACT: byte of this lane's cell inside the window's two segments

    // value of the reference's first-row H at position p, incl. the stale scoreFirstRow[] inheritance
    auto first_row = [&](int p) -> int {
        if (p >= tot) return 0;
        int pi = p;
        if (p >= pattern_len) {
            int j2 = p / seg_len, r2 = p - j2 * seg_len, l2 = r2 / num_vec, k2 = r2 - l2 * num_vec;
            pi = -1;
            for (int v = j2 * num_vec + k2 - 1; v >= 0; v--) {
                int q = (v / num_vec) * seg_len + l2 * num_vec + (v % num_vec);
                if (q < pattern_len) { pi = q; break; }
            }
            if (pi < 0) return 0;
        }
        int x = score_init - gap_open - pi * gap_ext;
        return x > 0 ? x : 0;
    };
#if SNAPGPU_AG_DUP == 1
    for (int rep_ = 0; rep_ < 2; rep_++) {
#else
    {
#endif
    for (int p0 = 0; p0 < tot; p0 += WAVE) {
        const int p = p0 + lane;
        if (p < tot) {
            fr16[p] = (int16_t)first_row(p);
            pcode[p] = (uint8_t)(p < pattern_len ? base_value(P(p)) : 5u);
        }
    }
    for (int i0 = 0; i0 < text_len; i0 += WAVE) {
        const int i = i0 + lane;
        if (i < text_len) tcode[i] = (uint8_t)base_value(T(i));
    }
    WAVE_SYNC();
    }

    // ---------------------------------------------------------------------------------------------------------------------------------
    // The row loop (round 4).  The single-end kernel is bound by VALU issue (profiles/r04b: SQ_ACTIVE_INST_VALU = 79 % of the SIMDs'
    // cycles, 4 cycles per wave64 instruction) and this loop is half of it, so a row is written for the fewest VECTOR instructions:
    //   * every per-lane predicate that depends on wave-uniform row state only (which lanes are inside the band, which hold vector 0 of a
    //     stripe, ...) is a 64-bit mask computed on the scalar unit and consumed through lane_in() -- no v_cmp, no v_cndmask chains;
    //   * the four traceback bits are kept as four scalar masks (ballots come out of the compares that are needed anyway), OR-ed on the
    //     scalar unit when lazy F revises them, and turned into the byte once, after lazy F;
    //   * per-lane constants of the call (k * ext, (p - 1) * ext + tag, the bpermute source of the lazy-F gather ...) live in registers,
    //     with "not this lane" folded in as a huge offset, so that a select becomes the max / subtract that follows it anyway;
    //   * the first-pass F chain is three zero-filled wave shifts and one v_max3 (stripes carry a tag that grows with the lane, so what is
    //     shifted in from an earlier stripe always loses);
    //   * the common lazy-F outcome -- no lane continues: the reference leaves each segment's loop after vector 0 -- is one scalar test,
    //     and only vector-0 lanes take the new H; anything else goes the general way (fold / more_rounds below, as in rounds 1-3).
    // ISA count of the common row (two segments): profiles/r04c.  Results: bit for bit those of the loop it replaces (ag_banded_win_v1).
    // ---------------------------------------------------------------------------------------------------------------------------------
    int best_global = -1, best_global_text = -1, best_local = -1, best_local_text = -1, best_local_pat = -1;
#if SNAPGPU_AG_DUP == 2
    for (int rep_ = 0; rep_ < 2; rep_++) {
    best_global = -1; best_global_text = -1; best_local = -1; best_local_text = -1; best_local_pat = -1;
#else
    {
#endif
    int wbase = 0, jbase = 0;
    int Hp = lane < tot ? (int)fr16[lane] : 0, Hm = 0, E = 0;
    int pbv = lane < tot ? (int)pcode[lane] : 5;
    // The best local score (:the row maximum that beats every earlier row's, its row, the HIGHEST position holding it) without a per-row reduction:
    // every position keeps  (its best H << 16) | (0xFFFF - the first row that reached it)  -- one shift-or, one select, one unsigned max per row, no
    // scalar instruction -- and the order of that word is the reference's: a larger H wins, then the earlier row.  The wave-wide maximum of the words,
    // taken once after the loop, is the best local score and its row; the highest position that holds that word is its column.  Positions that
    // slide out of the window leave their word's maximum in `dpk` (+ `dpos`); H <= 32767 (ag_sat16), rows < 65535.
    uint32_t lpk = 0u, dpk = 0u; int dpos = -1;
    int left_h = 0;
    // H / H-1 of the global-alignment cell (position pattern_len-1) once it has left the window: the
    // reference keeps reading its stale value on the row(s) after the band has passed the pattern end
    int gl_p = 0, gl_m = 0;


    // ---- masks of the call (wave-uniform)
    // (Only Kz -- the lanes with k == 0 -- is kept; the other masks of the call are two or three scalar instructions from seg_len / num_vec
    //  wherever they are used: kept as values they were 14 more SGPRs alive across the row loop, and the loop spilled and reloaded SGPRs --
    //  the traceback store's buffer descriptor among them -- with ~10 v_writelane / v_readlane per row.  A lambda that captured masks by
    //  reference even became a table in scratch memory, and everything downstream of its loads vector code.)
    const unsigned long long Kz = first_u64(BALLOT(k == 0));
    auto seg0 = [=]() -> unsigned long long { return seg_len >= 64 ? ~0ull : ((1ull << seg_len) - 1ull); };            // lanes of the window's first segment
    auto seg1 = [=]() -> unsigned long long { return seg_len >= 64 ? 0ull : (((1ull << seg_len) - 1ull) << seg_len); };   // ... of its second (2 * seg_len <= 64)
    auto kmask = [=](int n) -> unsigned long long {                                                                       // lanes with k < n (n = 0 .. num_vec <= 4)
        if (n >= num_vec) return ~0ull;
        return (Kz << n) - Kz;                 // every k == 0 lane b becomes the lanes b .. b + n - 1 (n < num_vec: the runs do not meet, nothing borrows)
    };
    auto vmask = [=](int wb) -> unsigned long long { const int nvl = tot - wb; return nvl >= 64 ? ~0ull : (nvl <= 0 ? 0ull : ((1ull << nvl) - 1ull)); };
    auto xmask = [=]() -> unsigned long long { return seg_len >= 64 ? 0ull : (((1ull << num_vec) - 1ull) << seg_len); };   // stripe 0 of the second segment: where the F carried over enters
    auto lmid = [=]() -> unsigned long long { return seg_len >= 64 ? 0ull : (((1ull << (6 * num_vec)) - 1ull) << (seg_len + num_vec)); };   // the second segment's stripes 1 .. 6
    unsigned long long V = vmask(0);                                                                 // lanes with p < tot
    // ---- per-lane constants of the call
    const int tagl = (segsel * 8 + l) * AG_BIG;                  // grows with the lane: what a wave shift brings in from an earlier stripe loses
    const int c_kext = -k * gap_ext;                             // F entering a stripe is 0: F(k) >= -k * ext
    const int c_lazy = l == 0 ? AG_HUGE : k * gap_ext;           // lazy F: f = max(f_in - c_lazy, 0)   (stripe 0 gets no F from the left)
    const int c_x = is_x_lane ? k * gap_ext : AG_HUGE;           // the F carried over from the first segment: max(fk, X0 - c_x)
    const int c_src = (segsel * seg_len + (l - 1) * num_vec - 1) * 4;      // lazy-F gather: byte address of lane (stripe l - 1, vector nk - 1) is c_src + 4 * nk
    int c_pt = lane * gap_ext + tagl;                            // p * ext + tag           (p = wbase + lane: + seg_len * ext per slide)
    int c_pm = k == 0 ? AG_HUGE : (lane - 1) * gap_ext + tagl;   // (p - 1) * ext + tag, "no F from the left" for vector 0
    int v_else = pbv == 5 ? -32768 : (pbv == 4 ? -1 : sub);      // profile entry of this lane's pattern base against a text base that differs from it
    int v_else_n = pbv == 5 ? -32768 : -1;                       // ... against an 'N' of the text
    const int c_prev = (l == 0 ? lane : lane - num_vec) * 4;     // lazy F, rounds 1 .. 6: the same vector one stripe to the left (stripe 0: itself)
    // lazy F, "does vector k have an SSE lane whose F goes on": lane k < num_vec holds the (at most 8) lanes of vector k of a segment -- k + l' * num_vec --
    // as a mask, so that the verdict of a round is ONE vector AND + compare of the round's ballot against it (lanes >= num_vec: 0) instead of three
    // shift / or pairs on the scalar unit, which is what bounds this kernel (round 6: ~6 of a round's ~20 scalar instructions)
    typename std::conditional<FULL, unsigned long long, uint32_t>::type kgrp = 0;
    if (lane < num_vec) for (int l2 = 0; l2 < 8; l2++) kgrp |= (decltype(kgrp))1 << (lane + l2 * num_vec);
    int c_ls = 0, c_g = 0;                                        // (follow nk like stepv: the closed form of the second segment's lazy F)
    unsigned long long KM0 = 0ull, KM1 = 0ull, KE1 = 0ull, KL1 = 0ull;   // masks that follow (nk0, nk1) like the per-lane values below: the band's lanes of either segment, the second segment's stripe ends / stripes 1 .. 7
    int nk_addr = 0, stepv = 0;                     // per-lane values that follow (nk0, nk1): the gather address of round 0, nk * ext of the lane's segment
    // ... and the masks that follow (nk0, nk1) AND the window (V changes when it slides; a slide resets nk_key): the band's valid lanes per segment, and kgrp
    // restricted to them (a round's ballot then needs no AND with the segment's lanes on the scalar unit)
    unsigned long long ins0 = 0ull, ins1 = 0ull, inseg_mask = 0ull;
    decltype(kgrp) kg0 = 0; unsigned long long kg1 = 0ull;     // (kg1 at the second segment's own lanes: a round's ballot is tested in place, not shifted down first)
    uint32_t full0 = 0u, full1 = 0u; int src7_0 = 0, c7x = 0;   // (1 << nk) - 1 per segment; round 1's stripe-6 lane and 7 * nk0 * ext of the first segment's X
    // The row's own scalars, kept incrementally (round 6: the scalar unit bounds the kernel; recomputed from i every row they were ~45 scalar instructions):
    //   be        band_end = min(i + w, pattern_len - 1): + 1 per row until it reaches the pattern's end;
    //   be_next   the band_end at which (nk0, nk1) next change -- the per-(nk0, nk1) block below runs when be reaches it (a slide, which moves the
    //             window under the band, forces it) -- so nk0 / nk1 / two and everything derived from them are only touched there;
    //   slide_at  the row on which the window slides next: (jbase + 1) * seg_len <= band_beg  <=>  i >= (jbase + 1) * seg_len + w;
    //   l0, hi    H(i-1, p-1) of window lane 0 for this row and, while the window is at the pattern's start, the first-column value of the next:
    //             score_init, then max(score_init - open - (i - 1) * ext, 0); after a slide the H that left the window on the slide's row, then 0.
    const int pe1 = pattern_len - 1;
    int be = FULL ? pe1 : (w < pe1 ? w : pe1) - 1;                // (the row's first statement adds the row's step)
    int be_next = -(1 << 30);
    int slide_at = FULL ? (1 << 30) : seg_len + w;
    int l0 = score_init, hi = score_init - gap_open > 0 ? score_init - gap_open : 0;
    int nk0 = 0, nk1 = 0, f7_lane = 0, e0_lane = 0; bool two = false;
    int tcv = 0;                                                 // base codes of 64 text rows, lane j: row (i & ~63) + j
    int row_off = 0;                                             // EXACT: i * nv_tot8, the row's offset in the reference object's array

    // open - ext, in a VECTOR register on purpose: every lazy-F round subtracts it, the row loop has more wave-uniform values than SGPRs, and as an
    // SGPR it was the one the allocator spilled -- a v_readlane per round to get it back.
#if defined(SNAPGPU_WAVE_EMU)
    const int d_open = gap_open - gap_ext;
#else
    int d_open;
    asm("v_mov_b32 %0, %1" : "=v"(d_open) : "s"(gap_open - gap_ext));
#endif
    EMU_STAT(8, 1);
    for (int i = 0; i < text_len; i++) {
        EMU_STAT(9, 1);
#if defined(SNAPGPU_AG_DUMMY) && !defined(SNAPGPU_WAVE_EMU)
        // measurement scaffolding (scripts/ab_bench.py): 32 more scalar (1) / vector (2) instructions per row -- which of the two the row loop's time follows
        {
#if SNAPGPU_AG_DUMMY == 1
            int ds_ = i;
            asm volatile(".rept 32\n s_add_u32 %0, %0, 1\n .endr" : "+s"(ds_) : : "scc");
#else
            int dv_ = lane;
            asm volatile(".rept 32\n v_add_u32 %0, %0, 1\n .endr" : "+v"(dv_));
#endif
        }
#endif
        if (__builtin_expect((i & 63) == 0, 0)) tcv = i + lane < text_len ? (int)tcode[i + lane] : 0;      // 64 rows' text codes per LDS read
        const int tb = __builtin_amdgcn_readlane(tcv, i & 63);
        if (!FULL) { be = be + 1 < pe1 ? be + 1 : pe1; }
        if (!FULL && __builtin_expect(i >= slide_at, 0)) {                           // slide the window by one segment (one row in twenty)
            EMU_STAT(10, 1);
            left_h = __builtin_amdgcn_readlane(Hp, seg_len - 1);
            if (pattern_len - 1 >= wbase && pattern_len - 1 < wbase + seg_len) {
                gl_p = __builtin_amdgcn_readlane(Hp, pattern_len - 1 - wbase);
                gl_m = __builtin_amdgcn_readlane(Hm, pattern_len - 1 - wbase);
            }
            {   // the departing positions' best word (a later group's positions are higher: it wins a tie)
                const uint32_t dv = lane < seg_len ? lpk : 0u;
                const uint32_t dmx = (uint32_t)__builtin_amdgcn_readlane(ag_prefix_max((int)dv), 63);      // (words are below 2^31: the signed maximum is the unsigned one)
                if (dmx != 0u && dmx >= dpk) {
                    const unsigned long long mk = BALLOT(dv == dmx);
                    dpk = dmx; dpos = wbase + 63 - __clzll((long long)mk);
                }
            }
            int nHp = __shfl_down(Hp, seg_len), nHm = __shfl_down(Hm, seg_len), nE = __shfl_down(E, seg_len);
            int npb = __shfl_down(pbv, seg_len);
            uint32_t nlpk = (uint32_t)__shfl_down((int)lpk, seg_len);
            wbase += seg_len; jbase++;
            if (lane >= WAVE - seg_len) {                       // positions entering the window
                const int p = wbase + lane;
                const int fr = p < tot ? (int)fr16[p] : 0;
                nHp = (i & 1) ? 0 : fr;                         // Hptr is H on even rows, Hminus1 on odd rows
                nHm = (i & 1) ? fr : 0;
                nE = 0;
                npb = p < tot ? (int)pcode[p] : 5;
                nlpk = 0u;
            }
            Hp = nHp; Hm = nHm; E = nE; pbv = npb; lpk = nlpk;
            V = vmask(wbase);
            c_pt += seg_len * gap_ext; c_pm += seg_len * gap_ext;
            v_else = pbv == 5 ? -32768 : (pbv == 4 ? -1 : sub);
            v_else_n = pbv == 5 ? -32768 : -1;
            slide_at += seg_len;
            be_next = -(1 << 30);                               // (V and wbase moved: the block below runs)
            l0 = left_h; hi = 0;
        }
        // The reference finishes segment j (first pass, then lazy F) before it starts segment j + 1, and the only thing that crosses
        // over is X, the F that left segment j's last stripe, which enters stripe 0 of segment j + 1.  Everything else of the two
        // first passes is independent, so both segments go through ONE first pass (each lane knows its segment), the few stripe-0
        // lanes of the second segment take X in afterwards (F only ever raises the values derived from it), and the first lazy-F round
        // -- the only one in all but a few rows -- runs for both segments at once.
        {
            if (__builtin_expect(be >= be_next, 0)) {               // (nk0, nk1) change here (one row in six), or the window has just moved
                EMU_STAT(14, 1);
                const int t0 = be - wbase + 1;                      // cells of the band from the window's start on (wbase = jbase * seg_len)
                nk0 = t0 < num_vec ? t0 : num_vec;
                nk1 = t0 - seg_len; if (nk1 > num_vec) nk1 = num_vec; if (nk1 < 0 || FULL) nk1 = 0;    // (FULL: one segment, known at compile time)
                two = !FULL && nk1 > 0;                             // (jbase + 1) * seg_len <= band_end
                // when they change next: while the first segment's vectors fill up (t0 < num_vec) and while the second's do (seg_len < t0 < seg_len + num_vec) every
                // row; in between at t0 = seg_len + 1; afterwards not before the window slides
                be_next = FULL ? (1 << 30) : (t0 < num_vec ? be + 1 : (t0 <= seg_len ? be + (seg_len + 1 - t0) : (t0 < seg_len + num_vec ? be + 1 : (1 << 30))));
                f7_lane = nk0 - 1 + 7 * num_vec; e0_lane = seg_len + nk1 - 1;
                const int nkl = lane_in(seg0()) ? nk0 : nk1;
                nk_addr = c_src + 4 * nkl;
                stepv = l == 0 ? 0 : nkl * gap_ext;
                c_ls = l * nkl * gap_ext;                           // a stripe end's origin: e_l + l * nk * ext
                c_g = l == 0 ? AG_HUGE : c_ls - stepv - c_kext;     // what lies between stripe 0's end and cell (l, k): ((l - 1) * nk + k) * ext
                KM0 = kmask(nk0) & seg0(); KM1 = kmask(nk1) & seg1();
                KE1 = nk1 > 0 ? ((Kz << (nk1 - 1)) & lmid()) & KM1 : 0ull;
                KL1 = KM1 & ~xmask();
                ins0 = KM0 & V; ins1 = KM1 & V;                      // valid && k < nk(segment), per segment
                inseg_mask = ins0 | ins1;
                kg0 = kgrp & (decltype(kgrp))ins0; kg1 = FULL ? 0ull : (((unsigned long long)kgrp << seg_len) & ins1);
                full0 = (1u << nk0) - 1u; full1 = (1u << nk1) - 1u;
                src7_0 = nk0 - 1 + 6 * num_vec; c7x = 7 * nk0 * gap_ext;
            }
        }
        const bool inseg = lane_in(inseg_mask);

        // ---------------- first pass, both segments (:483-531)
        const int lane0_in = l0;                                   // H(i-1, p-1) of window lane 0: the reference's segment-start rule (:461-476)
        l0 = hi; hi = hi > gap_ext ? hi - gap_ext : 0;
        const int h_in = ag_shr1(lane0_in, Hp);
        const int prof = tb > 3 ? v_else_n : (pbv == tb ? match : v_else);       // (tb > 3: an 'N' of the text)
        const int m = h_in > 0 ? ag_sat16(h_in + prof) : 0;
        const int e = E;
        unsigned long long M1 = BALLOT(e > m);                     // traceback bit 1: E wins over the diagonal
        const int hpp = m > e ? m : e;
        const int e2 = e - gap_ext;
        int tmp = m - gap_open; if (tmp < 0) tmp = 0;
        const unsigned long long M4 = BALLOT(e2 > tmp);            // bit 4: E extends
        // first-pass F along the (at most 4) vectors of a stripe: F(k) = max_{j<k} (tmp_j - (k-1-j)*ext), a max of tmp_j + p_j*ext over
        // the (at most 3) lanes to the left that belong to the same stripe
        // (always three shifts: with fewer vectors per stripe the extra ones bring cells of an earlier stripe, whose tag loses -- as it
        //  already does for the lanes with k < 3 of a four-vector stripe; three tests of num_vec per row cost more than the two instructions)
        int fk;
        {
            const int g = inseg ? tmp + c_pt : AG_NEG;
            const int a1 = ag_shr1z(g);
            const int a2 = ag_shr1z(a1);
            const int a3 = ag_shr1z(a2);
            int pm = a2 > a1 ? a2 : a1;
            pm = a3 > pm ? a3 : pm;
            if (FULL && num_vec > 4) {                               // (up to 8 vectors per stripe: four more cells to the left)
                const int a4 = ag_shr1z(a3), a5 = ag_shr1z(a4), a6 = ag_shr1z(a5), a7 = ag_shr1z(a6);
                const int q = a5 > a4 ? a5 : a4, q2 = a7 > a6 ? a7 : a6;
                pm = q > pm ? q : pm; pm = q2 > pm ? q2 : pm;
            }
            const int a = pm - c_pm;
            fk = a > c_kext ? a : c_kext;
        }
        int X0 = 0;
        if (two) {
            EMU_STAT(12, 1);
            // X after the first segment's lazy F, assuming -- as in all but a few rows -- that its first round is also its last:
            // the F that left stripe 7 in the first pass (:538).  It enters stripe 0 of the second segment (f = X, :571).
            const int f2p0 = fk - gap_ext;
            const int endv_a = f2p0 > tmp ? f2p0 : tmp;
            const int f7 = __builtin_amdgcn_readlane(endv_a, f7_lane);
            X0 = f7 > 0 ? f7 : 0;
            const int fkp = X0 - c_x;
            fk = fkp > fk ? fkp : fk;
        }
        unsigned long long M2 = BALLOT(fk > hpp);                  // bit 2: F wins
        const int hp = fk > hpp ? fk : hpp;
        const int f2p = fk - gap_ext;
        unsigned long long M32 = BALLOT(f2p > tmp);                // bit 32: F extends
        int endv = inseg ? (f2p > tmp ? f2p : tmp) : 0;
        Hm = inseg ? hp : Hm;
        E = inseg ? (e2 > tmp ? e2 : tmp) : E;

        // ---------------- lazy F (:534-569): up to 7 rounds per segment.  Round r brings each stripe the F that left the stripe r + 1 to
        // its left in the first pass, decayed by r whole stripes (the reference's per-round  vF = max(vF - nk*ext, 0)  composes to that),
        // and is applied to vectors 0 .. jlim, jlim = the first vector in which no SSE lane's F goes on (:560; all of them when every
        // vector has such a lane: the round is complete and the next one runs).  On real reads a row goes through ~3 rounds of its first
        // segment and all 7 of its second (whole SSE vectors are evaluated, so the stripes of the second segment that lie beyond the band
        // hold small stale H and F runs through all of them): the rounds, not the first pass, are most of a row, so a round is kept to a
        // gather, four max / subtract and one compare:
        //   * what a round offers lane (l, k) is u_r(l, k) = u_{r-1}(l - 1, k) - nk * ext: ONE gather from the lane num_vec to the left
        //     (stripe 0 reads itself and stays at "nothing"), f_r = max(u_r, 0);
        //   * only  Fx = max of the offers a lane took  is tracked.  H after lazy F is max(H_fp, Fx); traceback bit 2 (F wins) was set in some
        //     round iff Fx > H_fp; bit 32 (F goes on: f_r - ext > max(max(H, f_r) - open, 0) in the round that offered f_r) was set in some
        //     round iff Fx > T_fp = max(H_fp - (open - ext), ext) -- the FIRST offer above T_fp also beats every earlier offer minus
        //     (open - ext), and no offer below T_fp can go on; (open - ext = the gap-open penalty > 0: the caller sends a zero penalty to v1)
        //   * "F goes on" in round r, which decides jlim and whether round r + 1 runs, is f_r > Tr, Tr = max(T_fp, Fx - (open - ext)) kept
        //     incrementally.
        // X, the F that leaves the first segment's last stripe into stripe 0 of the second (:538, :571), grows with every round the first
        // segment runs; when it has, the second segment's stripe-0 cells are redone before that segment's rounds start.
        int T_fp = Hm - d_open; if (T_fp < gap_ext) T_fp = gap_ext;
        int Fx = 0;
        {
            const int u0 = __builtin_amdgcn_ds_bpermute(nk_addr, endv) - c_lazy;    // round 0's offer, both segments
            const int wv = endv + c_ls;                                              // a stripe end at its origin (the first segment's X; the second's closed form)
            int u = u0;
            int Fx0 = 0, Fx1 = 0;                                                    // the largest offer taken, first / second segment (every round runs over all lanes)
            // A complete round applies to every lane of the segment: no select, and lanes outside the segment may collect what they like
            // (they are masked once, below).  Only the round that stops the walk selects its vectors 0 .. jlim.
            // "Every vector 0 .. nk - 1 has a lane whose F goes on" is an OR over the (at most 8) stripes of the segment: the segment's
            // lanes fit 32 bits, three shift / or pairs fold them onto stripe 0.
            // X, what round r >= 1 brings to the first segment's end, is endv(stripe 7 - r, vector nk - 1) - r * step = wv there - 7 * step:
            // the rounds keep the largest wv they pass (a readlane and a max), the subtraction happens once.
            // (Written as nested calls, one instantiation per round, not as a loop with a break: unrolled, the loop's exits came out as
            //  flag registers set, tested and tested again -- five scalar instructions and two branches per round.)
            auto rounds = [&](auto sc, int nk, uint32_t full, unsigned long long ins_mask, auto kgs, int &Fxs) {
                constexpr int s = decltype(sc)::value;
                int src7 = src7_0;                                                   // (round 1 looks at stripe 6's last vector; first segment only)
                int Wm = -AG_HUGE;
                int Tr = T_fp;
                // (The next round's gather needs this round's OFFER, not its verdict: it is issued before the verdict's ballot-and-fold chain so
                //  that the two latencies overlap; when the verdict stops the walk one gather was for nothing.)
                auto round = [&](auto &self, auto rc, int u_cur) -> void {
                    constexpr int r = decltype(rc)::value;
#if defined(SNAPGPU_WAVE_EMU) && defined(SNAPGPU_AG_WIN_STATS)
                    { extern unsigned long long g_agwin_stats[64]; if (lane == 0) __atomic_fetch_add(&g_agwin_stats[16 + s * 8 + r], 1, __ATOMIC_RELAXED); }
#endif
                    int u_next = 0;
                    if constexpr (r < 6) u_next = __builtin_amdgcn_ds_bpermute(c_prev, u_cur);
                    const unsigned long long go = BALLOT(u_cur > Tr);                // (an offer <= 0 is never above Tr >= 0; lanes outside the segment's band: masked by kgs)
#if defined(SNAPGPU_AG_FOLD_SCALAR)
                    typename std::conditional<FULL, unsigned long long, uint32_t>::type cm = s == 0 ? go : (go >> seg_len);      // (a banded segment's lanes fit 32 bits)
#else
                    const auto cm = (decltype(kgs))go;                               // (first segment of a banded call: its lanes fit 32 bits; second: tested where they are)
#endif
#if defined(SNAPGPU_AG_FOLD_SCALAR)                                                    // (measurement builds: the fold of rounds 4 - 5)
                    cm &= (decltype(cm))(s == 0 ? ins_mask : (ins_mask >> seg_len));
                    cm |= cm >> num_vec; cm |= cm >> (2 * num_vec); cm |= cm >> (4 * num_vec);
                    const uint32_t low = (uint32_t)cm & full;
#else
                    const uint32_t low = (uint32_t)BALLOT((cm & kgs) != 0);          // bit k: vector k goes on (kgs: lane k < nk holds the band's valid lanes of vector k; every other lane 0)
#endif
                    if (__builtin_expect(low != full, 0)) {                          // the round stops at vector jlim: the walk ends
                        const int jlim = (int)__builtin_ctz(~low);
                        const int fm = (k <= jlim && lane_in(ins_mask)) ? u_cur : 0;   // (vectors 0 .. jlim of the segment: a vector compare against the lane's k, not a mask built on the scalar unit)
                        Fxs = fm > Fxs ? fm : Fxs;
                    } else {
                        Fxs = u_cur > Fxs ? u_cur : Fxs;
                        if constexpr (r < 6) {
                            const int tq = u_cur - d_open;
                            Tr = tq > Tr ? tq : Tr;
                            if (s == 0 && !FULL) {                                   // (kept whether or not the row has a second segment: X0 is only read when it has; FULL never has)
                                const int w7 = __builtin_amdgcn_readlane(wv, src7);
                                Wm = w7 > Wm ? w7 : Wm;
                                src7 -= num_vec;
                            }
                            self(self, std::integral_constant<int, r + 1>{}, u_next - stepv);
                        }
                    }
                };
                round(round, std::integral_constant<int, 0>{}, u);
                if (s == 0 && !FULL) { const int xr = Wm - c7x; if (xr > X0) X0 = xr; }
            };
            const int X_first = X0;
            if (nk0 > 0) rounds(std::integral_constant<int, 0>{}, nk0, full0, ins0, kg0, Fx0);
            int Fx = lane_in(ins0) ? Fx0 : 0;
            if (two) {
                if (X0 != X_first) {
                    EMU_STAT(11, 1);
                    // stripe 0 of the second segment again, with the final X (values derived from F only go up with it)
                    const unsigned long long xlm = xmask() & inseg_mask;
                    const bool xl = lane_in(xlm);
                    const int fkp = X0 - c_x;
                    const int fkx = fkp > fk ? fkp : fk;
                    const int hpx = fkx > hpp ? fkx : hpp;
                    const int evx = (fkx - gap_ext) > tmp ? (fkx - gap_ext) : tmp;
                    M2 = (M2 & ~xlm) | (BALLOT(fkx > hpp) & xlm);
                    M32 = (M32 & ~xlm) | (BALLOT((fkx - gap_ext) > tmp) & xlm);
                    Hm = xl ? hpx : Hm;
                    endv = xl ? evx : endv;
                    T_fp = Hm - d_open; if (T_fp < gap_ext) T_fp = gap_ext;
                }
                // The second segment usually runs all seven rounds -- its stripes 1 .. 7 lie beyond the band, hold small stale H, and the
                // F that entered stripe 0 from the first segment runs through all of them -- and then what every cell ends up with is
                // that one flow: e0, the F leaving stripe 0, decayed by the cells in between.  Two ballots say when that is so:
                //   (A) no later stripe end beats it at its own origin (e_j + j * nk * ext <= e0, j = 1 .. 6), so for every cell the offer
                //       that started in stripe 0 is the largest it gets;
                //   (B) that offer is above T_fp in every cell of stripes 1 .. 7: the cell (r + 1, k) then goes on in round r whatever it
                //       was offered before (smaller, by (A)), every vector has such a cell, all seven rounds are complete and apply to
                //       every cell -- and Fx is the stripe-0 offer itself.
                // Otherwise the rounds run as for the first segment.
                const int e0 = __builtin_amdgcn_readlane(endv, e0_lane);
                const unsigned long long endm = KE1 & V;
                const int g0 = e0 - c_g;
                const unsigned long long insL = KL1 & V;
                if ((BALLOT(wv > e0) & endm) == 0ull && (BALLOT(g0 > T_fp) & insL) == insL) {     // (endm: stripes 1 .. 6, whose endv the new X did not touch)
#if defined(SNAPGPU_WAVE_EMU) && defined(SNAPGPU_AG_WIN_STATS)
                    { extern unsigned long long g_agwin_stats[64]; if (lane == 0) __atomic_fetch_add(&g_agwin_stats[4], 1, __ATOMIC_RELAXED); }
#endif
                    Fx = lane_in(insL) ? g0 : Fx;
                } else {
                    if ((BALLOT(wv > e0) & endm) != 0ull) EMU_STAT(15, 1); else EMU_STAT(16, 1);
                    u = X0 != X_first ? __builtin_amdgcn_ds_bpermute(nk_addr, endv) - c_lazy : u0;     // round 0's offer again (gathered anew when X moved the stripe-0 ends)
                    rounds(std::integral_constant<int, 1>{}, nk1, full1, ins1, kg1, Fx1);
                    Fx = lane_in(ins1) ? Fx1 : Fx;
                }
            }
            M2 |= BALLOT(Fx > Hm);
            M32 |= BALLOT(Fx > T_fp);
            Hm = Fx > Hm ? Fx : Hm;
        }

        // the traceback byte, once, from the four masks
        const int btr = (lane_in(M1) ? 1 : 0) | (lane_in(M4) ? 4 : 0) | (lane_in(M2) ? 2 : 0) | (lane_in(M32) ? 32 : 0);
        // (uniform row pointer + zero-extended 32-bit lane offset: the form that selects the SGPR-base store; with a sign-extended
        //  lane the address lives in a VGPR pair, which the 80-VGPR build spills and reloads -- with a vmcnt(0) wait -- every row)
        if constexpr (EXACT) {
            if (inseg) bt_store(sink, (uint32_t)(row_off + wbase), (uint32_t)flat_lane, (uint32_t)btr | bt_tag);   // (i * nv_tot8 + jbase * seg_len, kept incrementally)
            row_off += nv_tot8;
        } else {
            bt_store(sink, (uint32_t)i * 64u, (uint32_t)lane, (uint32_t)btr);       // (lanes outside the band write what they have; the traceback never reads them)
        }
        if (be == pe1) {
            int gscore = pattern_len - 1 >= wbase ? __builtin_amdgcn_readlane(Hm, pattern_len - 1 >= wbase ? pattern_len - 1 - wbase : 0) : gl_m;
            if (gscore >= best_global) { best_global = gscore; best_global_text = i; }
        }
        // The loop ends on a row of zeros (:max_row == 0); otherwise every position notes its best (see lpk).
        {
            const uint32_t t = ((uint32_t)Hm << 16) | (uint32_t)(0xFFFF - i);
            const uint32_t tin = inseg ? t : 0u;
            if (BALLOT(tin > 0xFFFFu) == 0ull) break;                                // max_row == 0: no evaluated cell of the row holds anything
            lpk = tin > lpk ? tin : lpk;
        }
        { int t = Hm; Hm = Hp; Hp = t; }
        { int t = gl_m; gl_m = gl_p; gl_p = t; }
    }
    {   // the best local score, its row and column from the positions' words (the window's positions are higher than every departed one: they win a tie)
        const uint32_t wmx = (uint32_t)__builtin_amdgcn_readlane(ag_prefix_max((int)lpk), 63);
        uint32_t bpk = dpk; int bpos = dpos;
        if (wmx != 0u && wmx >= dpk) {
            const unsigned long long mk = BALLOT(lpk == wmx);
            bpk = wmx; bpos = wbase + 63 - __clzll((long long)mk);
        }
        if (bpk != 0u) { best_local = (int)(bpk >> 16); best_local_text = 0xFFFF - (int)(bpk & 0xFFFFu); best_local_pat = bpos; }
    }
    WAVE_SYNC();
    }

    // ---------------- local vs global (:643-730)
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

#if SNAPGPU_AG_DUP == 3
    const int text_off_dup = text_off, pat_off_dup = pat_off;
    for (int rep_ = 0; rep_ < 2; rep_++) { text_off = text_off_dup; pat_off = pat_off_dup; res.stale_reads = 0;
#else
    {
#endif
    if (score > score_init) {                                          // traceback, :732-815
        double prob = 1.0;
        int row = text_off, col = pat_off;
        int action = 0, prev_action = 0, action_count = 1, n_matches = 0, n_mismatches = 0, n_gaps = 0;
        while (row >= 0 && col >= 0) {
            EMU_STAT(13, 1);
            const int rt = row - lane, ct = col - lane;
            const bool ok = rt >= 0 && ct >= 0;
            bool computed = false;
            int wb = 0;
            if (ok) {
                int bb = rt - w > 0 ? rt - w : 0, be = rt + w < pattern_len - 1 ? rt + w : pattern_len - 1;
                int cj = ct / seg_len, ck = (ct - cj * seg_len) % num_vec;
                computed = cj >= bb / seg_len && cj <= be / seg_len && cj * seg_len + ck <= be;
                wb = (bb / seg_len) * seg_len;
            }
            int cell;
            if constexpr (EXACT) {
                int vi = 0, li = 0;
                if (ok) { const int cj = ct / seg_len, cr = ct - cj * seg_len; vi = cj * num_vec + cr % num_vec; li = cr / num_vec; }
                const uint32_t at = (uint32_t)rt * (uint32_t)nv_tot8 + (uint32_t)(vi * 8 + li);
                cell = (ok && at < bt_bytes) ? bt_cell((int)bt_scratch[at], bt_tag) : 0;
            } else {
                cell = computed ? (int)bt_scratch[(size_t)rt * 64 + (ct - wb)] : 0;
            }
            int pbyte = ok ? (int)P(ct) : 0, tbyte = ok ? (int)T(rt) : 0, qbyte = ok ? (int)Q(ct) : 0;
            int info = cell | ((ok && !computed) ? 0x100 : 0) | ((pbyte != tbyte) ? 0x200 : 0) | (qbyte << 16);
            for (int t = 0; t < WAVE && row >= 0 && col >= 0; t++) {
                // Along an alignment most steps are diagonal steps out of the match state, one after the other: such a run is taken at once --
                // how far it goes is a ballot, its matches and mismatches are population counts, and only its mismatches (whose quality
                // terms enter the FP64 product in path order) are visited one by one.  Everything else goes through the step below, as before:
                // 130 serial steps of ~18 scalar instructions per call were 7 % of the kernel's scalar instructions.
                if (action == 0 && prev_action == 0) {
                    const unsigned long long nz = BALLOT((info & 3) != 0) >> t;
                    int run = nz ? (int)__builtin_ctzll(nz) : WAVE - t;
                    const int room = (row < col ? row : col) + 1;
                    if (run > room) run = room;
                    if (run > 0) {
                        const unsigned long long rm = (run >= 64 ? ~0ull : ((1ull << run) - 1ull)) << t;
                        unsigned long long mm = BALLOT((info & 0x200) != 0) & rm;
                        res.stale_reads += (int)__popcll(BALLOT((info & 0x100) != 0) & rm);
                        const int nmm = (int)__popcll(mm);
                        n_mismatches += nmm; n_matches += run - nmm;
                        while (mm) {
                            const int tl = (int)__builtin_ctzll(mm); mm &= mm - 1ull;
                            prob *= tab->phred[(__builtin_amdgcn_readlane(info, tl) >> 16) & 0xff];
                        }
                        row -= run; col -= run; t += run - 1;
                        continue;
                    }
                }
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
    }
    return res;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The window form for WIDE bands (round 6): 32 < segLen <= 64, i.e. limits 13 .. 31 -- every banded call of the `-d 20` configuration
// (BASELINE configs[4]: limit 22, segLen 48) and the gapless-clipped Phase-4 calls of the paired-end path (limit 30, segLen 64).  The band's two
// segments no longer fit one wavefront side by side, so each lives in a register set of its own: lane L < segLen of set A is position
// wbase + L (segment jbase), of set B position wbase + segLen + L (segment jbase + 1) -- the same (stripe, vector) in both, so every per-lane
// constant of the call is shared.  A row is two segment passes, in the reference's order (segment j's first pass and lazy F, then X -- the F
// that left its last stripe -- into stripe 0 of segment j + 1, AffineGapVectorized.h:461-571); a slide moves B into A.  Until this form these
// calls went through ag_compute_reg<AGC, true>, whose row walks AGC chunks of 64 positions through LDS-staged lazy-F rounds (~1 600 wave
// instructions per row; on the `c5` leg 86 % of the paired kernel's wave cycles, profiles/r06b).  The statement of the arithmetic is
// ag_banded_win_v1's (no closed forms, any gap-open penalty); results: bit for bit those of ag_compute_reg / ag.h.
template <bool EXACT = false, typename PSeq, typename TSeq, typename QSeq>
static __device__ __forceinline__ AGResult ag_banded_win2(
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
    const int tot = num_seg * seg_len;
    LDS_AS int16_t  *fr16 = (LDS_AS int16_t *)(lds_rows + 8);               // [tot]   first_row(p)        (the tables of ag_banded_win)
    LDS_AS uint8_t  *pcode = (LDS_AS uint8_t *)(fr16 + ((tot + 7) & ~7));  // [tot]   base_value(P(p)), 5 beyond the pattern
    LDS_AS uint8_t  *tcode = pcode + ((tot + 15) & ~15);                   // [text_len] base_value(T(i))

    int end_bonus;
    if (!is_rc) end_bonus = dir == -1 ? prm.five_bonus : prm.three_bonus;
    else        end_bonus = dir == -1 ? prm.three_bonus : prm.five_bonus;

    // lane constants: stripe l and vector k of the lane's cell inside ITS segment (the same in both register sets)
    const bool seg_lane = lane < seg_len;
    const int l = lane / num_vec, k = lane - l * num_vec;
    const BtSink sink = bt_sink(bt_scratch, bt_bytes);
    const int nv_tot8 = num_vec * num_seg * 8;                   // EXACT: bytes per row of the reference's array
    const int flat_lane = k * 8 + l;                             // EXACT: byte of this lane's cell inside its segment
    unsigned long long kgrp = 0ull;                              // lane k < num_vec: the lanes of vector k in the segment's eight stripes
    if (lane < num_vec) for (int l2 = 0; l2 < 8; l2++) kgrp |= 1ull << (lane + l2 * num_vec);

    auto first_row = [&](int p) -> int {                         // the reference's first-row H at position p, incl. the stale scoreFirstRow[] inheritance
        if (p >= tot) return 0;
        int pi = p;
        if (p >= pattern_len) {
            int j2 = p / seg_len, r2 = p - j2 * seg_len, l2 = r2 / num_vec, k2 = r2 - l2 * num_vec;
            pi = -1;
            for (int v = j2 * num_vec + k2 - 1; v >= 0; v--) {
                int q = (v / num_vec) * seg_len + l2 * num_vec + (v % num_vec);
                if (q < pattern_len) { pi = q; break; }
            }
            if (pi < 0) return 0;
        }
        int x = score_init - gap_open - pi * gap_ext;
        return x > 0 ? x : 0;
    };
    for (int p0 = 0; p0 < tot; p0 += WAVE) {
        const int p = p0 + lane;
        if (p < tot) {
            fr16[p] = (int16_t)first_row(p);
            pcode[p] = (uint8_t)(p < pattern_len ? base_value(P(p)) : 5u);
        }
    }
    for (int i0 = 0; i0 < text_len; i0 += WAVE) {
        const int i = i0 + lane;
        if (i < text_len) tcode[i] = (uint8_t)base_value(T(i));
    }
    WAVE_SYNC();

    // a segment's registers as the reference's never-touched cells hold them on row i: Hptr is H (first-row values) on even rows, Hminus1 (zero) on odd ones
    auto fresh = [&](int seg_start, int row, int &Hp, int &Hm, int &E, int &pbv) {
        const int p = seg_start + lane;
        const bool v = seg_lane && p < tot;
        const int fr = v ? (int)fr16[p] : 0;
        Hp = (row & 1) ? 0 : fr; Hm = (row & 1) ? fr : 0; E = 0;
        pbv = v ? (int)pcode[p] : 5;
    };
    int wbase = 0, jbase = 0;
    int HpA, HmA, EA, pbvA, HpB, HmB, EB, pbvB;
    fresh(0, 0, HpA, HmA, EA, pbvA);
    fresh(seg_len, 0, HpB, HmB, EB, pbvB);
    int left_h = 0;
    int gl_p = 0, gl_m = 0;          // H / H-1 of the global-alignment cell (position pattern_len - 1) once it has left the window
    int best_global = -1, best_global_text = -1, best_local = -1, best_local_text = -1, best_local_pat = -1;
    // (as in ag_banded_win: the best local score through one word per position, the row's band scalars kept incrementally, 64 text codes per LDS read)
    uint32_t lpkA = 0u, lpkB = 0u, dpk = 0u; int dpos = -1;
    const int pe1 = pattern_len - 1;
    int be = (w < pe1 ? w : pe1) - 1;
    int slide_at = seg_len + w;
    int l0 = score_init, hi = score_init - gap_open > 0 ? score_init - gap_open : 0;
    int tcv = 0;

    for (int i = 0; i < text_len; i++) {
        if (__builtin_expect((i & 63) == 0, 0)) tcv = i + lane < text_len ? (int)tcode[i + lane] : 0;
        const int tb = __builtin_amdgcn_readlane(tcv, i & 63);
#if defined(SNAPGPU_WAVE_EMU) && defined(SNAPGPU_AG_WIN_STATS)
        { extern unsigned long long g_agwin_stats[64]; if (lane == 0) __atomic_fetch_add(&g_agwin_stats[5], 1, __ATOMIC_RELAXED); }
#endif
        be = be + 1 < pe1 ? be + 1 : pe1;
        const int band_end = be;
        if (__builtin_expect(i >= slide_at, 0)) {               // slide ((jbase + 1) * seg_len <= band_beg): segment jbase leaves, B becomes A, the next segment enters fresh
            {   // the departing segment's best word (a later segment's positions are higher: it wins a tie)
                const uint32_t dmx = (uint32_t)__builtin_amdgcn_readlane(ag_prefix_max((int)lpkA), 63);
                if (dmx != 0u && dmx >= dpk) { const unsigned long long mk = BALLOT(lpkA == dmx); dpk = dmx; dpos = wbase + 63 - __clzll((long long)mk); }
                lpkA = lpkB; lpkB = 0u;
            }
            left_h = __builtin_amdgcn_readlane(HpA, seg_len - 1);
            if (pattern_len - 1 >= wbase && pattern_len - 1 < wbase + seg_len) {
                gl_p = __builtin_amdgcn_readlane(HpA, pattern_len - 1 - wbase);
                gl_m = __builtin_amdgcn_readlane(HmA, pattern_len - 1 - wbase);
            }
            HpA = HpB; HmA = HmB; EA = EB; pbvA = pbvB;
            wbase += seg_len; jbase++;
            fresh(wbase + seg_len, i, HpB, HmB, EB, pbvB);
            slide_at += seg_len;
            l0 = left_h; hi = 0;
        }
        const bool two = (jbase + 1) * seg_len <= band_end;
        int nk0 = band_end - wbase + 1; if (nk0 > num_vec) nk0 = num_vec;
        int nk1 = 0;
        if (two) { nk1 = band_end - (wbase + seg_len) + 1; if (nk1 > num_vec) nk1 = num_vec; }

        // One segment of the row (AffineGapVectorized.h:483-569): first pass, then up to seven lazy-F rounds.  X_in: the F that enters stripe 0
        // (0 for the row's first segment); X: the F that has left the segment's last stripe so far (kept for the first segment only).
        int X = 0;
        auto seg_pass = [&](int seg_start, int nk, int lane0_in, int X_in, bool keep_X, int &Hp, int &Hm, int &E, int pbv, int &btr, bool &ins_out) {
            const int p = seg_start + lane;
            const bool inseg = seg_lane && p < tot && k < nk;
            const int h_in = ag_shr1(lane0_in, Hp);
            const int prof = pbv == 5 ? -32768 : ((tb > 3 || pbv > 3) ? -1 : (tb == pbv ? match : sub));
            const int m = h_in > 0 ? ag_sat16(h_in + prof) : 0;
            const int e = E;
            const int bt_e0 = e > m ? 1 : 0;
            const int hpp = m > e ? m : e;
            const int e2 = e - gap_ext;
            int tmp = m - gap_open; if (tmp < 0) tmp = 0;
            const int bt_e = bt_e0 | (e2 > tmp ? 4 : 0);
            // first-pass F along the (at most 8) vectors of a stripe: F(k) = max_{j<k} (tmp_j - (k-1-j)*ext), a prefix max of tmp_j + p_j*ext over
            // the lanes to the left that belong to the same stripe -- three shift-and-max steps
            int g = inseg ? tmp + p * gap_ext : AG_NEG;
            { const int a = ag_shr1(AG_NEG, g); if (k >= 1) g = a > g ? a : g; }
            { const int b = ag_shr1(AG_NEG, ag_shr1(AG_NEG, g)); if (k >= 2) g = b > g ? b : g; }
            if (num_vec > 4) { const int c = ag_shr1(AG_NEG, ag_shr1(AG_NEG, ag_shr1(AG_NEG, ag_shr1(AG_NEG, g)))); if (k >= 4) g = c > g ? c : g; }
            const int pm = ag_shr1(AG_NEG, g);
            int fk = -k * gap_ext;
            if (k >= 1) { const int a = pm - (p - 1) * gap_ext; fk = a > fk ? a : fk; }
            { const int fx = X_in - k * gap_ext; fk = (l == 0 && fx > fk) ? fx : fk; }          // (X_in >= 0: for X_in == 0 this is fk itself)
            int bt = bt_e | (fk > hpp ? 2 : 0);
            const int hp = fk > hpp ? fk : hpp;
            const int f2p = fk - gap_ext;
            bt |= f2p > tmp ? 32 : 0;
            const int endv = inseg ? (f2p > tmp ? f2p : tmp) : 0;
            Hm = inseg ? hp : Hm;
            E = inseg ? (e2 > tmp ? e2 : tmp) : E;
            btr = inseg ? bt : 0;
            ins_out = inseg;
            const unsigned long long ins_mask = BALLOT(inseg);
            const uint32_t full = (1u << nk) - 1u;
            const int decay_step = nk * gap_ext;
            int decay = 0;
            int src7 = nk - 1 + 7 * num_vec;                        // stripe 7 - r's last vector
            int src_addr = (nk - 1 + (l - 1) * num_vec) * 4;        // ds_bpermute byte address: the last vector of the stripe r + 1 to the left
            int ls = l - 1;
            if (!keep_X && gap_open > gap_ext) {
                // The second segment usually runs all seven rounds: its stripes 1 .. 7 lie beyond the band, hold small stale H, and the F that entered
                // stripe 0 runs through all of them -- what every cell then ends up with is that one flow, e0 (the F leaving stripe 0) decayed by the cells
                // in between.  The two tests of ag_banded_win say when that is so (its notes have the argument): (A) no later stripe end beats e0 at its
                // own origin, (B) the flow is above T_fp = max(H - open, ext) in every cell of stripes 1 .. 7 -- then the seven rounds are this select.
                const int e0 = __builtin_amdgcn_readlane(endv, nk - 1);
                const int wv = endv + l * decay_step;
                const int g0 = e0 - ((l - 1) * nk + k) * gap_ext;
                int T_fp = Hm - (gap_open - gap_ext); if (T_fp < gap_ext) T_fp = gap_ext;
                const bool insL = inseg && l >= 1;
                if (BALLOT(insL && l <= 6 && k == nk - 1 && wv > e0) == 0ull && BALLOT(insL && !(g0 > T_fp)) == 0ull) {
#if defined(SNAPGPU_WAVE_EMU) && defined(SNAPGPU_AG_WIN_STATS)
                    { extern unsigned long long g_agwin_stats[64]; if (lane == 0) __atomic_fetch_add(&g_agwin_stats[7], 1, __ATOMIC_RELAXED); }
#endif
                    btr |= (insL && g0 > Hm) ? 2 : 0;
                    btr |= insL ? 32 : 0;
                    Hm = (insL && g0 > Hm) ? g0 : Hm;
                    return;
                }
            }
            for (int r = 0; r < 7; r++, decay += decay_step, src7 -= num_vec, src_addr -= num_vec * 4, ls--) {
#if defined(SNAPGPU_WAVE_EMU) && defined(SNAPGPU_AG_WIN_STATS)
                { extern unsigned long long g_agwin_stats[64]; if (lane == 0) __atomic_fetch_add(&g_agwin_stats[keep_X ? 6 : 8], 1, __ATOMIC_RELAXED); }
#endif
                if (keep_X) {
                    const int f7 = __builtin_amdgcn_readlane(endv, src7) - decay;
                    if (f7 > X) X = f7;
                }
                int f_in = __builtin_amdgcn_ds_bpermute(src_addr, endv) - decay;
                if (ls < 0 || f_in < 0) f_in = 0;
                int f = f_in - k * gap_ext; if (f < 0) f = 0;
                const int hn = Hm > f ? Hm : f;
                const int t2 = hn > gap_open ? hn - gap_open : 0;
                const int f2 = f > gap_ext ? f - gap_ext : 0;
                const bool cont = inseg && (f2 > t2);
                unsigned long long cm = __builtin_amdgcn_ballot_w64(f2 > t2) & ins_mask;
                const uint32_t low = (uint32_t)BALLOT((cm & kgrp) != 0ull);                          // "some SSE lane of vector kk goes on" in bit kk (see ag_banded_win: kgrp)
                const int jstar = low == full ? 64 : (int)__builtin_ctz(~low);
                const bool complete = jstar >= nk;
                const int jlim = complete ? nk - 1 : jstar;
                const bool upd = inseg && k <= jlim;
                btr |= (upd && f > Hm) ? 2 : 0;
                Hm = (upd && f > Hm) ? f : Hm;
                btr |= (upd && cont) ? 32 : 0;
                if (!complete) break;
            }
        };
        const int lane0_in = l0;                                   // H(i-1, p-1) of the window's first cell: the reference's segment-start rule (:461-476); see ag_banded_win
        l0 = hi; hi = hi > gap_ext ? hi - gap_ext : 0;
        int btrA = 0, btrB = 0; bool insA = false, insB = false;
        seg_pass(wbase, nk0, lane0_in, 0, true, HpA, HmA, EA, pbvA, btrA, insA);
        if (two) {
            const int lastA = __builtin_amdgcn_readlane(HpA, seg_len - 1);     // H(i-1) of the cell before segment B's first
            const int X_in = X;
            seg_pass(wbase + seg_len, nk1, lastA, X_in, false, HpB, HmB, EB, pbvB, btrB, insB);
        }

        if constexpr (EXACT) {
            if (insA) bt_store(sink, (uint32_t)(i * nv_tot8 + jbase * seg_len), (uint32_t)flat_lane, (uint32_t)btrA | bt_tag);
            if (insB) bt_store(sink, (uint32_t)(i * nv_tot8 + (jbase + 1) * seg_len), (uint32_t)flat_lane, (uint32_t)btrB | bt_tag);
        } else {
            bt_store(sink, (uint32_t)i * 128u, (uint32_t)lane, (uint32_t)btrA);
            if (two) bt_store(sink, (uint32_t)i * 128u + 64u, (uint32_t)lane, (uint32_t)btrB);
        }
        if (band_end == pe1) {
            const int gp = pattern_len - 1 - wbase;
            int gscore;
            if (gp < 0) gscore = gl_m;
            else if (gp < seg_len) gscore = __builtin_amdgcn_readlane(HmA, gp);
            else gscore = __builtin_amdgcn_readlane(HmB, gp - seg_len);
            if (gscore >= best_global) { best_global = gscore; best_global_text = i; }
        }
        {
            const uint32_t rowc = (uint32_t)(0xFFFF - i);
            const uint32_t tA = insA ? (((uint32_t)HmA << 16) | rowc) : 0u, tB = insB ? (((uint32_t)HmB << 16) | rowc) : 0u;
            if (BALLOT((tA | tB) > 0xFFFFu) == 0ull) break;                          // max_row == 0
            lpkA = tA > lpkA ? tA : lpkA; lpkB = tB > lpkB ? tB : lpkB;
        }
        { int t = HmA; HmA = HpA; HpA = t; }
        { int t = HmB; HmB = HpB; HpB = t; }
        { int t = gl_m; gl_m = gl_p; gl_p = t; }
    }
    {   // the best local score, its row and column from the positions' words: segment B's positions are the highest, then A's, then the departed ones
        uint32_t bpk = dpk; int bpos = dpos;
        const uint32_t wa = (uint32_t)__builtin_amdgcn_readlane(ag_prefix_max((int)lpkA), 63), wb = (uint32_t)__builtin_amdgcn_readlane(ag_prefix_max((int)lpkB), 63);
        if (wa != 0u && wa >= bpk) { const unsigned long long mk = BALLOT(lpkA == wa); bpk = wa; bpos = wbase + 63 - __clzll((long long)mk); }
        if (wb != 0u && wb >= bpk) { const unsigned long long mk = BALLOT(lpkB == wb); bpk = wb; bpos = wbase + seg_len + 63 - __clzll((long long)mk); }
        if (bpk != 0u) { best_local = (int)(bpk >> 16); best_local_text = 0xFFFF - (int)(bpk & 0xFFFFu); best_local_pat = bpos; }
    }
    WAVE_SYNC();

    // ---------------- local vs global (:643-730)
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

    if (score > score_init) {                                          // traceback, :732-815
        double prob = 1.0;
        int row = text_off, col = pat_off;
        int action = 0, prev_action = 0, action_count = 1, n_matches = 0, n_mismatches = 0, n_gaps = 0;
        while (row >= 0 && col >= 0) {
            const int rt = row - lane, ct = col - lane;
            const bool ok = rt >= 0 && ct >= 0;
            bool computed = false;
            int wb = 0;
            if (ok) {
                int bb = rt - w > 0 ? rt - w : 0, be = rt + w < pattern_len - 1 ? rt + w : pattern_len - 1;
                int cj = ct / seg_len, ck = (ct - cj * seg_len) % num_vec;
                computed = cj >= bb / seg_len && cj <= be / seg_len && cj * seg_len + ck <= be;
                wb = (bb / seg_len) * seg_len;
            }
            int cell;
            if constexpr (EXACT) {
                int vi = 0, li = 0;
                if (ok) { const int cj = ct / seg_len, cr = ct - cj * seg_len; vi = cj * num_vec + cr % num_vec; li = cr / num_vec; }
                const uint32_t at = (uint32_t)rt * (uint32_t)nv_tot8 + (uint32_t)(vi * 8 + li);
                cell = (ok && at < bt_bytes) ? bt_cell((int)bt_scratch[at], bt_tag) : 0;
            } else {
                const int d = ct - wb;                                 // 0 .. 2 * seg_len - 1: segment A's bytes at 0, segment B's at 64
                cell = computed ? (int)bt_scratch[(size_t)rt * 128 + (d < seg_len ? d : 64 + d - seg_len)] : 0;
            }
            int pbyte = ok ? (int)P(ct) : 0, tbyte = ok ? (int)T(rt) : 0, qbyte = ok ? (int)Q(ct) : 0;
            int info = cell | ((ok && !computed) ? 0x100 : 0) | ((pbyte != tbyte) ? 0x200 : 0) | (qbyte << 16);
            for (int t = 0; t < WAVE && row >= 0 && col >= 0; t++) {
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

// AGC > 0: register formulation with AGC chunks of 64 positions (the host guarantees it fits);
// AGC == 0: the LDS formulation of ag.h (any pattern length up to RL).
template <int AGC, bool EXACT, typename PSeq, typename TSeq, typename QSeq>
static __device__ __forceinline__ AGResult ag_dispatch_inl(
    bool banded, int dir, const AGParams &prm, const PSeq &P, const QSeq &Q, int pattern_len,
    const TSeq &T, int text_len, int w, int score_init, bool is_rc, int use_clipping,
    int16_t *lds_rows, uint8_t *bt_scratch, uint32_t RL, const DevTables *tab, uint32_t bt_tag = 0)
{
    if constexpr (AGC > 0) {
        AGResult res; res.ag_score = -1; res.text_offset = -1; res.pattern_offset = -1; res.n_edits = -1;
        res.match_probability = 0.0; res.stale_reads = 0;
        int ww = w > 126 ? 126 : w;
        if (ww < 0) return res;                                           // :325 / :890
        int num_vec, seg_len, num_seg;
        ag_dims(banded, pattern_len, ww, &num_vec, &seg_len, &num_seg);
        // AGC 4 and 6 (reads beyond ~170 bp): since round 6 these kernels keep THREE chunks of 64 positions in registers like AGC 3 -- every banded call with a
        // band half-width up to 31 goes through the window forms below, whatever the pattern length, and what is left for the chunked register form are the
        // unbanded calls (pattern shorter than 3 * (2w + 1)): at most 134 positions at -d 20; BASELINE configs[4] has none beyond 128 (scripts/emu_paired_stats.py) --
        // and a call that needs more than 192 positions takes the LDS form of ag.h, which these contexts have the LDS for (ag_lds_bytes(RL)).  The kernels
        // that used to carry 256 / 384 positions of H, H-1, E in VGPRs (128 / 256 VGPRs, 4 / 2 waves per SIMD) run at AGC 3's occupancy.
        constexpr int CH = AGC > 3 ? 3 : AGC;
        if (num_vec > 1023 || num_seg > 255 ||
            (size_t)text_len * (size_t)(((num_seg * seg_len + 63) >> 6) * 64) > ag_scratch_bytes(RL) ||
            (EXACT && (size_t)text_len * (size_t)(num_seg * seg_len) > ag_scratch_bytes(RL))) {
            __builtin_trap();                                             // host sizing bug: fail loudly
        }
#if defined(SNAPGPU_WAVE_EMU) && defined(SNAPGPU_AG_WIN_STATS)
        {   // which form takes the call (scripts/emu_stats.py, emu_paired_stats.py): calls / rows / positions per form, and the unbanded register calls by chunk count
            extern unsigned long long g_agform_stats[64];
            const int tot_ = num_seg * seg_len;
            const int f = banded ? (2 * seg_len <= 64 ? 1 : (seg_len <= 64 ? 2 : 3)) : (tot_ <= 64 ? 4 : 6);
            if (lane_id() == 0) {
                __atomic_fetch_add(&g_agform_stats[f], 1, __ATOMIC_RELAXED); __atomic_fetch_add(&g_agform_stats[8 + f], (unsigned long long)text_len, __ATOMIC_RELAXED);
                __atomic_fetch_add(&g_agform_stats[16 + f], (unsigned long long)tot_, __ATOMIC_RELAXED);
                if (f == 6) { __atomic_fetch_add(&g_agform_stats[24 + ((tot_ + 63) >> 6)], 1, __ATOMIC_RELAXED); __atomic_fetch_add(&g_agform_stats[32 + ((tot_ + 63) >> 6)], (unsigned long long)text_len, __ATOMIC_RELAXED); }
                __atomic_fetch_add(&g_agform_stats[40 + (ww > 15 ? 15 : ww)], 1, __ATOMIC_RELAXED);
                __atomic_fetch_add(&g_agform_stats[56 + (ww >= 32 ? 1 : 0) + (tot_ > 192 ? 2 : 0) + (tot_ > 256 ? 2 : 0)], 1, __ATOMIC_RELAXED);     // [56] w < 32 && tot <= 192 ... [61] w >= 32 && tot > 256
            }
        }
#endif
        if (banded && 2 * seg_len <= 64 && prm.gap_open <= 0)        // (the rewritten row loop assumes a positive gap-open penalty: see its lazy-F notes)
            return ag_banded_win_v1<EXACT>(dir, prm, P, Q, pattern_len, T, text_len, ww, score_init, is_rc, use_clipping,
                                           lds_rows, bt_scratch, tab, num_vec, seg_len, num_seg, (uint32_t)ag_scratch_bytes(RL), bt_tag);
        if (banded && 2 * seg_len <= 64)        // the band's two segments fit one wavefront: sliding-window form
            return ag_banded_win<EXACT>(dir, prm, P, Q, pattern_len, T, text_len, ww, score_init, is_rc, use_clipping,
                                        lds_rows, bt_scratch, tab, num_vec, seg_len, num_seg, (uint32_t)ag_scratch_bytes(RL), bt_tag);
#if !defined(SNAPGPU_NO_AG_WIN2)
        if (banded && seg_len <= 64 && (size_t)text_len * 128u <= ag_scratch_bytes(RL))        // wide bands (limits 13 .. 31): one segment per register set
            return ag_banded_win2<EXACT>(dir, prm, P, Q, pattern_len, T, text_len, ww, score_init, is_rc, use_clipping,
                                         lds_rows, bt_scratch, tab, num_vec, seg_len, num_seg, (uint32_t)ag_scratch_bytes(RL), bt_tag);
#endif
        if (num_seg * seg_len > 64 * CH) {
            if constexpr (AGC > 3)
                return ag_compute<EXACT>(banded, dir, prm, P, Q, pattern_len, T, text_len, w, score_init, is_rc, use_clipping,
                                         lds_rows, bt_scratch, RL, tab, bt_tag);
            else __builtin_trap();                                        // host sizing bug: fail loudly
        }
        if (banded)
            return ag_compute_reg<CH, true, EXACT>(dir, prm, P, Q, pattern_len, T, text_len, ww, score_init, is_rc, use_clipping,
                                                   lds_rows, bt_scratch, tab, num_vec, seg_len, num_seg, (uint32_t)ag_scratch_bytes(RL), bt_tag);
        if (num_seg * seg_len <= 64 && prm.gap_open > 0)        // short pattern (e.g. the read's head before an early seed): the window form's row, band = everything
            return ag_banded_win<EXACT, true>(dir, prm, P, Q, pattern_len, T, text_len, ww, score_init, is_rc, use_clipping,
                                              lds_rows, bt_scratch, tab, num_vec, seg_len, num_seg, (uint32_t)ag_scratch_bytes(RL), bt_tag);
        if (num_seg * seg_len <= 64)            // ... with a zero gap-open penalty: the chunked form with one chunk
            return ag_compute_reg<1, false, EXACT>(dir, prm, P, Q, pattern_len, T, text_len, ww, score_init, is_rc, use_clipping,
                                                   lds_rows, bt_scratch, tab, num_vec, seg_len, num_seg, (uint32_t)ag_scratch_bytes(RL), bt_tag);
        return ag_compute_reg<CH, false, EXACT>(dir, prm, P, Q, pattern_len, T, text_len, ww, score_init, is_rc, use_clipping,
                                                lds_rows, bt_scratch, tab, num_vec, seg_len, num_seg, (uint32_t)ag_scratch_bytes(RL), bt_tag);
    } else {
        return ag_compute<EXACT>(banded, dir, prm, P, Q, pattern_len, T, text_len, w, score_init, is_rc, use_clipping,
                                 lds_rows, bt_scratch, RL, tab, bt_tag);
    }
}

// The affine-gap code as ONE function per kernel instead of one inlined copy per call site.  k_align_paired reaches ag_dispatch from 24
// places (Phases 3 and 4, the speculative Phase-4 scoring, the single-end aligner of the chimeric fallback ...), each copy ~40 KB of code:
// 1.3 MB of kernel against a 64 KB instruction cache shared by two CUs, every copy allocated out of the caller's 256 VGPRs (1 600 bytes of
// scratch per lane, 4 500 SGPR spills).  A call costs a few hundred cycles against the ~10^5 of the work it starts.  Arguments arrive in
// VGPRs under the device calling convention, so everything wave-uniform is made scalar again on entry.

template <int AGC, bool EXACT, typename PSeq, typename TSeq, typename QSeq>
static __device__ __attribute__((noinline)) AGResult ag_dispatch_fn(
    uint32_t flags, AGParams prm_in, PSeq P_in, QSeq Q_in, int pattern_len,
    TSeq T_in, int text_len, int w, int score_init, int use_clipping,
    int16_t *lds_rows, uint8_t *bt_scratch, uint32_t RL, const DevTables *tab)
{
    flags = first_u32(flags);
    const bool banded = (flags & 1u) != 0, is_rc = (flags & 2u) != 0;
    const uint32_t bt_tag = (flags >> 8) & 0xFFu;               // EXACT: the tag of the read whose image cells count (dev_common.h: bt_cell)
    const int dir = (flags & 4u) ? -1 : 1;
    AGParams prm;
    prm.match_reward = (int)first_u32((uint32_t)prm_in.match_reward); prm.sub_penalty = (int)first_u32((uint32_t)prm_in.sub_penalty);
    prm.gap_open = (int)first_u32((uint32_t)prm_in.gap_open); prm.gap_extend = (int)first_u32((uint32_t)prm_in.gap_extend);
    prm.five_bonus = (int)first_u32((uint32_t)prm_in.five_bonus); prm.three_bonus = (int)first_u32((uint32_t)prm_in.three_bonus);
    const PSeq P = seq_uniform(P_in); const QSeq Q = seq_uniform(Q_in); const TSeq T = seq_uniform(T_in);
    pattern_len = (int)first_u32((uint32_t)pattern_len); text_len = (int)first_u32((uint32_t)text_len);
    w = (int)first_u32((uint32_t)w); score_init = (int)first_u32((uint32_t)score_init); use_clipping = (int)first_u32((uint32_t)use_clipping);
    lds_rows = (int16_t *)(uintptr_t)first_u64((uint64_t)(uintptr_t)lds_rows);
    bt_scratch = (uint8_t *)(uintptr_t)first_u64((uint64_t)(uintptr_t)bt_scratch);
    RL = first_u32(RL);
    tab = (const DevTables *)(uintptr_t)first_u64((uint64_t)(uintptr_t)tab);
    return ag_dispatch_inl<AGC, EXACT>(banded, dir, prm, P, Q, pattern_len, T, text_len, w, score_init, is_rc, use_clipping,
                                       lds_rows, bt_scratch, RL, tab, bt_tag);
}

template <int AGC, bool EXACT = false, typename PSeq, typename TSeq, typename QSeq>
static __device__ __forceinline__ AGResult ag_dispatch(
    bool banded, int dir, const AGParams &prm, const PSeq &P, const QSeq &Q, int pattern_len,
    const TSeq &T, int text_len, int w, int score_init, bool is_rc, int use_clipping,
    int16_t *lds_rows, uint8_t *bt_scratch, uint32_t RL, const DevTables *tab, uint32_t bt_tag = 0)
{
#if !defined(SNAPGPU_AG_LV_FUNCTIONS)
    return ag_dispatch_inl<AGC, EXACT>(banded, dir, prm, P, Q, pattern_len, T, text_len, w, score_init, is_rc, use_clipping, lds_rows, bt_scratch, RL, tab, bt_tag);
#else
    const uint32_t flags = (banded ? 1u : 0u) | (is_rc ? 2u : 0u) | (dir == -1 ? 4u : 0u) | (bt_tag << 8);
    return ag_dispatch_fn<AGC, EXACT, PSeq, TSeq, QSeq>(flags, prm, P, Q, pattern_len, T, text_len, w, score_init, use_clipping, lds_rows, bt_scratch, RL, tab);
#endif
}
