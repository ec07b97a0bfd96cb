// lv.h -- Landau-Vishkin edit distance with match probability, one problem per wavefront.
//
// Restates LandauVishkin<TEXT_DIRECTION>::computeEditDistance (SNAPLib/LandauVishkin.h:100-351)
// and countPerfectMatch (:377-407); tie-breaking rules summarised in SURVEY.md Appendix A.3.
//
// GPU mapping: L[e][d] is the furthest-reaching pattern offset with e edits on diagonal d.
// Lane r owns the diagonal with *iteration rank* r in the reference's visiting order
// d = 0,+1,-1,+2,-2,... (rank(d) = 2d-1 for d>0, -2d otherwise), so "first diagonal in
// visiting order that reaches the end" is simply the lowest set bit of a ballot.  All
// 2e+1 diagonals of a level are extended concurrently; the level's results go to an LDS
// triangle (row e starts at e*e) from which the next level and the backtrace read.
// The text direction of the template parameter becomes an accessor stride, so the forward
// (<1>) and backward (<-1>) variants are the same code.
//
// Outputs for e <= k do not depend on k (k only bounds the level loop), which is what lets
// callers evaluate with one limit and threshold afterwards.
#pragma once
#include "dev_common.h"
#include "planes.h"

#define LV_ACT_X 0
#define LV_ACT_D 1
#define LV_ACT_I 2

struct ByteSeq {                 // s(i) = p[i*stride]
    const uint8_t *p;
    int stride;
    __device__ __forceinline__ uint8_t operator()(int i) const { return p[i * stride]; }
};

// The same, for bytes that are known to lie in LDS -- the kernels' reads, qualities and reference window do.  Landau-Vishkin and affine gap
// are functions (ag_win.h: ag_dispatch_fn); across a call a generic pointer is all the compiler knows and every P(i) / T(i) becomes a FLAT
// load (both counters, the vector-memory path); as LdsSeq it is a ds_read_u8.
struct LdsSeq {
    LDS_AS const uint8_t *p;
    int stride;
    __device__ __forceinline__ uint8_t operator()(int i) const { return p[i * stride]; }
};
// (No "is this pointer LDS" test here: the callers pass base + origin with origin = -1 for an empty backward half, and a generic pointer one
//  byte below LDS address 0 -- wave 0's read buffer starts there -- is outside the LDS aperture although its low 32 bits are the right LDS
//  address.  Such a test stopped the paired-end kernel on hardware, profiles/r04w.)
static __device__ __forceinline__ LdsSeq lds_seq(const ByteSeq &s) { return LdsSeq{(LDS_AS const uint8_t *)s.p, s.stride}; }

#define LV_PLANES_FROM 3               // first level whose bitmaps come from planes (lv_compute_inl)

struct LVResult {
    int    score;                // edit distance, or -1 (ScoreAboveLimit)
    double match_probability;
    int    net_indel;
    int    total_indels;
    int    text_span;
};

static __device__ __forceinline__ int lv_rank(int d) { return d > 0 ? 2 * d - 1 : -2 * d; }
static __device__ __forceinline__ int lv_diag(int r) { return (r & 1) ? (r + 1) >> 1 : -(r >> 1); }

// LDS needed: (kmax+1)^2 uint16 for L/A, (kmax+1) uint32 for the backtrace, and one mismatch bitmap of pcap bits per
// diagonal (2*kmax+1 of them), pcap = the longest pattern.
static __host__ __device__ __forceinline__ uint32_t lv_mask_words(uint32_t pcap) { return (pcap + 63) >> 6; }
static __host__ __device__ __forceinline__ uint32_t lv_tri_bytes(uint32_t kmax) {
    uint32_t tri = (kmax + 1) * (kmax + 1) * 2;
    tri = (tri + 3) & ~3u;
    return (tri + (kmax + 1) * 4 + 7) & ~7u;
}
static __host__ __device__ __forceinline__ uint32_t lv_lds_bytes(uint32_t kmax, uint32_t pcap) {
    return lv_tri_bytes(kmax) + (2 * kmax + 1) * lv_mask_words(pcap) * 8;
}

// cell = ((L + 2) << 2) | action ; unset cells read as L = -2
static __device__ __forceinline__ uint16_t lv_pack(int L, int act) { return (uint16_t)(((L + 2) << 2) | act); }

// Where the level triangle, the backtrace words and the mismatch bitmaps live: LDS (the normal case) or, TRI_LDS = false, a per-wave buffer
// in HBM -- the paired-end kernel keeps an LDS triangle for limits up to AlignCfg::kmax only and sends the rare call with a larger limit
// (indel-hinted candidates, up to 48 / 60: SURVEY.md 8 preamble) there, so that its LDS footprint allows 4 waves per SIMD.
template <bool TRI_LDS> struct LvMem;
template <> struct LvMem<true>  { typedef LDS_AS uint16_t U16; typedef LDS_AS uint32_t U32; typedef LDS_AS unsigned long long U64; typedef LDS_AS uint8_t U8; };
template <> struct LvMem<false> { typedef uint16_t U16; typedef uint32_t U32; typedef unsigned long long U64; typedef uint8_t U8; };

// planes != NULL: a call that is still running at level LV_PLANES_FROM gets ALL its remaining bitmaps from bit planes of the pattern and the
// text (planes.h), every lane building the one of its own diagonal, instead of two byte-compared bitmaps per level: the perfect-match prefix
// and the first levels -- where most calls end -- stay with the bytes, which are cheaper there (profiles/r03e).
template <bool TRI_LDS = true, typename PSeq, typename TSeq, typename QSeq>
static __device__ __forceinline__ LVResult lv_compute_inl(
    const PSeq &P, const QSeq &Q, int pattern_len, const TSeq &T, int text_len, int k,
    uint16_t *lds_tri_generic, uint32_t kmax, const DevTables *tab, uint32_t pcap, const LvPlanes *planes = nullptr)
{
    typedef typename LvMem<TRI_LDS>::U16 M16; typedef typename LvMem<TRI_LDS>::U32 M32; typedef typename LvMem<TRI_LDS>::U64 M64;
    typedef typename LvMem<TRI_LDS>::U8 M8;
    const int lane = lane_id();
    M16 *lds_tri = (M16 *)lds_tri_generic;
    LVResult res;
    res.score = -1; res.match_probability = 0.0; res.net_indel = 0; res.total_indels = 0; res.text_span = 0;
    if (k < 0) return res;                                  // LandauVishkin.h:117
    if (k > 126) k = 126;                                   // :142
    if (k > (int)kmax) k = (int)kmax;
    res.match_probability = 1.0;
    EMU_STAT(0, 1);

    M32 *bt = (M32 *)((M8 *)lds_tri + (((kmax + 1) * (kmax + 1) * 2 + 3) & ~3u));
    // Mismatch bitmaps, one per diagonal (row = visiting rank), built as a level first needs the diagonal: bit i is set when
    // pattern[i] != text[d+i] or i is past the end of the comparison, so "extend a run from x" is a count-trailing-zeros
    // instead of a byte loop (the reference's countPerfectMatch compares 8 bytes at a time for the same reason, :377-407).
    M64 *mask = (M64 *)((M8 *)lds_tri + lv_tri_bytes(kmax));
    const int nw = (int)lv_mask_words(pcap);
    const int nwu = (pattern_len + 63) >> 6;                // words that can hold a compared position
    auto build_mask = [&](int r) {
        const int d = lv_diag(r);
        const int tl = text_len - d;
        const int end = pattern_len < tl ? pattern_len : tl;
        // (the words are collected per lane -- lane w keeps word w -- and stored by ONE masked store: a store by lane 0 per word was an exec save / restore on
        //  the scalar unit per word)
        unsigned long long mine = 0ull;
        for (int w = 0; w < nwu; w++) {
            const int i = w * 64 + lane;
            const bool mm = i >= end || d + i < 0 || P(i) != T(d + i);     // (text before its start is never part of a run: L(e,d) >= -d)
            const unsigned long long m = BALLOT(mm);
            mine = lane == (w & 63) ? m : mine;
            if ((w & 63) == 63 || w == nwu - 1) { const int w0 = w & ~63; if (w0 + lane <= w) mask[r * nw + w0 + lane] = mine; }
        }
    };

    // ---- e = 0: the perfect-match prefix, compared 64 bytes per step by the whole wave
    const int end0 = pattern_len < text_len ? pattern_len : text_len;
    int run0 = 0;
    while (true) {
        int i = run0 + lane;
        bool same = (i < end0) && (P(i) == T(i));
        uint64_t stopm = BALLOT(!same);
        if (stopm) { run0 += __ffsll((long long)stopm) - 1; break; }
        run0 += WAVE;
    }
    if (run0 > end0) run0 = end0;
    if (run0 == end0) {                                     // :170-185
        EMU_STAT(1, 1);
        int result = pattern_len > end0 ? pattern_len - end0 : 0;
        res.match_probability = tab->perfect[pattern_len];
        if (result > k) { res.score = -1; return res; }
        res.text_span = pattern_len;
        res.score = result;
        return res;
    }
    if (lane == 0) lds_tri[0] = lv_pack(run0, LV_ACT_X);
    WAVE_SYNC();

    int last_best_rank = -1;
    int e;
    for (e = 1; e <= k; e++) {
        const M16 *prev_row = lds_tri + (e - 1) * (e - 1);
        M16 *row = lds_tri + e * e;
        int x_rank = 1 << 30, any_rank = 1 << 30;
        EMU_STAT(2, 1);
        if (planes == nullptr || e < LV_PLANES_FROM) {
            if (e == 1) build_mask(0);
            build_mask(2 * e - 1);
            build_mask(2 * e);
            WAVE_SYNC();
        } else if (e == LV_PLANES_FROM) {
            EMU_STAT(3, 1);
            const int n_pw = nwu, n_sw = ((pattern_len + 2 * k + 63) >> 6) + 1;
            lv_planes_prepare(*planes, k, pattern_len, n_pw, n_sw);
            WAVE_SYNC();
            if (lane <= 2 * k) {
                const int d = lv_diag(lane);
                const int tl = text_len - d;
                const int end = pattern_len < tl ? pattern_len : tl;
                for (int w = 0; w < nwu; w++) mask[lane * nw + w] = lv_planes_word(*planes, k, d, w, end, n_pw, n_sw);
            }
            WAVE_SYNC();
        }
        for (int r0 = 0; r0 <= 2 * e; r0 += WAVE) {
            int r = r0 + lane;
            bool live = r <= 2 * e;
            int d = lv_diag(r);
            bool reached = false; int act = LV_ACT_X;
            if (live) {
                int ad = d < 0 ? -d : d;
                int Lc = (ad <= e - 1) ? ((int)(prev_row[lv_rank(d)] >> 2) - 2) : -2;
                int dl = d - 1, dr = d + 1;
                int adl = dl < 0 ? -dl : dl, adr = dr < 0 ? -dr : dr;
                int Ll = (adl <= e - 1) ? ((int)(prev_row[lv_rank(dl)] >> 2) - 2) : -2;
                int Lr = (adr <= e - 1) ? ((int)(prev_row[lv_rank(dr)] >> 2) - 2) : -2;
                int tl = text_len - d;
                const int end = pattern_len < tl ? pattern_len : tl;
                const M64 *m = mask + r * nw;
                // position of the first mismatch at or after x on this diagonal (x itself when it cannot start a run)
                auto extend = [&](int x) -> int {
                    if (x < 0 || x >= end) return x;
                    int w = x >> 6;
                    unsigned long long v = m[w] >> (x & 63);
                    if (v) return x + (int)__builtin_ctzll(v);
                    for (w++; w < nwu; w++) { v = m[w]; if (v) return w * 64 + (int)__builtin_ctzll(v); }
                    return end;
                };
                int best = extend(Lc + 1);                   // substitution ("up")
                // a run started at or before `best` cannot end beyond it, so the other two moves only need extending
                // when they start beyond it -- which is also exactly when the reference's `> best` tests can succeed
                int left = Ll;                               // deletion
                if (left > best) { left = extend(left); best = left; act = LV_ACT_D; }
                int right = Lr + 1;                          // insertion
                if (right > best) { right = extend(right); best = right; act = LV_ACT_I; }
                reached = (best == pattern_len);
                row[r] = lv_pack(best, act);
            }
            const uint64_t ma = BALLOT(reached);
            if (__builtin_expect(ma != 0ull, 0)) {                 // (one level per call: the other levels pay one compare here, not the two selects)
                const uint64_t mx = BALLOT(reached && act == LV_ACT_X);
                if (mx && x_rank == (1 << 30)) x_rank = r0 + __ffsll((long long)mx) - 1;
                if (any_rank == (1 << 30)) any_rank = r0 + __ffsll((long long)ma) - 1;
            }
        }
        WAVE_SYNC();
        if (x_rank != (1 << 30)) { last_best_rank = x_rank; break; }       // :243-248 (goto got_answer)
        if (any_rank != (1 << 30)) { last_best_rank = any_rank; break; }   // :253-264
    }
    if (last_best_rank < 0) { EMU_STAT(4, 1); res.score = -1; return res; }                // :267-269 (probability stays 1.0)

    // ---- backtrace (:286-304): uniform, every lane walks the same path
    {
        int cur_d = lv_diag(last_best_rank);
        for (int cur_e = e; cur_e >= 1; cur_e--) {
            uint16_t cell = lds_tri[cur_e * cur_e + lv_rank(cur_d)];
            int act = cell & 3;
            int Lcur = (int)(cell >> 2) - 2;
            int pd = act == LV_ACT_I ? cur_d + 1 : act == LV_ACT_D ? cur_d - 1 : cur_d;
            int apd = pd < 0 ? -pd : pd;
            int Lprev = (apd <= cur_e - 1) ? ((int)(lds_tri[(cur_e - 1) * (cur_e - 1) + lv_rank(pd)] >> 2) - 2) : -2;
            int matched = act == LV_ACT_D ? Lcur - Lprev : Lcur - Lprev - 1;
            if (lane == 0) bt[cur_e] = ((uint32_t)(matched & 0xffff) << 2) | (uint32_t)act;
            cur_d = pd;
        }
        WAVE_SYNC();
    }

    // ---- forward pass over the actions (:306-342): FP64 products in the reference's order
    {
        double prob = 1.0;
        int net = 0, total = 0, span = 0;
        int cur_e = 1;
        int offset = run0;
        while (cur_e <= e) {
            uint32_t b = bt[cur_e];
            int action = (int)(b & 3);
            int matched = (int)(int16_t)(b >> 2);
            int count = 1;
            while (cur_e + 1 <= e && matched == 0 && (int)(bt[cur_e + 1] & 3) == action) {
                count++;
                cur_e++;
                matched = (int)(int16_t)(bt[cur_e] >> 2);
            }
            if (action == LV_ACT_I) {
                prob *= tab->indel[count];
                offset += count; net += count; total += count;
            } else if (action == LV_ACT_D) {
                prob *= tab->indel[count];
                offset -= count; net -= count; total += count; span += count;
            } else {
                for (int i = 0; i < count; i++) {
                    int qi = offset < 0 ? 0 : offset;
                    if (qi > pattern_len - 1) qi = pattern_len - 1;
                    prob *= tab->phred[Q(qi)];
                    offset++;
                }
            }
            offset += matched;
            cur_e++;
        }
        prob *= tab->perfect[pattern_len - e];
        span += pattern_len;
        res.match_probability = prob;
        res.net_indel = net; res.total_indels = total; res.text_span = span;
    }
    EMU_STAT(5, 1); EMU_STAT(6, e);
    res.score = e;
    return res;
}

// One copy of the Landau-Vishkin code per kernel (see ag_win.h: ag_dispatch_fn for the why); wave-uniform arguments are made scalar on entry.
static __device__ __forceinline__ ByteSeq seq_uniform(const ByteSeq &s) {
    return ByteSeq{(const uint8_t *)(uintptr_t)first_u64((uint64_t)(uintptr_t)s.p), (int)first_u32((uint32_t)s.stride)};
}
static __device__ __forceinline__ LdsSeq seq_uniform(const LdsSeq &s) {
#if defined(SNAPGPU_WAVE_EMU)
    return LdsSeq{(const uint8_t *)(uintptr_t)first_u64((uint64_t)(uintptr_t)s.p), (int)first_u32((uint32_t)s.stride)};
#else
    return LdsSeq{(LDS_AS const uint8_t *)(uintptr_t)first_u32((uint32_t)(uintptr_t)s.p), (int)first_u32((uint32_t)s.stride)};
#endif
}
template <class S> static __device__ __forceinline__ S seq_uniform(const S &s) { return s; }

// (the call without bit planes -- every call of the default kernels -- has its own entry: nine dwords of plane arguments fewer to move into VGPRs at the
//  call site and back into SGPRs on entry, twenty times per read)
template <bool TRI_LDS, typename PSeq, typename TSeq, typename QSeq>
static __device__ __attribute__((noinline)) LVResult lv_compute_fn_np(
    PSeq P_in, QSeq Q_in, int pattern_len, TSeq T_in, int text_len, int k, uint16_t *lds_tri, uint32_t kmax, const DevTables *tab, uint32_t pcap)
{
    const PSeq P = seq_uniform(P_in); const QSeq Q = seq_uniform(Q_in); const TSeq T = seq_uniform(T_in);
    pattern_len = (int)first_u32((uint32_t)pattern_len); text_len = (int)first_u32((uint32_t)text_len); k = (int)first_u32((uint32_t)k);
    lds_tri = (uint16_t *)(uintptr_t)first_u64((uint64_t)(uintptr_t)lds_tri);
    kmax = first_u32(kmax); pcap = first_u32(pcap);
    tab = (const DevTables *)(uintptr_t)first_u64((uint64_t)(uintptr_t)tab);
    return lv_compute_inl<TRI_LDS>(P, Q, pattern_len, T, text_len, k, lds_tri, kmax, tab, pcap, nullptr);
}

template <bool TRI_LDS, typename PSeq, typename TSeq, typename QSeq>
static __device__ __attribute__((noinline)) LVResult lv_compute_fn(
    PSeq P_in, QSeq Q_in, int pattern_len, TSeq T_in, int text_len, int k,
    uint16_t *lds_tri, uint32_t kmax, const DevTables *tab, uint32_t pcap, uint64_t rp_lds, uint64_t tp_lds, uint64_t work_lds, uint32_t plane_geom, int p_org, int t_org)
{
    const PSeq P = seq_uniform(P_in); const QSeq Q = seq_uniform(Q_in); const TSeq T = seq_uniform(T_in);
    pattern_len = (int)first_u32((uint32_t)pattern_len); text_len = (int)first_u32((uint32_t)text_len); k = (int)first_u32((uint32_t)k);
    lds_tri = (uint16_t *)(uintptr_t)first_u64((uint64_t)(uintptr_t)lds_tri);
    kmax = first_u32(kmax); pcap = first_u32(pcap);
    tab = (const DevTables *)(uintptr_t)first_u64((uint64_t)(uintptr_t)tab);
    // bit planes (planes.h): LDS addresses of the pattern's and the text's planes (~0: none), words per plane << 16 / << 24 | stride + 1
    rp_lds = first_u64(rp_lds); tp_lds = first_u64(tp_lds); work_lds = first_u64(work_lds); plane_geom = first_u32(plane_geom);
    p_org = (int)first_u32((uint32_t)p_org); t_org = (int)first_u32((uint32_t)t_org);
    if (rp_lds != ~0ull) {
        LvPlanes lp;
        const int pw = (int)((plane_geom >> 16) & 0x7fu), tw = (int)(plane_geom >> 24);
        const LDS_AS unsigned long long *rb = (const LDS_AS unsigned long long *)(uintptr_t)rp_lds, *tb = (const LDS_AS unsigned long long *)(uintptr_t)tp_lds;
        lp.p0 = rb; lp.p1 = rb + pw; lp.pn = rb + 2 * pw; lp.po = rb + 3 * pw;
        lp.t0 = tb; lp.t1 = tb + tw; lp.tn = tb + 2 * tw;
        lp.p_org = p_org; lp.t_org = t_org; lp.st = (int)(plane_geom & 0xffu) - 1; lp.p_words = pw; lp.t_words = tw;
        lp.work = (LDS_AS unsigned long long *)(uintptr_t)work_lds; lp.plain = ((plane_geom >> 23) & 1u) != 0;
        return lv_compute_inl<TRI_LDS>(P, Q, pattern_len, T, text_len, k, lds_tri, kmax, tab, pcap, &lp);
    }
    return lv_compute_inl<TRI_LDS>(P, Q, pattern_len, T, text_len, k, lds_tri, kmax, tab, pcap);
}

template <bool TRI_LDS = true, typename PSeq, typename TSeq, typename QSeq>
static __device__ __forceinline__ LVResult lv_compute(
    const PSeq &P, const QSeq &Q, int pattern_len, const TSeq &T, int text_len, int k,
    uint16_t *lds_tri, uint32_t kmax, const DevTables *tab, uint32_t pcap, const LvPlanes *planes = nullptr)
{
#if !defined(SNAPGPU_AG_LV_FUNCTIONS) || (defined(SNAPGPU_LV_INLINE) && !defined(PAIRED_AGC))      // (SNAPGPU_LV_INLINE: LV inlined in the single-end kernels, affine gap still a function)
    return lv_compute_inl<TRI_LDS>(P, Q, pattern_len, T, text_len, k, lds_tri, kmax, tab, pcap, planes);
#else
    // (the planes' LDS pointers cross the call as 32-bit LDS addresses)
    if (planes != nullptr)
        return lv_compute_fn<TRI_LDS, PSeq, TSeq, QSeq>(P, Q, pattern_len, T, text_len, k, lds_tri, kmax, tab, pcap,
                                                        (uint64_t)(uintptr_t)planes->p0, (uint64_t)(uintptr_t)planes->t0, (uint64_t)(uintptr_t)planes->work,
                                                        (uint32_t)(planes->st + 1) | ((uint32_t)planes->p_words << 16) | (planes->plain ? 1u << 23 : 0u) | ((uint32_t)planes->t_words << 24),
                                                        planes->p_org, planes->t_org);
    return lv_compute_fn_np<TRI_LDS, PSeq, TSeq, QSeq>(P, Q, pattern_len, T, text_len, k, lds_tri, kmax, tab, pcap);
#endif
}

