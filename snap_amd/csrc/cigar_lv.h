// cigar_lv.h -- the CIGAR of a written read, one wavefront per read (result -> SAM record).
//
// Restates SAMFormat::computeCigar, Landau-Vishkin variant (SNAPLib/SAM.cpp:2354-2467), over
// LandauVishkinWithCigar::computeEditDistanceNormalized (SNAPLib/LandauVishkin.cpp:507-648) and
// LandauVishkinWithCigar::computeEditDistance (:141-505) with BAM_CIGAR_OPS output (writeCigar :124-131).
// This is NOT the scoring Landau-Vishkin of lv.h: diagonals are visited 0, -1, +1, -2, +2, ... (:222), a cell takes the
// predecessor that reaches furthest and, on a tie, the one with fewer indels so far (totalIndels[][], :264-270), the answer
// of a level is the first diagonal that ends the pattern without indels, otherwise the one with the fewest (:276-292), and
// an alignment whose e edits can all be substitutions is written without looking at the table at all (:305-368).
//
// GPU mapping: the cells of level e depend only on level e-1, so the 2e+1 diagonals of a level are computed concurrently,
// lane r = the diagonal with visiting rank r; "first in visiting order" is the lowest lane of a ballot.  Cells live in a
// per-wave slab of HBM scratch (L2-resident; row e starts at e*e, one dword per cell: L+2 | totalIndels | action), read and
// pattern bytes in LDS.  Backtrace and emission are short scalar loops (at most e steps) run wave-uniformly; lane 0 stores.
#pragma once
#include "dev_common.h"

#define LVC_MAX_K 127                        // MAX_K, LandauVishkin.h:11
#define LVC_ACT_D 0                          // PrevDelta = -1: the cell came from diagonal d-1
#define LVC_ACT_X 1
#define LVC_ACT_I 2                          // PrevDelta = +1
#define LVC_OP_M 0u                          // BAMAlignment::CigarToCode (Bam.cpp)
#define LVC_OP_I 1u
#define LVC_OP_D 2u
#define LVC_OP_EQ 7u
#define LVC_OP_X 8u

static __host__ __device__ __forceinline__ uint32_t lvc_scratch_bytes() { return (LVC_MAX_K * LVC_MAX_K + LVC_MAX_K + 1) * 4u; }   // cells of levels 0..126 + backtrace
static __host__ __device__ __forceinline__ uint32_t lvc_lds_bytes(uint32_t RL) { return ((RL + 15) & ~15u) + ((RL + LVC_MAX_K + 15) & ~15u); }

static __device__ __forceinline__ int lvc_rank(int d) { return d < 0 ? -2 * d - 1 : 2 * d; }          // 0, -1, +1, -2, +2, ...
static __device__ __forceinline__ int lvc_diag(int r) { return (r & 1) ? -((r + 1) >> 1) : (r >> 1); }
static __device__ __forceinline__ uint32_t lvc_cell(int L, int ti, int act) { return (uint32_t)(L + 2) | ((uint32_t)ti << 16) | ((uint32_t)act << 24); }
static __device__ __forceinline__ int lvc_L(uint32_t c) { return (int)(c & 0xffffu) - 2; }
static __device__ __forceinline__ int lvc_TI(uint32_t c) { return (int)((c >> 16) & 0xffu); }
static __device__ __forceinline__ int lvc_act(uint32_t c) { return (int)((c >> 24) & 3u); }

struct LVCOut {                              // the op list being written (lane 0 stores; n is wave-uniform)
    uint32_t *ops;
    int cap, n;
    bool ok;
    __device__ __forceinline__ void put(int count, uint32_t code, int lane) {                        // writeCigar, :77-82 / :124-131
        if (count <= 0 || !ok) return;
        if (n >= cap || count >= (1 << 28)) { ok = false; return; }
        if (lane == 0) ops[n] = ((uint32_t)count << 4) | code;
        n++;
    }
};

struct LVCResult {
    int score;                               // edit distance, -1 (more than k), -2 (op buffer too small)
    int net_indel;
    int n_ops;
};

// computeEditDistance (:141-505).  pat[0, plen) and txt[0, tlen) are in LDS; bytes outside compare unequal, which is what the
// reference sees whenever tlen >= plen + k (computeCigar passes plen + MAX_K).
static __device__ __forceinline__ LVCResult lvc_compute(const uint8_t *pat, int plen, const uint8_t *txt, int tlen, int k, bool use_m,
                                                        uint32_t *cells, uint32_t *ops, int ops_cap)
{
    const int lane = lane_id();
    LVCResult res; res.score = -1; res.net_indel = 0; res.n_ops = 0;
    LVCOut out; out.ops = ops; out.cap = ops_cap; out.n = 0; out.ok = true;
    if (k >= LVC_MAX_K) k = LVC_MAX_K - 1;                                                            // :163
    uint32_t *bt = cells + LVC_MAX_K * LVC_MAX_K;
    const int end = plen < tlen ? plen : tlen;

    // ---- L[0][0]: the exact-match run (:170-186), 64 bytes per step
    int run0 = 0;
    while (run0 < end) {
        const int i = run0 + lane;
        const bool same = i < end && pat[i] == txt[i];
        const unsigned long long stopm = BALLOT(!same);
        if (stopm) { run0 += __ffsll((long long)stopm) - 1; break; }
        run0 += WAVE;
    }
    if (run0 > end) run0 = end;
    if (run0 == end) {                                                                                // :187-213
        if (use_m) out.put(plen, LVC_OP_M, lane);
        else { out.put(end, LVC_OP_EQ, lane); if (plen > end) out.put(plen - end, LVC_OP_X, lane); }
        res.score = out.ok ? 0 : -2; res.n_ops = out.n;
        return res;
    }
    if (lane == 0) cells[0] = lvc_cell(run0, 0, LVC_ACT_X);
    WAVE_SYNC();

    int e, ans_rank = -1;
    for (e = 1; e <= k; e++) {
        const uint32_t *prev = cells + (e - 1) * (e - 1);
        uint32_t *row = cells + e * e;
        int best_key = 0x7fffffff;                          // (indels << 8 | rank) of the best end-reaching diagonal of this level
        bool zero_found = false;
        for (int r0 = 0; r0 <= 2 * e; r0 += WAVE) {
            const int r = r0 + lane;
            const bool act_lane = r <= 2 * e;
            int bestbest = -1, bestdelta = 0, bbi = LVC_MAX_K + 1;
            if (act_lane) {
                const int d = lvc_diag(r);
                for (int dx = 0; dx < 3; dx++) {
                    // PrevDelta (:66-69): straight first, then the neighbour nearer to diagonal 0
                    const int delta = dx == 0 ? 0 : (d > 0 ? (dx == 1 ? -1 : +1) : (dx == 1 ? +1 : -1));
                    const int dp = d + delta;
                    if (dp < -(e - 1) || dp > e - 1) continue;                                        // never written: L = -2 (:14-22)
                    const uint32_t c = prev[lvc_rank(dp)];
                    int best = lvc_L(c) + (delta >= 0 ? 1 : 0);
                    const int bi = lvc_TI(c) + (delta != 0 ? 1 : 0);
                    if (best < 0) continue;
                    if (best < plen && d + best >= 0 && d + best < tlen && pat[best] == txt[d + best]) {   // :239-262
                        const int e2 = plen < tlen - d ? plen : tlen - d;
                        int x = best + 1;
                        while (x < plen && d + x < tlen && pat[x] == txt[d + x]) x++;
                        best = x < e2 ? x : e2;
                    }
                    if (best > bestbest || (best == bestbest && bi < bbi)) { bestbest = best; bestdelta = delta; bbi = bi; }
                }
                row[r] = lvc_cell(bestbest, bbi, bestdelta + 1);
            }
            // :276-292 -- in visiting order: the first diagonal that ends the pattern with no indels wins at once, otherwise the
            // first one with the fewest indels
            const bool reached = act_lane && bestbest == plen;
            const unsigned long long zm = BALLOT(reached && bbi == 0);
            if (zm) { ans_rank = r0 + __ffsll((long long)zm) - 1; zero_found = true; break; }
            int key = reached ? ((bbi << 8) | lane) : 0x7fffffff;
            for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(key, o); key = t < key ? t : key; }
            if (key != 0x7fffffff) {
                const int kk = ((key >> 8) << 8) | (r0 + (key & 0xff));
                if ((kk >> 8) < (best_key >> 8) || best_key == 0x7fffffff) best_key = kk;             // strictly fewer indels only (:288)
            }
        }
        WAVE_SYNC();
        if (zero_found) break;
        if (best_key != 0x7fffffff) { ans_rank = best_key & 0xff; break; }
    }
    if (ans_rank < 0) return res;                                                                     // more than k edits (:300)
    const int ans_d = lvc_diag(ans_rank);

    // ---- can e substitutions explain it? (:305-312)
    int straight = plen - end;
    for (int i0 = 0; i0 < end; i0 += WAVE) {
        const int i = i0 + lane;
        straight += __popcll(BALLOT(i < end && pat[i] != txt[i]));
    }
    if (straight == e) {                                                                              // :313-368
        if (use_m) out.put(plen, LVC_OP_M, lane);
        else {
            // runs of = / X over [0, end), then the tail of the pattern past the text as X
            int streak_start = 0;
            bool matching = pat[0] == txt[0];
            for (int i0 = 0; i0 < end && out.ok; i0 += WAVE) {
                const int i = i0 + lane;
                const unsigned long long eqm = BALLOT(i < end && pat[i] == txt[i]);
                const int nb = end - i0 < WAVE ? end - i0 : WAVE;
                int b = 0;
                while (b < nb) {                            // next position whose state differs from `matching`
                    unsigned long long want = matching ? ~eqm : eqm;
                    want = b < 64 ? (want >> b) << b : 0ull;
                    if (nb < 64) want &= (1ull << nb) - 1ull;
                    if (!want) break;
                    const int i_change = i0 + (__ffsll((long long)want) - 1);
                    out.put(i_change - streak_start, matching ? LVC_OP_EQ : LVC_OP_X, lane);
                    matching = !matching; streak_start = i_change;
                    b = i_change - i0;
                }
            }
            if (plen > streak_start) {
                if (!matching) out.put(plen - streak_start, LVC_OP_X, lane);
                else { out.put(end - streak_start, LVC_OP_EQ, lane); if (plen > end) out.put(plen - end, LVC_OP_X, lane); }
            }
        }
        res.score = out.ok ? e : -2; res.n_ops = out.n;
        return res;
    }

    // ---- trace back (:394-420): bt[ce] = action | matched << 2
    {
        int cur_d = ans_d;
        for (int ce = e; ce >= 1; ce--) {
            const uint32_t c = first_u32(cells[ce * ce + lvc_rank(cur_d)]);
            const int a = lvc_act(c);
            const int pd = a == LVC_ACT_I ? cur_d + 1 : (a == LVC_ACT_D ? cur_d - 1 : cur_d);
            const uint32_t pc = first_u32(cells[(ce - 1) * (ce - 1) + lvc_rank(pd)]);
            const int matched = lvc_L(c) - lvc_L(pc) - (a == LVC_ACT_D ? 0 : 1);
            if (lane == 0) bt[ce] = (uint32_t)a | ((uint32_t)matched << 2);
            cur_d = pd;
        }
        WAVE_SYNC();
    }
    // ---- emit forwards (:422-497)
    int net_indel = 0, acc_m = 0;
    {
        const int l00 = lvc_L(first_u32(cells[0]));
        if (use_m) acc_m = l00; else out.put(l00, LVC_OP_EQ, lane);
        int ce = 1;
        while (ce <= e && out.ok) {
            uint32_t b = first_u32(bt[ce]);
            const int action = (int)(b & 3u);
            int count = 1;
            while (ce + 1 <= e && (b >> 2) == 0) {
                const uint32_t nx = first_u32(bt[ce + 1]);
                if ((int)(nx & 3u) != action) break;
                count++; ce++; b = nx;
            }
            const int matched = (int)(b >> 2);
            if (action == LVC_ACT_I) net_indel -= count; else if (action == LVC_ACT_D) net_indel += count;
            const uint32_t code = action == LVC_ACT_I ? LVC_OP_I : (action == LVC_ACT_D ? LVC_OP_D : LVC_OP_X);
            if (use_m) {
                if (action == LVC_ACT_X) acc_m += count;
                else { if (acc_m != 0) { out.put(acc_m, LVC_OP_M, lane); acc_m = 0; } out.put(count, code, lane); }
            } else out.put(count, code, lane);
            if (matched > 0) { if (use_m) acc_m += matched; else out.put(matched, LVC_OP_EQ, lane); }
            ce++;
        }
        if (use_m && acc_m != 0) out.put(acc_m, LVC_OP_M, lane);
    }
    res.score = out.ok ? e : -2; res.net_indel = net_indel; res.n_ops = out.n;
    return res;
}

// One item of SAMFormat::computeCigar (SAM.cpp:2354-2467) + computeEditDistanceNormalized's leading-indel convention (:607-622).
struct CigarItemOut { int n_ops, edit_distance, add_front_clipping; long long extra_after; };

static __device__ __forceinline__ CigarItemOut cigar_lv_item(const DevIndex &ix, const uint8_t *data, long long data_len, long long extra_before,
                                                             long long loc, bool use_m, uint8_t *lds_pat, uint8_t *lds_txt,
                                                             uint32_t *cells, uint32_t *ops, int ops_cap)
{
    const int lane = lane_id();
    CigarItemOut o; o.n_ops = 0; o.edit_distance = 0; o.add_front_clipping = 0; o.extra_after = 0;
    loc += extra_before; data += extra_before; data_len -= extra_before;                               // :2381-2383
    // getContigAtLocation (Genome.cpp:574-594)
    int lo = 0, hi = (int)ix.n_contigs - 1, c = -1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if ((long long)first_u64(ix.contig_begin[mid]) <= loc) { c = mid; lo = mid + 1; } else hi = mid - 1;
    }
    if (c < 0 || data_len < 0) { o.n_ops = -1; return o; }
    const long long nb = (long long)ix.n_bases;
    const long long cend = c == (int)ix.n_contigs - 1 ? nb : (long long)first_u64(ix.contig_begin[c + 1]);
    const long long real_end = cend - (long long)ix.chromosome_padding;
    if (loc + data_len > real_end) o.extra_after = loc + data_len - real_end;                          // :2387-2395
    {   // getSubstring(genomeLocation, dataLength) == NULL -> "*" (:2397-2408; Genome.h:339-367)
        bool ok;
        if (loc > nb || loc + data_len > nb + 1000) ok = false;
        else if (data_len <= (long long)ix.chromosome_padding && first_u32(ix.genome[loc]) != 'n') ok = true;
        else if (data_len == 0) ok = true;
        else ok = cend > loc + data_len;
        if (!ok) { o.n_ops = -1; return o; }
    }
    // stage the read and the reference window [loc, loc + data_len + MAX_K) once; bytes past the padded genome read as 0
    const long long readable = nb + (long long)ix.genome_pad - loc;
    const int full_t = (int)data_len + LVC_MAX_K;
    for (int i = lane; i < (int)data_len; i += WAVE) lds_pat[i] = data[i];
    for (int i = lane; i < full_t; i += WAVE) lds_txt[i] = i < readable ? ix.genome[loc + i] : (uint8_t)0;
    WAVE_SYNC();
    for (long long pass = 0; pass <= data_len; pass++) {                                               // the first call and the loop of :2435-2460
        const int plen = (int)(data_len - o.extra_after);
        const LVCResult r = lvc_compute(lds_pat, plen, lds_txt, plen + LVC_MAX_K, LVC_MAX_K - 1, use_m, cells, ops, ops_cap);
        WAVE_SYNC();
        o.edit_distance = r.score; o.n_ops = r.score < 0 ? 0 : r.n_ops; o.add_front_clipping = 0;
        if (r.score >= 0 && r.n_ops > 0) {                                                             // LandauVishkin.cpp:607-622
            const uint32_t op0 = first_u32(ops[0]);
            if ((op0 & 0xfu) == LVC_OP_D) {
                o.add_front_clipping = (int)(op0 >> 4);
                if (o.add_front_clipping != 0) { o.edit_distance = 0; o.n_ops = 0; }
            } else if ((op0 & 0xfu) == LVC_OP_I) o.add_front_clipping = -(int)(op0 >> 4);
        }
        const int net_indel = r.net_indel;                                                             // (0 unless the table was traced back)
        if (pass == 0 && o.add_front_clipping != 0) return o;                                          // SAM.cpp:2425-2431
        long long nw = loc + data_len + net_indel - real_end; if (nw < 0) nw = 0;                      // :2434 / :2459
        if (nw == o.extra_after) return o;
        o.extra_after = nw;
    }
    return o;
}
