// cigar_k.hip -- kernel of snapgpu_compute_cigar_lv: SAMFormat::computeCigar (Landau-Vishkin variant) for a batch of written
// reads, one wavefront per read, persistent grid.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c cigar_k.hip
#include <hip/hip_runtime.h>
#include "cigar_lv.h"
#include "cigar_ag.h"
#include "sam_fields.h"
#include "adjust.h"
#include "cigar_args.h"

__global__ __launch_bounds__(256) void k_cigar_lv(CigarArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t wave_slot = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave_in_block;
    const uint32_t per_wave = lvc_lds_bytes(a.RL);
    uint8_t *lds_pat = lds + (size_t)wave_in_block * per_wave;
    uint8_t *lds_txt = lds_pat + ((a.RL + 15) & ~15u);
    uint32_t *cells = (uint32_t *)(a.scratch + (size_t)wave_slot * lvc_scratch_bytes());
    while (true) {
        uint32_t i = 0;
        if (lane == 0) i = atomicAdd(a.work_counter, 1u);
        i = first_u32(i);
        if (i >= a.n) break;
        const uint64_t off = first_u64(a.off[i]);
        const long long len = (long long)(int)first_u32((uint32_t)a.len[i]);
        const long long loc = (long long)first_u64((uint64_t)a.loc[i]);
        const long long xb = (long long)(int)first_u32((uint32_t)a.extra_before[i]);
        uint32_t *ops = a.ops + (size_t)i * a.ops_stride;
        const CigarItemOut o = cigar_lv_item(a.ix, a.data + off, len, xb, loc, a.use_m != 0, lds_pat, lds_txt, cells, ops, (int)a.ops_stride);
        if (lane == 0) {
            a.n_ops[i] = o.n_ops; a.edit_distance[i] = o.edit_distance; a.add_front_clipping[i] = o.add_front_clipping;
            a.extra_after[i] = o.extra_after;
        }
        WAVE_SYNC();
    }
}

extern "C" void snapgpu_launch_cigar_lv(const CigarArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL(k_cigar_lv, dim3(blocks), dim3(256), lds_bytes, s, *a);
}

// SAMFormat::computeCigar, affine-gap variant: one wavefront per read (cigar_ag.h)
__global__ __launch_bounds__(256) void k_cigar_ag(CigarAGArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t wave_slot = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave_in_block;
    uint8_t *my = lds + (size_t)wave_in_block * agc_lds_bytes(a.RL);
    uint8_t *scratch = a.scratch + (size_t)wave_slot * a.scratch_stride;
    AGCParams prm; prm.match = a.prm.match; prm.sub = a.prm.sub; prm.gap_open = a.prm.gap_open; prm.gap_ext = a.prm.gap_ext;
    while (true) {
        uint32_t i = 0;
        if (lane == 0) i = atomicAdd(a.work_counter, 1u);
        i = first_u32(i);
        if (i >= a.n) break;
        const uint64_t off = first_u64(a.off[i]);
        const long long len = (long long)(int)first_u32((uint32_t)a.len[i]);
        const long long loc = (long long)first_u64((uint64_t)a.loc[i]);
        const long long xb = (long long)(int)first_u32((uint32_t)a.extra_before[i]);
        const int k = (int)first_u32((uint32_t)a.score[i]);
        uint32_t *ops = a.ops + (size_t)i * a.ops_stride;
        const CigarAGItemOut o = cigar_ag_item(a.ix, prm, a.data + off, a.quals + off, len, k, xb, loc, a.use_m != 0, my, a.RL, scratch, ops, (int)a.ops_stride);
        if (lane == 0) {
            a.n_ops[i] = o.n_ops; a.edit_distance[i] = o.edit_distance; a.add_front_clipping[i] = o.add_front_clipping;
            a.extra_after[i] = o.extra_after; a.tail_ins[i] = o.tail_ins; a.stale[i] = o.stale;
        }
        WAVE_SYNC();
    }
}

extern "C" void snapgpu_launch_cigar_ag(const CigarAGArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL(k_cigar_ag, dim3(blocks), dim3(256), lds_bytes, s, *a);
}

// result -> computed fields of the SAM record (sam_fields.h): one wavefront per read
#ifndef SAMF_WAVES
#define SAMF_WAVES 8            // waves per SIMD the SAM-field kernels are built for (blocks of four waves: as many blocks per CU).  The kernels are
                                // latency-bound (8 of 64 lanes in the affine-gap CIGAR): 4.32 M reads/s at 4 (131 VGPRs, rounds 2-3), 5.31 M at 6, 5.73 M at 8 (profiles/r04n)
#endif
__global__ __launch_bounds__(256, SAMF_WAVES) void k_sam_fields(SamFieldsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t wave_slot = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave_in_block;
    uint8_t *my = lds + (size_t)wave_in_block * agc_lds_bytes(a.RL);
    uint8_t *scratch = a.scratch + (size_t)wave_slot * a.scratch_stride;
    uint8_t *oriented = scratch;                                                      // 2 * RL bytes
    uint32_t *lv_cells = (uint32_t *)(scratch + ((2 * a.RL + 255) & ~255u));
    uint8_t *ag_scratch = (uint8_t *)lv_cells + ((lvc_scratch_bytes() + 255) & ~255u);
    AGCParams prm; prm.match = a.prm.match; prm.sub = a.prm.sub; prm.gap_open = a.prm.gap_open; prm.gap_ext = a.prm.gap_ext;
    while (true) {
        uint32_t i = 0;
        if (lane == 0) i = atomicAdd(a.work_counter, 1u);
        i = first_u32(i);
        if (i >= a.n) break;
        const uint64_t b = first_u64(a.offsets[i]), e = first_u64(a.offsets[i + 1]);
        snapgpu_single_result r;                                                  // the fields the writer looks at, as wave-uniform values
        {
            const snapgpu_single_result *rp = &a.results[i];
            r.status = (int32_t)first_u32((uint32_t)rp->status); r.direction = (int32_t)first_u32((uint32_t)rp->direction);
            r.location = (int64_t)first_u64((uint64_t)rp->location); r.orig_location = 0;
            r.score = (int32_t)first_u32((uint32_t)rp->score); r.score_prior_to_clipping = 0;
            r.mapq = (int32_t)first_u32((uint32_t)rp->mapq);
            r.clipping_for_read_adjustment = (int32_t)first_u32((uint32_t)rp->clipping_for_read_adjustment);
            r.used_affine_gap_scoring = (int32_t)first_u32((uint32_t)rp->used_affine_gap_scoring);
            r.bases_clipped_before = (int32_t)first_u32((uint32_t)rp->bases_clipped_before);
            r.bases_clipped_after = (int32_t)first_u32((uint32_t)rp->bases_clipped_after);
            r.ag_score = 0; r.supplementary = (int32_t)first_u32((uint32_t)rp->supplementary); r.seed_offset = 0; r.match_probability = 0.0;
            r.probability_all_candidates = 0.0; r.popular_seeds_skipped = 0; r.reserved = 0;
        }
        uint32_t *ops = a.ops + (size_t)i * a.ops_stride;
        const int F0 = (int)first_u32((uint32_t)a.front_clip[i]), D0 = (int)first_u32((uint32_t)a.data_len[i]);
        const SamfPre *pre = a.pre ? (const SamfPre *)(a.pre + (size_t)i * a.pre_stride) : nullptr;
        const SamFieldsOut o = sam_fields_single_item(a.ix, prm, a.use_affine_gap != 0, a.use_m != 0, a.bases + b, a.quals + b, (int)(e - b), F0, D0, r,
                                                      my, a.RL, oriented, lv_cells, ag_scratch, ops, (int)a.ops_stride, false, pre);
        if (lane == 0) {
            a.flag[i] = o.flag; a.contig[i] = o.contig; a.pos[i] = o.pos; a.mapq[i] = o.mapq; a.n_ops[i] = o.n_ops; a.nm[i] = o.nm; a.stale[i] = o.stale;
        }
        WAVE_SYNC();
    }
}

extern "C" void snapgpu_launch_sam_fields(const SamFieldsArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL(k_sam_fields, dim3(blocks), dim3(256), lds_bytes, s, *a);
}

// ---- the banded row loops of a batch's records, eight reads to a wavefront (cigar_ag.h: SamfPre) -------------------------------------------
// One read per 8-lane group g = lane / 8, lane el = lane % 8 = the SSE element.  A group takes part when its read's FIRST affine-gap cigar call
// (sam_fields_single_item's attempt 0 -> cigar_ag_item's pass 0) is a banded call with at most two vectors per segment (edit distance <= 7)
// and nothing unusual about it; what it computes is agc_banded (cigar_ag.h) with the group's lanes as the eight lanes that store: the same
// operations on the same values, in the reference's order (a segment's vectors one after the other, as the SSE code visits them).
// LDS per group: the oriented, clipped read; the reference text the rows need; H of the previous row, H of this row, E (int16 per position).
static __host__ __device__ __forceinline__ uint32_t samf_dp8_hn(uint32_t RL) { return (RL + 15u) & ~15u; }
static __host__ __device__ __forceinline__ uint32_t samf_dp8_group_bytes(uint32_t RL) {
    return ((RL + 15u) & ~15u) + ((samf_pre_rows(RL) + 15u) & ~15u) + 3u * 2u * samf_dp8_hn(RL);
}
extern "C" size_t snapgpu_samf_dp8_lds_per_wave(uint32_t RL) { return (size_t)8 * samf_dp8_group_bytes(RL); }

__global__ __launch_bounds__(256) void k_samf_dp8(SamFieldsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id(), el = lane & 7, g = lane >> 3;
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t RL = a.RL, HN = samf_dp8_hn(RL), GB = samf_dp8_group_bytes(RL);
    uint8_t *mine = lds + (size_t)wave_in_block * 8u * GB + (size_t)g * GB;
    uint8_t *pat = mine, *txt = mine + ((RL + 15u) & ~15u);
    int16_t *Hb = (int16_t *)(txt + ((samf_pre_rows(RL) + 15u) & ~15u));          // [2][HN], then E[HN]
    int16_t *E = Hb + 2 * HN;
    const int open = a.prm.gap_open, ext = a.prm.gap_ext, score_init = AGC_MAX_READ_LENGTH;
    const unsigned long long gmask = 0xffull << (8 * g);
    const long long nb = (long long)a.ix.n_bases;
    while (true) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(a.pre_counter, 8u);
        base = first_u32(base);
        if (base >= a.n) break;
        const uint32_t r = base + (uint32_t)g;
        // ---- is this read's first cigar call the case this kernel runs?  (sam_fields_single_item attempt 0, cigar_ag_item pass 0)
        bool elig = r < a.n;
        int plen = 0, w = 0, dir = 0, bcb = 0, U = 0;
        long long loc = 0;
        uint64_t rb = 0;
        SamfPre *pre = (SamfPre *)(a.pre + (size_t)(elig ? r : base) * a.pre_stride);
        if (elig) {
            const snapgpu_single_result *rp = &a.results[r];
            rb = a.offsets[r]; U = (int)(a.offsets[r + 1] - rb);
            const int status = rp->status, score = rp->score, F0 = a.front_clip[r], D0 = a.data_len[r], addF = rp->clipping_for_read_adjustment;
            loc = rp->location; dir = rp->direction;
            const bool ag_branch = a.use_affine_gap != 0 && (rp->used_affine_gap_scoring != 0 || score > 0);
            const int front = F0 + addF, dlen = D0 - addF;
            int clipped = dlen, bca;
            if (dir == 1) { bcb = U - clipped - front; bca = front; } else { bcb = front; bca = U - clipped - bcb; }
            bcb += rp->bases_clipped_before; bca += rp->bases_clipped_after; clipped -= rp->bases_clipped_before + rp->bases_clipped_after;
            elig = status != SNAPGPU_NotFound && ag_branch && loc >= 0 && loc < nb && score >= 0 && score <= SAMF_PRE_MAX_W && U <= (int)RL &&
                   clipped >= 3 * (2 * score + 1) && bcb >= 0 && bcb + clipped <= U && dlen >= 0;
            if (elig) {
                int lo = 0, hi = (int)a.ix.n_contigs - 1, c = -1;                   // Genome::getContigAtLocation
                while (lo <= hi) { const int mid = (lo + hi) >> 1; if ((long long)a.ix.contig_begin[mid] <= loc) { c = mid; lo = mid + 1; } else hi = mid - 1; }
                const long long cend = c < 0 ? 0 : (c == (int)a.ix.n_contigs - 1 ? nb : (long long)a.ix.contig_begin[c + 1]);
                // inside its contig as the writer sees it (no `extra`), clear of the contig's end as the cigar sees it (no extra_after), substring there
                elig = c >= 0 && loc + dlen <= cend && loc + clipped <= cend - (long long)a.ix.chromosome_padding && cend > loc + clipped;
            }
            plen = clipped; w = score;
        }
        const int rows_cap = (int)samf_pre_rows(RL);
        if (el == 0 && r < a.n) pre->valid = 0;
        if (!BALLOT(elig)) continue;
        // the group's shape: vectors per segment (1 or 2), segment length, segments
        const int bw = 2 * w + 1 < plen ? 2 * w + 1 : plen;
        const int nv = elig ? (bw + 7) >> 3 : 1, seg_len = nv * 8, num_seg = elig ? (plen + seg_len - 1) / seg_len : 0;
        // ---- stage the group's read (as SAM prints it, clipped) and reference text
        {
            // (loop bounds are the wave's maximum over the eligible groups, so that the wave's control flow stays uniform)
            int m = elig ? plen : 0;
            for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(m, o); m = t > m ? t : m; }
            const int top = (int)first_u32((uint32_t)m);
            for (int j = el; j < top; j += 8) {
                if (elig && j < plen) {
                    const int x = bcb + j;
                    pat[j] = dir == 1 ? rc_base(a.bases[rb + (uint64_t)(U - 1 - x)]) : a.bases[rb + (uint64_t)x];
                }
            }
            const long long readable = nb + (long long)a.ix.genome_pad - loc;
            for (int j = el; j < top + LVC_MAX_K && j < rows_cap; j += 8) {     // (every row the loop below can reach: plen + LVC_MAX_K of them, as cigar_ag_item stages)
                if (elig && j < plen + LVC_MAX_K) txt[j] = j < readable ? a.ix.genome[loc + j] : (uint8_t)0;
            }
        }
        // ---- first row (AffineGapVectorized.cpp:611-628): a lane's scoreFirstRow keeps its last value past the pattern's end
        {
            int sfr = 0;
            int ns_top = num_seg;
            for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(ns_top, o); ns_top = t > ns_top ? t : ns_top; }
            ns_top = (int)first_u32((uint32_t)ns_top);
            for (int sg = 0; sg < ns_top; sg++) {
                for (int v = 0; v < 2; v++) {
                    if (elig && sg < num_seg && v < nv) {
                        const int p = sg * seg_len + el * nv + v, x8 = (sg * nv + v) * 8 + el;
                        if (p < plen) { const int x = score_init - (open + p * ext); sfr = x > 0 ? x : 0; }
                        Hb[x8] = (int16_t)sfr; Hb[HN + x8] = 0; E[x8] = 0;
                    }
                }
            }
        }
        WAVE_SYNC();
        int score = score_init, text_used = -1, idle_rows = 0, cur = 0;
        bool going = elig;
        const int tlen = plen + LVC_MAX_K;
        uint8_t *btout = (uint8_t *)(pre + 1);
        for (int i = 0; ; i++) {
            if (going && (i >= tlen || i >= rows_cap)) going = false;
            if (!BALLOT(going)) break;
            int16_t *Hp = Hb + (cur ? HN : 0), *Hm = Hb + (cur ? 0 : HN);
            const int tb = going ? (int)base_value(txt[i]) : 4;
            int f = 0, X = 0;
            const int band_beg = i - w > 0 ? i - w : 0;
            const int band_end = i + w < plen - 1 ? i + w : plen - 1;
            const int seg_beg = band_beg / seg_len, seg_end = band_end / seg_len;
            for (int jj = 0; jj < 2; jj++) {
                const int j = seg_beg + jj;
                const bool act = going && j <= seg_end;
                // diagonal input: the previous row's H of the segment's LAST vector one element down; element 0 takes h_init (:639-657)
                int h = __shfl_up(act ? (int)Hp[(j * nv + nv - 1) * 8 + el] : 0, 1);
                if (el == 0) {
                    if (j == 0) h = (int)(int16_t)(i > 0 ? score_init - (open + (i - 1) * ext) : score_init);
                    else if (band_beg > j * seg_len) h = 0;
                    else h = act ? (int)Hp[(j * nv - 1) * 8 + 7] : 0;
                }
                int hv[2] = {0, 0}, btk[2] = {0, 0};
                bool ka[2];
#pragma unroll
                for (int k = 0; k < 2; k++) {                               // first pass over the segment's vectors (:659-735)
                    ka[k] = act && k < nv && j * seg_len + k <= band_end;
                    if (ka[k]) {
                        const int vi = (j * nv + k) * 8 + el, p = j * seg_len + el * nv + k;
                        int prof;
                        if (p >= plen) prof = -32768;
                        else { const int pb = (int)base_value(pat[p]); prof = (tb > 3 || pb > 3) ? -1 : (tb == pb ? a.prm.match : a.prm.sub); }
                        const int m = agc_sat(h + prof);
                        int e = (int)E[vi];
                        int bt = e > m ? 1 : 0;
                        int hh = m > e ? m : e;
                        { const int t = f > hh ? 2 : 0; bt = t | (bt & ~t); }
                        hh = hh > f ? hh : f;
                        const int hnext = (int)Hp[vi];
                        e = agc_sat(e - ext);
                        const int temp = agc_sat(m - open);
                        if (e > temp) bt |= 4;
                        e = e > temp ? e : temp;
                        E[vi] = (int16_t)e;
                        f = agc_sat(f - ext);
                        if (f > temp) bt |= 32;
                        f = f > temp ? f : temp;
                        hv[k] = hh; btk[k] = bt; h = hnext;
                    }
                }
                // lazy F (:737-781): up to seven rounds; a round stops after the first vector in which no element's F goes on
                bool lz = act;
                for (int kk = 0; kk < 7; kk++) {
                    if (!BALLOT(lz)) break;
                    const int f7 = __shfl(f, (lane & ~7) + 7);
                    int fin = __shfl_up(f, 1);
                    if (lz) {
                        X = X > f7 ? X : f7;
                        f = el == 0 ? 0 : fin;
                    }
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        const bool kv = lz && ka[v];
                        if (kv) {
                            int hh = hv[v], bt = btk[v];
                            { const int t = f > hh ? 2 : 0; bt = t | (bt & ~t); }
                            hh = hh > f ? hh : f;
                            const int temp = agc_sat(hh - open);
                            f = agc_sat(f - ext);
                            if (f > temp) bt |= 32;
                            hv[v] = hh; btk[v] = bt;
                        }
                        const bool cont = kv && f > agc_sat(hv[v] - open);
                        const unsigned long long cm = BALLOT(cont);
                        if (kv && (cm & gmask) == 0ull) lz = false;         // (the vectors after it are not touched in this round, nor in any round after)
                    }
                }
                if (act) {
#pragma unroll
                    for (int k = 0; k < 2; k++)
                        if (ka[k]) { Hm[(j * nv + k) * 8 + el] = (int16_t)hv[k]; btout[(size_t)i * SAMF_PRE_ROW + (size_t)(jj * seg_len + k * 8 + el)] = (uint8_t)btk[k]; }
                    f = el == 0 ? X : 0;                                    // :783
                }
            }
            WAVE_SYNC();
            if (going && band_end == plen - 1) {                            // :803-815
                const int vec = (band_end / seg_len) * nv + (band_end % seg_len) % nv, e_i = (band_end % seg_len) / nv;
                const int gsc = (int)Hm[vec * 8 + e_i];
                if (gsc > score) { score = gsc; text_used = i; }
            }
            cur ^= 1;
            if (going && seg_beg > seg_end && ++idle_rows == 2) going = false;       // (see agc_banded)
            WAVE_SYNC();
        }
        if (elig && el == 0) {
            pre->plen = plen; pre->w = w; pre->bcb = bcb; pre->loc = loc; pre->score = score; pre->text_used = text_used; pre->dir = dir; pre->rows = rows_cap;
            pre->valid = 1;
        }
        WAVE_SYNC();
    }
}

extern "C" void snapgpu_launch_samf_dp8(const SamFieldsArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL(k_samf_dp8, dim3(blocks), dim3(256), lds_bytes, s, *a);
}

// paired-end writer: both reads of a pair by one wavefront, then SAMFormat::fillMateInfo for each (sam_fields.h)
__global__ __launch_bounds__(256, SAMF_WAVES) void k_sam_fields_paired(SamFieldsPairedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t wave_slot = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave_in_block;
    uint8_t *my = lds + (size_t)wave_in_block * agc_lds_bytes(a.RL);
    uint8_t *scratch = a.scratch + (size_t)wave_slot * a.scratch_stride;
    uint8_t *oriented = scratch;
    uint32_t *lv_cells = (uint32_t *)(scratch + ((2 * a.RL + 255) & ~255u));
    uint8_t *ag_scratch = (uint8_t *)lv_cells + ((lvc_scratch_bytes() + 255) & ~255u);
    AGCParams prm; prm.match = a.prm.match; prm.sub = a.prm.sub; prm.gap_open = a.prm.gap_open; prm.gap_ext = a.prm.gap_ext;
    while (true) {
        uint32_t i = 0;
        if (lane == 0) i = atomicAdd(a.work_counter, 1u);
        i = first_u32(i);
        if (i >= a.n_pairs) break;
        const snapgpu_paired_result *pr = &a.results[i];
        const bool aligned_as_pair = first_u32((uint32_t)pr->aligned_as_pair) != 0;
        SamFieldsOut o[2];
        for (int w = 0; w < 2; w++) {
            const uint32_t ri = 2 * i + (uint32_t)w;
            const uint64_t b = first_u64(a.offsets[ri]), e = first_u64(a.offsets[ri + 1]);
            snapgpu_single_result r;                                              // this mate's part of the PairedAlignmentResult
            r.status = (int32_t)first_u32((uint32_t)pr->status[w]); r.direction = (int32_t)first_u32((uint32_t)pr->direction[w]);
            r.location = (int64_t)first_u64((uint64_t)pr->location[w]); r.orig_location = 0;
            r.score = (int32_t)first_u32((uint32_t)pr->score[w]); r.score_prior_to_clipping = 0;
            r.mapq = (int32_t)first_u32((uint32_t)pr->mapq[w]);
            r.clipping_for_read_adjustment = (int32_t)first_u32((uint32_t)pr->clipping_for_read_adjustment[w]);
            r.used_affine_gap_scoring = (int32_t)first_u32((uint32_t)pr->used_affine_gap_scoring[w]);
            r.bases_clipped_before = (int32_t)first_u32((uint32_t)pr->bases_clipped_before[w]);
            r.bases_clipped_after = (int32_t)first_u32((uint32_t)pr->bases_clipped_after[w]);
            r.ag_score = 0; r.supplementary = (int32_t)first_u32((uint32_t)pr->supplementary[w]); r.seed_offset = 0; r.match_probability = 0.0;
            r.probability_all_candidates = 0.0; r.popular_seeds_skipped = 0; r.reserved = 0;
            const int F0 = (int)first_u32((uint32_t)a.front_clip[ri]), D0 = (int)first_u32((uint32_t)a.data_len[ri]);
            o[w] = sam_fields_single_item(a.ix, prm, a.use_affine_gap != 0, a.use_m != 0, a.bases + b, a.quals + b, (int)(e - b), F0, D0, r,
                                          my, a.RL, oriented, lv_cells, ag_scratch, a.ops + (size_t)ri * a.ops_stride, (int)a.ops_stride, true);
            WAVE_SYNC();
        }
        for (int w = 0; w < 2; w++) {
            const SamMateOut m = sam_fill_mate_info(a.ix, o[w], o[1 - w], w == 0, aligned_as_pair);
            const uint32_t ri = 2 * i + (uint32_t)w;
            if (lane == 0) {
                a.flag[ri] = m.flag; a.contig[ri] = m.contig; a.pos[ri] = m.pos; a.mapq[ri] = o[w].mapq; a.n_ops[ri] = o[w].n_ops; a.nm[ri] = o[w].nm;
                a.rnext[ri] = m.rnext; a.pnext[ri] = m.pnext; a.tlen[ri] = m.tlen; a.stale[ri] = o[w].stale;
            }
        }
        if (lane == 0) {                                                          // ReadWriter.cpp:481-488: numerical order of the final locations
            const unsigned long long l0 = o[0].final_loc < 0 ? ~0ull : (unsigned long long)o[0].final_loc, l1 = o[1].final_loc < 0 ? ~0ull : (unsigned long long)o[1].final_loc;
            a.first_written[i] = l0 <= l1 ? 0 : 1;
        }
        WAVE_SYNC();
    }
}

extern "C" void snapgpu_launch_sam_fields_paired(const SamFieldsPairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL(k_sam_fields_paired, dim3(blocks), dim3(256), lds_bytes, s, *a);
}

// AlignmentAdjuster::AdjustAlignment (adjust.h) for a batch: one wavefront per result, persistent grid
__global__ __launch_bounds__(256) void k_adjust_alignments(AdjustArgs a)
{
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t wave_slot = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave_in_block;
    uint8_t *mine = a.scratch + (size_t)wave_slot * a.scratch_stride;
    uint8_t *fwd = mine, *rc = mine + ((a.RL + 255) & ~255u);
    const AdjustScratch sc = adjust_scratch_at(mine + 2 * (size_t)((a.RL + 255) & ~255u), a.RL);
    const AdjustIx aix = adjust_ix(a.ix);
    while (true) {
        uint32_t i = 0;
        if (lane == 0) i = atomicAdd(a.work_counter, 1u);
        i = first_u32(i);
        if (i >= a.n) break;
        const uint64_t off = first_u64(a.off[i]);
        const int len = (int)first_u32((uint32_t)a.len[i]);
        for (int j = lane; j < len; j += WAVE) { const uint8_t b = a.data[off + j]; fwd[j] = b; rc[len - 1 - j] = rc_base(b); }
        WAVE_SYNC(); __threadfence_block();
        snapgpu_single_result *r = a.results + i;
        const AdjustOut o = adjust_alignment(aix, fwd, rc, len, (int)first_u32((uint32_t)r->status), (int)first_u32((uint32_t)r->direction),
                                             (long long)first_u64((uint64_t)r->location), (int)first_u32((uint32_t)r->score), SNAPGPU_InvalidGenomeLocation32, sc);
        WAVE_SYNC();
        if (lane == 0) { r->status = o.status; r->location = o.location; r->score = o.score; r->clipping_for_read_adjustment = o.clipping; }
        WAVE_SYNC();
    }
}

extern "C" void snapgpu_launch_adjust_alignments(const AdjustArgs *a, uint32_t blocks, hipStream_t s)
{
    hipLaunchKernelGGL(k_adjust_alignments, dim3(blocks), dim3(256), 0, s, *a);
}
