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
        const SamFieldsOut o = sam_fields_single_item(a.ix, prm, a.use_affine_gap != 0, a.use_m != 0, a.bases + b, a.quals + b, (int)(e - b), F0, D0, r,
                                                      my, a.RL, oriented, lv_cells, ag_scratch, ops, (int)a.ops_stride);
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
