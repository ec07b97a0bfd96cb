// cigar_args.h -- kernel arguments of k_cigar_lv (cigar_k.hip), shared with the host side (snapgpu.hip).
#pragma once
#include "dev_common.h"
#include "../../include/snapgpu.h"

struct AGCParamsPOD { int match, sub, gap_open, gap_ext; };

struct CigarArgs {
    DevIndex ix;
    uint32_t n, RL, ops_stride, use_m;
    const uint8_t *data; const uint64_t *off; const int32_t *len; const int64_t *loc; const int32_t *extra_before;
    uint8_t *scratch;                 // waves * lvc_scratch_bytes()
    uint32_t *work_counter;
    uint32_t *ops; int32_t *n_ops; int32_t *edit_distance; int32_t *add_front_clipping; int64_t *extra_after;
};

struct CigarAGArgs {
    DevIndex ix;
    AGCParamsPOD prm;
    uint32_t n, RL, ops_stride, use_m;
    const uint8_t *data; const uint8_t *quals; const uint64_t *off; const int32_t *len; const int64_t *loc; const int32_t *extra_before;
    const int32_t *score;             // the alignment's edit distance: k of computeGlobalScoreNormalized
    uint8_t *scratch; uint64_t scratch_stride;
    uint32_t *work_counter;
    uint32_t *ops; int32_t *n_ops; int32_t *edit_distance; int32_t *add_front_clipping; int64_t *extra_after; int32_t *tail_ins; int32_t *stale;
};

struct SamFieldsArgs {
    DevIndex ix;
    AGCParamsPOD prm;
    uint32_t n, RL, ops_stride, use_m, use_affine_gap;
    const uint8_t *bases; const uint8_t *quals; const uint64_t *offsets;       // the reads as they came from the file (unclipped)
    const int32_t *front_clip; const int32_t *data_len;                          // Read::clip's result: bases clipped in front, bases kept
    const snapgpu_single_result *results;
    uint8_t *scratch; uint64_t scratch_stride;
    uint32_t *work_counter;
    int32_t *flag; int32_t *contig; int64_t *pos; int32_t *mapq; uint32_t *ops; int32_t *n_ops; int32_t *nm; int32_t *stale;
    // the banded row loops run ahead of the records, eight reads to a wavefront (cigar_ag.h: SamfPre; k_samf_dp8): n * pre_stride bytes, or NULL
    uint8_t *pre; uint64_t pre_stride; uint32_t *pre_counter;
};

struct SamFieldsPairedArgs {
    DevIndex ix;
    AGCParamsPOD prm;
    uint32_t n_pairs, RL, ops_stride, use_m, use_affine_gap;
    const uint8_t *bases; const uint8_t *quals; const uint64_t *offsets;       // 2 * n_pairs + 1: read 0 and read 1 of each pair, unclipped
    const int32_t *front_clip; const int32_t *data_len;                          // [2 * n_pairs]
    const snapgpu_paired_result *results;                                        // [n_pairs]
    uint8_t *scratch; uint64_t scratch_stride;
    uint32_t *work_counter;
    // per read [2 * n_pairs]
    int32_t *flag; int32_t *contig; int64_t *pos; int32_t *mapq; uint32_t *ops; int32_t *n_ops; int32_t *nm; int32_t *rnext; int64_t *pnext; int64_t *tlen; int32_t *stale;
    int32_t *first_written;                                                      // [n_pairs]: which read's record comes first in the file
};

extern "C" void snapgpu_launch_sam_fields_paired(const SamFieldsPairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
extern "C" void snapgpu_launch_sam_fields(const SamFieldsArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
extern "C" void snapgpu_launch_samf_dp8(const SamFieldsArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
extern "C" size_t snapgpu_samf_dp8_lds_per_wave(uint32_t RL);
extern "C" void snapgpu_launch_cigar_ag(const CigarAGArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
// snapgpu_adjust_alignments: AlignmentAdjuster::AdjustAlignment for a batch of results (adjust.h), one wavefront per result
struct AdjustArgs {
    DevIndex ix;
    uint32_t n, RL;
    const uint8_t *data; const uint64_t *off; const int32_t *len;
    snapgpu_single_result *results;   // in / out: status, direction, location, score -> status, location, score, clipping_for_read_adjustment
    uint8_t *scratch; uint64_t scratch_stride;        // per wave: the read, its reverse complement (RL bytes each), then adjust_scratch_bytes(RL)
    uint32_t *work_counter;
};
extern "C" void snapgpu_launch_adjust_alignments(const AdjustArgs *a, uint32_t blocks, hipStream_t s);
extern "C" void snapgpu_launch_cigar_lv(const CigarArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
