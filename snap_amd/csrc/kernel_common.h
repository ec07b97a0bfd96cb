// kernel_common.h -- kernel argument structs and the per-wave LDS carve-out shared by the kernel
// translation units (snapgpu.hip: single-end + batch primitives; paired_k.hip: paired end) and the host side.
#pragma once
#include "align_single.h"

struct AlignArgs {
    DevIndex ix;
    AlignCfg cfg;
    const DevTables *tab;
    uint8_t *scratch;                 // n_wave_slots * cfg.scratch_stride
    const uint8_t *bases, *quals;
    const uint64_t *offsets;
    uint32_t n_reads;
    snapgpu_single_result *primary, *first_alt;
    uint32_t *work_counter;
    unsigned long long *counters;     // snapgpu_counters layout
    // secondary results (k_align_single<.., true> only)
    SecCfg sec_cfg;
    uint8_t *sec_scratch;             // n_wave_slots * sec_stride_bytes
    uint64_t sec_stride_bytes;
    snapgpu_single_result *secondary; // [n_reads * sec_out_stride]
    uint32_t sec_out_stride;
    uint32_t *n_secondary;            // [n_reads]: how many the read has (may exceed sec_out_stride: only that many are stored)
    // exact replay of flagged reads (DESIGN.md "Reference nondeterminism"): the fast pass appends every read whose banded affine-gap
    // traceback left the band to flag_list; k_align_single<0, SEC, true> then redoes exactly those (work item i = read remap[i]) with
    // the reference's traceback arrays kept per wave in `persist` (2 x ag_scratch_bytes(RL) per wave slot)
    uint32_t *flag_list, *flag_count;
    const uint32_t *remap, *n_remap;
    uint32_t is_replay;               // this launch redoes reads an earlier launch of the same call already counted (the exact replay)
    uint8_t *persist; uint64_t persist_stride;
    // heavy-first dequeue (order.h): work item i of the MAIN pass is read order[i] (a permutation of 0 .. n_reads); NULL = batch order
    const uint32_t *order;
    // TIMED instantiation only (SNAPGPU_PHASE_TIMERS=1; NULL otherwise): launch diagnostics, snapgpu_debug_launch_profile
    //   [0, 64)                       reads by floor(log2(wave cycles spent on the read))
    //   [64, 64 + S)                  clock when wave slot s took its first read      (S = wave slots of the launch)
    //   [64 + S, 64 + 2S)             clock when wave slot s ran out of reads
    //   [64 + 2S, 64 + 3S)            cycles of the slot's most expensive read << 24 | its affine-gap calls (capped)
    unsigned long long *dbg; uint32_t dbg_slots;
    // help for heavy reads (se_help.h): slots, one record array of se_spec_cap entries per slot, [done reads | idle waves]; NULL: none
    SEHelpSlot *se_slots; uint32_t se_n_slots; SESpec *se_spec; uint32_t se_spec_cap; uint32_t *se_ctl; uint32_t se_eager;
    uint32_t se_keep;                 // every se_keep-th wave stays on as a helper when it runs out of reads (1: all)
    // Read::clip's outcome applied on the device (snapgpu_align_sam_single: one upload serves the aligner and the SAM-field kernel, which
    // wants the unclipped read): read i is bases[offsets[i] + front_clip[i] .. + data_len[i]); skip[i] != 0: the read is not given to the
    // aligner at all (SingleAligner.cpp:211-232) and gets the result the reference writes for it.  NULL: offsets say it all.
    const int32_t *front_clip, *data_len; const uint8_t *skip;
};

extern "C" {
void snapgpu_launch_single_sec_3(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_single_sec_4(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_single_sec_6(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_single_sec_0(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_single_exact_3(const AlignArgs *a, int sec, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_single_exact_0(const AlignArgs *a, int sec, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_single_exact_4(const AlignArgs *a, int sec, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_single_exact_6(const AlignArgs *a, int sec, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_single_exact_3_timed(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_single_3_timed(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
// single_planes_k.hip: the instantiations that carry the plane Landau-Vishkin (SNAPGPU_LV_PLANES=1)
void snapgpu_launch_single_planes_3(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_single_planes_4(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_single_planes_6(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_single_planes_0(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_single_exact_planes_3(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
void snapgpu_launch_single_exact_planes_0(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s);
}

static __device__ __forceinline__ uint32_t align_up(uint32_t v, uint32_t a) { return (v + a - 1) & ~(a - 1); }

// LDS carve-out per wave; must match lds_bytes_per_wave() on the host.
struct LdsLayout {
    uint32_t rd0, rd1, ql0, ql1, gw, seed_used, wl_next, wl_prev, lv, ag, shared, rp, tp, lvp, total;
};

// ag_lds: bytes of LDS the affine-gap code of the kernel variant needs (AlignCfg::ag_lds; 0 = no affine-gap buffers)
static __host__ __device__ __forceinline__ LdsLayout lds_layout(uint32_t RL, uint32_t num_weight_lists, uint32_t kmax, uint32_t ag_lds) {
    LdsLayout L; uint32_t o = 0;
    L.rd0 = o; o += RL; L.rd1 = o; o += RL; L.ql0 = o; o += RL; L.ql1 = o; o += RL;
    L.gw = o; o += (RL + 2 * WIN_PAD + 15) & ~15u;
    L.seed_used = o; o += (((RL + 31) / 32) * 4 + 15) & ~15u;
    L.wl_next = o; o += (num_weight_lists * 2 + 15) & ~15u;
    L.wl_prev = o; o += (num_weight_lists * 2 + 15) & ~15u;
    L.lv = o; o += (lv_lds_bytes(kmax, RL) + 15) & ~15u;
    L.ag = o; o += (ag_lds + 15) & ~15u;
    L.shared = o; o += ((uint32_t)sizeof(WaveShared) + 15) & ~15u;
    L.rp = o; o += (2 * 4 * read_plane_words(RL) * 8 + 15) & ~15u;          // read planes: [direction][code bit 0, code bit 1, N, other][word]
    L.tp = o; o += (3 * text_plane_blocks(RL, WIN_PAD) * 8 + 15) & ~15u;    // text planes of the candidate window: [plane][block]
    L.lvp = o; o += (lv_plane_work_words(RL) * 8 + 15) & ~15u;              // Landau-Vishkin's prepared plane words (planes.h)
    L.total = o;
    return L;
}

