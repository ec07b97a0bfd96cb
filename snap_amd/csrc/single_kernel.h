// single_kernel.h -- the single-end kernel: one wavefront per read, persistent grid, atomic work counter.
// SEC = with secondary results (-om); a separate instantiation so that the default kernel carries none of it.
#pragma once
#include "kernel_common.h"

// Latency-bound kernel.  The 192-position affine-gap variant (reads up to ~170 bp) asks for 6 waves per SIMD (80 VGPRs, ~75 dwords
// of cold state spilled): measured 3.96 M reads/s against 3.74-3.80 M at 4 and 3.79 M at 5 waves (profiles/r01g).  The variants
// with more affine-gap state in registers stay at 4 (128 VGPRs); their LDS footprint caps occupancy first anyway.
#ifndef SNAPGPU_WAVES_PER_SIMD
#define SNAPGPU_WAVES_PER_SIMD(AGC) ((AGC) == 3 ? 6 : 4)            // (AGC 4 / 6 -- which since round 6 keep three chunks in registers like AGC 3 -- at 6: 250 bp / -d 20 reads 5.10 against 5.29 M reads/s at 4, profiles/r06g)
#endif
template <int AGC, bool SEC, bool EXACT = false, bool TIMED = false, bool PLANES = false>
__global__ __launch_bounds__(256, SNAPGPU_WAVES_PER_SIMD(AGC)) void k_align_single(AlignArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id();
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // uniform: keeps the LDS/scratch pointers in SGPRs
    const uint32_t wave_slot = blockIdx.x * (blockDim.x >> 6) + (uint32_t)wave_in_block;
    const LdsLayout L = lds_layout(a.cfg.RL, a.cfg.num_weight_lists, a.cfg.kmax, a.cfg.ag_lds);
    uint8_t *my = lds + (size_t)wave_in_block * L.total;

    WaveShared *ws = (WaveShared *)(my + L.shared);
#if !defined(SNAPGPU_REGDIR)
    Aligner<AGC, SEC, EXACT, TIMED, PLANES> al(a.ix, a.tab, a.cfg, ws);
#else
    // (measurement builds: the candidate table's directory in a vector register -- align_single.h: REGDIR.  It takes findElement's two dependent HBM loads
    //  away and was 1.5 % SLOWER on the bench batch, 15.0 against 15.3 M reads/s, profiles/r06f: five more spilled VGPRs, and the loads it removes are not what
    //  the hit phase waits for)
    Aligner<AGC, SEC, EXACT, TIMED, PLANES, true> al(a.ix, a.tab, a.cfg, ws);
#endif
    al.rd[0] = my + L.rd0; al.rd[1] = my + L.rd1;
    al.ql[0] = my + L.ql0; al.ql[1] = my + L.ql1;
    al.gw = my + L.gw;
    al.seed_used = (uint32_t *)(my + L.seed_used);
    al.wl_next = (uint16_t *)(my + L.wl_next);
    al.wl_prev = (uint16_t *)(my + L.wl_prev);
    al.lv_tri = (uint16_t *)(my + L.lv);
    al.ag_rows = (int16_t *)(my + L.ag);
    al.rp = (unsigned long long *)(my + L.rp); al.tp = (unsigned long long *)(my + L.tp); al.lvp = (unsigned long long *)(my + L.lvp);
    uint8_t *sc = a.scratch + (size_t)wave_slot * a.cfg.scratch_stride;
    al.heads = (uint16_t *)sc;
    al.pool = (Elem *)(sc + (size_t)a.cfg.ht_size * 2);
    al.ag_scratch = sc + (size_t)a.cfg.ht_size * 2 + (size_t)a.cfg.pool_size * sizeof(Elem);
    al.ag_persist0 = al.ag_persist1 = nullptr; al.ag_hw0 = al.ag_hw1 = 0;
    if constexpr (EXACT) {              // (the host zeroes the images when it allocates them; from then on each unit clears what the last one wrote)
        al.ag_persist0 = a.persist + (size_t)wave_slot * a.persist_stride;
        al.ag_persist1 = al.ag_persist0 + a.persist_stride / 2;
    }
    if constexpr (SEC) {            // secondary-result scratch of this wave (snapgpu_enable_secondary)
        uint8_t *ss = a.sec_scratch + (size_t)wave_slot * a.sec_stride_bytes;
        al.sec_cfg = a.sec_cfg;
        al.sec = (snapgpu_single_result *)ss;
        al.sec_key = (uint32_t *)(ss + (size_t)a.sec_cfg.cap * sizeof(snapgpu_single_result));
        al.sec_ord = al.sec_key + 2 * (size_t)a.sec_cfg.cap;
        al.n_sec = 0;
        al.adj_scratch = a.sec_cfg.adjust ? ss + a.sec_cfg.adj_off : nullptr;
    }
    al.cnt() = WaveCounters{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t n_done = 0;
    // help for heavy reads (se_help.h): not in the exact replay, not without the context's arrays
    const bool se_on = !EXACT && a.se_slots != nullptr && a.cfg.se_items_cap != 0;
    al.se_slots = se_on ? a.se_slots : nullptr; al.se_n_slots = a.se_n_slots; al.se_spec = a.se_spec; al.se_spec_cap = a.se_spec_cap;
    al.se_ctl = a.se_ctl; al.se_eager = a.se_eager; al.se_diag = a.counters + 14;
    al.se_items = (uint32_t *)(sc + a.cfg.se_off); al.se_first = al.se_items + a.cfg.se_items_cap;
    al.se_slot = -1; al.se_n = 0; al.se_tried = 0; al.cur_read = 0; al.se_mine = nullptr;

    // EXACT kernels run either over a list of flagged reads (remap: the replay pass behind the register variants for long reads) or, as
    // the main pass of the 192-position variant, over the whole batch
    const uint32_t n_total = a.remap ? first_u32(*a.n_remap) : a.n_reads;
    unsigned long long dbg_worst = 0;
    if constexpr (TIMED) { if (a.dbg && lane == 0 && wave_slot < a.dbg_slots) a.dbg[64 + wave_slot] = wave_clock(); }
    while (true) {
        uint32_t i = 0;
        if (lane == 0) i = atomicAdd(a.work_counter, 1u);
        i = first_u32(i);
        if (i >= n_total) break;
        if (a.remap) i = first_u32(a.remap[i]);
        else if (a.order) i = first_u32(a.order[i]);
        al.new_read_images();           // EXACT: a newly constructed reference aligner: both traceback arrays read as zero
        al.cur_read = i;
        uint64_t b = first_u64(a.offsets[i]), e = first_u64(a.offsets[i + 1]);
        if (a.front_clip) {
            b += (uint64_t)first_u32((uint32_t)a.front_clip[i]); e = b + (uint64_t)first_u32((uint32_t)a.data_len[i]);
            if (a.skip && first_u32((uint32_t)a.skip[i]) != 0u) {         // not given to the aligner (SingleAligner.cpp:215-225): NotFound, no location, score -1
                const int nd = (int)(sizeof(snapgpu_single_result) / 4);
                if (lane < nd) { ((uint32_t *)&a.primary[i])[lane] = 0u; if (a.first_alt) ((uint32_t *)&a.first_alt[i])[lane] = 0u; }
                WAVE_SYNC();                                              // (one wave's stores reach memory in issue order: the fields below land on the zeros)
                if (lane == 0) { a.primary[i].status = SNAPGPU_NotFound; a.primary[i].location = SNAPGPU_InvalidGenomeLocation32; a.primary[i].score = -1; }
                if (se_on && lane == 0) atomicAdd(&a.se_ctl[0], 1u);
                continue;
            }
        }
        const uint64_t dbg_t0 = TIMED ? wave_clock() : 0; const uint64_t dbg_ag0 = al.cnt().ag;
        al.align_read(a.bases + b, a.quals + b, (int)(e - b));
        WAVE_SYNC();
#if defined(SNAPGPU_TEST_BREAK_PARITY)
        // TEST BUILDS ONLY (snap_amd/ab/libsnapgpu_broken.so, tests/test_zzzz_gpu_bench.py): a kernel that answers differently from the
        // reference, so that the bench line's "fails loudly" path can be exercised on the hardware
        if (i % 997u == 7u && lane == 0) ws->primary.mapq ^= 1;
        WAVE_SYNC();
#endif
        if constexpr (TIMED) {
            if (a.dbg) {
                const unsigned long long cyc = wave_clock() - dbg_t0, nag = al.cnt().ag - dbg_ag0;
                if (lane == 0) atomicAdd(&a.dbg[63 - __clzll((long long)(cyc | 1ull))], 1ull);
                const unsigned long long packed = (cyc << 24) | (nag > 0xffffffull ? 0xffffffull : nag);
                if (packed > dbg_worst) dbg_worst = packed;
            }
        }
        if constexpr (!EXACT) {         // traceback left the band somewhere: the exact pass redoes this read
            if (a.flag_list && (ws->primary.reserved & 0x40000000u) && lane == 0) a.flag_list[atomicAdd(a.flag_count, 1u)] = i;
        }
        if (EXACT && lane == 0) ws->primary.reserved |= 0x80000000u;       // this record is the exact pass's answer
        WAVE_SYNC();
        {   // results: LDS -> global, one dword per lane
            const uint32_t *src = (const uint32_t *)&ws->primary;
            uint32_t *dst = (uint32_t *)&a.primary[i];
            const int nd = (int)(sizeof(snapgpu_single_result) / 4);
            if (lane < nd) dst[lane] = src[lane];
            if (a.first_alt) {
                const uint32_t *src2 = (const uint32_t *)&ws->first_alt;
                uint32_t *dst2 = (uint32_t *)&a.first_alt[i];
                if (lane < nd) dst2[lane] = src2[lane];
            }
        }
        WAVE_SYNC();
        if constexpr (SEC) {        // secondary results: sec[sec_ord[k]] -> secondary[i * stride + k], 22 dwords per record, two records per pass
            const uint32_t n_sec = al.n_sec;
            if (lane == 0) a.n_secondary[i] = al.sec_overflow ? 0xFFFFFFFFu : n_sec;       // (the host turns the marker into an error)
            const uint32_t n_out = n_sec < a.sec_out_stride ? n_sec : a.sec_out_stride;
            const int nd = (int)(sizeof(snapgpu_single_result) / 4);
            for (uint32_t k0 = 0; k0 < n_out; k0 += 2) {
                const uint32_t k = k0 + (uint32_t)(lane >> 5);
                const int w = lane & 31;
                if (k < n_out && w < nd) {
                    const uint32_t *src = (const uint32_t *)&al.sec[al.sec_ord[k]];
                    uint32_t *dst = (uint32_t *)&a.secondary[(size_t)i * a.sec_out_stride + k];
                    dst[w] = src[w];
                }
            }
            WAVE_SYNC();
        }
        n_done++;
        if (se_on && lane == 0) atomicAdd(&a.se_ctl[0], 1u);
    }
    if constexpr (!EXACT) {
        // Out of reads: until every read of the launch is done, evaluate candidates of the reads that have published their lists.
        // Only every a.se_keep-th wave stays on as a helper; the others leave, so that their slots go to whatever launch is queued behind
        // this one (another feeder's batch: measured with three feeders and every wave staying, 130 -> 179 ms per batch, profiles/r03g).
        if (se_on && (a.se_keep <= 1u || wave_slot % a.se_keep == 0u)) {
            if (lane == 0) atomicAdd(&a.se_ctl[1], 1u);                   // one more idle wave: forced walks start publishing
            const uint64_t t_idle0 = wave_clock();
            // (Polling is done with plain device-scope LOADS of words that only atomics write -- a stale value merely postpones a decision, and
            //  the attach below re-reads the state through the atomic unit -- and an idle wave sleeps ~30 us between looks while nothing is
            //  published: thousands of idle waves scanning the slots with read-modify-write atomics were measured to slow the whole launch
            //  down by half, profiles/r03f.)
            for (uint32_t round = 0;; round++) {
                if (XW::ld(a.se_ctl[0]) >= n_total) break;
                if ((round & 63u) == 0u && wave_clock() - t_idle0 > 48000000000ull) {      // ~20 s: stop waiting, leave a trace
                    if (lane == 0) atomicAdd(&a.counters[14], 1ull << 32);
                    break;
                }
                bool any = false;
                if (XW::ld(a.se_ctl[2]) != 0u) {                              // lists open right now
                    const uint32_t s_first = (wave_slot * 13u + round * 7u) % a.se_n_slots;      // (waves start their scan in different places)
                    for (uint32_t k = 0; k < a.se_n_slots && !any; k++) {
                        const uint32_t s = s_first + k < a.se_n_slots ? s_first + k : s_first + k - a.se_n_slots;
                        SEHelpSlot *slot = &a.se_slots[s];
                        if (XW::ld(slot->state) != 1u) continue;
                        if (XW::ld(slot->next) >= XW::ld(slot->n)) continue;
                        if (XW::ld(slot->helpers) >= SE_HELP_MAX_HELPERS) continue;
                        // attach, THEN look at the state again (the owner does the mirror image: close, then look at the attach count)
                        uint32_t before = 0;
                        if (lane == 0) before = atomicAdd(&slot->helpers, 1u);
                        before = first_u32(before);
                        if (before < SE_HELP_MAX_HELPERS && XW::aload(&slot->state) == 1u) {
                            XW::fence_acquire();
                            const uint32_t r = XW::ld(slot->read);
                            if (r < a.n_reads) {
                                uint64_t rb = first_u64(a.offsets[r]), re = first_u64(a.offsets[r + 1]);
                                if (a.front_clip) { rb += (uint64_t)first_u32((uint32_t)a.front_clip[r]); re = rb + (uint64_t)first_u32((uint32_t)a.data_len[r]); }
                                const int len = (int)(re - rb);
                                if (len >= (int)a.ix.seed_len && len <= (int)a.cfg.RL) {
                                    al.read_len = len;
                                    (void)al.load_read(a.bases + rb, a.quals + rb, len);
                                    WAVE_SYNC();
                                    if (PLANES && a.ix.planes != nullptr) al.build_read_planes(len);
                                    al.se_help_slot(slot);
                                    any = true;
                                }
                            }
                        }
                        if (lane == 0) atomicSub(&slot->helpers, 1u);
                    }
                }
#ifdef SNAPGPU_WAVE_EMU
                // (emulator: see paired_dev.h -- a wave that waits can keep the reads it waits for from ever starting)
                if (!getenv("SNAPGPU_EMU_HELP_SPIN")) break;
#endif
                if (!any) { for (int z = 0; z < 8; z++) XW::nap(); }
            }
        }
    }
    if constexpr (TIMED) {
        if (a.dbg && lane == 0 && wave_slot < a.dbg_slots) { a.dbg[64 + a.dbg_slots + wave_slot] = wave_clock(); a.dbg[64 + 2 * a.dbg_slots + wave_slot] = dbg_worst; }
    }
    if constexpr (EXACT) {              // leave the images zeroed for the next launch (the high-water marks live in registers)
        if (al.ag_hw0) wave_zero16(al.ag_persist0, ((size_t)al.ag_hw0 + 15) & ~(size_t)15);
        if (al.ag_hw1) wave_zero16(al.ag_persist1, ((size_t)al.ag_hw1 + 15) & ~(size_t)15);
    }
    if (lane == 0 && !a.is_replay) {          // (a replayed read was already counted by the fast pass)
        atomicAdd(&a.counters[0], (unsigned long long)n_done);
        atomicAdd(&a.counters[1], (unsigned long long)al.cnt().lookups);
        atomicAdd(&a.counters[2], (unsigned long long)al.cnt().slots);
        atomicAdd(&a.counters[3], (unsigned long long)al.cnt().hits);
        atomicAdd(&a.counters[4], (unsigned long long)al.cnt().overflow_lists);
        atomicAdd(&a.counters[5], (unsigned long long)al.cnt().lv);
        atomicAdd(&a.counters[6], (unsigned long long)al.cnt().ag);
        atomicAdd(&a.counters[7], (unsigned long long)al.cnt().lv_ref_bytes);
        atomicAdd(&a.counters[8], (unsigned long long)al.cnt().cyc_lookup);
        atomicAdd(&a.counters[9], (unsigned long long)al.cnt().cyc_hits);
        atomicAdd(&a.counters[10], (unsigned long long)al.cnt().cyc_lv);
        atomicAdd(&a.counters[11], (unsigned long long)al.cnt().cyc_ag);
        atomicAdd(&a.counters[12], (unsigned long long)al.cnt().cyc_total);
    }
}

