// snapgpu.hip -- gfx950 kernels and the C-ABI host side of libsnapgpu.so (include/snapgpu.h).
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
// There is deliberately no CPU fallback anywhere in this file: without a HIP device
// snapgpu_create fails with SNAPGPU_E_NODEVICE.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <deque>
#include <mutex>
#include <chrono>
#include <atomic>
#include <memory>
#include <thread>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>

#include "../../include/snapgpu.h"
#include "dev_common.h"
#include "probe.h"
#include "lv.h"
#include "ag_win.h"
#include "align_single.h"
#include "kernel_common.h"
#include "single_kernel.h"
#include "order.h"
#include "lookup16.h"
#include "planes.h"
#include "paired_args.h"
#include "cigar_lv.h"
#include "cigar_ag.h"
#include "cigar_args.h"

// =====================================================================================
// kernels
// =====================================================================================

// k_align_single: single_kernel.h

// One wave per seed: GenomeIndex::lookupSeed32 for a batch of seeds.
// hits == NULL: hit counts only, and the hit lists are still READ (max_hits_out of them at most, as BaseAligner consumes them) so that the
// launch moves the bytes a lookup is entitled to -- the probe-only roofline measurement of bench.py.  counters (snapgpu_counters layout,
// may be NULL): lookups, slots examined, hits read, overflow lists dereferenced.
__global__ __launch_bounds__(256) void k_lookup_seeds(DevIndex ix, uint32_t n, const uint8_t *seeds,
                                                      long long *n_hits, uint32_t *hits, uint32_t max_hits_out, unsigned long long *counters)
{
    const int lane = lane_id();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    unsigned long long c_lookups = 0, c_slots = 0, c_hits = 0, c_lists = 0;
    uint32_t sink = 0;
    if (ix.bucket_blob != nullptr) {
        // Device-native layout (bucket.h): FOUR seeds per wavefront pass -- lanes 16 s + 8 d + e read dword pair e of the bucket of seed s,
        // strand d, so the eight probes of a pass are eight independent 64-byte lines in flight; the eight overflow headers follow as
        // one load, then each list is read 64 hits per load.
        const int g = lane >> 3, e = lane & 7, sidx = g >> 1, dir = g & 1;
        const uint32_t key_bits = ix.key_bytes * 8;
        const uint32_t n_bases32 = (uint32_t)ix.n_bases;
        for (uint32_t base = wave * 4; base < n; base += n_waves * 4) {
            uint64_t my_bases = 0; bool my_valid = false;
            for (int s = 0; s < 4; s++) {
                if (base + (uint32_t)s >= n) break;
                SeedBits sb = pack_seed(seeds + (size_t)(base + (uint32_t)s) * ix.seed_len, ix.seed_len);
                if (sidx == s) { my_bases = dir ? sb.rc : sb.bases; my_valid = sb.valid; }
            }
            const uint32_t seed_i = base + (uint32_t)sidx;
            const bool in_range = seed_i < n;
            const bool active = in_range && my_valid;
            const uint32_t key = (uint32_t)(my_bases & ((1ull << key_bits) - 1));
            const uint32_t table = (uint32_t)(my_bases >> key_bits);
            uint32_t lines = 0;
            const uint32_t v = bucket_probe8(ix.bucket_blob, ix.bucket_offset, ix.n_buckets, table, key, active, &lines);
            // decode (GenomeIndex.cpp:2160-2202): singleton / absent / overflow list; the group leader fetches the list's count word
            long long nh = active ? 0 : -1;
            uint32_t ofs = 0; bool is_list = false;
            if (active && v != BUCKET_INVALID) {
                if ((uint64_t)v < ix.n_bases) nh = 1;
                else if (v != 0xfffffffeu) { ofs = v - n_bases32; is_list = true; }
            }
            uint32_t cnt = 0;
            if (is_list && e == 0) cnt = ix.overflow[ofs];
            cnt = (uint32_t)__shfl((int)cnt, g * 8);
            if (is_list) nh = (long long)(int32_t)cnt;
            if (in_range && e == 0) n_hits[2 * (size_t)seed_i + dir] = nh;
            const unsigned long long leaders = BALLOT(active && e == 0);
            c_lookups += (unsigned long long)__popcll(BALLOT(active && e == 0 && dir == 0));
            // bytes model: one 64-byte line = 8 slots of 8 bytes per bucket read
            for (int gg = 0; gg < 8; gg++) {
                const bool g_act = (leaders >> (gg * 8)) & 1ull;
                if (!g_act) continue;
                c_slots += 8ull * (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)lines, gg * 8);
                const long long g_nh = ((long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)nh >> 32), gg * 8) << 32) |
                                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)nh, gg * 8);
                const uint32_t g_v = (uint32_t)__builtin_amdgcn_readlane((int)v, gg * 8);
                const uint32_t g_ofs = (uint32_t)__builtin_amdgcn_readlane((int)ofs, gg * 8);
                const uint32_t g_seed = base + (uint32_t)(gg >> 1);
                if (g_nh > 1) c_lists++;
                const long long lim = g_nh < (long long)max_hits_out ? g_nh : (long long)max_hits_out;
                if (lim <= 0) continue;
                c_hits += (unsigned long long)lim;
                uint32_t *dst = hits ? hits + (size_t)(2 * (size_t)g_seed + (uint32_t)(gg & 1)) * max_hits_out : nullptr;
                for (long long j = lane; j < lim; j += WAVE) {
                    const uint32_t hv = g_nh == 1 ? g_v : ix.overflow[g_ofs + 1 + (uint32_t)j];
                    if (dst) dst[j] = hv; else sink ^= hv;
                }
            }
        }
    } else
    for (uint32_t i = wave; i < n; i += n_waves) {
        SeedBits seed = pack_seed(seeds + (size_t)i * ix.seed_len, ix.seed_len);
        if (!seed.valid) {
            if (lane == 0) { n_hits[2 * i] = -1; n_hits[2 * i + 1] = -1; }
            continue;
        }
        HitList hl[2];
        lookup_seed(ix, seed, hl);
        c_lookups++; c_slots += hl[0].slots + hl[1].slots;
        for (int d = 0; d < 2; d++) {
            if (lane == 0) n_hits[2 * i + d] = hl[d].n_hits;
            int64_t lim = hl[d].n_hits < (int64_t)max_hits_out ? hl[d].n_hits : (int64_t)max_hits_out;
            if (hl[d].n_hits > 1) c_lists++;
            c_hits += (unsigned long long)lim;
            uint32_t *dst = hits ? hits + (size_t)(2 * i + d) * max_hits_out : nullptr;
            for (int64_t j = lane; j < lim; j += WAVE) {
                const uint32_t v = hl[d].n_hits == 1 ? hl[d].singleton : hl[d].hits[j];
                if (dst) dst[j] = v; else sink ^= v;
            }
        }
    }
    if (sink == 0xDEADBEEFu && n == 0xFFFFFFFFu) n_hits[0] = (long long)sink;       // keeps the hit loads alive when nothing is stored
    if (counters && lane == 0) {
        atomicAdd(&counters[1], c_lookups); atomicAdd(&counters[2], c_slots); atomicAdd(&counters[3], c_hits); atomicAdd(&counters[4], c_lists);
    }
}

struct LVBatchArgs {
    int dir; uint32_t n; uint32_t kmax; uint32_t pcap;
    const uint8_t *texts; const uint32_t *text_off; const int32_t *text_len;
    const uint8_t *patterns; const uint8_t *quals; const uint32_t *pat_off; const int32_t *pat_len; const int32_t *k;
    int32_t *score; double *prob; int32_t *net_indel; int32_t *total_indels; int32_t *text_span;
    const DevTables *tab;
};

// One wave per problem: LandauVishkin<dir>::computeEditDistance.
__global__ __launch_bounds__(256) void k_lv_batch(LVBatchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id();
    const int wave_in_block = (int)(threadIdx.x >> 6);
    const uint32_t per_wave = (lv_lds_bytes(a.kmax, a.pcap) + 15) & ~15u;
    uint16_t *tri = (uint16_t *)(lds + (size_t)wave_in_block * per_wave);
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = wave; i < a.n; i += n_waves) {
        int plen = a.pat_len[i], tlen = a.text_len[i], k = a.k[i];
        const uint8_t *p = a.patterns + a.pat_off[i];
        const uint8_t *q = a.quals + a.pat_off[i];
        const uint8_t *t = a.texts + a.text_off[i];
        LVResult r;
        if (a.dir == 1) {
            ByteSeq P{p, 1}, Q{q, 1}, T{t, 1};
            r = lv_compute(P, Q, plen, T, tlen, k, tri, a.kmax, a.tab, a.pcap);
        } else {
            ByteSeq P{p, 1}, Q{q, 1}, T{t - 1, -1};     // the reference does text-- then walks backwards
            r = lv_compute(P, Q, plen, T, tlen, k, tri, a.kmax, a.tab, a.pcap);
        }
        if (lane == 0) {
            a.score[i] = r.score; a.prob[i] = r.match_probability; a.net_indel[i] = r.net_indel;
            a.total_indels[i] = r.total_indels; a.text_span[i] = r.text_span;
        }
    }
}

struct AGBatchArgs {
    int dir; uint32_t n; uint32_t RL;
    AGParams prm;
    const uint8_t *texts; const uint32_t *text_off; const int32_t *text_len;
    const uint8_t *patterns; const uint8_t *quals; const uint32_t *pat_off; const int32_t *pat_len;
    const int32_t *w; const int32_t *score_init; const uint8_t *is_rc; const uint8_t *banded; const uint8_t *use_clip;
    uint8_t *scratch;
    int32_t *ag_score; int32_t *text_offset; int32_t *pattern_offset; int32_t *n_edits; double *prob;
    int32_t *stale;
    const DevTables *tab;
};

// One wave per problem: AffineGapVectorized<dir>::computeScore / computeScoreBanded.
template <int AGC>
__global__ __launch_bounds__(64) void k_ag_batch(AGBatchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id();
    int16_t *rows = (int16_t *)lds;
    uint8_t *bt = a.scratch + (size_t)blockIdx.x * ag_scratch_bytes(a.RL);
    for (uint32_t i = blockIdx.x; i < a.n; i += gridDim.x) {
        int plen = a.pat_len[i], tlen = a.text_len[i];
        const uint8_t *p = a.patterns + a.pat_off[i];
        const uint8_t *q = a.quals + a.pat_off[i];
        const uint8_t *t = a.texts + a.text_off[i];
        AGResult r;
        if (a.dir == 1) {
            ByteSeq P{p, 1}, Q{q, 1}, T{t, 1};
            r = ag_dispatch<AGC>(a.banded[i] != 0, 1, a.prm, P, Q, plen, T, tlen, a.w[i], a.score_init[i], a.is_rc[i] != 0,
                           (int)a.use_clip[i], rows, bt, a.RL, a.tab);
        } else {
            ByteSeq P{p, 1}, Q{q, 1}, T{t - 1, -1};
            r = ag_dispatch<AGC>(a.banded[i] != 0, -1, a.prm, P, Q, plen, T, tlen, a.w[i], a.score_init[i], a.is_rc[i] != 0,
                           (int)a.use_clip[i], rows, bt, a.RL, a.tab);
        }
        if (lane == 0) {
            a.ag_score[i] = r.ag_score; a.text_offset[i] = r.text_offset; a.pattern_offset[i] = r.pattern_offset;
            a.n_edits[i] = r.n_edits; a.prob[i] = r.match_probability;
            if (a.stale) a.stale[i] = r.stale_reads;
        }
    }
}

// The same problems as CALLS IN ORDER ON ONE OBJECT: one wave, the exact form (ag.h: EXACT) over one image of the object's traceback array
// that starts zeroed and is kept from call to call -- what a newly constructed AffineGapVectorized<dir> answers for the sequence, the
// out-of-band traceback steps that read what an earlier call left behind included.  (Test entry: snapgpu_affine_gap_sequence.)
template <int AGC>
__global__ __launch_bounds__(64) void k_ag_sequence(AGBatchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id();
    int16_t *rows = (int16_t *)lds;
    uint8_t *bt = a.scratch;
    for (uint32_t i = 0; i < a.n; i++) {
        int plen = a.pat_len[i], tlen = a.text_len[i];
        const uint8_t *p = a.patterns + a.pat_off[i];
        const uint8_t *q = a.quals + a.pat_off[i];
        const uint8_t *t = a.texts + a.text_off[i];
        AGResult r;
        if (a.dir == 1) {
            ByteSeq P{p, 1}, Q{q, 1}, T{t, 1};
            r = ag_dispatch<AGC, true>(a.banded[i] != 0, 1, a.prm, P, Q, plen, T, tlen, a.w[i], a.score_init[i], a.is_rc[i] != 0,
                                       (int)a.use_clip[i], rows, bt, a.RL, a.tab);
        } else {
            ByteSeq P{p, 1}, Q{q, 1}, T{t - 1, -1};
            r = ag_dispatch<AGC, true>(a.banded[i] != 0, -1, a.prm, P, Q, plen, T, tlen, a.w[i], a.score_init[i], a.is_rc[i] != 0,
                                       (int)a.use_clip[i], rows, bt, a.RL, a.tab);
        }
        WAVE_SYNC(); __threadfence_block();
        if (lane == 0) {
            a.ag_score[i] = r.ag_score; a.text_offset[i] = r.text_offset; a.pattern_offset[i] = r.pattern_offset;
            a.n_edits[i] = r.n_edits; a.prob[i] = r.match_probability;
            if (a.stale) a.stale[i] = r.stale_reads;
        }
    }
}

// =====================================================================================
// host side
// =====================================================================================

static thread_local std::string g_last_error;
static thread_local const struct snapgpu_ctx *g_share_buckets_from = nullptr;      // snapgpu_create_replica(share_index): adopt these bucket tables

struct DevPool {
    struct Slot { void *p; size_t cap; bool busy; unsigned long long stamp; };
    std::vector<Slot> slots;
    std::mutex m;
    void *acquire(size_t bytes, hipError_t *err) {
        std::lock_guard<std::mutex> l(m);
        int best = -1;
        for (size_t i = 0; i < slots.size(); i++)
            if (!slots[i].busy && slots[i].cap >= bytes && slots[i].cap <= 4 * bytes + 65536 && (best < 0 || slots[i].cap < slots[(size_t)best].cap)) best = (int)i;
        if (best >= 0) { slots[(size_t)best].busy = true; *err = hipSuccess; return slots[(size_t)best].p; }
        void *p = nullptr;
        const size_t cap = bytes + bytes / 4 + 256;
        *err = hipMalloc(&p, cap);
        if (*err != hipSuccess) {                    // (make room: drop what is idle, try once more)
            for (auto &s : slots) if (!s.busy && s.p) { (void)hipFree(s.p); s.p = nullptr; s.cap = 0; }
            *err = hipMalloc(&p, cap);
            if (*err != hipSuccess) return nullptr;
        }
        slots.push_back(Slot{p, cap, true, 0ull});
        return p;
    }
    // (a buffer comes back with whatever its last user left in it: DevBuf memory is UNINITIALISED, every kernel clears what it needs)
    // Idle buffers are kept for the next call of a similar size, but not without bound: batches of varying size (a last short one, a retry
    // with a larger cigar stride, one-read calls of the host record loop) would otherwise leave gigabytes of per-wave scratch pinned next to
    // the index.  More than MAX_IDLE idle buffers, or more than MAX_IDLE_BYTES of them: the largest idle ones go.
    static const size_t MAX_IDLE = 40, MAX_IDLE_BYTES = (size_t)24 << 30;     // (a 1 M-read call of snapgpu_align_sam_single parks 18 buffers, ~8 GB: 5.9 GB of row-loop results among them)
    unsigned long long tick = 0;
    void release(void *p) {
        // Too many idle buffers: the SMALLEST go (cheap to allocate again; dropping the largest, as rounds 4 - 5 did, meant a hipMalloc / hipFree of the multi-GB
        // row-loop buffer per call for a context that alternates between call shapes).  Too many idle BYTES: the least recently released go.  hipFree
        // synchronises the device: it is called after the pool's lock is dropped.
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> l(m);
            for (auto &s : slots) if (s.p == p) { s.busy = false; s.stamp = ++tick; break; }
            for (;;) {
                size_t n_idle = 0, bytes = 0; int small = -1, old = -1;
                for (size_t i = 0; i < slots.size(); i++) if (!slots[i].busy && slots[i].p) {
                    n_idle++; bytes += slots[i].cap;
                    if (small < 0 || slots[i].cap < slots[(size_t)small].cap) small = (int)i;
                    if (old < 0 || slots[i].stamp < slots[(size_t)old].stamp) old = (int)i;
                }
                int victim = -1;
                if (n_idle > MAX_IDLE) victim = small; else if (bytes > MAX_IDLE_BYTES) victim = old;
                if (victim < 0) break;
                drop.push_back(slots[(size_t)victim].p);
                slots.erase(slots.begin() + victim);
            }
        }
        for (void *q : drop) (void)hipFree(q);
    }
    void free_all() {
        std::lock_guard<std::mutex> l(m);
        for (auto &s : slots) if (s.p) (void)hipFree(s.p);
        slots.clear();
    }
};
struct snapgpu_ctx {
    DevPool pool;                           // device buffers of the host-pointer entry points, kept between calls
    int device = -1;
    hipStream_t stream = nullptr;
    DevIndex ix{};
    bool owns_index = true;
    void *d_hash = nullptr, *d_overflow = nullptr, *d_genome_padded = nullptr;
    void *d_table_offset = nullptr, *d_table_size = nullptr, *d_contig_begin = nullptr;
    void *d_bucket_blob = nullptr, *d_bucket_offset = nullptr, *d_n_buckets = nullptr;      // device-native hash layout (bucket.h)
    bool owns_buckets = true; uint64_t bucket_bytes = 0;
    unsigned long long *d_planes = nullptr; bool owns_planes = true; uint64_t plane_bytes = 0;     // bit-plane shadow of the genome (planes.h)
    void *d_proj = nullptr;           // [proj_begin u64 x n][cigar_start u32 x (n+1)][cigar_ops u32 x m][proj_rc u8 x n]
    PEProj proj{};
    DevTables *d_tab = nullptr;
    DevTables h_tab{};
    snapgpu_params params{};
    AlignCfg cfg{};
    uint8_t *d_scratch = nullptr;
    uint32_t n_wave_slots = 0;
    uint32_t *d_work = nullptr;
    unsigned long long *d_counters = nullptr;
    // staging for the host-pointer entry points
    void *d_stage[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t stage_cap[5] = {0, 0, 0, 0, 0};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double kernel_ms = 0.0;
    uint64_t kernel_launches = 0;
    int num_cus = 0;
    int lookup_blocks_per_cu = 0;     // resident blocks per CU of k_lookup_seeds20 on this context's device (launch_lookup)
    int ag_variant = 0;               // chunks of 64 striped positions the affine-gap kernel variant holds in registers (0 = LDS form)
    const int32_t *clip_front = nullptr, *clip_len = nullptr; const uint8_t *clip_skip = nullptr;    // snapgpu_align_sam_single: Read::clip's outcome for the launch in hand (device)
    // secondary results (snapgpu_enable_secondary)
    bool secondary = false;
    SecCfg sec_cfg{};
    uint8_t *d_sec_scratch = nullptr;
    uint64_t sec_stride_bytes = 0;
    void *d_sec_stage[2] = {nullptr, nullptr};       // secondary records, counts (host-pointer entry point)
    size_t sec_stage_cap[2] = {0, 0};
    // paired-end path (snapgpu_enable_paired)
    bool paired = false;
    snapgpu_paired_params pparams{};
    PairedArgs pargs{};               // everything but the per-call pointers
    PairedArgs pargs_big{};           // second pass: a few waves with 32x larger candidate buffers
    uint8_t *d_pscratch = nullptr, *d_pscratch_big = nullptr;
    uint32_t *d_flag_list = nullptr; size_t flag_list_cap = 0;
    // exact replay of flagged reads / pairs: the reference's traceback arrays per replay wave (2 per read, 4 per pair)
    uint8_t *d_exact_persist = nullptr; uint64_t exact_persist_stride = 0; uint32_t exact_slots = 0;
    uint8_t *d_pexact_persist = nullptr; uint64_t pexact_persist_stride = 0; uint32_t pexact_slots = 0;
    // the exact kernel beside the paired main pass (launch_paired, opt-in): its stream, fork / join events
    hipStream_t replay_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; int replay_beside = 0;
    // 192-position variant: the exact kernels ARE the main pass (every wave keeps the images; each unit clears what the last one wrote)
    bool always_exact = false, p_always_exact = false;
    // Phase-4 help (paired_dev.h): [done counter | slots] and the per-slot PEHelpSpec arrays
    uint8_t *d_help = nullptr; size_t help_bytes = 0; uint32_t n_help = 0; PEHelpSpec *d_help_spec = nullptr; uint32_t help_spec_cap = 0; uint32_t help_min = 0;
    uint32_t p_wave_slots = 0, p_big_slots = 0, p_lds_per_wave = 0;
    int p_ag_variant = 0;
    // paired-end path with secondary results (both snapgpu_enable_paired and snapgpu_enable_secondary called): its own slabs
    // experimental: dequeue heavy pairs first (SNAPGPU_PAIRED_HEAVY_FIRST=1 at snapgpu_enable_paired; paired_dev.h)
    bool heavy_first = false;
    // heavy-first dequeue of the single-end path (order.h).  -1: where the context has the GPU to itself (one feeder: 226 -> 210 ms per launch,
    // profiles/r03b); with three feeders the tails of the launches overlap anyway and the ordering pass plus "every launch starts with its
    // heaviest reads" cost 12 % (7.93 M -> 8.93 M reads/s without it, profiles/r03q).  SNAPGPU_SINGLE_HEAVY_FIRST=0 / 1 decides for all launches.
    int single_heavy_first = -1;
    bool phase_timers = false;        // SNAPGPU_PHASE_TIMERS=1: launch the instantiation that carries the s_memtime phase timers
    uint32_t *d_order = nullptr, *d_wbucket = nullptr, *d_whist = nullptr; size_t order_cap = 0;
    // help for heavy reads (se_help.h): on unless SNAPGPU_SINGLE_HELP=0 at snapgpu_create (then the 192-position variant runs the exact
    // form as its main pass again, as in round 2)
    bool single_help = true, single_help_eager = false; uint32_t single_help_keep = 3;
    // contexts over this index on this device (this one and the feeder replicas that share its blobs).  The help is for a context that
    // has the GPU to itself: with several batches in flight the next batch's blocks want the wave slots idle helpers would sit on, and the
    // tails overlap anyway (measured: three feeders 130 ms per batch without, 140-147 ms with: profiles/r03g, r03h) -- such launches run the
    // exact form as their one pass, as in round 2.  SNAPGPU_SINGLE_HELP=1 forces the help on whatever the number of feeders, =0 off.
    std::shared_ptr<std::atomic<int>> feeders;
    uint32_t stop_on_first_hit = 0, explore_popular_seeds = 0;      // snapgpu_set_aligner_flags: the single-end launches' -f / -x
    int single_help_forced = -1;
    SEHelpSlot *d_se_slots = nullptr; SESpec *d_se_spec = nullptr; uint32_t *d_se_ctl = nullptr; uint32_t se_spec_cap = 0;
    unsigned long long *d_dbg = nullptr;          // phase_timers: launch diagnostics of the last single-end launch (kernel_common.h: AlignArgs::dbg)
    bool paired_sec = false;
    PairedArgs pargs_sec{}, pargs_sec_big{};
    uint8_t *d_pscratch_sec = nullptr, *d_pscratch_sec_big = nullptr;
    uint32_t p_sec_slots = 0, p_sec_big_slots = 0;
    int paired_share = 1;               // calls in flight on the device when this context's current paired-end call began (PairedInFlight)
    void *d_psec_stage[4] = {nullptr, nullptr, nullptr, nullptr};      // paired secondary, counts, single secondary, counts
    size_t psec_stage_cap[4] = {0, 0, 0, 0};
    // what snapgpu_create was given, minus the blobs: lets snapgpu_create_replica build another context over the same index
    snapgpu_index_view view_meta{};
    std::vector<uint64_t> h_table_offset, h_table_size, h_contig_begin, h_proj_begin;
    std::vector<uint8_t> h_proj_rc; std::vector<uint32_t> h_cigar_start, h_cigar_ops;
    std::string err;
};

#define HIPCHK(ctx, call, code)                                                                   \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            std::string m = std::string(#call) + ": " + hipGetErrorString(_e);                    \
            if (ctx) (ctx)->err = m;                                                              \
            g_last_error = m;                                                                     \
            return (code);                                                                        \
        }                                                                                         \
    } while (0)

static int fail(snapgpu_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    g_last_error = msg;
    return code;
}

// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and a hardware queue runs its kernels in
// order: two feeders whose streams land on one queue take turns, whatever their grids ask for.  Seen in the driver's bench command, whose paired-end
// legs come after the single-end leg's contexts (profiles/r06zzz: blocking call 7.6 s around 4.4 s of kernels, 410 k reads/s where the same leg alone
// ran at 544 k) and with GPU_MAX_HW_QUEUES=2 (362 k, profiles/r06v).  So: a context has ONE stream unless it asked for the replay beside the main
// pass, and the library asks for 8 hardware queues when it is loaded before the runtime starts (a value in the environment wins).
__attribute__((constructor)) static void snapgpu_hw_queues(void) { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

extern "C" int snapgpu_abi_version(void) { return SNAPGPU_ABI_VERSION; }

extern "C" int snapgpu_set_aligner_flags(snapgpu_ctx *ctx, int stop_on_first_hit, int explore_popular_seeds)
{
    if (!ctx) return SNAPGPU_E_INVALID;
    ctx->stop_on_first_hit = stop_on_first_hit ? 1u : 0u;
    ctx->explore_popular_seeds = explore_popular_seeds ? 1u : 0u;
    return SNAPGPU_OK;
}
// (index_build.hip reports through the same per-thread message as the functions of this file)
extern "C" void snapgpu_set_last_error(const char *msg) { g_last_error = msg ? msg : ""; }

extern "C" const char *snapgpu_last_error(const snapgpu_ctx *ctx) {
    if (ctx && !ctx->err.empty()) return ctx->err.c_str();
    return g_last_error.c_str();
}

extern "C" void snapgpu_default_params(snapgpu_params *p) {      // AlignerOptions.cpp:39-117 (single-end)
    memset(p, 0, sizeof(*p));
    p->max_hits = 300; p->max_k = 27; p->num_seeds = 25; p->seed_coverage = 0.0;
    p->min_weight_to_check = 1; p->extra_search_depth = 1; p->use_affine_gap = 1;
    p->match_reward = 1; p->sub_penalty = 4; p->gap_open_penalty = 6; p->gap_extend_penalty = 1;
    p->five_prime_end_bonus = 10; p->three_prime_end_bonus = 7;
    p->alt_awareness = 1; p->emit_alt_alignments = 0; p->max_score_gap_to_prefer_non_alt = 64;
    p->max_read_len = 400;
}

// __builtin_powi as libgcc evaluates it: the reference is built as C++98, where
// pow(double, int) resolves to std::pow(double, int) == __builtin_powi (BaseAligner.cpp:1314).
static double powi_like_libgcc(double x, int m) {
    unsigned n = m < 0 ? -(unsigned)m : (unsigned)m;
    double y = (n % 2) ? x : 1.0;
    while (n >>= 1) {
        x = x * x;
        if (n % 2) y *= x;
    }
    return m < 0 ? 1.0 / y : y;
}

static void build_tables(DevTables &t, unsigned seed_len) {
    const double SNP_PROB = 0.001, GAP_OPEN_PROB = 0.001, GAP_EXTEND_PROB = 0.5;   // BaseAligner.h:368-370
    // LandauVishkin.cpp:734-760
    t.indel[0] = 1.0;
    t.indel[1] = GAP_OPEN_PROB;
    for (int i = 2; i < N_INDEL_PROB; i++) t.indel[i] = t.indel[i - 1] * GAP_EXTEND_PROB;
    const double mutationRate = SNP_PROB;
    for (int i = 0; i < 33; i++) t.phred[i] = mutationRate;
    for (int i = 33; i <= 93 + 33; i++) t.phred[i] = 1.0 - (1.0 - pow(10.0, -1.0 * (i - 33.0) / 10.0)) * (1.0 - mutationRate);
    for (int i = 93 + 33 + 1; i < 256; i++) t.phred[i] = mutationRate;
    t.perfect[0] = 1.0;
    for (int i = 1; i < N_PERFECT_PROB; i++) t.perfect[i] = t.perfect[i - 1] * (1 - SNP_PROB);
    t.seed_prob = powi_like_libgcc(1 - SNP_PROB, (int)seed_len);
    t.seed_prob_pow = pow(1 - SNP_PROB, (double)seed_len);      // (host libm, as the reference's: see DevTables)

    // MAPQ thresholds: threshold[m] = largest x with (int)(-10*log10(x)) >= m, found by bisection
    // over the ordered bit patterns of positive doubles with the HOST log10 (mapq.h:53-59).
    t.mapq_threshold[0] = INFINITY;
    for (int m = 1; m <= 70; m++) {
        uint64_t lo, hi;                 // invariant: f(lo) >= m, f(hi) < m
        double dlo = 1e-300, dhi = 1.0;
        memcpy(&lo, &dlo, 8); memcpy(&hi, &dhi, 8);
        while (hi - lo > 1) {
            uint64_t mid = lo + (hi - lo) / 2;
            double x; memcpy(&x, &mid, 8);
            int f = (int)(-10 * log10(x));
            if (f >= m) lo = mid; else hi = mid;
        }
        double x; memcpy(&x, &lo, 8);
        t.mapq_threshold[m] = x;
    }
    t.mapq_threshold[71] = 0.0;

    // GetWrappedNextSeedToTest: the table SeedSequencer builds with a FIFO of intervals
    // (SeedSequencer.cpp:36-103), indexed by wrapCount exactly as SeedSequencer.h:40-43 does.
    memset(t.wrapped_seed, 0, sizeof(t.wrapped_seed));
    if (seed_len >= 2 && seed_len <= 32) {
        std::vector<unsigned> offsets(seed_len, 0);
        std::deque<std::pair<unsigned, unsigned>> work;
        work.push_back({1u, seed_len - 1});
        unsigned filled = 1;
        while (!work.empty()) {
            auto it = work.front(); work.pop_front();
            unsigned sel = (it.first + it.second) / 2;
            offsets[sel] = filled++;
            if (it.second > sel) work.push_back({sel + 1, it.second});
            if (it.first < sel) work.push_back({it.first, sel - 1});
        }
        for (unsigned i = 0; i < seed_len; i++) t.wrapped_seed[i] = offsets[i];
    }
}

static uint32_t next_pow2(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

static int ensure_stage(snapgpu_ctx *ctx, int which, size_t bytes) {
    if (ctx->stage_cap[which] >= bytes) return 0;
    if (ctx->d_stage[which]) (void)hipFree(ctx->d_stage[which]);
    ctx->d_stage[which] = nullptr; ctx->stage_cap[which] = 0;
    size_t cap = bytes + bytes / 4 + 4096;
    HIPCHK(ctx, hipMalloc(&ctx->d_stage[which], cap), SNAPGPU_E_NOMEM);
    ctx->stage_cap[which] = cap;
    return 0;
}

extern "C" void snapgpu_destroy(snapgpu_ctx *ctx) {
    if (!ctx) return;
    if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
    if (ctx->owns_index) {
        if (ctx->d_hash) (void)hipFree(ctx->d_hash);
        if (ctx->d_overflow) (void)hipFree(ctx->d_overflow);
        if (ctx->d_genome_padded) (void)hipFree(ctx->d_genome_padded);
    }
    if (ctx->owns_buckets) {
        if (ctx->d_bucket_blob) (void)hipFree(ctx->d_bucket_blob);
        if (ctx->d_bucket_offset) (void)hipFree(ctx->d_bucket_offset);
        if (ctx->d_n_buckets) (void)hipFree(ctx->d_n_buckets);
    }
    if (ctx->d_sec_scratch) (void)hipFree(ctx->d_sec_scratch);
    for (int i = 0; i < 2; i++) if (ctx->d_sec_stage[i]) (void)hipFree(ctx->d_sec_stage[i]);
    if (ctx->d_table_offset) (void)hipFree(ctx->d_table_offset);
    if (ctx->d_table_size) (void)hipFree(ctx->d_table_size);
    if (ctx->d_contig_begin) (void)hipFree(ctx->d_contig_begin);
    if (ctx->d_proj) (void)hipFree(ctx->d_proj);
    if (ctx->d_tab) (void)hipFree(ctx->d_tab);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_pscratch) (void)hipFree(ctx->d_pscratch);
    if (ctx->d_pscratch_big) (void)hipFree(ctx->d_pscratch_big);
    if (ctx->d_pscratch_sec) (void)hipFree(ctx->d_pscratch_sec);
    if (ctx->d_pscratch_sec_big) (void)hipFree(ctx->d_pscratch_sec_big);
    for (int i = 0; i < 4; i++) if (ctx->d_psec_stage[i]) (void)hipFree(ctx->d_psec_stage[i]);
    if (ctx->d_flag_list) (void)hipFree(ctx->d_flag_list);
    if (ctx->replay_stream) (void)hipStreamDestroy(ctx->replay_stream);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->d_exact_persist) (void)hipFree(ctx->d_exact_persist);
    if (ctx->d_pexact_persist) (void)hipFree(ctx->d_pexact_persist);
    if (ctx->d_help) (void)hipFree(ctx->d_help);
    if (ctx->d_help_spec) (void)hipFree(ctx->d_help_spec);
    if (ctx->feeders) ctx->feeders->fetch_sub(1);
    if (ctx->d_dbg) (void)hipFree(ctx->d_dbg);
    if (ctx->d_se_slots) (void)hipFree(ctx->d_se_slots);
    if (ctx->d_se_spec) (void)hipFree(ctx->d_se_spec);
    if (ctx->d_se_ctl) (void)hipFree(ctx->d_se_ctl);
    if (ctx->d_planes && ctx->owns_planes) (void)hipFree(ctx->d_planes);
    if (ctx->d_order) (void)hipFree(ctx->d_order);
    if (ctx->d_wbucket) (void)hipFree(ctx->d_wbucket);
    if (ctx->d_whist) (void)hipFree(ctx->d_whist);
    ctx->pool.free_all();
    if (ctx->d_work) (void)hipFree(ctx->d_work);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    for (int i = 0; i < 5; i++) if (ctx->d_stage[i]) (void)hipFree(ctx->d_stage[i]);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// Device-native hash layout (bucket.h): built on the GPU from the reference's slot arrays once they are in HBM
// (snapgpu_create for uploaded / adopted blobs, snapgpu_broadcast_index for the replicas it fills).
// The bit-plane shadow of the genome (planes.h), from the byte genome already in HBM -- however it got there.
// SNAPGPU_LV_PLANES=<non-zero> and not SNAPGPU_NO_PLANES: one parse for the context that builds the shadow and the feeders that adopt it
static bool lv_planes_wanted() {
    const char *e = getenv("SNAPGPU_LV_PLANES");
    return !getenv("SNAPGPU_NO_PLANES") && e && atoi(e) != 0;
}

// Built, and used by Landau-Vishkin, when SNAPGPU_LV_PLANES=1 (and not SNAPGPU_NO_PLANES=1).  NOT the default: measured on the bench batch
// (profiles/r03e) the plane form stages 2.7x fewer reference bytes per scored location but spends MORE instructions -- 1.59 M wave cycles
// per read against 1.48 M, Landau-Vishkin 22 % of them against 16 %: most calls end after one or two levels, where the byte form has
// built three or five bitmaps and the plane form all 2k + 1 -- for the same launch time.  DESIGN.md section 16.
static int build_planes(snapgpu_ctx *ctx)
{
    if (!lv_planes_wanted()) return SNAPGPU_OK;
    const uint64_t n_bytes = ctx->ix.n_bases + 2 * (uint64_t)ctx->ix.genome_pad;
    const uint64_t n_blocks = (n_bytes + 63) / 64 + 32;
    if (!ctx->d_planes) {
        ctx->plane_bytes = n_blocks * 24;
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_planes, (size_t)ctx->plane_bytes), SNAPGPU_E_NOMEM);
        ctx->owns_planes = true;
    }
    hipLaunchKernelGGL(k_genome_planes, dim3((unsigned)ctx->num_cus * 8), dim3(256), 0, ctx->stream, (const uint8_t *)ctx->d_genome_padded, n_bytes, n_blocks, ctx->d_planes);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream), SNAPGPU_E_LAUNCH);
    ctx->ix.planes = ctx->d_planes;
    return SNAPGPU_OK;
}

static int build_buckets(snapgpu_ctx *ctx)
{
    DevIndex &ix = ctx->ix;
    if (ix.large || ix.entry_bytes != 8 || ix.key_bytes > 4 || getenv("SNAPGPU_NO_BUCKETS")) return SNAPGPU_OK;      // this shape keeps the slot walk
    const uint32_t nt = ix.n_hash_tables;
    std::vector<uint64_t> nb(nt), boff(nt);
    uint64_t total = 0;
    for (uint32_t t = 0; t < nt; t++) { nb[t] = bucket_count_for(ctx->h_table_size[t]); boff[t] = total * BUCKET_BYTES; total += nb[t]; }
    if (!ctx->d_bucket_blob) {
        ctx->bucket_bytes = total * BUCKET_BYTES;
        HIPCHK(ctx, hipMalloc(&ctx->d_bucket_blob, (size_t)ctx->bucket_bytes + 64), SNAPGPU_E_NOMEM);
        HIPCHK(ctx, hipMalloc(&ctx->d_bucket_offset, (size_t)nt * 8), SNAPGPU_E_NOMEM);
        HIPCHK(ctx, hipMalloc(&ctx->d_n_buckets, (size_t)nt * 8), SNAPGPU_E_NOMEM);
        HIPCHK(ctx, hipMemcpy(ctx->d_bucket_offset, boff.data(), (size_t)nt * 8, hipMemcpyHostToDevice), SNAPGPU_E_NODEVICE);
        HIPCHK(ctx, hipMemcpy(ctx->d_n_buckets, nb.data(), (size_t)nt * 8, hipMemcpyHostToDevice), SNAPGPU_E_NODEVICE);
    }
    const uint32_t key_mask = ix.key_bytes >= 4 ? 0xffffffffu : ((1u << (8 * ix.key_bytes)) - 1u);
    hipLaunchKernelGGL(k_bucket_init, dim3((unsigned)ctx->num_cus * 8), dim3(256), 0, ctx->stream, (uint32_t *)ctx->d_bucket_blob, total);
    for (uint32_t t = 0; t < nt; t++) {
        if (ctx->h_table_size[t] == 0) continue;
        hipLaunchKernelGGL(k_bucket_build, dim3((unsigned)ctx->num_cus * 4), dim3(256), 0, ctx->stream,
                           (const uint32_t *)((const uint8_t *)ctx->d_hash + ctx->h_table_offset[t]), ctx->h_table_size[t], key_mask,
                           (uint32_t *)((uint8_t *)ctx->d_bucket_blob + boff[t]), nb[t]);
    }
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream), SNAPGPU_E_LAUNCH);
    ix.bucket_blob = (const uint8_t *)ctx->d_bucket_blob;
    ix.bucket_offset = (const uint64_t *)ctx->d_bucket_offset;
    ix.n_buckets = (const uint64_t *)ctx->d_n_buckets;
    return SNAPGPU_OK;
}

extern "C" int snapgpu_create(const snapgpu_index_view *idx, const snapgpu_params *p, int device, snapgpu_ctx **out)
{
    if (!idx || !p || !out) return fail(nullptr, SNAPGPU_E_INVALID, "snapgpu_create: null argument");
    *out = nullptr;
    if (idx->location_size != 4)
        return fail(nullptr, SNAPGPU_E_UNSUPPORTED, "index uses 64-bit genome locations; only the 32-bit lookup path (seed >= 20) is implemented");
    if (idx->seed_len < 2 || idx->seed_len > 32) return fail(nullptr, SNAPGPU_E_INVALID, "seed length out of range");
    if (idx->key_bytes < 2 || idx->key_bytes > 8) return fail(nullptr, SNAPGPU_E_INVALID, "hash key size out of range");
    if (idx->genome_pad < 1000) return fail(nullptr, SNAPGPU_E_INVALID, "genome_pad must be >= 1000");
    if (p->max_read_len < idx->seed_len || p->max_read_len > 1000)
        return fail(nullptr, SNAPGPU_E_INVALID, "max_read_len must be in [seed_len, 1000] (MAX_READ_LENGTH, Read.h:49)");
    if (p->sub_penalty <= p->gap_extend_penalty) return fail(nullptr, SNAPGPU_E_INVALID, "sub_penalty must exceed gap_extend_penalty");
    if (p->sub_penalty > p->gap_open_penalty + p->gap_extend_penalty)
        return fail(nullptr, SNAPGPU_E_INVALID, "subPenalty > gapOpen + gapExtend (BaseAligner.cpp:141)");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, SNAPGPU_E_NODEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(nullptr, SNAPGPU_E_NODEVICE, "device index out of range");

    snapgpu_ctx *ctx = new snapgpu_ctx();
    ctx->device = device;
    ctx->params = *p;
#define CRCHK(call, code) do { hipError_t _e = (call); if (_e != hipSuccess) { \
        fail(nullptr, code, std::string(#call) + ": " + hipGetErrorString(_e)); snapgpu_destroy(ctx); return code; } } while (0)
    CRCHK(hipSetDevice(device), SNAPGPU_E_NODEVICE);
    hipDeviceProp_t prop;
    CRCHK(hipGetDeviceProperties(&prop, device), SNAPGPU_E_NODEVICE);
    ctx->num_cus = prop.multiProcessorCount;
    CRCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking), SNAPGPU_E_NODEVICE);
    CRCHK(hipEventCreate(&ctx->ev0), SNAPGPU_E_NODEVICE);
    CRCHK(hipEventCreate(&ctx->ev1), SNAPGPU_E_NODEVICE);

    // ---- index blobs
    DevIndex &ix = ctx->ix;
    ix.n_bases = idx->n_bases; ix.overflow_size = idx->overflow_table_size;
    ix.first_alt_location = idx->first_alt_location; ix.n_contigs = idx->n_contigs;
    ix.seed_len = idx->seed_len; ix.key_bytes = idx->key_bytes; ix.large = idx->large_hash_table ? 1 : 0;
    ix.entry_bytes = 4 * (ix.large ? 2 : 1) + ix.key_bytes; ix.n_hash_tables = idx->n_hash_tables;
    ix.chromosome_padding = idx->chromosome_padding; ix.genome_pad = idx->genome_pad;
    const size_t genome_total = (size_t)idx->n_bases + 2 * (size_t)idx->genome_pad;
    const size_t overflow_words = idx->overflow_table_size ? (size_t)idx->overflow_table_size : 1;
    if (idx->on_device) {
        ctx->owns_index = false;
        ctx->d_hash = (void *)idx->hash_blob;
        ctx->d_overflow = (void *)idx->overflow;
        ctx->d_genome_padded = (void *)(idx->genome - idx->genome_pad);
    } else {
        CRCHK(hipMalloc(&ctx->d_hash, (size_t)idx->hash_blob_bytes + 16), SNAPGPU_E_NOMEM);
        CRCHK(hipMalloc(&ctx->d_overflow, overflow_words * 4 + 16), SNAPGPU_E_NOMEM);
        CRCHK(hipMalloc(&ctx->d_genome_padded, genome_total + 16), SNAPGPU_E_NOMEM);
        if (idx->hash_blob) CRCHK(hipMemcpy(ctx->d_hash, idx->hash_blob, (size_t)idx->hash_blob_bytes, hipMemcpyHostToDevice), SNAPGPU_E_NODEVICE);
        if (idx->overflow) CRCHK(hipMemcpy(ctx->d_overflow, idx->overflow, overflow_words * 4, hipMemcpyHostToDevice), SNAPGPU_E_NODEVICE);
        if (idx->genome) CRCHK(hipMemcpy(ctx->d_genome_padded, idx->genome - idx->genome_pad, genome_total, hipMemcpyHostToDevice), SNAPGPU_E_NODEVICE);
    }
    ix.hash_blob = (const uint8_t *)ctx->d_hash;
    ix.overflow = (const uint32_t *)ctx->d_overflow;
    ix.genome = (const uint8_t *)ctx->d_genome_padded + idx->genome_pad;
    CRCHK(hipMalloc(&ctx->d_table_offset, (size_t)idx->n_hash_tables * 8), SNAPGPU_E_NOMEM);
    CRCHK(hipMalloc(&ctx->d_table_size, (size_t)idx->n_hash_tables * 8), SNAPGPU_E_NOMEM);
    CRCHK(hipMemcpy(ctx->d_table_offset, idx->table_offset, (size_t)idx->n_hash_tables * 8, hipMemcpyHostToDevice), SNAPGPU_E_NODEVICE);
    CRCHK(hipMemcpy(ctx->d_table_size, idx->table_size, (size_t)idx->n_hash_tables * 8, hipMemcpyHostToDevice), SNAPGPU_E_NODEVICE);
    size_t nc = idx->n_contigs ? idx->n_contigs : 1;
    CRCHK(hipMalloc(&ctx->d_contig_begin, nc * 8), SNAPGPU_E_NOMEM);
    if (idx->n_contigs) CRCHK(hipMemcpy(ctx->d_contig_begin, idx->contig_begin, (size_t)idx->n_contigs * 8, hipMemcpyHostToDevice), SNAPGPU_E_NODEVICE);
    ix.table_offset = (const uint64_t *)ctx->d_table_offset;
    ix.table_size = (const uint64_t *)ctx->d_table_size;
    ix.contig_begin = (const uint64_t *)ctx->d_contig_begin;
    ix.bucket_blob = nullptr; ix.bucket_offset = nullptr; ix.n_buckets = nullptr; ix.planes = nullptr;
    {   // ALT-to-primary projections (used by the paired-end path's ALT liftover); absent data = "location 0, no CIGAR"
        const size_t n = idx->n_contigs;
        const uint32_t n_ops = (idx->contig_cigar_start && idx->cigar_ops && n) ? idx->contig_cigar_start[n] : 0;
        std::vector<uint8_t> img(n * 8 + (n + 1) * 4 + ((size_t)n_ops + 1) * 4 + n + 16, 0);
        uint64_t *pb = (uint64_t *)img.data();
        uint32_t *cs = (uint32_t *)(img.data() + n * 8);
        uint32_t *ops = cs + (n + 1);
        uint8_t *rc = (uint8_t *)(ops + (size_t)n_ops + 1);
        for (size_t i = 0; i < n; i++) {
            pb[i] = idx->contig_proj_begin ? idx->contig_proj_begin[i] : 0;
            rc[i] = idx->contig_proj_rc ? idx->contig_proj_rc[i] : 0;
            cs[i] = n_ops ? idx->contig_cigar_start[i] : 0;
        }
        cs[n] = n_ops;
        for (uint32_t i = 0; i < n_ops; i++) ops[i] = idx->cigar_ops[i];
        CRCHK(hipMalloc(&ctx->d_proj, img.size()), SNAPGPU_E_NOMEM);
        CRCHK(hipMemcpy(ctx->d_proj, img.data(), img.size(), hipMemcpyHostToDevice), SNAPGPU_E_NODEVICE);
        uint8_t *d = (uint8_t *)ctx->d_proj;
        ctx->proj.contig_begin = (const uint64_t *)ctx->d_contig_begin;
        ctx->proj.proj_begin = (const uint64_t *)d;
        ctx->proj.cigar_start = (const uint32_t *)(d + n * 8);
        ctx->proj.cigar_ops = ctx->proj.cigar_start + (n + 1);
        ctx->proj.proj_rc = (const uint8_t *)(ctx->proj.cigar_ops + (size_t)n_ops + 1);
        ctx->proj.n_contigs = idx->n_contigs;
    }

    {   // keep the description of the index (not the blobs) for snapgpu_create_replica
        ctx->view_meta = *idx;
        const size_t n = idx->n_contigs;
        ctx->h_table_offset.assign(idx->table_offset, idx->table_offset + idx->n_hash_tables);
        ctx->h_table_size.assign(idx->table_size, idx->table_size + idx->n_hash_tables);
        if (n) ctx->h_contig_begin.assign(idx->contig_begin, idx->contig_begin + n);
        if (idx->contig_proj_begin && n) ctx->h_proj_begin.assign(idx->contig_proj_begin, idx->contig_proj_begin + n);
        if (idx->contig_proj_rc && n) ctx->h_proj_rc.assign(idx->contig_proj_rc, idx->contig_proj_rc + n);
        if (idx->contig_cigar_start && idx->cigar_ops && n) {
            ctx->h_cigar_start.assign(idx->contig_cigar_start, idx->contig_cigar_start + n + 1);
            ctx->h_cigar_ops.assign(idx->cigar_ops, idx->cigar_ops + idx->contig_cigar_start[n]);
        }
        ctx->view_meta.hash_blob = nullptr; ctx->view_meta.overflow = nullptr; ctx->view_meta.genome = nullptr;
        ctx->view_meta.table_offset = nullptr; ctx->view_meta.table_size = nullptr; ctx->view_meta.contig_begin = nullptr;
        ctx->view_meta.contig_proj_begin = nullptr; ctx->view_meta.contig_proj_rc = nullptr; ctx->view_meta.contig_cigar_start = nullptr;
        ctx->view_meta.cigar_ops = nullptr;
    }

    if (g_share_buckets_from && g_share_buckets_from->d_bucket_blob && idx->on_device && !getenv("SNAPGPU_NO_BUCKETS")) {     // second feeder context on this GPU: adopt
        ctx->owns_buckets = false;
        ctx->d_bucket_blob = g_share_buckets_from->d_bucket_blob; ctx->d_bucket_offset = g_share_buckets_from->d_bucket_offset;
        ctx->d_n_buckets = g_share_buckets_from->d_n_buckets; ctx->bucket_bytes = g_share_buckets_from->bucket_bytes;
        ix.bucket_blob = (const uint8_t *)ctx->d_bucket_blob; ix.bucket_offset = (const uint64_t *)ctx->d_bucket_offset;
        ix.n_buckets = (const uint64_t *)ctx->d_n_buckets;
    } else if (idx->on_device || idx->hash_blob != nullptr) {      // (blobs left unfilled wait for snapgpu_broadcast_index)
        const int brc = build_buckets(ctx);
        if (brc != SNAPGPU_OK) { snapgpu_destroy(ctx); return brc; }
    }
    if (g_share_buckets_from && g_share_buckets_from->d_planes && idx->on_device && lv_planes_wanted()) {       // a feeder context: adopt the shadow too
        ctx->d_planes = g_share_buckets_from->d_planes; ctx->owns_planes = false; ctx->plane_bytes = g_share_buckets_from->plane_bytes;
        ix.planes = ctx->d_planes;
    } else if (idx->on_device || idx->genome != nullptr) {
        const int prc = build_planes(ctx);
        if (prc != SNAPGPU_OK) { snapgpu_destroy(ctx); return prc; }
    }

    // ---- tables
    build_tables(ctx->h_tab, idx->seed_len);
    CRCHK(hipMalloc((void **)&ctx->d_tab, sizeof(DevTables)), SNAPGPU_E_NOMEM);
    CRCHK(hipMemcpy(ctx->d_tab, &ctx->h_tab, sizeof(DevTables), hipMemcpyHostToDevice), SNAPGPU_E_NODEVICE);

    // ---- aligner configuration (BaseAligner ctor, BaseAligner.cpp:173-183, with maxReadSize = MAX_READ_LENGTH
    //      as SingleAligner.cpp:138 passes it)
    AlignCfg &c = ctx->cfg;
    c.max_hits = p->max_hits; c.max_k = p->max_k; c.num_seeds = p->num_seeds;
    c.min_weight = p->min_weight_to_check < 1 ? 1 : p->min_weight_to_check;   // max(1u, ...), BaseAligner.cpp:83
    c.extra_depth = p->extra_search_depth; c.use_ag = p->use_affine_gap ? 1 : 0; c.ag_buffers = c.use_ag;
    c.match_reward = (int)p->match_reward; c.sub_penalty = (int)p->sub_penalty;
    c.gap_open = (int)p->gap_open_penalty; c.gap_extend = (int)p->gap_extend_penalty;
    c.five_bonus = (int)p->five_prime_end_bonus; c.three_bonus = (int)p->three_prime_end_bonus;
    c.alt_aware = p->alt_awareness ? 1 : 0; c.emit_alt = p->emit_alt_alignments ? 1 : 0;
    c.max_gap_alt = p->max_score_gap_to_prefer_non_alt; c.seed_coverage = p->seed_coverage;
    uint32_t max_seeds_ctor = p->num_seeds != 0 ? p->num_seeds : (uint32_t)(int)(p->seed_coverage * 1000 / idx->seed_len);
    c.num_weight_lists = max_seeds_ctor + 1;
    if (c.num_weight_lists < 2 || c.num_weight_lists > 0x3FF) { snapgpu_destroy(ctx); return fail(nullptr, SNAPGPU_E_UNSUPPORTED, "number of seeds out of the supported range [1, 1022]"); }
    c.RL = (p->max_read_len + 15) & ~15u;
    uint32_t kmax = p->max_k + p->extra_search_depth; if (kmax > 126) kmax = 126;
    c.kmax = kmax;
    uint64_t pool = (uint64_t)p->max_hits * max_seeds_ctor;
    if (pool < 64) pool = 64;
    if (pool > 60000) { snapgpu_destroy(ctx); return fail(nullptr, SNAPGPU_E_UNSUPPORTED, "max_hits * num_seeds > 60000 candidate buckets per read is not supported"); }
    c.pool_size = (uint32_t)pool;
    c.ht_size = next_pow2((uint32_t)pool * 2);
    c.ag_numvec_max = (c.RL + 7) / 8;
    {   // affine-gap kernel variant: patterns are at most max_read_len - seed_len long, limits at most kmax
        int need = ag_max_positions((int)p->max_read_len - (int)idx->seed_len, (int)kmax);
        ctx->ag_variant = need <= 192 ? 3 : need <= 256 ? 4 : need <= 384 ? 6 : 0;
        if (getenv("SNAPGPU_AG_LDS")) ctx->ag_variant = 0;
    }
    size_t ag_bytes = c.ag_buffers ? ag_scratch_bytes(c.RL) : 0;
    if (const char *e = getenv("SNAPGPU_SINGLE_HELP")) { ctx->single_help_forced = atoi(e) != 0 ? 1 : 0; ctx->single_help = ctx->single_help_forced != 0; }
    if (g_share_buckets_from && g_share_buckets_from->feeders && idx->on_device) { ctx->feeders = g_share_buckets_from->feeders; ctx->feeders->fetch_add(1); }
    else ctx->feeders = std::make_shared<std::atomic<int>>(1);
    if (const char *e = getenv("SNAPGPU_SINGLE_HELP_EAGER")) ctx->single_help_eager = atoi(e) != 0;
    if (const char *e = getenv("SNAPGPU_SINGLE_HELP_KEEP")) { int v = atoi(e); if (v >= 1 && v <= 64) ctx->single_help_keep = (uint32_t)v; }
    c.se_items_cap = ctx->single_help ? SE_HELP_ITEMS_CAP : 0u;
    c.se_off = ((size_t)c.ht_size * 2 + (size_t)c.pool_size * sizeof(Elem) + ag_bytes + 255) & ~(size_t)255;
    c.scratch_stride = (c.se_off + ((size_t)c.se_items_cap + (c.se_items_cap ? c.pool_size : 0u)) * 4 + 255) & ~(size_t)255;
    c.ag_lds = !c.ag_buffers ? 0u : (ctx->ag_variant == 3 ? ag_lds_bytes_reg(c.RL, 3) : ag_lds_bytes(c.RL));
    LdsLayout L = lds_layout(c.RL, c.num_weight_lists, c.kmax, c.ag_lds);
    c.lds_per_wave = L.total;

    // waves in flight: a fixed number per CU, each with its own scratch slab
    int waves_per_cu = ctx->ag_variant == 3 ? 24 : 16;               // 4 SIMDs x the kernel's waves per SIMD (single_kernel.h)
    if (const char *e = getenv("SNAPGPU_WAVES_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 32) waves_per_cu = v; }
    // LDS limit: 160 KiB per CU
    while (waves_per_cu > 1 && (size_t)waves_per_cu * L.total > 160 * 1024) waves_per_cu--;
    // (every align kernel is launched as 4 waves per workgroup with 4 * L.total bytes of dynamic LDS: refuse here, not with an opaque launch error)
    if ((size_t)4 * L.total > 160 * 1024) { snapgpu_destroy(ctx); return fail(nullptr, SNAPGPU_E_UNSUPPORTED, "per-read LDS state exceeds 40 KiB per wave (max_read_len / num_seeds / max_k too large)"); }
    ctx->n_wave_slots = (uint32_t)ctx->num_cus * (uint32_t)waves_per_cu;
    ctx->n_wave_slots = (ctx->n_wave_slots + 3) & ~3u;
    size_t scratch_total = (size_t)ctx->n_wave_slots * c.scratch_stride;
    CRCHK(hipMalloc((void **)&ctx->d_scratch, scratch_total), SNAPGPU_E_NOMEM);
    // the head tables must start zeroed (one fill of the whole slab is cheaper than a fill per wave)
    CRCHK(hipMemsetAsync(ctx->d_scratch, 0, scratch_total, ctx->stream), SNAPGPU_E_NODEVICE);
    if (c.use_ag) {
        // Exactness of the banded affine-gap traceback.  192-position variant (reads up to ~170 bp): every wave keeps
        // the images of the reference objects' traceback arrays and the exact kernel is the only pass.  Longer reads: fast pass with the
        // arrays forgotten between calls + a replay of the flagged reads on 64 waves (kernel_common.h: AlignArgs::persist).
        // (with the help for heavy reads on -- the default since round 3 -- the main pass is the fast form: an idle wave cannot take part in
        //  a walk that is ordered through the traceback arrays its calls share)
        // (whether a launch runs the exact form as its main pass is decided per launch -- launch_align -- so the images are there for every wave)
        ctx->always_exact = ctx->ag_variant == 3 && !getenv("SNAPGPU_NO_ALWAYS_EXACT");
        ctx->exact_slots = ctx->always_exact ? ctx->n_wave_slots : (ctx->n_wave_slots < 64 ? ctx->n_wave_slots : 64);
        ctx->exact_persist_stride = 2 * (uint64_t)((ag_bytes + 255) & ~(size_t)255);
        CRCHK(hipMalloc((void **)&ctx->d_exact_persist, (size_t)ctx->exact_slots * ctx->exact_persist_stride), SNAPGPU_E_NOMEM);
        CRCHK(hipMemsetAsync(ctx->d_exact_persist, 0, (size_t)ctx->exact_slots * ctx->exact_persist_stride, ctx->stream), SNAPGPU_E_NODEVICE);
    }
    if (const char *e = getenv("SNAPGPU_SINGLE_HEAVY_FIRST")) ctx->single_heavy_first = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("SNAPGPU_PHASE_TIMERS")) ctx->phase_timers = atoi(e) != 0;
    if (ctx->single_help) {
        ctx->se_spec_cap = c.se_items_cap;
        CRCHK(hipMalloc((void **)&ctx->d_se_slots, SE_HELP_SLOTS * sizeof(SEHelpSlot)), SNAPGPU_E_NOMEM);
        CRCHK(hipMalloc((void **)&ctx->d_se_spec, (size_t)SE_HELP_SLOTS * ctx->se_spec_cap * sizeof(SESpec)), SNAPGPU_E_NOMEM);
        CRCHK(hipMalloc((void **)&ctx->d_se_ctl, 256), SNAPGPU_E_NOMEM);
    }
    CRCHK(hipMalloc((void **)&ctx->d_work, 256), SNAPGPU_E_NOMEM);
    CRCHK(hipMalloc((void **)&ctx->d_counters, sizeof(snapgpu_counters)), SNAPGPU_E_NOMEM);
    CRCHK(hipMemsetAsync(ctx->d_counters, 0, sizeof(snapgpu_counters), ctx->stream), SNAPGPU_E_NODEVICE);
    CRCHK(hipStreamSynchronize(ctx->stream), SNAPGPU_E_NODEVICE);
#undef CRCHK
    *out = ctx;
    return SNAPGPU_OK;
}

// ---- several contexts over one index: more feeder threads per GPU, more GPUs (SNAPLib/ParallelTask.h:128-138 is the
// reference's equivalent: one aligner per thread over the shared g_index)
extern "C" int snapgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int snapgpu_create_replica(const snapgpu_ctx *src, int device, int share_index, snapgpu_ctx **out)
{
    return snapgpu_create_replica_with_params(src, device, share_index, nullptr, out);
}

extern "C" int snapgpu_create_replica_with_params(const snapgpu_ctx *src, int device, int share_index, const snapgpu_params *p, snapgpu_ctx **out)
{
    if (!src || !out) return fail(nullptr, SNAPGPU_E_INVALID, "snapgpu_create_replica: null argument");
    *out = nullptr;
    if (share_index && device != src->device) return fail(nullptr, SNAPGPU_E_INVALID, "snapgpu_create_replica: index blobs can only be shared on the device that holds them");
    snapgpu_index_view v = src->view_meta;
    v.table_offset = src->h_table_offset.data(); v.table_size = src->h_table_size.data();
    v.contig_begin = src->h_contig_begin.empty() ? nullptr : src->h_contig_begin.data();
    v.contig_proj_begin = src->h_proj_begin.empty() ? nullptr : src->h_proj_begin.data();
    v.contig_proj_rc = src->h_proj_rc.empty() ? nullptr : src->h_proj_rc.data();
    v.contig_cigar_start = src->h_cigar_start.empty() ? nullptr : src->h_cigar_start.data();
    v.cigar_ops = src->h_cigar_ops.empty() ? nullptr : src->h_cigar_ops.data();
    if (share_index) {          // a second feeder context on the same GPU: adopt the blobs, own streams / scratch / staging
        v.on_device = 1;
        v.hash_blob = (const uint8_t *)src->d_hash; v.overflow = (const uint32_t *)src->d_overflow;
        v.genome = (const uint8_t *)src->d_genome_padded + src->view_meta.genome_pad;
    } else {                    // same-size blobs on `device`, left for snapgpu_broadcast_index to fill
        v.on_device = 0;
        v.hash_blob = nullptr; v.overflow = nullptr; v.genome = nullptr;
    }
    g_share_buckets_from = share_index ? src : nullptr;
    const int rc = snapgpu_create(&v, p ? p : &src->params, device, out);
    g_share_buckets_from = nullptr;
    return rc;
}

// RCCL through dlopen: libsnapgpu.so must load on a box without librccl (single-GPU use); the symbols are resolved on first use.
// (the handful of RCCL declarations used, spelled out here instead of #include <rccl/rccl.h>: values as in rccl.h of ROCm 7.2)
#include <dlfcn.h>
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
static const ncclResult_t ncclSuccess = 0;
static const ncclDataType_t ncclUint8 = 1;
struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string &why) {
        if (h) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (h) break; }
        if (!h) { why = std::string("dlopen(librccl): ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart"); GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        Broadcast = (decltype(Broadcast))dlsym(h, "ncclBroadcast"); GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Broadcast || !GetErrorString) { why = "librccl lacks an expected symbol"; return false; }
        return true;
    }
};
static RcclApi g_rccl;
static std::once_flag g_rccl_once; static bool g_rccl_ok = false; static std::string g_rccl_why;      // (load() once, whatever thread gets there first)

// The index of ctxs[0] into the (same-size, so far unfilled) blobs of ctxs[1 .. n): one ncclBroadcast per blob, root 0, all ranks driven
// from this thread inside one group call (ncclCommInitAll communicators: one process, one rank per GPU).  Over xGMI a ring broadcast is
// bound by one link (~153 GB/s peak), i.e. ~0.2-0.3 s for GRCh38's ~30 GB.
extern "C" int snapgpu_broadcast_index(snapgpu_ctx **ctxs, int n)
{
    if (!ctxs || n < 1) return fail(nullptr, SNAPGPU_E_INVALID, "snapgpu_broadcast_index: bad argument");
    for (int i = 0; i < n; i++) if (!ctxs[i]) return fail(nullptr, SNAPGPU_E_INVALID, "snapgpu_broadcast_index: null context");
    snapgpu_ctx *root = ctxs[0];
    const size_t hash_bytes = (size_t)root->view_meta.hash_blob_bytes;
    const size_t ovf_bytes = (size_t)(root->view_meta.overflow_table_size ? root->view_meta.overflow_table_size : 1) * 4;
    const size_t gen_bytes = (size_t)root->view_meta.n_bases + 2 * (size_t)root->view_meta.genome_pad;
    for (int i = 1; i < n; i++) {
        const snapgpu_index_view &a = root->view_meta, &b = ctxs[i]->view_meta;
        if (a.hash_blob_bytes != b.hash_blob_bytes || a.overflow_table_size != b.overflow_table_size || a.n_bases != b.n_bases ||
            a.genome_pad != b.genome_pad || a.seed_len != b.seed_len || a.n_hash_tables != b.n_hash_tables)
            return fail(root, SNAPGPU_E_INVALID, "snapgpu_broadcast_index: context " + std::to_string(i) + " was not created over the same index (snapgpu_create_replica)");
        if (!ctxs[i]->owns_index) return fail(root, SNAPGPU_E_INVALID, "snapgpu_broadcast_index: context " + std::to_string(i) + " shares another context's blobs");
        for (int j = 0; j < i; j++) if (ctxs[j]->device == ctxs[i]->device) return fail(root, SNAPGPU_E_INVALID, "snapgpu_broadcast_index: two contexts on one device");
    }
    if (n == 1) return SNAPGPU_OK;
    std::call_once(g_rccl_once, [] { g_rccl_ok = g_rccl.load(g_rccl_why); });
    if (!g_rccl_ok) return fail(root, SNAPGPU_E_UNSUPPORTED, "snapgpu_broadcast_index: RCCL is not available (" + g_rccl_why + ")");
    std::vector<int> devs((size_t)n);
    struct Comms {                     // destroyed on every way out of this function
        std::vector<ncclComm_t> v;
        ~Comms() { for (auto c : v) if (c) g_rccl.CommDestroy(c); }
    } comms_owner; comms_owner.v.assign((size_t)n, nullptr);
    std::vector<ncclComm_t> &comms = comms_owner.v;
    for (int i = 0; i < n; i++) devs[(size_t)i] = ctxs[i]->device;
#define NCCLCHK(call) do { ncclResult_t _r = (call); if (_r != ncclSuccess) { \
        std::string m = std::string(#call) + ": " + g_rccl.GetErrorString(_r); return fail(root, SNAPGPU_E_LAUNCH, m); } } while (0)
    NCCLCHK(g_rccl.CommInitAll(comms.data(), n, devs.data()));
    struct Blob { void *snapgpu_ctx::*p; size_t bytes; };
    const Blob blobs[3] = {{&snapgpu_ctx::d_hash, hash_bytes}, {&snapgpu_ctx::d_overflow, ovf_bytes}, {&snapgpu_ctx::d_genome_padded, gen_bytes}};
    for (const Blob &b : blobs) {
        NCCLCHK(g_rccl.GroupStart());
        for (int i = 0; i < n; i++) {
            HIPCHK(root, hipSetDevice(ctxs[i]->device), SNAPGPU_E_NODEVICE);
            NCCLCHK(g_rccl.Broadcast(ctxs[i]->*(b.p), ctxs[i]->*(b.p), b.bytes, ncclUint8, 0, comms[(size_t)i], ctxs[i]->stream));
        }
        NCCLCHK(g_rccl.GroupEnd());
    }
    for (int i = 0; i < n; i++) {
        HIPCHK(root, hipSetDevice(ctxs[i]->device), SNAPGPU_E_NODEVICE);
        HIPCHK(root, hipStreamSynchronize(ctxs[i]->stream), SNAPGPU_E_LAUNCH);
    }
    for (auto &c : comms) { if (c) g_rccl.CommDestroy(c); c = nullptr; }
#undef NCCLCHK
    for (int i = 1; i < n; i++) {           // the replicas' own device-native tables, now that their slot arrays are there
        HIPCHK(root, hipSetDevice(ctxs[i]->device), SNAPGPU_E_NODEVICE);
        const int brc = build_buckets(ctxs[i]);
        if (brc != SNAPGPU_OK) return brc;
        const int prc = build_planes(ctxs[i]);
        if (prc != SNAPGPU_OK) return prc;
    }
    return SNAPGPU_OK;
}

// ---- host-side index loader (C++ mirror of snap_amd/index.py; formats in SURVEY.md Appendix B)
static bool read_file(const std::string &path, std::vector<uint8_t> &out, std::string &err) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    fseek(f, 0, SEEK_END); long long n = ftell(f); fseek(f, 0, SEEK_SET);
    out.resize((size_t)n);
    size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    if ((long long)got != n) { err = "short read on " + path; return false; }
    return true;
}

// ---- the streamed loader (round 6): the files go disk / page cache -> page-locked pieces -> HBM without ever being whole in host memory.  The first
// loader read each file into a std::vector (a zero fill and a copy of 31 GB), copied the hash file into one blob (another 27 GB) and handed pageable
// memory to hipMemcpy: 15.7 s for the GRCh38-scale directory (2 GB/s) -- five times what the GPU takes to BUILD that index.  Here a pool of threads each
// owns one page-locked piece and a stream: pread a slice of a file into the piece, hipMemcpyAsync it to its place in the device arrays, next slice.  The
// hash file's tables land where the blob layout wants them (the 32 + valueSize header bytes of each table are skipped: HashTable.cpp:98-175).
// Directories with 5 .. 8-byte locations take the first loader (their slots are narrowed on the host).
struct LoadJob { int fd; uint64_t file_off; uint8_t *dev; uint64_t bytes; };

static int stream_files_to_device(const std::vector<LoadJob> &jobs, int device, std::string &err)
{
    const uint64_t PIECE = 64ull << 20;
    struct Slice { int fd; uint64_t off; uint8_t *dev; uint64_t n; };
    std::vector<Slice> slices;
    for (const LoadJob &j : jobs)
        for (uint64_t o = 0; o < j.bytes; o += PIECE) slices.push_back(Slice{j.fd, j.file_off + o, j.dev + o, j.bytes - o < PIECE ? j.bytes - o : PIECE});
    int n_threads = 12;
    if (const char *e = getenv("SNAPGPU_LOAD_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) n_threads = v; }
    uint64_t total_bytes = 0, largest = 0;
    for (const Slice &sl : slices) { total_bytes += sl.n; if (sl.n > largest) largest = sl.n; }
    {   // a small directory: no more threads (and page-locked pieces) than it has 64 MB of data for, pieces no larger than its largest slice
        const uint64_t by_size = (total_bytes + PIECE - 1) / PIECE;
        if ((uint64_t)n_threads > by_size) n_threads = (int)(by_size ? by_size : 1);
    }
    const size_t piece_bytes = (size_t)(((largest ? largest : 1) + 4095) & ~(uint64_t)4095);
    const bool want_pin = !(getenv("SNAPGPU_LOAD_PIN") && atoi(getenv("SNAPGPU_LOAD_PIN")) == 0);
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    std::mutex err_mu;
    auto worker = [&]() {
        if (hipSetDevice(device) != hipSuccess) { failed = 1; return; }
        void *buf = nullptr;
        hipStream_t st = nullptr;
        bool pinned = false;
        if (posix_memalign(&buf, 4096, piece_bytes) != 0 || !buf) { failed = 1; return; }
        pinned = want_pin && hipHostRegister(buf, piece_bytes, 0) == hipSuccess;   // (not page-locked: the copy still works, slower)
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { failed = 1; if (pinned) (void)hipHostUnregister(buf); free(buf); return; }
        while (!failed.load()) {
            const size_t k = next.fetch_add(1);
            if (k >= slices.size()) break;
            const Slice &s = slices[k];
            uint64_t got = 0;
            while (got < s.n) {
                const ssize_t r = pread(s.fd, (uint8_t *)buf + got, (size_t)(s.n - got), (off_t)(s.off + got));
                if (r <= 0) { std::lock_guard<std::mutex> g(err_mu); err = "short read while loading the index directory"; failed = 1; break; }
                got += (uint64_t)r;
            }
            if (failed.load()) break;
            if (hipMemcpyAsync(s.dev, buf, (size_t)s.n, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                std::lock_guard<std::mutex> g(err_mu); err = "host-to-device copy failed while loading the index directory"; failed = 1; break;
            }
        }
        (void)hipStreamDestroy(st);
        if (pinned) (void)hipHostUnregister(buf);
        free(buf);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) th.emplace_back(worker);
    for (auto &t : th) t.join();
    return failed.load() ? SNAPGPU_E_INVALID : SNAPGPU_OK;
}

// returns SNAPGPU_OK with *out set, an error, or -1000 = "not this loader's case" (wide locations: the caller falls back)
static int create_from_directory_streamed(const std::string &dir, const snapgpu_params *p, int device, snapgpu_ctx **out)
{
    std::string err;
    std::vector<uint8_t> hdr;
    if (!read_file(dir + "/GenomeIndex", hdr, err)) return fail(nullptr, SNAPGPU_E_INVALID, err);
    hdr.push_back(0);
    int major = 0, minor = 0, n_tables = 0, seed_len = 0, padding = 0, key_bytes = 0, small = 0, loc_size = 0;
    long long overflow_size = 0, hash_file_size = 0;
    if (sscanf((const char *)hdr.data(), "%d %d %d %lld %d %d %d %lld %d %d", &major, &minor, &n_tables, &overflow_size, &seed_len,
               &padding, &key_bytes, &hash_file_size, &small, &loc_size) != 10)               // GenomeIndex.cpp:1879
        return fail(nullptr, SNAPGPU_E_INVALID, "malformed GenomeIndex header");
    if (major != 7 || loc_size != 4 || n_tables <= 0) return -1000;
    struct Fd { int fd = -1; ~Fd() { if (fd >= 0) close(fd); } } fg, fo, fh;
    fg.fd = open((dir + "/Genome").c_str(), O_RDONLY); fo.fd = open((dir + "/OverflowTable").c_str(), O_RDONLY); fh.fd = open((dir + "/GenomeIndexHash").c_str(), O_RDONLY);
    if (fg.fd < 0 || fo.fd < 0 || fh.fd < 0) return fail(nullptr, SNAPGPU_E_INVALID, "cannot open the files of " + dir);
    auto size_of = [](int fd) -> long long { struct stat sb; return fstat(fd, &sb) == 0 ? (long long)sb.st_size : -1; };
    const long long gen_size = size_of(fg.fd), ovf_size = size_of(fo.fd), hash_size = size_of(fh.fd);
    if (ovf_size != overflow_size * 4) return fail(nullptr, SNAPGPU_E_INVALID, "OverflowTable size does not match the header");

    // Genome: "nBases nContigs flags\n", one line per contig, then nBases raw bytes (Genome.cpp:203-229): the head is read until its lines are all there
    std::vector<uint8_t> head;
    size_t pos = 0;
    long long n_bases = 0; int n_contigs = 0, gflags = 0;
    std::vector<uint64_t> contig_begin, proj_begin; std::vector<uint8_t> proj_rc; std::vector<uint32_t> cigar_start, cigar_ops;
    uint64_t first_alt = ~0ull >> 2;
    for (size_t want = 1u << 20;; want *= 4) {
        const size_t n = (long long)want < gen_size ? want : (size_t)gen_size;
        head.resize(n);
        size_t got = 0;
        while (got < n) { const ssize_t r = pread(fg.fd, head.data() + got, n - got, (off_t)got); if (r <= 0) return fail(nullptr, SNAPGPU_E_INVALID, "short read on Genome"); got += (size_t)r; }
        pos = 0;
        bool complete = true;
        auto next_line = [&](std::string &line) -> bool {
            size_t e = pos; while (e < head.size() && head[e] != '\n') e++;
            if (e >= head.size()) return false;
            line.assign((const char *)head.data() + pos, e - pos); pos = e + 1; return true;
        };
        std::string line;
        if (!next_line(line)) { if ((long long)n == gen_size) return fail(nullptr, SNAPGPU_E_INVALID, "malformed Genome header"); continue; }
        if (sscanf(line.c_str(), "%lld %d %d", &n_bases, &n_contigs, &gflags) < 2 || n_contigs < 0) return fail(nullptr, SNAPGPU_E_INVALID, "malformed Genome header");
        contig_begin.assign((size_t)n_contigs, 0); proj_begin.assign((size_t)n_contigs, 0); proj_rc.assign((size_t)n_contigs, 0);
        cigar_start.assign((size_t)n_contigs + 1, 0); cigar_ops.clear(); first_alt = ~0ull >> 2;
        for (int i = 0; i < n_contigs; i++) {
            long long begin = 0, pbegin = 0; int cflags = 0, orig = 0, pflags = 0, name_len = 0, cigar_len = 0, consumed = 0;
            if (!next_line(line)) { complete = false; break; }
            if (sscanf(line.c_str(), "%lld %x %d %lld %x %d %d %n", &begin, &cflags, &orig, &pbegin, &pflags, &name_len, &cigar_len, &consumed) < 7)
                return fail(nullptr, SNAPGPU_E_INVALID, "malformed contig line in Genome");
            contig_begin[(size_t)i] = (uint64_t)begin; proj_begin[(size_t)i] = (uint64_t)pbegin; proj_rc[(size_t)i] = (uint8_t)(pflags & 1);
            if ((cflags & 1) && (uint64_t)begin < first_alt) first_alt = (uint64_t)begin;
            const size_t cpos = (size_t)consumed + (size_t)name_len + 1;
            if (cpos <= line.size()) {
                const char *c = line.c_str() + cpos, *cend = line.c_str() + line.size();
                while (c < cend) {
                    int count = 0, used = 0; char act = 0;
                    if (sscanf(c, "%d%c%n", &count, &act, &used) != 2) break;
                    cigar_ops.push_back(((uint32_t)count << 8) | (uint32_t)(uint8_t)act);
                    c += used;
                }
            }
            cigar_start[(size_t)i + 1] = (uint32_t)cigar_ops.size();
        }
        if (complete) break;
        if ((long long)n == gen_size) return fail(nullptr, SNAPGPU_E_INVALID, "Genome file truncated");
    }
    if (cigar_ops.empty()) cigar_ops.push_back(0);
    if (gen_size - (long long)pos < n_bases) return fail(nullptr, SNAPGPU_E_INVALID, "Genome file truncated");
    std::vector<uint8_t>().swap(head);

    // GenomeIndexHash: per table a 32 + valueSize byte header, then tableSize slots
    const uint32_t value_count = small ? 1 : 2, entry = 4 * value_count + (uint32_t)key_bytes;
    std::vector<uint64_t> toff((size_t)n_tables), tsz((size_t)n_tables), fpos((size_t)n_tables);
    uint64_t hp = 0, blob_bytes = 0;
    for (int t = 0; t < n_tables; t++) {
        uint8_t h36[36];
        if ((long long)hp + 36 > hash_size || pread(fh.fd, h36, 36, (off_t)hp) != 36) return fail(nullptr, SNAPGPU_E_INVALID, "GenomeIndexHash truncated");
        uint32_t magic, ks, vs, vc; uint64_t table_size;
        memcpy(&magic, h36, 4); memcpy(&table_size, h36 + 4, 8); memcpy(&ks, h36 + 20, 4); memcpy(&vs, h36 + 24, 4); memcpy(&vc, h36 + 28, 4);
        if (magic != 0xb111b010u || ks != (uint32_t)key_bytes || vs != 4u || vc != value_count)
            return fail(nullptr, SNAPGPU_E_INVALID, "hash table header does not match the index header");
        hp += 32 + vs;
        const uint64_t nbytes = table_size * entry;
        if ((long long)(hp + nbytes) > hash_size) return fail(nullptr, SNAPGPU_E_INVALID, "GenomeIndexHash truncated");
        toff[(size_t)t] = blob_bytes; tsz[(size_t)t] = table_size; fpos[(size_t)t] = hp;
        blob_bytes += nbytes; hp += nbytes;
    }
    blob_bytes += 16;

    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, SNAPGPU_E_NODEVICE, "no such HIP device");
    const uint32_t pad = 1024;
    const size_t genome_total = (size_t)n_bases + 2 * (size_t)pad;
    const size_t overflow_words = overflow_size ? (size_t)overflow_size : 1;
    uint8_t *d_hash = nullptr, *d_ovf = nullptr, *d_gen = nullptr;
    auto drop = [&]() { if (d_hash) (void)hipFree(d_hash); if (d_ovf) (void)hipFree(d_ovf); if (d_gen) (void)hipFree(d_gen); };
    if (hipMalloc((void **)&d_hash, blob_bytes + 16) != hipSuccess || hipMalloc((void **)&d_ovf, overflow_words * 4 + 16) != hipSuccess ||
        hipMalloc((void **)&d_gen, genome_total + 16) != hipSuccess) { drop(); return fail(nullptr, SNAPGPU_E_NOMEM, "out of device memory for the index"); }
    if (hipMemset(d_gen, 'n', pad) != hipSuccess || hipMemset(d_gen + pad + (size_t)n_bases, 'n', pad + 16) != hipSuccess ||
        hipMemset(d_hash + blob_bytes - 16, 0, 32) != hipSuccess || hipMemset(d_ovf, 0, overflow_words * 4 + 16 < 64 ? overflow_words * 4 + 16 : 64) != hipSuccess ||
        hipMemset(d_ovf + overflow_words * 4, 0, 16) != hipSuccess) { drop(); return fail(nullptr, SNAPGPU_E_NODEVICE, "hipMemset failed"); }
    std::vector<LoadJob> jobs;
    for (int t = 0; t < n_tables; t++) if (tsz[(size_t)t]) jobs.push_back(LoadJob{fh.fd, fpos[(size_t)t], d_hash + toff[(size_t)t], tsz[(size_t)t] * entry});
    if (n_bases) jobs.push_back(LoadJob{fg.fd, (uint64_t)pos, d_gen + pad, (uint64_t)n_bases});
    if (overflow_size) jobs.push_back(LoadJob{fo.fd, 0, d_ovf, (uint64_t)overflow_size * 4});
    const int src = stream_files_to_device(jobs, device, err);
    if (src != SNAPGPU_OK) { drop(); return fail(nullptr, src, err); }

    snapgpu_index_view v; memset(&v, 0, sizeof(v));
    v.seed_len = (uint32_t)seed_len; v.key_bytes = (uint32_t)key_bytes; v.n_hash_tables = (uint32_t)n_tables;
    v.large_hash_table = small ? 0 : 1; v.location_size = 4; v.chromosome_padding = (uint32_t)padding;
    v.overflow_table_size = (uint64_t)overflow_size;
    v.hash_blob = d_hash; v.hash_blob_bytes = blob_bytes; v.table_offset = toff.data(); v.table_size = tsz.data();
    v.overflow = (const uint32_t *)d_ovf; v.genome = d_gen + pad; v.n_bases = (uint64_t)n_bases;
    v.genome_pad = pad; v.contig_begin = contig_begin.data(); v.n_contigs = (uint32_t)n_contigs;
    v.first_alt_location = first_alt; v.on_device = 1;
    v.contig_proj_begin = proj_begin.data(); v.contig_proj_rc = proj_rc.data(); v.contig_cigar_start = cigar_start.data(); v.cigar_ops = cigar_ops.data();
    const int rc = snapgpu_create(&v, p, device, out);
    if (rc != SNAPGPU_OK) { drop(); return rc; }
    (*out)->owns_index = true;             // (adopted device arrays that nobody else holds: this context frees them)
    return SNAPGPU_OK;
}

extern "C" int snapgpu_create_from_directory(const char *index_dir, const snapgpu_params *p, int device, snapgpu_ctx **out)
{
    if (!index_dir || !p || !out) return fail(nullptr, SNAPGPU_E_INVALID, "snapgpu_create_from_directory: null argument");
    std::string dir(index_dir), err;
    if (!getenv("SNAPGPU_LOAD_CLASSIC")) {              // (the first loader: whole files through host vectors; still what wide-location directories take)
        const int src = create_from_directory_streamed(dir, p, device, out);
        if (src != -1000) return src;
    }
    std::vector<uint8_t> hdr, gen, ovf, hash;
    if (!read_file(dir + "/GenomeIndex", hdr, err) || !read_file(dir + "/Genome", gen, err) ||
        !read_file(dir + "/OverflowTable", ovf, err) || !read_file(dir + "/GenomeIndexHash", hash, err))
        return fail(nullptr, SNAPGPU_E_INVALID, err);
    hdr.push_back(0);
    int major = 0, minor = 0, n_tables = 0, seed_len = 0, padding = 0, key_bytes = 0, small = 0, loc_size = 0;
    long long overflow_size = 0, hash_file_size = 0;
    if (sscanf((const char *)hdr.data(), "%d %d %d %lld %d %d %d %lld %d %d", &major, &minor, &n_tables, &overflow_size, &seed_len,
               &padding, &key_bytes, &hash_file_size, &small, &loc_size) != 10)               // GenomeIndex.cpp:1879
        return fail(nullptr, SNAPGPU_E_INVALID, "malformed GenomeIndex header");
    if (major != 7) return fail(nullptr, SNAPGPU_E_UNSUPPORTED, "index major version is not 7 (GenomeIndex.h:170)");
    // locationSize 5 .. 8 (GenomeIndex.cpp:446-453: what the indexer picks for seeds shorter than 20, or -locationSize): the reference then
    // goes through lookupSeed / overflowTable64 (GenomeIndex.cpp:2205-2328), whose answers are those of lookupSeed32 over the same tables
    // with wider values.  Where every value fits 32 bits -- genome + overflow table below 2^32 - 2 entries -- the tables are NARROWED here,
    // slot for slot (same table sizes, same keys, so the same probe sequences), and the device keeps its one 32-bit layout; a genome that
    // needs the wider values is refused.
    if (loc_size < 4 || loc_size > 8) return fail(nullptr, SNAPGPU_E_INVALID, "GenomeIndex header: location size out of range");
    const bool wide = loc_size > 4;
    if ((long long)ovf.size() != overflow_size * (wide ? 8 : 4)) return fail(nullptr, SNAPGPU_E_INVALID, "OverflowTable size does not match the header");

    // Genome: "nBases nContigs flags\n", one line per contig, then nBases raw bytes (Genome.cpp:203-229)
    size_t pos = 0;
    auto next_line = [&](std::string &line) -> bool {
        size_t e = pos; while (e < gen.size() && gen[e] != '\n') e++;
        if (e >= gen.size()) return false;
        line.assign((const char *)gen.data() + pos, e - pos); pos = e + 1; return true;
    };
    std::string line;
    long long n_bases = 0; int n_contigs = 0, gflags = 0;
    if (!next_line(line) || sscanf(line.c_str(), "%lld %d %d", &n_bases, &n_contigs, &gflags) < 2)
        return fail(nullptr, SNAPGPU_E_INVALID, "malformed Genome header");
    std::vector<uint64_t> contig_begin((size_t)n_contigs), proj_begin((size_t)n_contigs);
    std::vector<uint8_t> proj_rc((size_t)n_contigs);
    std::vector<uint32_t> cigar_start((size_t)n_contigs + 1, 0), cigar_ops;
    uint64_t first_alt = ~0ull >> 2;
    for (int i = 0; i < n_contigs; i++) {
        // "begin flags originalNumber projBegin projFlags nameLen cigarLen name cigar" (Genome.cpp:226, 353-403)
        long long begin = 0, pbegin = 0; int cflags = 0, orig = 0, pflags = 0, name_len = 0, cigar_len = 0, consumed = 0;
        if (!next_line(line) || sscanf(line.c_str(), "%lld %x %d %lld %x %d %d %n", &begin, &cflags, &orig, &pbegin, &pflags, &name_len, &cigar_len, &consumed) < 7)
            return fail(nullptr, SNAPGPU_E_INVALID, "malformed contig line in Genome");
        contig_begin[(size_t)i] = (uint64_t)begin;
        proj_begin[(size_t)i] = (uint64_t)pbegin;
        proj_rc[(size_t)i] = (uint8_t)(pflags & 1);                                             // GENOME_FLAG_ALT_PROJ_CONTIG_IS_RC
        if ((cflags & 1) && (uint64_t)begin < first_alt) first_alt = (uint64_t)begin;           // GENOME_FLAG_CONTIG_IS_ALT
        size_t cpos = (size_t)consumed + (size_t)name_len + 1;
        if (cpos <= line.size()) {
            const char *c = line.c_str() + cpos, *cend = line.c_str() + line.size();
            while (c < cend) {                                                                  // repeated sscanf("%d%c")
                int count = 0, used = 0; char act = 0;
                if (sscanf(c, "%d%c%n", &count, &act, &used) != 2) break;
                cigar_ops.push_back(((uint32_t)count << 8) | (uint32_t)(uint8_t)act);
                c += used;
            }
        }
        cigar_start[(size_t)i + 1] = (uint32_t)cigar_ops.size();
    }
    if (cigar_ops.empty()) cigar_ops.push_back(0);
    if (gen.size() - pos < (size_t)n_bases) return fail(nullptr, SNAPGPU_E_INVALID, "Genome file truncated");
    const uint32_t pad = 1024;
    std::vector<uint8_t> genome_padded((size_t)n_bases + 2 * pad, (uint8_t)'n');
    memcpy(genome_padded.data() + pad, gen.data() + pos, (size_t)n_bases);
    std::vector<uint8_t>().swap(gen);

    if (wide) {
        if ((unsigned long long)n_bases + (unsigned long long)overflow_size >= 0xfffffffeull)
            return fail(nullptr, SNAPGPU_E_UNSUPPORTED, "index uses 64-bit genome locations and its genome + overflow table do not fit 32-bit values: only indexes whose values fit are supported (they are narrowed on load)");
        std::vector<uint8_t> narrow((size_t)overflow_size * 4);
        for (long long i = 0; i < overflow_size; i++) {
            uint64_t v; memcpy(&v, &ovf[(size_t)i * 8], 8);
            if (v > 0xffffffffull) return fail(nullptr, SNAPGPU_E_UNSUPPORTED, "OverflowTable entry does not fit 32 bits");
            const uint32_t w = (uint32_t)v; memcpy(&narrow[(size_t)i * 4], &w, 4);
        }
        ovf.swap(narrow);
    }

    // GenomeIndexHash: per table a 36-byte header then tableSize slots (HashTable.cpp:98-175)
    const uint32_t value_count = small ? 1 : 2, entry = 4 * value_count + (uint32_t)key_bytes;
    const uint32_t src_entry = (uint32_t)loc_size * value_count + (uint32_t)key_bytes;
    const uint64_t all_ones = loc_size == 8 ? ~0ull : ((1ull << (8 * loc_size)) - 1ull);
    std::vector<uint64_t> toff((size_t)n_tables), tsz((size_t)n_tables);
    std::vector<uint8_t> blob; blob.reserve(hash.size());
    size_t hp = 0;
    for (int t = 0; t < n_tables; t++) {
        if (hp + 36 > hash.size()) return fail(nullptr, SNAPGPU_E_INVALID, "GenomeIndexHash truncated");
        uint32_t magic, ks, vs, vc; uint64_t table_size;
        memcpy(&magic, &hash[hp], 4); memcpy(&table_size, &hash[hp + 4], 8);
        memcpy(&ks, &hash[hp + 20], 4); memcpy(&vs, &hash[hp + 24], 4); memcpy(&vc, &hash[hp + 28], 4);
        if (magic != 0xb111b010u || ks != (uint32_t)key_bytes || vs != (uint32_t)loc_size || vc != value_count)
            return fail(nullptr, SNAPGPU_E_INVALID, "hash table header does not match the index header");
        hp += 32 + vs;
        size_t nbytes = (size_t)table_size * src_entry;
        if (hp + nbytes > hash.size()) return fail(nullptr, SNAPGPU_E_INVALID, "GenomeIndexHash truncated");
        toff[(size_t)t] = blob.size(); tsz[(size_t)t] = table_size;
        if (!wide) blob.insert(blob.end(), hash.begin() + (long)hp, hash.begin() + (long)(hp + nbytes));
        else {              // slot for slot: [value x value_count][key] with values of loc_size bytes -> 4 bytes (all ones = unused, all ones - 1 = "the other strand only")
            const size_t at = blob.size();
            blob.resize(at + (size_t)table_size * entry);
            for (uint64_t sl = 0; sl < table_size; sl++) {
                const uint8_t *src = &hash[hp + (size_t)sl * src_entry];
                uint8_t *dst = &blob[at + (size_t)sl * entry];
                for (uint32_t k = 0; k < value_count; k++) {
                    uint64_t v = 0; memcpy(&v, src + (size_t)k * (size_t)loc_size, (size_t)loc_size);
                    uint32_t w;
                    if (v == all_ones) w = 0xffffffffu;
                    else if (v == all_ones - 1) w = 0xfffffffeu;
                    else if (v < 0xfffffffeull) w = (uint32_t)v;
                    else return fail(nullptr, SNAPGPU_E_UNSUPPORTED, "hash table value does not fit 32 bits");
                    memcpy(dst + 4 * (size_t)k, &w, 4);
                }
                memcpy(dst + 4 * (size_t)value_count, src + (size_t)loc_size * value_count, (size_t)key_bytes);
            }
        }
        hp += nbytes;
    }
    std::vector<uint8_t>().swap(hash);
    blob.resize(blob.size() + 16, 0);
    if (ovf.empty()) ovf.resize(4, 0);

    snapgpu_index_view v; memset(&v, 0, sizeof(v));
    v.seed_len = (uint32_t)seed_len; v.key_bytes = (uint32_t)key_bytes; v.n_hash_tables = (uint32_t)n_tables;
    v.large_hash_table = small ? 0 : 1; v.location_size = 4 /* (narrowed above where the files have wider values) */; v.chromosome_padding = (uint32_t)padding;
    v.overflow_table_size = (uint64_t)overflow_size;
    v.hash_blob = blob.data(); v.hash_blob_bytes = blob.size(); v.table_offset = toff.data(); v.table_size = tsz.data();
    v.overflow = (const uint32_t *)ovf.data(); v.genome = genome_padded.data() + pad; v.n_bases = (uint64_t)n_bases;
    v.genome_pad = pad; v.contig_begin = contig_begin.data(); v.n_contigs = (uint32_t)n_contigs;
    v.first_alt_location = first_alt; v.on_device = 0;
    v.contig_proj_begin = proj_begin.data(); v.contig_proj_rc = proj_rc.data(); v.contig_cigar_start = cigar_start.data(); v.cigar_ops = cigar_ops.data();
    return snapgpu_create(&v, p, device, out);
}

extern "C" int snapgpu_index_device_ptrs(snapgpu_ctx *ctx, void **hash_blob, void **overflow, void **genome_with_pad) {
    if (!ctx) return SNAPGPU_E_INVALID;
    if (hash_blob) *hash_blob = ctx->d_hash;
    if (overflow) *overflow = ctx->d_overflow;
    if (genome_with_pad) *genome_with_pad = ctx->d_genome_padded;
    return SNAPGPU_OK;
}

// the index-probe kernel: sixteen probes per wave pass (lookup16.h) for the shape the north star uses, k_lookup_seeds for every other
static void launch_lookup(snapgpu_ctx *ctx, uint32_t n, const void *d_seeds, void *d_n_hits, void *d_hits, uint32_t max_hits_out,
                          unsigned long long *d_counters, hipStream_t s)
{
    const uint32_t maxb = (uint32_t)ctx->num_cus * 8;                                   // 32 waves per CU
    if (ctx->ix.bucket_blob && ctx->ix.seed_len == 20 && ctx->ix.key_bytes == 4 && ((uintptr_t)d_seeds & 3) == 0 && !getenv("SNAPGPU_LOOKUP8")) {
        uint32_t blocks = (n + 31) / 32; if (blocks > maxb) blocks = maxb;
        // as many blocks as are resident at once (see the kernel): every wave then does an equal share of the passes from the start
        // (per context: contexts live on different devices and are driven by different threads)
        if (ctx->lookup_blocks_per_cu == 0) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_lookup_seeds20, 256, 0) != hipSuccess || nb <= 0) nb = 4;
            if (const char *e = getenv("SNAPGPU_LOOKUP_BLOCKS_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 8) nb = v; }
            ctx->lookup_blocks_per_cu = nb > 8 ? 8 : nb;
        }
        const uint32_t fit = (uint32_t)ctx->num_cus * (uint32_t)ctx->lookup_blocks_per_cu;
        if (blocks > fit) blocks = fit;
        hipLaunchKernelGGL(k_lookup_seeds20, dim3(blocks), dim3(256), 0, s, ctx->ix, n, (const uint8_t *)d_seeds, (long long *)d_n_hits,
                           (uint32_t *)d_hits, max_hits_out, d_counters);
    } else {
        uint32_t blocks = (n + 3) / 4; if (blocks > maxb) blocks = maxb;
        hipLaunchKernelGGL(k_lookup_seeds, dim3(blocks), dim3(256), 0, s, ctx->ix, n, (const uint8_t *)d_seeds, (long long *)d_n_hits,
                           (uint32_t *)d_hits, max_hits_out, d_counters);
    }
}

extern "C" int snapgpu_lookup_seeds(snapgpu_ctx *ctx, uint32_t n, const char *seeds, int64_t *n_hits,
                                    uint32_t *hits, uint32_t max_hits_out)
{
    if (!ctx || !seeds || !n_hits || !hits || max_hits_out == 0) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_lookup_seeds: bad argument");
    if (n == 0) return SNAPGPU_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    size_t sb = (size_t)n * ctx->ix.seed_len, nb = (size_t)n * 2 * 8, hb = (size_t)n * 2 * max_hits_out * 4;
    int rc;
    if ((rc = ensure_stage(ctx, 0, sb)) || (rc = ensure_stage(ctx, 1, nb)) || (rc = ensure_stage(ctx, 2, hb))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage[0], seeds, sb, hipMemcpyHostToDevice, ctx->stream), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemsetAsync(ctx->d_stage[2], 0, hb, ctx->stream), SNAPGPU_E_LAUNCH);
    launch_lookup(ctx, n, ctx->d_stage[0], ctx->d_stage[1], ctx->d_stage[2], max_hits_out, nullptr, ctx->stream);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(n_hits, ctx->d_stage[1], nb, hipMemcpyDeviceToHost, ctx->stream), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(hits, ctx->d_stage[2], hb, hipMemcpyDeviceToHost, ctx->stream), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream), SNAPGPU_E_LAUNCH);
    return SNAPGPU_OK;
}

static int finish_timing(snapgpu_ctx *ctx);
// Device-pointer form of snapgpu_lookup_seeds: seeds, hit counts and (optionally) hits already in HBM.  d_hits == NULL: counts only
// (the lists are read, not stored).  Timed with hipEvents like the align kernels (snapgpu_kernel_time) and counted into
// snapgpu_counters (lookups, slots, hits, overflow lists): the index-probe kernel on its own, for its HBM roofline (bench.py).
extern "C" int snapgpu_lookup_seeds_device(snapgpu_ctx *ctx, uint32_t n, const void *d_seeds, void *d_n_hits, void *d_hits,
                                           uint32_t max_hits_out, void *stream)
{
    if (!ctx || !d_seeds || !d_n_hits || max_hits_out == 0) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_lookup_seeds_device: bad argument");
    if (n == 0) return SNAPGPU_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s), SNAPGPU_E_LAUNCH);
    launch_lookup(ctx, n, d_seeds, d_n_hits, d_hits, max_hits_out, ctx->d_counters, s);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s), SNAPGPU_E_LAUNCH);
    if (!stream) return finish_timing(ctx);
    return SNAPGPU_OK;
}

// small RAII helper for per-call device buffers of the (test-oriented) batch primitives
// Device buffers of the host-pointer entry points.  Until round 4 each call hipMalloc'ed its buffers and hipFree'd them on return -- fifteen
// of them in snapgpu_sam_fields_single, half a gigabyte of per-wave scratch among them, and every hipFree synchronises the device: at 65 536
// reads per batch snapgpu-sam spent ~200 ms per batch there against ~25 ms of kernels (profiles/r04j: 0.58 M reads/s end to end).  Now a
// context keeps what it has allocated: a buffer goes back to the context's pool when the call returns and the next call of a similar size
// takes it again.  (Calls on one context are serial, and every entry point synchronises its stream before it returns.)
static thread_local DevPool *t_pool = nullptr;
// blocks per CU of the SAM-field kernels' persistent grids (SNAPGPU_SAMF_BLOCKS_PER_CU: measurement knob; default 8 = the kernels' launch bounds)
static uint32_t samf_blocks_per_cu() { static int v = 0; if (!v) { const char *e = getenv("SNAPGPU_SAMF_BLOCKS_PER_CU"); v = e ? atoi(e) : 8; if (v < 1 || v > 8) v = 8; } return (uint32_t)v; }       // the pool of the context whose entry point this thread is in
struct PoolScope {
    DevPool *prev;
    explicit PoolScope(DevPool *p) : prev(t_pool) { t_pool = p; }
    ~PoolScope() { t_pool = prev; }
};
struct DevBuf {
    void *p = nullptr;
    DevPool *from = nullptr;
    ~DevBuf() { if (p) { if (from) from->release(p); else (void)hipFree(p); } }
    hipError_t put(const void *src, size_t bytes, hipStream_t s, size_t slack = 0) {      // slack: bytes allocated beyond what is copied
        hipError_t e;
        if (t_pool) { from = t_pool; p = from->acquire(bytes + slack ? bytes + slack : 16, &e); }
        else e = hipMalloc(&p, bytes + slack ? bytes + slack : 16);
        if (e != hipSuccess) return e;
        if (src && bytes) return hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, s);
        return hipSuccess;
    }
};

extern "C" int snapgpu_landau_vishkin(snapgpu_ctx *ctx, int dir, uint32_t n,
                                      const char *texts, uint64_t texts_bytes, const uint32_t *text_off, const int32_t *text_len,
                                      const char *patterns, const char *quals, uint64_t patterns_bytes,
                                      const uint32_t *pat_off, const int32_t *pat_len, const int32_t *k,
                                      int32_t *score, double *match_probability, int32_t *net_indel,
                                      int32_t *total_indels, int32_t *text_span)
{
    if (!ctx || (dir != 1 && dir != -1)) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_landau_vishkin: bad argument");
    if (n == 0) return SNAPGPU_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    uint32_t kmax = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (pat_len[i] < 0 || pat_len[i] > 1000) return fail(ctx, SNAPGPU_E_INVALID, "pattern length out of range");
        int kk = k[i] > 126 ? 126 : k[i];
        if (kk > (int)kmax) kmax = (uint32_t)kk;
    }
    hipStream_t s = ctx->stream;
    PoolScope pool_scope(&ctx->pool);
    DevBuf dt, dto, dtl, dp, dq, dpo, dpl, dk, ds, dpr, dni, dti, dts;
    HIPCHK(ctx, dt.put(texts, texts_bytes, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dto.put(text_off, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dtl.put(text_len, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dp.put(patterns, patterns_bytes, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dq.put(quals, patterns_bytes, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dpo.put(pat_off, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dpl.put(pat_len, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dk.put(k, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, ds.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dpr.put(nullptr, (size_t)n * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dni.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dti.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dts.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    LVBatchArgs a;
    a.dir = dir; a.n = n; a.kmax = kmax;
    uint32_t pcap = 64;
    for (uint32_t i = 0; i < n; i++) if ((uint32_t)pat_len[i] > pcap) pcap = (uint32_t)pat_len[i];
    a.pcap = (pcap + 63) & ~63u;
    a.texts = (const uint8_t *)dt.p; a.text_off = (const uint32_t *)dto.p; a.text_len = (const int32_t *)dtl.p;
    a.patterns = (const uint8_t *)dp.p; a.quals = (const uint8_t *)dq.p; a.pat_off = (const uint32_t *)dpo.p;
    a.pat_len = (const int32_t *)dpl.p; a.k = (const int32_t *)dk.p;
    a.score = (int32_t *)ds.p; a.prob = (double *)dpr.p; a.net_indel = (int32_t *)dni.p;
    a.total_indels = (int32_t *)dti.p; a.text_span = (int32_t *)dts.p; a.tab = ctx->d_tab;
    uint32_t per_wave = (lv_lds_bytes(kmax, a.pcap) + 15) & ~15u;
    uint32_t waves_per_block = 4;
    while (waves_per_block > 1 && (size_t)waves_per_block * per_wave > 64 * 1024) waves_per_block >>= 1;
    if ((size_t)waves_per_block * per_wave > 64 * 1024) return fail(ctx, SNAPGPU_E_UNSUPPORTED, "k too large for the LDS triangle");
    uint32_t blocks = (n + waves_per_block - 1) / waves_per_block; uint32_t maxb = (uint32_t)ctx->num_cus * 8; if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(k_lv_batch, dim3(blocks), dim3(64 * waves_per_block), waves_per_block * per_wave, s, a);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(score, ds.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(match_probability, dpr.p, (size_t)n * 8, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(net_indel, dni.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(total_indels, dti.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(text_span, dts.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    return SNAPGPU_OK;
}

// kernel time of the SAM-side launches, into the same accumulator snapgpu_kernel_time reads (events recorded around the launch)
static int sam_side_kernel_time(snapgpu_ctx *ctx)
{
    float ms = 0.f;
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1), SNAPGPU_E_LAUNCH);
    ctx->kernel_ms += ms; ctx->kernel_launches++;
    return SNAPGPU_OK;
}

// SAMFormat::computeCigar, Landau-Vishkin variant, for a batch of written reads (cigar_lv.h, cigar_k.hip).
extern "C" int snapgpu_compute_cigar_lv(snapgpu_ctx *ctx, uint32_t n, const char *data, uint64_t data_bytes, const uint64_t *off,
                                        const int32_t *len, const int64_t *loc, const int32_t *extra_before, int use_m,
                                        uint32_t *ops, uint32_t ops_stride, int32_t *n_ops, int32_t *edit_distance,
                                        int32_t *add_front_clipping, int64_t *extra_clipped_after)
{
    if (!ctx || (n && (!data || !off || !len || !loc || !extra_before || !ops || !n_ops || !edit_distance || !add_front_clipping || !extra_clipped_after)))
        return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_compute_cigar_lv: null argument");
    if (ops_stride == 0) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_compute_cigar_lv: ops_stride must be positive");
    if (n == 0) return SNAPGPU_OK;
    uint32_t RL = 64;
    for (uint32_t i = 0; i < n; i++) {
        if (len[i] < 0 || len[i] > 4000) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_compute_cigar_lv: read length out of range");
        if (extra_before[i] < 0 || extra_before[i] > len[i]) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_compute_cigar_lv: extra_before out of range");
        if (off[i] + (uint64_t)len[i] > data_bytes) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_compute_cigar_lv: read outside the data buffer");
        if (loc[i] < 0 || (uint64_t)loc[i] >= ctx->ix.n_bases) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_compute_cigar_lv: location outside the genome");
        if ((uint32_t)len[i] > RL) RL = (uint32_t)len[i];
    }
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    hipStream_t s = ctx->stream;
    const uint32_t per_wave = lvc_lds_bytes(RL);
    uint32_t blocks = (uint32_t)ctx->num_cus * 4;                    // persistent grid: 16 waves per CU
    const uint32_t need = (n + 3) / 4; if (blocks > need) blocks = need;
    PoolScope pool_scope(&ctx->pool);
    DevBuf dd, doff, dlen, dloc, dxb, dscr, dops, dno, ded, dafc, dxa;
    HIPCHK(ctx, dd.put(data, data_bytes, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, doff.put(off, (size_t)n * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dlen.put(len, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dloc.put(loc, (size_t)n * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dxb.put(extra_before, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dscr.put(nullptr, (size_t)blocks * 4 * lvc_scratch_bytes(), s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dops.put(nullptr, (size_t)n * ops_stride * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dno.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, ded.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dafc.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dxa.put(nullptr, (size_t)n * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, hipMemsetAsync(dops.p, 0, (size_t)n * ops_stride * 4, s), SNAPGPU_E_LAUNCH);
    CigarArgs a;
    a.ix = ctx->ix; a.n = n; a.RL = RL; a.ops_stride = ops_stride; a.use_m = use_m ? 1u : 0u;
    a.data = (const uint8_t *)dd.p; a.off = (const uint64_t *)doff.p; a.len = (const int32_t *)dlen.p; a.loc = (const int64_t *)dloc.p;
    a.extra_before = (const int32_t *)dxb.p; a.scratch = (uint8_t *)dscr.p; a.work_counter = ctx->d_work;
    a.ops = (uint32_t *)dops.p; a.n_ops = (int32_t *)dno.p; a.edit_distance = (int32_t *)ded.p;
    a.add_front_clipping = (int32_t *)dafc.p; a.extra_after = (int64_t *)dxa.p;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_work, 0, 4, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s), SNAPGPU_E_LAUNCH);
    snapgpu_launch_cigar_lv(&a, blocks, (size_t)4 * per_wave, s);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(ops, dops.p, (size_t)n * ops_stride * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(n_ops, dno.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(edit_distance, ded.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(add_front_clipping, dafc.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(extra_clipped_after, dxa.p, (size_t)n * 8, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    return sam_side_kernel_time(ctx);
}

// AlignmentAdjuster::AdjustAlignment for a batch of results (adjust.h, cigar_k.hip)
extern "C" int snapgpu_adjust_alignments(snapgpu_ctx *ctx, uint32_t n, const char *data, uint64_t data_bytes, const uint64_t *off, const int32_t *len,
                                         snapgpu_single_result *results)
{
    if (!ctx || (n && (!data || !off || !len || !results))) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_adjust_alignments: null argument");
    if (n == 0) return SNAPGPU_OK;
    uint32_t RL = 64;
    for (uint32_t i = 0; i < n; i++) {
        if (len[i] < 1 || len[i] > 4000) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_adjust_alignments: read length out of range");
        if (off[i] + (uint64_t)len[i] > data_bytes) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_adjust_alignments: read outside the data buffer");
        if (results[i].status != SNAPGPU_NotFound && (results[i].location < 0 || (uint64_t)results[i].location >= ctx->ix.n_bases))
            return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_adjust_alignments: location outside the genome");
        if (results[i].direction != 0 && results[i].direction != 1) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_adjust_alignments: direction must be 0 (forward) or 1 (reverse complement)");
        if ((uint32_t)len[i] > RL) RL = (uint32_t)len[i];
    }
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    hipStream_t s = ctx->stream;
    uint32_t blocks = (uint32_t)ctx->num_cus * 4;
    const uint32_t need = (n + 3) / 4; if (blocks > need) blocks = need;
    const uint64_t stride = 2 * (uint64_t)((RL + 255) & ~255u) + ((adjust_scratch_bytes(RL) + 255) & ~(uint64_t)255);
    PoolScope pool_scope(&ctx->pool);
    DevBuf dd, doff, dlen, dres, dscr;
    HIPCHK(ctx, dd.put(data, data_bytes, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, doff.put(off, (size_t)n * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dlen.put(len, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dres.put(results, (size_t)n * sizeof(snapgpu_single_result), s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dscr.put(nullptr, (size_t)blocks * 4 * stride, s), SNAPGPU_E_NOMEM);
    AdjustArgs a;
    a.ix = ctx->ix; a.n = n; a.RL = RL; a.data = (const uint8_t *)dd.p; a.off = (const uint64_t *)doff.p; a.len = (const int32_t *)dlen.p;
    a.results = (snapgpu_single_result *)dres.p; a.scratch = (uint8_t *)dscr.p; a.scratch_stride = stride; a.work_counter = ctx->d_work;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_work, 0, 4, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s), SNAPGPU_E_LAUNCH);
    snapgpu_launch_adjust_alignments(&a, blocks, s);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(results, dres.p, (size_t)n * sizeof(snapgpu_single_result), hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    return sam_side_kernel_time(ctx);
}

// SAMFormat::computeCigar, affine-gap variant, for a batch of written reads (cigar_ag.h, cigar_k.hip).
extern "C" int snapgpu_compute_cigar_ag(snapgpu_ctx *ctx, uint32_t n, const char *data, const char *quals, uint64_t data_bytes,
                                        const uint64_t *off, const int32_t *len, const int64_t *loc, const int32_t *extra_before,
                                        const int32_t *score, int use_m, uint32_t *ops, uint32_t ops_stride, int32_t *n_ops,
                                        int32_t *edit_distance, int32_t *add_front_clipping, int64_t *extra_clipped_after,
                                        int32_t *back_clipping_missed, int32_t *reference_history_dependent)
{
    if (!ctx || (n && (!data || !quals || !off || !len || !loc || !extra_before || !score || !ops || !n_ops || !edit_distance ||
                       !add_front_clipping || !extra_clipped_after || !back_clipping_missed || !reference_history_dependent)))
        return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_compute_cigar_ag: null argument");
    if (ops_stride == 0) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_compute_cigar_ag: ops_stride must be positive");
    if (n == 0) return SNAPGPU_OK;
    uint32_t RL = 64;
    for (uint32_t i = 0; i < n; i++) {
        if (len[i] < 0 || len[i] > AGC_MAX_READ_LENGTH) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_compute_cigar_ag: read length out of range");
        if (extra_before[i] < 0 || extra_before[i] > len[i]) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_compute_cigar_ag: extra_before out of range");
        if (off[i] + (uint64_t)len[i] > data_bytes) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_compute_cigar_ag: read outside the data buffer");
        if (loc[i] < 0 || (uint64_t)loc[i] >= ctx->ix.n_bases) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_compute_cigar_ag: location outside the genome");
        if (score[i] < 0) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_compute_cigar_ag: negative score");
        if ((uint32_t)len[i] > RL) RL = (uint32_t)len[i];
    }
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    hipStream_t s = ctx->stream;
    const uint32_t per_wave = agc_lds_bytes(RL);
    uint32_t waves_per_block = 4;
    if ((size_t)waves_per_block * per_wave > 64 * 1024) return fail(ctx, SNAPGPU_E_UNSUPPORTED, "snapgpu_compute_cigar_ag: reads too long for the LDS rows");
    uint32_t blocks = (uint32_t)ctx->num_cus * 4;
    const uint32_t need = (n + 3) / 4; if (blocks > need) blocks = need;
    const uint64_t scratch_stride = (agc_scratch_bytes(RL) + 255) & ~(uint64_t)255;
    PoolScope pool_scope(&ctx->pool);
    DevBuf dd, dq, doff, dlen, dloc, dxb, dsc, dscr, dops, dno, ded, dafc, dxa, dti, dst;
    HIPCHK(ctx, dd.put(data, data_bytes, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dq.put(quals, data_bytes, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, doff.put(off, (size_t)n * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dlen.put(len, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dloc.put(loc, (size_t)n * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dxb.put(extra_before, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dsc.put(score, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dscr.put(nullptr, (size_t)blocks * 4 * scratch_stride, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dops.put(nullptr, (size_t)n * ops_stride * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dno.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, ded.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dafc.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dxa.put(nullptr, (size_t)n * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dti.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dst.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, hipMemsetAsync(dops.p, 0, (size_t)n * ops_stride * 4, s), SNAPGPU_E_LAUNCH);
    CigarAGArgs a;
    a.ix = ctx->ix;
    // AffineGapVectorizedWithCigar's constructor (AffineGapVectorized.cpp:15-23): subPenalty negated, gapOpen = open + extend
    a.prm.match = (int)ctx->params.match_reward; a.prm.sub = -(int)ctx->params.sub_penalty;
    a.prm.gap_open = (int)ctx->params.gap_open_penalty + (int)ctx->params.gap_extend_penalty; a.prm.gap_ext = (int)ctx->params.gap_extend_penalty;
    a.n = n; a.RL = RL; a.ops_stride = ops_stride; a.use_m = use_m ? 1u : 0u;
    a.data = (const uint8_t *)dd.p; a.quals = (const uint8_t *)dq.p; a.off = (const uint64_t *)doff.p; a.len = (const int32_t *)dlen.p;
    a.loc = (const int64_t *)dloc.p; a.extra_before = (const int32_t *)dxb.p; a.score = (const int32_t *)dsc.p;
    a.scratch = (uint8_t *)dscr.p; a.scratch_stride = scratch_stride; a.work_counter = ctx->d_work;
    a.ops = (uint32_t *)dops.p; a.n_ops = (int32_t *)dno.p; a.edit_distance = (int32_t *)ded.p; a.add_front_clipping = (int32_t *)dafc.p;
    a.extra_after = (int64_t *)dxa.p; a.tail_ins = (int32_t *)dti.p; a.stale = (int32_t *)dst.p;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_work, 0, 4, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s), SNAPGPU_E_LAUNCH);
    snapgpu_launch_cigar_ag(&a, blocks, (size_t)waves_per_block * per_wave, s);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(ops, dops.p, (size_t)n * ops_stride * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(n_ops, dno.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(edit_distance, ded.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(add_front_clipping, dafc.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(extra_clipped_after, dxa.p, (size_t)n * 8, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(back_clipping_missed, dti.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(reference_history_dependent, dst.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    return sam_side_kernel_time(ctx);
}

// The banded row loops of a batch's records ahead of the records themselves, eight reads to a wavefront (cigar_ag.h: SamfPre, cigar_k.hip:
// k_samf_dp8): fills a.pre / a.pre_stride / a.pre_counter and launches the kernel on `s`.  `buf` keeps the per-read results alive until the
// caller has synchronised.  SNAPGPU_SAMF_DP8=0 (measurement knob) or reads beyond 400 bp: nothing is launched and a.pre stays NULL.
static bool samf_dp8_enabled() { static int v = -1; if (v < 0) { const char *e = getenv("SNAPGPU_SAMF_DP8"); v = e ? (atoi(e) != 0 ? 1 : 0) : 1; } return v == 1; }
static int launch_samf_dp8(snapgpu_ctx *ctx, SamFieldsArgs &a, DevBuf &buf, hipStream_t s)
{
    a.pre = nullptr; a.pre_stride = 0; a.pre_counter = ctx->d_work + 2;
    if (!samf_dp8_enabled() || !a.use_affine_gap || a.RL > 400 || a.n == 0) return SNAPGPU_OK;
    // Best effort (a.pre == NULL is a valid mode: k_sam_fields then runs the row loops itself): not beyond the 64 KiB of LDS per workgroup that k_sam_fields is
    // held to, not beyond a bounded pre-buffer (5 - 13 KB per read: a fraction of what is free, 12 GiB at most), and an allocation that fails is not an error.
    const size_t stride = samf_pre_stride(a.RL);
    const size_t lds = 4 * snapgpu_samf_dp8_lds_per_wave(a.RL);
    if (lds > 64 * 1024) return SNAPGPU_OK;
    const size_t want = (size_t)a.n * stride;
    if (want > ((size_t)12 << 30)) return SNAPGPU_OK;
    if (buf.put(nullptr, want, s) != hipSuccess) { (void)hipGetLastError(); buf.p = nullptr; return SNAPGPU_OK; }
    a.pre = (uint8_t *)buf.p; a.pre_stride = stride;
    uint32_t per_cu = (uint32_t)((size_t)160 * 1024 / (lds ? lds : 1)); if (per_cu > 8) per_cu = 8; if (per_cu < 1) per_cu = 1;
    uint32_t blocks = (uint32_t)ctx->num_cus * per_cu;
    const uint32_t need = (a.n + 31) / 32; if (blocks > need) blocks = need;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_work + 2, 0, 4, s), SNAPGPU_E_LAUNCH);
    snapgpu_launch_samf_dp8(&a, blocks, lds, s);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    return SNAPGPU_OK;
}

// result -> FLAG / RNAME index / POS / MAPQ / CIGAR / NM of the SAM record (sam_fields.h, cigar_k.hip)
extern "C" int snapgpu_sam_fields_single(snapgpu_ctx *ctx, uint32_t n, const char *bases, const char *quals, const uint64_t *offsets,
                                         const int32_t *front_clip, const int32_t *data_len, const snapgpu_single_result *results, int use_m,
                                         int32_t *flag, int32_t *contig, int64_t *pos, int32_t *mapq, uint32_t *ops, uint32_t ops_stride,
                                         int32_t *n_ops, int32_t *nm, int32_t *reference_history_dependent)
{
    if (!ctx || (n && (!bases || !quals || !offsets || !front_clip || !data_len || !results || !flag || !contig || !pos || !mapq || !ops || !n_ops ||
                       !nm || !reference_history_dependent)))
        return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_sam_fields_single: null argument");
    if (ops_stride < 3) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_sam_fields_single: ops_stride must be at least 3");
    if (n == 0) return SNAPGPU_OK;
    uint32_t RL = 64;
    for (uint32_t i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > AGC_MAX_READ_LENGTH)
            return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_sam_fields_single: read length out of range");
        const int64_t U = (int64_t)(offsets[i + 1] - offsets[i]);
        if (front_clip[i] < 0 || data_len[i] < 0 || (int64_t)front_clip[i] + data_len[i] > U)
            return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_sam_fields_single: clipping outside the read");
        if (results[i].status != SNAPGPU_NotFound && (results[i].location < 0 || (uint64_t)results[i].location >= ctx->ix.n_bases))
            return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_sam_fields_single: location outside the genome");
        if ((uint32_t)U > RL) RL = (uint32_t)U;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    hipStream_t s = ctx->stream;
    const uint32_t per_wave = agc_lds_bytes(RL);
    if ((size_t)4 * per_wave > 64 * 1024) return fail(ctx, SNAPGPU_E_UNSUPPORTED, "snapgpu_sam_fields_single: reads too long for the LDS rows");
    uint32_t blocks = (uint32_t)ctx->num_cus * samf_blocks_per_cu();
    const uint32_t need = (n + 3) / 4; if (blocks > need) blocks = need;
    const uint64_t scratch_stride = ((2 * (uint64_t)RL + 255) & ~(uint64_t)255) + ((lvc_scratch_bytes() + 255) & ~255u) + ((agc_scratch_bytes(RL) + 255) & ~(uint64_t)255);
    const uint64_t total = offsets[n];
    PoolScope pool_scope(&ctx->pool);
    DevBuf db, dq, doff, dfc, ddl, dres, dscr, dflag, dctg, dpos, dmq, dops, dno, dnm, dst;
    HIPCHK(ctx, db.put(bases, total, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dq.put(quals, total, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, doff.put(offsets, (size_t)(n + 1) * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dfc.put(front_clip, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, ddl.put(data_len, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dres.put(results, (size_t)n * sizeof(snapgpu_single_result), s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dscr.put(nullptr, (size_t)blocks * 4 * scratch_stride, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dflag.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dctg.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dpos.put(nullptr, (size_t)n * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dmq.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dops.put(nullptr, (size_t)n * ops_stride * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dno.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dnm.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dst.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, hipMemsetAsync(dops.p, 0, (size_t)n * ops_stride * 4, s), SNAPGPU_E_LAUNCH);
    SamFieldsArgs a;
    a.ix = ctx->ix;
    a.prm.match = (int)ctx->params.match_reward; a.prm.sub = -(int)ctx->params.sub_penalty;
    a.prm.gap_open = (int)ctx->params.gap_open_penalty + (int)ctx->params.gap_extend_penalty; a.prm.gap_ext = (int)ctx->params.gap_extend_penalty;
    a.n = n; a.RL = RL; a.ops_stride = ops_stride; a.use_m = use_m ? 1u : 0u; a.use_affine_gap = ctx->params.use_affine_gap ? 1u : 0u;
    a.bases = (const uint8_t *)db.p; a.quals = (const uint8_t *)dq.p; a.offsets = (const uint64_t *)doff.p;
    a.front_clip = (const int32_t *)dfc.p; a.data_len = (const int32_t *)ddl.p; a.results = (const snapgpu_single_result *)dres.p;
    a.scratch = (uint8_t *)dscr.p; a.scratch_stride = scratch_stride; a.work_counter = ctx->d_work;
    a.flag = (int32_t *)dflag.p; a.contig = (int32_t *)dctg.p; a.pos = (int64_t *)dpos.p; a.mapq = (int32_t *)dmq.p;
    a.ops = (uint32_t *)dops.p; a.n_ops = (int32_t *)dno.p; a.nm = (int32_t *)dnm.p; a.stale = (int32_t *)dst.p;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_work, 0, 4, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s), SNAPGPU_E_LAUNCH);
    DevBuf dpre;
    { const int prc = launch_samf_dp8(ctx, a, dpre, s); if (prc) return prc; }
    snapgpu_launch_sam_fields(&a, blocks, (size_t)4 * per_wave, s);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(flag, dflag.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(contig, dctg.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(pos, dpos.p, (size_t)n * 8, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(mapq, dmq.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(ops, dops.p, (size_t)n * ops_stride * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(n_ops, dno.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(nm, dnm.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(reference_history_dependent, dst.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    return sam_side_kernel_time(ctx);
}

// device-pointer form of snapgpu_sam_fields_single: reads, clipping and results already in HBM (e.g. right after
// snapgpu_align_single_device), outputs left in HBM.  max_read_len sizes the per-wave LDS rows and the scratch slab.
extern "C" int snapgpu_sam_fields_single_device(snapgpu_ctx *ctx, uint32_t n, uint32_t max_read_len, const void *d_bases, const void *d_quals,
                                                const void *d_offsets, const void *d_front_clip, const void *d_data_len, const void *d_results, int use_m,
                                                void *d_flag, void *d_contig, void *d_pos, void *d_mapq, void *d_ops, uint32_t ops_stride,
                                                void *d_n_ops, void *d_nm, void *d_reference_history_dependent, void *stream)
{
    if (!ctx || (n && (!d_bases || !d_quals || !d_offsets || !d_front_clip || !d_data_len || !d_results || !d_flag || !d_contig || !d_pos || !d_mapq ||
                       !d_ops || !d_n_ops || !d_nm || !d_reference_history_dependent)))
        return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_sam_fields_single_device: null argument");
    if (ops_stride < 3) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_sam_fields_single_device: ops_stride must be at least 3");
    if (max_read_len == 0 || max_read_len > AGC_MAX_READ_LENGTH) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_sam_fields_single_device: max_read_len out of range");
    if (n == 0) return SNAPGPU_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const uint32_t RL = max_read_len < 64 ? 64 : max_read_len;
    const uint32_t per_wave = agc_lds_bytes(RL);
    if ((size_t)4 * per_wave > 64 * 1024) return fail(ctx, SNAPGPU_E_UNSUPPORTED, "snapgpu_sam_fields_single_device: reads too long for the LDS rows");
    uint32_t blocks = (uint32_t)ctx->num_cus * samf_blocks_per_cu();
    const uint32_t need = (n + 3) / 4; if (blocks > need) blocks = need;
    const uint64_t scratch_stride = ((2 * (uint64_t)RL + 255) & ~(uint64_t)255) + ((lvc_scratch_bytes() + 255) & ~255u) + ((agc_scratch_bytes(RL) + 255) & ~(uint64_t)255);
    PoolScope pool_scope(&ctx->pool);
    DevBuf dscr;
    HIPCHK(ctx, dscr.put(nullptr, (size_t)blocks * 4 * scratch_stride, s), SNAPGPU_E_NOMEM);
    SamFieldsArgs a;
    a.ix = ctx->ix;
    a.prm.match = (int)ctx->params.match_reward; a.prm.sub = -(int)ctx->params.sub_penalty;
    a.prm.gap_open = (int)ctx->params.gap_open_penalty + (int)ctx->params.gap_extend_penalty; a.prm.gap_ext = (int)ctx->params.gap_extend_penalty;
    a.n = n; a.RL = RL; a.ops_stride = ops_stride; a.use_m = use_m ? 1u : 0u; a.use_affine_gap = ctx->params.use_affine_gap ? 1u : 0u;
    a.bases = (const uint8_t *)d_bases; a.quals = (const uint8_t *)d_quals; a.offsets = (const uint64_t *)d_offsets;
    a.front_clip = (const int32_t *)d_front_clip; a.data_len = (const int32_t *)d_data_len; a.results = (const snapgpu_single_result *)d_results;
    a.scratch = (uint8_t *)dscr.p; a.scratch_stride = scratch_stride; a.work_counter = ctx->d_work;
    a.flag = (int32_t *)d_flag; a.contig = (int32_t *)d_contig; a.pos = (int64_t *)d_pos; a.mapq = (int32_t *)d_mapq;
    a.ops = (uint32_t *)d_ops; a.n_ops = (int32_t *)d_n_ops; a.nm = (int32_t *)d_nm; a.stale = (int32_t *)d_reference_history_dependent;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_work, 0, 4, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemsetAsync(d_ops, 0, (size_t)n * ops_stride * 4, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s), SNAPGPU_E_LAUNCH);
    DevBuf dpre;
    { const int prc = launch_samf_dp8(ctx, a, dpre, s); if (prc) return prc; }
    snapgpu_launch_sam_fields(&a, blocks, (size_t)4 * per_wave, s);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);           // (the scratch slab is freed on return; a caller-owned slab is the next step)
    return sam_side_kernel_time(ctx);
}

static int launch_align(snapgpu_ctx *ctx, uint32_t n, const void *d_bases, const void *d_quals, const void *d_offsets,
                        void *d_primary, void *d_first_alt, hipStream_t s, void *d_secondary, uint32_t sec_out_stride, void *d_n_secondary);

// The single-end path of a SAM writer in one call, device-resident in between: ONE upload of the batch (the unclipped reads, Read::clip's
// outcome, which reads the aligner is given), BaseAligner::AlignRead over the clipped reads, the SAM fields of every read from the results
// where the align kernel left them, one download of the fields.  What snapgpu_align_single followed by snapgpu_sam_fields_single does with
// two uploads of the reads and a round trip of the results (profiles/r04zy: the feeders' time was those copies and calls).
extern "C" int snapgpu_align_sam_single(snapgpu_ctx *ctx, uint32_t n, const char *bases, const char *quals, const uint64_t *offsets,
                                        const int32_t *front_clip, const int32_t *data_len, const uint8_t *skip, int use_m,
                                        snapgpu_single_result *results, snapgpu_single_result *first_alt,
                                        int32_t *flag, int32_t *contig, int64_t *pos, int32_t *mapq, uint32_t *ops, uint32_t ops_stride,
                                        int32_t *n_ops, int32_t *nm, int32_t *reference_history_dependent)
{
    if (!ctx || (n && (!bases || !quals || !offsets || !front_clip || !data_len || !skip || !flag || !contig || !pos || !mapq || !ops || !n_ops || !nm ||
                       !reference_history_dependent)))
        return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_align_sam_single: null argument");
    if (ops_stride < 3) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_align_sam_single: ops_stride must be at least 3");
    if (ctx->secondary || ctx->paired) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_align_sam_single: a plain single-end context is needed (no snapgpu_enable_secondary / _paired)");
    if (n == 0) return SNAPGPU_OK;
    uint32_t RL = 64;
    for (uint32_t i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > AGC_MAX_READ_LENGTH)
            return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_align_sam_single: read length out of range");
        const int64_t U = (int64_t)(offsets[i + 1] - offsets[i]);
        if (front_clip[i] < 0 || data_len[i] < 0 || (int64_t)front_clip[i] + data_len[i] > U)
            return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_align_sam_single: clipping outside the read");
        if (!skip[i] && (uint32_t)data_len[i] > ctx->params.max_read_len)
            return fail(ctx, SNAPGPU_E_INVALID, "read longer than max_read_len given at snapgpu_create (BaseAligner.cpp:354-358)");
        if ((uint32_t)U > RL) RL = (uint32_t)U;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    hipStream_t s = ctx->stream;
    const uint32_t per_wave = agc_lds_bytes(RL);
    if ((size_t)4 * per_wave > 64 * 1024) return fail(ctx, SNAPGPU_E_UNSUPPORTED, "snapgpu_align_sam_single: reads too long for the LDS rows");
    uint32_t blocks = (uint32_t)ctx->num_cus * samf_blocks_per_cu();
    const uint32_t need = (n + 3) / 4; if (blocks > need) blocks = need;
    const uint64_t scratch_stride = ((2 * (uint64_t)RL + 255) & ~(uint64_t)255) + ((lvc_scratch_bytes() + 255) & ~255u) + ((agc_scratch_bytes(RL) + 255) & ~(uint64_t)255);
    const uint64_t total = offsets[n];
    PoolScope pool_scope(&ctx->pool);
    DevBuf db, dq, doff, dfc, ddl, dsk, dres, dalt, dscr, dflag, dctg, dpos, dmq, dops, dno, dnm, dst;
    HIPCHK(ctx, db.put(bases, total, s, 16), SNAPGPU_E_NOMEM);          // (16 bytes of slack behind the reads, as snapgpu_align_single's staging has)
    HIPCHK(ctx, dq.put(quals, total, s, 16), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, doff.put(offsets, (size_t)(n + 1) * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dfc.put(front_clip, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, ddl.put(data_len, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dsk.put(skip, (size_t)n, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dres.put(nullptr, (size_t)n * sizeof(snapgpu_single_result), s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dalt.put(nullptr, (size_t)n * sizeof(snapgpu_single_result), s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dscr.put(nullptr, (size_t)blocks * 4 * scratch_stride, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dflag.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dctg.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dpos.put(nullptr, (size_t)n * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dmq.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dops.put(nullptr, (size_t)n * ops_stride * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dno.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dnm.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dst.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, hipMemsetAsync(dops.p, 0, (size_t)n * ops_stride * 4, s), SNAPGPU_E_LAUNCH);
    ctx->clip_front = (const int32_t *)dfc.p; ctx->clip_len = (const int32_t *)ddl.p; ctx->clip_skip = (const uint8_t *)dsk.p;
    int rc = launch_align(ctx, n, db.p, dq.p, doff.p, dres.p, first_alt ? dalt.p : nullptr, s, nullptr, 0, nullptr);
    ctx->clip_front = ctx->clip_len = nullptr; ctx->clip_skip = nullptr;
    if (rc) return rc;
    rc = finish_timing(ctx);                // (the align launch's own hipEvent time, before the events are reused)
    if (rc) return rc;
    SamFieldsArgs a;
    a.ix = ctx->ix;
    a.prm.match = (int)ctx->params.match_reward; a.prm.sub = -(int)ctx->params.sub_penalty;
    a.prm.gap_open = (int)ctx->params.gap_open_penalty + (int)ctx->params.gap_extend_penalty; a.prm.gap_ext = (int)ctx->params.gap_extend_penalty;
    a.n = n; a.RL = RL; a.ops_stride = ops_stride; a.use_m = use_m ? 1u : 0u; a.use_affine_gap = ctx->params.use_affine_gap ? 1u : 0u;
    a.bases = (const uint8_t *)db.p; a.quals = (const uint8_t *)dq.p; a.offsets = (const uint64_t *)doff.p;
    a.front_clip = (const int32_t *)dfc.p; a.data_len = (const int32_t *)ddl.p; a.results = (const snapgpu_single_result *)dres.p;
    a.scratch = (uint8_t *)dscr.p; a.scratch_stride = scratch_stride; a.work_counter = ctx->d_work;
    a.flag = (int32_t *)dflag.p; a.contig = (int32_t *)dctg.p; a.pos = (int64_t *)dpos.p; a.mapq = (int32_t *)dmq.p;
    a.ops = (uint32_t *)dops.p; a.n_ops = (int32_t *)dno.p; a.nm = (int32_t *)dnm.p; a.stale = (int32_t *)dst.p;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_work, 0, 4, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s), SNAPGPU_E_LAUNCH);
    DevBuf dpre;
    { const int prc = launch_samf_dp8(ctx, a, dpre, s); if (prc) return prc; }
    snapgpu_launch_sam_fields(&a, blocks, (size_t)4 * per_wave, s);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s), SNAPGPU_E_LAUNCH);
    if (results) HIPCHK(ctx, hipMemcpyAsync(results, dres.p, (size_t)n * sizeof(snapgpu_single_result), hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    if (first_alt) HIPCHK(ctx, hipMemcpyAsync(first_alt, dalt.p, (size_t)n * sizeof(snapgpu_single_result), hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(flag, dflag.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(contig, dctg.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(pos, dpos.p, (size_t)n * 8, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(mapq, dmq.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(ops, dops.p, (size_t)n * ops_stride * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(n_ops, dno.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(nm, dnm.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(reference_history_dependent, dst.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    return sam_side_kernel_time(ctx);
}

// paired-end writer: results -> the computed fields of both SAM records of each pair (sam_fields.h, cigar_k.hip)
extern "C" int snapgpu_sam_fields_paired(snapgpu_ctx *ctx, uint32_t n_pairs, const char *bases, const char *quals, const uint64_t *offsets,
                                         const int32_t *front_clip, const int32_t *data_len, const snapgpu_paired_result *results, int use_m,
                                         int32_t *flag, int32_t *contig, int64_t *pos, int32_t *mapq, uint32_t *ops, uint32_t ops_stride,
                                         int32_t *n_ops, int32_t *nm, int32_t *rnext, int64_t *pnext, int64_t *tlen, int32_t *first_written,
                                         int32_t *reference_history_dependent)
{
    if (!ctx || (n_pairs && (!bases || !quals || !offsets || !front_clip || !data_len || !results || !flag || !contig || !pos || !mapq || !ops ||
                             !n_ops || !nm || !rnext || !pnext || !tlen || !first_written || !reference_history_dependent)))
        return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_sam_fields_paired: null argument");
    if (ops_stride < 3) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_sam_fields_paired: ops_stride must be at least 3");
    if (n_pairs == 0) return SNAPGPU_OK;
    const uint32_t n = 2 * n_pairs;
    uint32_t RL = 64;
    for (uint32_t i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > AGC_MAX_READ_LENGTH)
            return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_sam_fields_paired: read length out of range");
        const int64_t U = (int64_t)(offsets[i + 1] - offsets[i]);
        if (front_clip[i] < 0 || data_len[i] < 0 || (int64_t)front_clip[i] + data_len[i] > U)
            return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_sam_fields_paired: clipping outside the read");
        const snapgpu_paired_result &r = results[i >> 1];
        if (r.status[i & 1] != SNAPGPU_NotFound && (r.location[i & 1] < 0 || (uint64_t)r.location[i & 1] >= ctx->ix.n_bases))
            return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_sam_fields_paired: location outside the genome");
        if ((uint32_t)U > RL) RL = (uint32_t)U;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    hipStream_t s = ctx->stream;
    const uint32_t per_wave = agc_lds_bytes(RL);
    if ((size_t)4 * per_wave > 64 * 1024) return fail(ctx, SNAPGPU_E_UNSUPPORTED, "snapgpu_sam_fields_paired: reads too long for the LDS rows");
    uint32_t blocks = (uint32_t)ctx->num_cus * samf_blocks_per_cu();
    const uint32_t need = (n_pairs + 3) / 4; if (blocks > need) blocks = need;
    const uint64_t scratch_stride = ((2 * (uint64_t)RL + 255) & ~(uint64_t)255) + ((lvc_scratch_bytes() + 255) & ~255u) + ((agc_scratch_bytes(RL) + 255) & ~(uint64_t)255);
    const uint64_t total = offsets[n];
    PoolScope pool_scope(&ctx->pool);
    DevBuf db, dq, doff, dfc, ddl, dres, dscr, dflag, dctg, dpos, dmq, dops, dno, dnm, drn, dpn, dtl, dfw, dst;
    HIPCHK(ctx, db.put(bases, total, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dq.put(quals, total, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, doff.put(offsets, (size_t)(n + 1) * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dfc.put(front_clip, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, ddl.put(data_len, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dres.put(results, (size_t)n_pairs * sizeof(snapgpu_paired_result), s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dscr.put(nullptr, (size_t)blocks * 4 * scratch_stride, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dflag.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dctg.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dpos.put(nullptr, (size_t)n * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dmq.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dops.put(nullptr, (size_t)n * ops_stride * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dno.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dnm.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, drn.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dpn.put(nullptr, (size_t)n * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dtl.put(nullptr, (size_t)n * 8, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dfw.put(nullptr, (size_t)n_pairs * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dst.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, hipMemsetAsync(dops.p, 0, (size_t)n * ops_stride * 4, s), SNAPGPU_E_LAUNCH);
    SamFieldsPairedArgs a;
    a.ix = ctx->ix;
    a.prm.match = (int)ctx->params.match_reward; a.prm.sub = -(int)ctx->params.sub_penalty;
    a.prm.gap_open = (int)ctx->params.gap_open_penalty + (int)ctx->params.gap_extend_penalty; a.prm.gap_ext = (int)ctx->params.gap_extend_penalty;
    a.n_pairs = n_pairs; a.RL = RL; a.ops_stride = ops_stride; a.use_m = use_m ? 1u : 0u; a.use_affine_gap = ctx->params.use_affine_gap ? 1u : 0u;
    a.bases = (const uint8_t *)db.p; a.quals = (const uint8_t *)dq.p; a.offsets = (const uint64_t *)doff.p;
    a.front_clip = (const int32_t *)dfc.p; a.data_len = (const int32_t *)ddl.p; a.results = (const snapgpu_paired_result *)dres.p;
    a.scratch = (uint8_t *)dscr.p; a.scratch_stride = scratch_stride; a.work_counter = ctx->d_work;
    a.flag = (int32_t *)dflag.p; a.contig = (int32_t *)dctg.p; a.pos = (int64_t *)dpos.p; a.mapq = (int32_t *)dmq.p;
    a.ops = (uint32_t *)dops.p; a.n_ops = (int32_t *)dno.p; a.nm = (int32_t *)dnm.p; a.rnext = (int32_t *)drn.p; a.pnext = (int64_t *)dpn.p;
    a.tlen = (int64_t *)dtl.p; a.first_written = (int32_t *)dfw.p; a.stale = (int32_t *)dst.p;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_work, 0, 4, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s), SNAPGPU_E_LAUNCH);
    snapgpu_launch_sam_fields_paired(&a, blocks, (size_t)4 * per_wave, s);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s), SNAPGPU_E_LAUNCH);
    struct { void *h; void *d; size_t b; } outs[] = {
        {flag, dflag.p, (size_t)n * 4}, {contig, dctg.p, (size_t)n * 4}, {pos, dpos.p, (size_t)n * 8}, {mapq, dmq.p, (size_t)n * 4},
        {ops, dops.p, (size_t)n * ops_stride * 4}, {n_ops, dno.p, (size_t)n * 4}, {nm, dnm.p, (size_t)n * 4}, {rnext, drn.p, (size_t)n * 4},
        {pnext, dpn.p, (size_t)n * 8}, {tlen, dtl.p, (size_t)n * 8}, {first_written, dfw.p, (size_t)n_pairs * 4},
        {reference_history_dependent, dst.p, (size_t)n * 4}};
    for (auto &o : outs) HIPCHK(ctx, hipMemcpyAsync(o.h, o.d, o.b, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    return sam_side_kernel_time(ctx);
}

static int affine_gap_batch(snapgpu_ctx *ctx, int dir, uint32_t n,
                            const char *texts, uint64_t texts_bytes, const uint32_t *text_off, const int32_t *text_len,
                            const char *patterns, const char *quals, uint64_t patterns_bytes,
                            const uint32_t *pat_off, const int32_t *pat_len,
                            const int32_t *w, const int32_t *score_init, const uint8_t *is_rc,
                            const uint8_t *banded, const uint8_t *use_clip,
                            int32_t *ag_score, int32_t *text_offset, int32_t *pattern_offset,
                            int32_t *n_edits, double *match_probability, bool sequence, int32_t *stale_steps)
{
    if (!ctx || (dir != 1 && dir != -1)) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_affine_gap: bad argument");
    if (n == 0) return SNAPGPU_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    uint32_t RL = 16;
    for (uint32_t i = 0; i < n; i++) {
        if (pat_len[i] < 1 || pat_len[i] > 1000) return fail(ctx, SNAPGPU_E_INVALID, "pattern length out of range");
        if (text_len[i] < 0 || text_len[i] > pat_len[i] + 127) return fail(ctx, SNAPGPU_E_INVALID, "text length must be in [0, pattern_len + MAX_K]");
        if (score_init[i] < 0 || score_init[i] > 16000) return fail(ctx, SNAPGPU_E_INVALID, "score_init out of range");
        if ((uint32_t)pat_len[i] > RL) RL = (uint32_t)pat_len[i];
    }
    RL = (RL + 15) & ~15u;
    hipStream_t s = ctx->stream;
    const uint32_t waves_per_block = 1;
    uint32_t blocks = n; uint32_t maxb = (uint32_t)ctx->num_cus * 8; if (blocks > maxb) blocks = maxb;
    if (sequence) blocks = 1;
    PoolScope pool_scope(&ctx->pool);
    DevBuf dt, dto, dtl, dp, dq, dpo, dpl, dw, dsi, drc, dbd, dcl, dscratch, o1, o2, o3, o4, o5, o6;
    HIPCHK(ctx, dt.put(texts, texts_bytes, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dto.put(text_off, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dtl.put(text_len, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dp.put(patterns, patterns_bytes, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dq.put(quals, patterns_bytes, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dpo.put(pat_off, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dpl.put(pat_len, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dw.put(w, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dsi.put(score_init, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, drc.put(is_rc, (size_t)n, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dbd.put(banded, (size_t)n, s), SNAPGPU_E_NOMEM);
    std::vector<uint8_t> zeros;
    if (!use_clip) { zeros.assign(n, 0); use_clip = zeros.data(); }
    HIPCHK(ctx, dcl.put(use_clip, (size_t)n, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, dscratch.put(nullptr, (size_t)(sequence ? 3 : blocks * waves_per_block) * ag_scratch_bytes(RL), s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, o1.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, o2.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, o3.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, o4.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, o5.put(nullptr, (size_t)n * 8, s), SNAPGPU_E_NOMEM);
    if (stale_steps) HIPCHK(ctx, o6.put(nullptr, (size_t)n * 4, s), SNAPGPU_E_NOMEM);
    if (sequence) HIPCHK(ctx, hipMemsetAsync(dscratch.p, 0, ag_scratch_bytes(RL), s), SNAPGPU_E_LAUNCH);      // a newly constructed object: the array reads as zero
    AGBatchArgs a;
    a.dir = dir; a.n = n; a.RL = RL;
    a.prm = AGParams{ctx->cfg.match_reward, ctx->cfg.sub_penalty, ctx->cfg.gap_open, ctx->cfg.gap_extend, ctx->cfg.five_bonus, ctx->cfg.three_bonus};
    a.texts = (const uint8_t *)dt.p; a.text_off = (const uint32_t *)dto.p; a.text_len = (const int32_t *)dtl.p;
    a.patterns = (const uint8_t *)dp.p; a.quals = (const uint8_t *)dq.p; a.pat_off = (const uint32_t *)dpo.p;
    a.pat_len = (const int32_t *)dpl.p; a.w = (const int32_t *)dw.p; a.score_init = (const int32_t *)dsi.p;
    a.is_rc = (const uint8_t *)drc.p; a.banded = (const uint8_t *)dbd.p; a.use_clip = (const uint8_t *)dcl.p;
    a.scratch = (uint8_t *)dscratch.p;
    a.ag_score = (int32_t *)o1.p; a.text_offset = (int32_t *)o2.p; a.pattern_offset = (int32_t *)o3.p;
    a.n_edits = (int32_t *)o4.p; a.prob = (double *)o5.p; a.tab = ctx->d_tab; a.stale = stale_steps ? (int32_t *)o6.p : nullptr;
    uint32_t lds = (ag_lds_bytes(RL) + 15) & ~15u;
    // variant by the largest striped layout in this batch (chunks of 64 positions)
    int need = 0;
    for (uint32_t i = 0; i < n; i++) {
        int nv, sl, ns; int ww = w[i] > 126 ? 126 : (w[i] < 0 ? 0 : w[i]);
        ag_dims(banded[i] != 0, pat_len[i], ww, &nv, &sl, &ns);
        if (ns * sl > need) need = ns * sl;
    }
    if (getenv("SNAPGPU_AG_LDS")) need = 1 << 20;
    if (sequence) {
        if (need <= 192) hipLaunchKernelGGL(k_ag_sequence<3>, dim3(1), dim3(64), lds, s, a);
        else             hipLaunchKernelGGL(k_ag_sequence<0>, dim3(1), dim3(64), lds, s, a);          // (the exact replay has these two forms)
    } else
    if (need <= 192)      hipLaunchKernelGGL(k_ag_batch<3>, dim3(blocks), dim3(64 * waves_per_block), waves_per_block * lds, s, a);
    else if (need <= 256) hipLaunchKernelGGL(k_ag_batch<4>, dim3(blocks), dim3(64 * waves_per_block), waves_per_block * lds, s, a);
    else if (need <= 384) hipLaunchKernelGGL(k_ag_batch<6>, dim3(blocks), dim3(64 * waves_per_block), waves_per_block * lds, s, a);
    else                  hipLaunchKernelGGL(k_ag_batch<0>, dim3(blocks), dim3(64 * waves_per_block), waves_per_block * lds, s, a);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(ag_score, o1.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(text_offset, o2.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(pattern_offset, o3.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(n_edits, o4.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(match_probability, o5.p, (size_t)n * 8, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    if (stale_steps) HIPCHK(ctx, hipMemcpyAsync(stale_steps, o6.p, (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    return SNAPGPU_OK;
}

extern "C" int snapgpu_affine_gap(snapgpu_ctx *ctx, int dir, uint32_t n,
                                  const char *texts, uint64_t texts_bytes, const uint32_t *text_off, const int32_t *text_len,
                                  const char *patterns, const char *quals, uint64_t patterns_bytes,
                                  const uint32_t *pat_off, const int32_t *pat_len,
                                  const int32_t *w, const int32_t *score_init, const uint8_t *is_rc,
                                  const uint8_t *banded, const uint8_t *use_clip,
                                  int32_t *ag_score, int32_t *text_offset, int32_t *pattern_offset,
                                  int32_t *n_edits, double *match_probability)
{
    return affine_gap_batch(ctx, dir, n, texts, texts_bytes, text_off, text_len, patterns, quals, patterns_bytes, pat_off, pat_len, w, score_init, is_rc,
                            banded, use_clip, ag_score, text_offset, pattern_offset, n_edits, match_probability, false, nullptr);
}

extern "C" int snapgpu_affine_gap_sequence(snapgpu_ctx *ctx, int dir, uint32_t n,
                                           const char *texts, uint64_t texts_bytes, const uint32_t *text_off, const int32_t *text_len,
                                           const char *patterns, const char *quals, uint64_t patterns_bytes,
                                           const uint32_t *pat_off, const int32_t *pat_len,
                                           const int32_t *w, const int32_t *score_init, const uint8_t *is_rc,
                                           const uint8_t *banded, const uint8_t *use_clip,
                                           int32_t *ag_score, int32_t *text_offset, int32_t *pattern_offset,
                                           int32_t *n_edits, double *match_probability, int32_t *stale_steps)
{
    return affine_gap_batch(ctx, dir, n, texts, texts_bytes, text_off, text_len, patterns, quals, patterns_bytes, pat_off, pat_len, w, score_init, is_rc,
                            banded, use_clip, ag_score, text_offset, pattern_offset, n_edits, match_probability, true, stale_steps);
}

// heavy-first dequeue order of a batch (order.h): ctx->d_order[0 .. n_units) is ready on stream s when this returns
static int launch_unit_order(snapgpu_ctx *ctx, const void *d_bases, const void *d_offsets, uint32_t n_units, uint32_t rpu, uint32_t max_hits, hipStream_t s)
{
    if (ctx->order_cap < n_units) {
        if (ctx->d_order) (void)hipFree(ctx->d_order);
        if (ctx->d_wbucket) (void)hipFree(ctx->d_wbucket);
        ctx->d_order = ctx->d_wbucket = nullptr; ctx->order_cap = 0;
        size_t cap = (size_t)n_units + n_units / 4 + 1024;
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_order, cap * 4), SNAPGPU_E_NOMEM);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_wbucket, cap * 4), SNAPGPU_E_NOMEM);
        ctx->order_cap = cap;
    }
    if (!ctx->d_whist) HIPCHK(ctx, hipMalloc((void **)&ctx->d_whist, 64 * 4), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, hipMemsetAsync(ctx->d_whist, 0, 34 * sizeof(uint32_t), s), SNAPGPU_E_LAUNCH);
    hipLaunchKernelGGL(k_unit_weights<0>, dim3((unsigned)ctx->num_cus * 8), dim3(256), 0, s, ctx->ix, (const uint8_t *)d_bases, (const uint64_t *)d_offsets,
                       n_units, rpu, max_hits, ctx->d_wbucket, ctx->d_whist);
    hipLaunchKernelGGL(k_unit_weight_prefix<0>, dim3(1), dim3(64), 0, s, ctx->d_whist);
    hipLaunchKernelGGL(k_unit_weight_scatter<0>, dim3((n_units + 255) / 256), dim3(256), 0, s, (const uint32_t *)ctx->d_wbucket, n_units, ctx->d_whist, ctx->d_order);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    return SNAPGPU_OK;
}

static int launch_align(snapgpu_ctx *ctx, uint32_t n, const void *d_bases, const void *d_quals, const void *d_offsets,
                        void *d_primary, void *d_first_alt, hipStream_t s,
                        void *d_secondary = nullptr, uint32_t sec_out_stride = 0, void *d_n_secondary = nullptr)
{
    AlignArgs a;
    a.ix = ctx->ix; a.cfg = ctx->cfg; a.tab = ctx->d_tab; a.scratch = ctx->d_scratch;
    a.bases = (const uint8_t *)d_bases; a.quals = (const uint8_t *)d_quals; a.offsets = (const uint64_t *)d_offsets;
    a.n_reads = n; a.primary = (snapgpu_single_result *)d_primary; a.first_alt = (snapgpu_single_result *)d_first_alt;
    a.cfg.stop_on_first_hit = ctx->stop_on_first_hit; a.cfg.explore_popular_seeds = ctx->explore_popular_seeds;
    a.work_counter = ctx->d_work; a.counters = ctx->d_counters;
    a.sec_cfg = SecCfg{-1, -1, 0, 0}; a.sec_scratch = nullptr; a.sec_stride_bytes = 0; a.secondary = nullptr; a.sec_out_stride = 0; a.n_secondary = nullptr;
    a.flag_list = nullptr; a.flag_count = nullptr; a.remap = nullptr; a.n_remap = nullptr; a.persist = nullptr; a.persist_stride = 0;
    a.is_replay = 0; a.order = nullptr; a.dbg = nullptr; a.dbg_slots = 0;
    a.se_slots = nullptr; a.se_n_slots = 0; a.se_spec = nullptr; a.se_spec_cap = 0; a.se_ctl = nullptr; a.se_eager = 0; a.se_keep = 1;
    a.front_clip = ctx->clip_front; a.data_len = ctx->clip_len; a.skip = ctx->clip_skip;
    const bool use_help = ctx->single_help && ctx->d_se_slots && !d_n_secondary &&
                          (ctx->single_help_forced == 1 || (ctx->feeders && ctx->feeders->load() <= 1));
    if (use_help) {                                                       // (fresh protocol state for the launch that is about to start)
        HIPCHK(ctx, hipMemsetAsync(ctx->d_se_slots, 0, SE_HELP_SLOTS * sizeof(SEHelpSlot), s), SNAPGPU_E_LAUNCH);
        HIPCHK(ctx, hipMemsetAsync(ctx->d_se_ctl, 0, 256, s), SNAPGPU_E_LAUNCH);
        a.se_slots = ctx->d_se_slots; a.se_n_slots = SE_HELP_SLOTS; a.se_spec = ctx->d_se_spec; a.se_spec_cap = ctx->se_spec_cap;
        a.se_ctl = ctx->d_se_ctl; a.se_eager = ctx->single_help_eager ? 1u : 0u; a.se_keep = ctx->single_help_keep;
    }
    if (ctx->phase_timers) {
        const size_t words = 64 + 3 * (size_t)ctx->n_wave_slots;
        if (!ctx->d_dbg) HIPCHK(ctx, hipMalloc((void **)&ctx->d_dbg, words * 8), SNAPGPU_E_NOMEM);
        HIPCHK(ctx, hipMemsetAsync(ctx->d_dbg, 0, words * 8, s), SNAPGPU_E_LAUNCH);
        a.dbg = ctx->d_dbg; a.dbg_slots = ctx->n_wave_slots;
    }
    const bool always_exact = ctx->always_exact && ctx->d_exact_persist != nullptr && !use_help;
    const bool exact = !always_exact && ctx->d_exact_persist != nullptr && !getenv("SNAPGPU_NO_EXACT_REPLAY");
    if (exact) {
        if (ctx->flag_list_cap < n) {
            if (ctx->d_flag_list) (void)hipFree(ctx->d_flag_list);
            ctx->d_flag_list = nullptr; ctx->flag_list_cap = 0;
            size_t cap = (size_t)n + n / 4 + 1024;
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_flag_list, cap * 8), SNAPGPU_E_NOMEM);     // (second half: launch_paired's list of pairs for the exact kernel beside the main pass)
            ctx->flag_list_cap = cap;
        }
        a.flag_list = ctx->d_flag_list; a.flag_count = ctx->d_work + 4;
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->d_work, 0, 32, s), SNAPGPU_E_LAUNCH);
    uint32_t blocks = ctx->n_wave_slots / 4;
    uint32_t need = (n + 3) / 4; if (blocks > need) blocks = need;
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s), SNAPGPU_E_LAUNCH);
    const bool heavy_first = ctx->single_heavy_first == 1 || (ctx->single_heavy_first < 0 && ctx->feeders && ctx->feeders->load() <= 1);
    if (heavy_first && n > ctx->n_wave_slots) {                 // (inside the timed region: it is part of the pass)
        const int orc = launch_unit_order(ctx, d_bases, d_offsets, n, 1, ctx->cfg.max_hits, s);
        if (orc) return orc;
        a.order = ctx->d_order;
    }
    if (d_n_secondary) {
        if (!ctx->d_sec_scratch) HIPCHK(ctx, hipMalloc((void **)&ctx->d_sec_scratch, ctx->sec_stride_bytes * ctx->n_wave_slots), SNAPGPU_E_NOMEM);
        a.sec_cfg = ctx->sec_cfg; a.sec_scratch = ctx->d_sec_scratch; a.sec_stride_bytes = ctx->sec_stride_bytes;
        a.secondary = (snapgpu_single_result *)d_secondary; a.sec_out_stride = sec_out_stride; a.n_secondary = (uint32_t *)d_n_secondary;
    }
    // (the instantiations that carry the plane Landau-Vishkin: plain launches of a context created under SNAPGPU_LV_PLANES=1)
    const bool planes_k = ctx->ix.planes != nullptr && !d_n_secondary && !ctx->phase_timers;
    if (always_exact) {                 // one pass, exact by construction
        a.persist = ctx->d_exact_persist; a.persist_stride = ctx->exact_persist_stride;
        if (ctx->phase_timers && !d_n_secondary) snapgpu_launch_single_exact_3_timed(&a, blocks, 4 * ctx->cfg.lds_per_wave, s);
        else if (planes_k) snapgpu_launch_single_exact_planes_3(&a, blocks, 4 * ctx->cfg.lds_per_wave, s);
        else snapgpu_launch_single_exact_3(&a, d_n_secondary ? 1 : 0, blocks, 4 * ctx->cfg.lds_per_wave, s);
    } else if (planes_k) {
        switch (ctx->ag_variant) {
        case 3:  snapgpu_launch_single_planes_3(&a, blocks, 4 * ctx->cfg.lds_per_wave, s); break;
        case 4:  snapgpu_launch_single_planes_4(&a, blocks, 4 * ctx->cfg.lds_per_wave, s); break;
        case 6:  snapgpu_launch_single_planes_6(&a, blocks, 4 * ctx->cfg.lds_per_wave, s); break;
        default: snapgpu_launch_single_planes_0(&a, blocks, 4 * ctx->cfg.lds_per_wave, s); break;
        }
    } else if (d_n_secondary) {
        switch (ctx->ag_variant) {
        case 3:  snapgpu_launch_single_sec_3(&a, blocks, 4 * ctx->cfg.lds_per_wave, s); break;
        case 4:  snapgpu_launch_single_sec_4(&a, blocks, 4 * ctx->cfg.lds_per_wave, s); break;
        case 6:  snapgpu_launch_single_sec_6(&a, blocks, 4 * ctx->cfg.lds_per_wave, s); break;
        default: snapgpu_launch_single_sec_0(&a, blocks, 4 * ctx->cfg.lds_per_wave, s); break;
        }
    } else
    switch (ctx->ag_variant) {
    case 3:  if (ctx->phase_timers) snapgpu_launch_single_3_timed(&a, blocks, 4 * ctx->cfg.lds_per_wave, s);
             else hipLaunchKernelGGL((k_align_single<3, false>), dim3(blocks), dim3(256), 4 * ctx->cfg.lds_per_wave, s, a);
             break;
    case 4:  hipLaunchKernelGGL((k_align_single<4, false>), dim3(blocks), dim3(256), 4 * ctx->cfg.lds_per_wave, s, a); break;
    case 6:  hipLaunchKernelGGL((k_align_single<6, false>), dim3(blocks), dim3(256), 4 * ctx->cfg.lds_per_wave, s, a); break;
    default: hipLaunchKernelGGL((k_align_single<0, false>), dim3(blocks), dim3(256), 4 * ctx->cfg.lds_per_wave, s, a); break;
    }
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    if (exact) {        // redo the flagged reads (usually none: the launch then ends at once) as a newly constructed reference aligner would
        AlignArgs x = a;
        x.flag_list = nullptr; x.flag_count = nullptr; x.remap = ctx->d_flag_list; x.n_remap = ctx->d_work + 4; x.work_counter = ctx->d_work + 3;
        x.is_replay = 1; x.order = nullptr; x.se_slots = nullptr; x.se_ctl = nullptr;
        x.persist = ctx->d_exact_persist; x.persist_stride = ctx->exact_persist_stride;
        if (planes_k) { if (ctx->ag_variant == 3) snapgpu_launch_single_exact_planes_3(&x, ctx->exact_slots / 4, 4 * ctx->cfg.lds_per_wave, s);
                        else snapgpu_launch_single_exact_planes_0(&x, ctx->exact_slots / 4, 4 * ctx->cfg.lds_per_wave, s); }
        else if (ctx->ag_variant == 3) snapgpu_launch_single_exact_3(&x, d_n_secondary ? 1 : 0, ctx->exact_slots / 4, 4 * ctx->cfg.lds_per_wave, s);
        else if (ctx->ag_variant == 4) snapgpu_launch_single_exact_4(&x, d_n_secondary ? 1 : 0, ctx->exact_slots / 4, 4 * ctx->cfg.lds_per_wave, s);
        else if (ctx->ag_variant == 6) snapgpu_launch_single_exact_6(&x, d_n_secondary ? 1 : 0, ctx->exact_slots / 4, 4 * ctx->cfg.lds_per_wave, s);
        else snapgpu_launch_single_exact_0(&x, d_n_secondary ? 1 : 0, ctx->exact_slots / 4, 4 * ctx->cfg.lds_per_wave, s);
        HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s), SNAPGPU_E_LAUNCH);
    return SNAPGPU_OK;
}

static int finish_timing(snapgpu_ctx *ctx) {
    float ms = 0;
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1), SNAPGPU_E_LAUNCH);
    ctx->kernel_ms += ms; ctx->kernel_launches++;
    return SNAPGPU_OK;
}

extern "C" int snapgpu_align_single_device(snapgpu_ctx *ctx, uint32_t n, const void *d_bases, const void *d_quals,
                                           const void *d_offsets, void *d_primary, void *d_first_alt, void *stream)
{
    if (!ctx || !d_bases || !d_quals || !d_offsets || !d_primary) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_align_single_device: null argument");
    if (n == 0) return SNAPGPU_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    int rc = launch_align(ctx, n, d_bases, d_quals, d_offsets, d_primary, d_first_alt, s);
    if (rc) return rc;
    if (!stream) return finish_timing(ctx);      // own stream: synchronous, and the launch is timed
    return SNAPGPU_OK;
}

extern "C" int snapgpu_align_single(snapgpu_ctx *ctx, uint32_t n, const char *bases, const char *quals,
                                    const uint64_t *offsets, snapgpu_single_result *primary, snapgpu_single_result *first_alt)
{
    if (!ctx || !bases || !quals || !offsets || !primary) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_align_single: null argument");
    if (n == 0) return SNAPGPU_OK;
    for (uint32_t i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i]) return fail(ctx, SNAPGPU_E_INVALID, "offsets must be non-decreasing");
        if (offsets[i + 1] - offsets[i] > ctx->params.max_read_len)
            return fail(ctx, SNAPGPU_E_INVALID, "read longer than max_read_len given at snapgpu_create (BaseAligner.cpp:354-358)");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    size_t nb = (size_t)offsets[n];
    int rc;
    if ((rc = ensure_stage(ctx, 0, nb + 16)) || (rc = ensure_stage(ctx, 1, nb + 16)) || (rc = ensure_stage(ctx, 2, ((size_t)n + 1) * 8)) ||
        (rc = ensure_stage(ctx, 3, (size_t)n * sizeof(snapgpu_single_result))) || (rc = ensure_stage(ctx, 4, (size_t)n * sizeof(snapgpu_single_result)))) return rc;
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage[0], bases, nb, hipMemcpyHostToDevice, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage[1], quals, nb, hipMemcpyHostToDevice, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage[2], offsets, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, s), SNAPGPU_E_LAUNCH);
    rc = launch_align(ctx, n, ctx->d_stage[0], ctx->d_stage[1], ctx->d_stage[2], ctx->d_stage[3], first_alt ? ctx->d_stage[4] : nullptr, s);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(primary, ctx->d_stage[3], (size_t)n * sizeof(snapgpu_single_result), hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    if (first_alt) HIPCHK(ctx, hipMemcpyAsync(first_alt, ctx->d_stage[4], (size_t)n * sizeof(snapgpu_single_result), hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    return finish_timing(ctx);
}

// =====================================================================================
// secondary results (-om / -omax / -mpc)
// =====================================================================================

static int setup_paired_secondary(snapgpu_ctx *ctx);

extern "C" int snapgpu_enable_secondary(snapgpu_ctx *ctx, const snapgpu_secondary_params *sp)
{
    if (!ctx || !sp) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_enable_secondary: null argument");
    if (sp->max_edit_distance < 0) return fail(ctx, SNAPGPU_E_INVALID, "-om must be >= 0 (AlignerOptions.cpp:604)");
    if (sp->max_edit_distance > (int)ctx->params.extra_search_depth)
        return fail(ctx, SNAPGPU_E_INVALID, "the max edit distance for secondary alignments (-om) cannot be bigger than the extra search depth (-D) (AlignerContext.cpp:784)");
    if (sp->max_results <= 0) return fail(ctx, SNAPGPU_E_INVALID, "-omax must be strictly positive (AlignerOptions.cpp:621)");
    if (sp->max_per_contig == 0 || sp->max_per_contig < -1) return fail(ctx, SNAPGPU_E_INVALID, "-mpc must be strictly positive, or -1 for no limit (AlignerOptions.cpp:650)");
    if (sp->adjust_alignments && ctx->paired)
        return fail(ctx, SNAPGPU_E_UNSUPPORTED, "-ae (AlignmentAdjuster before the -om filter) is implemented for BaseAligner::AlignRead (BaseAligner.cpp:2444-2463), not for the paired-end aligners (IntersectingPairedEndAligner.cpp:1298-1320)");
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    // updateBestScore appends at most one entry per ScoreSet per scored candidate, and a read scores at most one candidate per
    // seed hit it applied: maxSeedsToUse * maxHits hits, two score sets (BaseAligner.cpp:451, 627, 1445-1468).
    uint64_t seeds = ctx->params.num_seeds ? ctx->params.num_seeds
                   : (uint64_t)(2 * ctx->params.seed_coverage * ctx->params.max_read_len / ctx->ix.seed_len) + 1;
    // (the seed loop tests nSeedsApplied[F] + nSeedsApplied[RC] < maxSeedsToUse and one pass can apply both directions: seeds + 1)
    uint64_t cap = 2 * (seeds + 1) * ctx->params.max_hits + 2;
    if (cap > (1u << 20)) return fail(ctx, SNAPGPU_E_UNSUPPORTED, "2 * seeds * max_hits > 2^20 secondary candidates per read is not supported");
    uint64_t stride = cap * sizeof(snapgpu_single_result) + cap * 3 * 4;
    stride = (stride + 255) & ~(uint64_t)255;
    const uint64_t adj_off = stride;
    if (sp->adjust_alignments) stride += (adjust_scratch_bytes(ctx->cfg.RL) + 255) & ~(uint64_t)255;     // adjust.h: the adjuster's Landau-Vishkin table, op list, staged read and window
    if (ctx->d_sec_scratch) { (void)hipFree(ctx->d_sec_scratch); ctx->d_sec_scratch = nullptr; }
    ctx->secondary = false;
    // (the per-wave lists of the SINGLE-END launches: ~1.5 MB x 6 144 wave slots at the defaults.  A context that has the paired-end path
    //  enabled keeps its single-end aligner's lists in the paired slab and gets these on its first single-end launch with secondary results.)
    if (!ctx->paired) HIPCHK(ctx, hipMalloc((void **)&ctx->d_sec_scratch, stride * ctx->n_wave_slots), SNAPGPU_E_NOMEM);
    ctx->sec_cfg = SecCfg{sp->max_edit_distance, sp->max_per_contig, sp->max_results, (uint32_t)cap, sp->adjust_alignments ? 1u : 0u, adj_off};
    ctx->sec_stride_bytes = stride;
    ctx->secondary = true;
    return setup_paired_secondary(ctx);       // (if the paired-end path is enabled too)
}

extern "C" int snapgpu_align_single_secondary_device(snapgpu_ctx *ctx, uint32_t n, const void *d_bases, const void *d_quals,
                                                     const void *d_offsets, void *d_primary, void *d_first_alt,
                                                     void *d_secondary, uint32_t secondary_stride, void *d_n_secondary, void *stream)
{
    if (!ctx || !d_bases || !d_quals || !d_offsets || !d_primary || !d_n_secondary || (secondary_stride && !d_secondary))
        return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_align_single_secondary_device: null argument");
    if (!ctx->secondary) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_enable_secondary has not been called on this context");
    if (n == 0) return SNAPGPU_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    int rc = launch_align(ctx, n, d_bases, d_quals, d_offsets, d_primary, d_first_alt, s, d_secondary, secondary_stride, d_n_secondary);
    if (rc) return rc;
    if (!stream) return finish_timing(ctx);
    return SNAPGPU_OK;
}

static int ensure_sec_stage(snapgpu_ctx *ctx, int which, size_t bytes) {
    if (ctx->sec_stage_cap[which] >= bytes) return 0;
    if (ctx->d_sec_stage[which]) (void)hipFree(ctx->d_sec_stage[which]);
    ctx->d_sec_stage[which] = nullptr; ctx->sec_stage_cap[which] = 0;
    size_t cap = bytes + bytes / 4 + 4096;
    HIPCHK(ctx, hipMalloc(&ctx->d_sec_stage[which], cap), SNAPGPU_E_NOMEM);
    ctx->sec_stage_cap[which] = cap;
    return 0;
}

extern "C" int snapgpu_align_single_secondary(snapgpu_ctx *ctx, uint32_t n, const char *bases, const char *quals,
                                              const uint64_t *offsets, snapgpu_single_result *primary, snapgpu_single_result *first_alt,
                                              snapgpu_single_result *secondary, uint32_t secondary_stride, uint32_t *n_secondary)
{
    if (!ctx || !bases || !quals || !offsets || !primary || !n_secondary || (secondary_stride && !secondary))
        return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_align_single_secondary: null argument");
    if (!ctx->secondary) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_enable_secondary has not been called on this context");
    if (n == 0) return SNAPGPU_OK;
    for (uint32_t i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i]) return fail(ctx, SNAPGPU_E_INVALID, "offsets must be non-decreasing");
        if (offsets[i + 1] - offsets[i] > ctx->params.max_read_len)
            return fail(ctx, SNAPGPU_E_INVALID, "read longer than max_read_len given at snapgpu_create (BaseAligner.cpp:354-358)");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    size_t nb = (size_t)offsets[n];
    const size_t sec_bytes = (size_t)n * secondary_stride * sizeof(snapgpu_single_result);
    int rc;
    if ((rc = ensure_stage(ctx, 0, nb + 16)) || (rc = ensure_stage(ctx, 1, nb + 16)) || (rc = ensure_stage(ctx, 2, ((size_t)n + 1) * 8)) ||
        (rc = ensure_stage(ctx, 3, (size_t)n * sizeof(snapgpu_single_result))) || (rc = ensure_stage(ctx, 4, (size_t)n * sizeof(snapgpu_single_result))) ||
        (rc = ensure_sec_stage(ctx, 0, sec_bytes + 16)) || (rc = ensure_sec_stage(ctx, 1, (size_t)n * 4))) return rc;
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage[0], bases, nb, hipMemcpyHostToDevice, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage[1], quals, nb, hipMemcpyHostToDevice, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage[2], offsets, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, s), SNAPGPU_E_LAUNCH);
    if (sec_bytes) HIPCHK(ctx, hipMemsetAsync(ctx->d_sec_stage[0], 0, sec_bytes, s), SNAPGPU_E_LAUNCH);
    rc = launch_align(ctx, n, ctx->d_stage[0], ctx->d_stage[1], ctx->d_stage[2], ctx->d_stage[3], first_alt ? ctx->d_stage[4] : nullptr, s,
                      ctx->d_sec_stage[0], secondary_stride, ctx->d_sec_stage[1]);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(primary, ctx->d_stage[3], (size_t)n * sizeof(snapgpu_single_result), hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    if (first_alt) HIPCHK(ctx, hipMemcpyAsync(first_alt, ctx->d_stage[4], (size_t)n * sizeof(snapgpu_single_result), hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    if (sec_bytes) HIPCHK(ctx, hipMemcpyAsync(secondary, ctx->d_sec_stage[0], sec_bytes, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(n_secondary, ctx->d_sec_stage[1], (size_t)n * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    if ((rc = finish_timing(ctx))) return rc;
    bool truncated = false;
    for (uint32_t i = 0; i < n; i++) {
        if (n_secondary[i] == 0xFFFFFFFFu) return fail(ctx, SNAPGPU_E_UNSUPPORTED, "a read produced more secondary candidates than 2 * (seeds + 1) * max_hits");
        if (n_secondary[i] > secondary_stride) truncated = true;
    }
    return truncated ? SNAPGPU_W_SECONDARY_TRUNCATED : SNAPGPU_OK;
}

// =====================================================================================
// paired end
// =====================================================================================

extern "C" void snapgpu_default_paired_params(snapgpu_paired_params *pp) {       // PairedAligner.cpp:55-57, 227-242; AlignerOptions.cpp:103-110
    memset(pp, 0, sizeof(*pp));
    pp->min_spacing = 0; pp->max_spacing = 1000; pp->force_spacing = 0; pp->max_big_hits = 4000; pp->max_candidate_pool_size = 1000000;
    pp->num_seeds = 8; pp->seed_coverage = 0.0; pp->max_k_for_indels = 40; pp->min_read_length = 50; pp->use_soft_clipping = 1;
    pp->flatten_mapq_at_or_below = 3; pp->min_score_realignment = 3; pp->min_score_gap_realignment_alt = 3; pp->min_ag_score_improvement = 24;
    pp->enable_hamming_scoring_base_aligner = 1; pp->max_single_seeds = 25;
}

// Builds the per-wave state of IntersectingPairedEndAligner + ChimericPairedEndAligner (PairedAligner.cpp:556-625).
// per-wave scratch slab of the paired-end kernel:
// [single-end: heads | buckets | AG traceback] [single AG candidates] [cand] [mate0] [mate1] [anchor] [paired AG candidates]
// and, for the secondary-result variant, [paired secondary list | order | keys] [single-end secondary list | keys | order]
static void paired_lay_out(PairedArgs &x, bool sec) {
    const AlignCfg &sc = x.scfg;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t ag_bytes = sc.ag_buffers ? ag_scratch_bytes(sc.RL) : 0;
    size_t off = up((size_t)sc.ht_size * 2 + (size_t)sc.pool_size * sizeof(Elem) + ag_bytes);
    x.off_single_agc = off; off += up((size_t)x.single_agc_cap * sizeof(snapgpu_single_result));
    x.off_cand = off;   off += up((size_t)x.pcfg.pool_size * sizeof(PECand));
    x.off_mate0 = off;  off += up((size_t)(x.pcfg.pool_size / 2 + 1) * sizeof(PEMate));
    x.off_mate1 = off;  off += up((size_t)(x.pcfg.pool_size / 2 + 1) * sizeof(PEMate));
    x.off_anchor = off; off += up((size_t)x.pcfg.pool_size * sizeof(PEAnchor));
    x.off_agc = off;    off += up((size_t)(x.pcfg.ag_cand_cap + 1) * sizeof(snapgpu_paired_result));
    x.off_agc_order = off; off += up((size_t)(x.pcfg.ag_cand_cap + 1) * 4);
    x.off_lv_big = off; off += up((size_t)lv_lds_bytes(x.kmax_lv, sc.RL) + 64);
    if (sec) {
        x.off_sec = off;     off += up((size_t)(x.pcfg.sec_cap + 1) * sizeof(snapgpu_paired_result));
        x.off_sec_ord = off; off += up((size_t)(x.pcfg.sec_cap + 1) * 4);
        x.off_sec_key = off; off += up((size_t)(2 * x.pcfg.sec_cap + 2) * 4);
        x.off_ssec = off;    off += up((size_t)x.ssec_cfg.cap * (sizeof(snapgpu_single_result) + 12));
    }
    x.stride = off;
}

// The paired-end kernel with secondary results has its own (fewer, larger) slabs, so that the default path is untouched.
static int setup_paired_secondary(snapgpu_ctx *ctx)
{
    if (ctx->d_pscratch_sec) { (void)hipFree(ctx->d_pscratch_sec); ctx->d_pscratch_sec = nullptr; }
    if (ctx->d_pscratch_sec_big) { (void)hipFree(ctx->d_pscratch_sec_big); ctx->d_pscratch_sec_big = nullptr; }
    ctx->paired_sec = false;
    if (!ctx->paired || !ctx->secondary) return SNAPGPU_OK;
    if (ctx->sec_cfg.adjust)
        return fail(ctx, SNAPGPU_E_UNSUPPORTED, "-ae (AlignmentAdjuster before the -om filter) is implemented for BaseAligner::AlignRead, not for the paired-end aligners (IntersectingPairedEndAligner.cpp:1298-1320)");
    PairedArgs &a = ctx->pargs_sec;
    a = ctx->pargs;
    a.pcfg.om = ctx->sec_cfg.om; a.pcfg.mpc = ctx->sec_cfg.mpc; a.pcfg.omax = ctx->sec_cfg.omax;
    a.pcfg.sec_cap = 4096;
    // the single-end aligner of the chimeric fallback: at most one entry per ScoreSet per scored candidate (see snapgpu_enable_secondary)
    uint64_t seeds = a.scfg.num_seeds ? a.scfg.num_seeds : (uint64_t)(2 * a.scfg.seed_coverage * ctx->params.max_read_len / ctx->ix.seed_len) + 1;
    uint64_t cap = 2 * (seeds + 1) * a.scfg.max_hits + 2;
    if (cap > (1u << 20)) return fail(ctx, SNAPGPU_E_UNSUPPORTED, "2 * seeds * max_hits > 2^20 secondary candidates per read is not supported");
    a.ssec_cfg = SecCfg{ctx->sec_cfg.om, ctx->sec_cfg.mpc, ctx->sec_cfg.omax, (uint32_t)cap};
    paired_lay_out(a, true);
    PairedArgs &big = ctx->pargs_sec_big;
    big = a;
    big.pcfg.ag_cand_cap = a.pcfg.ag_cand_cap * 8;
    big.single_agc_cap = a.single_agc_cap * 32;
    big.pcfg.sec_cap = a.pcfg.sec_cap * 32;
    paired_lay_out(big, true);
    size_t free_b = 0, total_b = 0;
    HIPCHK(ctx, hipMemGetInfo(&free_b, &total_b), SNAPGPU_E_NODEVICE);
    uint32_t slots = ctx->p_wave_slots;
    while (slots > 64 && (size_t)slots * a.stride > free_b / 2) slots /= 2;
    slots &= ~3u;
    if ((size_t)slots * a.stride > free_b / 2) return fail(ctx, SNAPGPU_E_NOMEM, "not enough device memory for the paired-end secondary-result lists");
    ctx->p_sec_slots = slots;
    ctx->p_sec_big_slots = 64;
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_pscratch_sec, (size_t)slots * a.stride), SNAPGPU_E_NOMEM);
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_pscratch_sec_big, (size_t)ctx->p_sec_big_slots * big.stride), SNAPGPU_E_NOMEM);
    for (uint32_t w = 0; w < slots; w++)              // only the single-end head tables must start zeroed
        HIPCHK(ctx, hipMemsetAsync(ctx->d_pscratch_sec + (size_t)w * a.stride, 0, (size_t)a.scfg.ht_size * 2, ctx->stream), SNAPGPU_E_NODEVICE);
    for (uint32_t w = 0; w < ctx->p_sec_big_slots; w++)
        HIPCHK(ctx, hipMemsetAsync(ctx->d_pscratch_sec_big + (size_t)w * big.stride, 0, (size_t)a.scfg.ht_size * 2, ctx->stream), SNAPGPU_E_NODEVICE);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream), SNAPGPU_E_NODEVICE);
    a.scratch = ctx->d_pscratch_sec;
    big.scratch = ctx->d_pscratch_sec_big;
    ctx->paired_sec = true;
    return SNAPGPU_OK;
}

extern "C" int snapgpu_enable_paired(snapgpu_ctx *ctx, const snapgpu_paired_params *pp)
{
    if (!ctx || !pp) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_enable_paired: null argument");
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    const snapgpu_params &p = ctx->params;
    if (pp->max_spacing > 100000) return fail(ctx, SNAPGPU_E_INVALID, "max_spacing out of range");
    if (ctx->d_pscratch) { (void)hipFree(ctx->d_pscratch); ctx->d_pscratch = nullptr; }
    if (ctx->d_pscratch_big) { (void)hipFree(ctx->d_pscratch_big); ctx->d_pscratch_big = nullptr; }
    ctx->paired = false;
    ctx->pparams = *pp;
    PairedArgs &a = ctx->pargs;
    memset(&a, 0, sizeof(a));
    a.ix = ctx->ix; a.tab = ctx->d_tab;

    // the single-end aligner inside ChimericPairedEndAligner: maxK/2, maxSeedsSingleEnd (ChimericPairedEndAligner.cpp:81-88)
    AlignCfg sc = ctx->cfg;
    sc.max_k = p.max_k / 2;
    sc.num_seeds = pp->max_single_seeds;
    uint32_t max_seeds_ctor = sc.num_seeds != 0 ? sc.num_seeds : (uint32_t)(int)(p.seed_coverage * 1000 / ctx->ix.seed_len);
    sc.num_weight_lists = max_seeds_ctor + 1;
    if (sc.num_weight_lists < 2 || sc.num_weight_lists > 0x3FF) return fail(ctx, SNAPGPU_E_UNSUPPORTED, "number of single-end seeds out of the supported range [1, 1022]");
    uint64_t spool = (uint64_t)p.max_hits * max_seeds_ctor;
    if (spool < 64) spool = 64;
    if (spool > 60000) return fail(ctx, SNAPGPU_E_UNSUPPORTED, "max_hits * max_single_seeds > 60000 candidate buckets per read is not supported");
    sc.pool_size = (uint32_t)spool;
    sc.ht_size = next_pow2((uint32_t)spool * 2);
    // LV limits: computeScoreLimit <= min(126, extraSearchDepth + maxK + maxKForIndels - 1) (IntersectingPairedEndAligner.cpp:3975-3988)
    uint32_t kmax_lv = p.max_k + p.extra_search_depth + (pp->max_k_for_indels ? pp->max_k_for_indels - 1 : 0);
    if (kmax_lv > 126) kmax_lv = 126;
    // The LDS triangle serves limits up to sc.kmax; the calls beyond that (indel-hinted candidates only) use a per-wave HBM buffer
    // (paired_dev.h: DevPL::lv).  22 = the smallest kmax whose LDS block (2 232 bytes) also holds the 512 counters of the Phase-4 counting
    // sort; a limit the single-end aligner of the fallback can reach (max_k + extra_search_depth) always stays in LDS.
    uint32_t kmax_lds = p.max_k + p.extra_search_depth; if (kmax_lds < 22) kmax_lds = 22;
    if (kmax_lds > kmax_lv) kmax_lds = kmax_lv;
    if (kmax_lv < 22) kmax_lv = kmax_lds = 22;
    if (const char *e = getenv("SNAPGPU_PAIRED_LV_LDS_KMAX")) { uint32_t v = (uint32_t)atoi(e); if (v >= 22 && v <= kmax_lv) kmax_lds = v; }
    sc.kmax = kmax_lds;
    a.kmax_lv = kmax_lv;
    sc.ag_buffers = (sc.use_ag || (pp->use_soft_clipping && pp->enable_hamming_scoring_base_aligner)) ? 1u : 0u;
    a.scfg = sc;

    PECfg &c = a.pcfg;
    memset(&c, 0, sizeof(c));
    c.proj = ctx->proj;
    c.max_k = (int)p.max_k; c.extra_depth = (int)p.extra_search_depth; c.max_k_for_indels = (int)pp->max_k_for_indels;
    c.max_gap_alt = p.max_score_gap_to_prefer_non_alt; c.use_ag = p.use_affine_gap ? 1 : 0; c.alt_aware = p.alt_awareness ? 1 : 0;
    c.emit_alt = p.emit_alt_alignments ? 1 : 0; c.use_soft_clip = pp->use_soft_clipping ? 1 : 0; c.force_spacing = pp->force_spacing ? 1 : 0;
    c.match_reward = (int)p.match_reward; c.sub_penalty = (int)p.sub_penalty; c.gap_open = (int)p.gap_open_penalty;
    c.gap_extend = (int)p.gap_extend_penalty; c.five_bonus = (int)p.five_prime_end_bonus; c.three_bonus = (int)p.three_prime_end_bonus;
    c.min_spacing = pp->min_spacing; c.max_spacing = pp->max_spacing; c.max_big_hits = pp->max_big_hits;
    c.num_seeds = pp->num_seeds < PE_MAX_SEEDS ? pp->num_seeds : PE_MAX_SEEDS;                       // __min(MAX_MAX_SEEDS, ...), :61
    c.seed_coverage = pp->seed_coverage; c.min_read_length = pp->min_read_length; c.flatten_mapq = pp->flatten_mapq_at_or_below;
    c.min_score_realign = pp->min_score_realignment; c.min_score_gap_realign_alt = pp->min_score_gap_realignment_alt;
    c.min_ag_improve = pp->min_ag_score_improvement; c.enable_hamming_base = pp->enable_hamming_scoring_base_aligner ? 1 : 0;
    c.seed_len = (int)ctx->ix.seed_len;
    c.om = -1; c.mpc = -1; c.omax = 0x7fffffff; c.sec_cap = 0;              // no secondary results on the default path (setup_paired_secondary)
    // maxSeedsToUse of the constructor (:69-74) sizes the pools (:141)
    uint32_t ctor_seeds = c.num_seeds != 0 ? c.num_seeds : (uint32_t)(1000 * pp->seed_coverage / ctx->ix.seed_len);
    if (ctor_seeds < 1) ctor_seeds = 1;
    c.max_seeds = ctor_seeds < PE_MAX_SEEDS ? ctor_seeds : PE_MAX_SEEDS;
    if (c.num_seeds == 0) c.max_seeds = PE_MAX_SEEDS;
    uint64_t pool = (uint64_t)pp->max_big_hits * ctor_seeds * 2;
    if (pool > pp->max_candidate_pool_size) pool = pp->max_candidate_pool_size;
    if (const char *e = getenv("SNAPGPU_PAIRED_POOL")) { uint64_t v = strtoull(e, nullptr, 10); if (v >= 64 && v < pool) pool = v; }
    if (pool < 64) pool = 64;
    c.pool_size = (uint32_t)pool;
    // PairedAligner.cpp:571 starts at 4096 and doubles on overflow.  First pass: 16384 (3.4 MB per wave), so that all but the very
    // heaviest pairs finish alongside the rest of the batch instead of serially in the second pass; second pass: 8x that.
    uint32_t first_cap = 16384;
    if (const char *e = getenv("SNAPGPU_PAIRED_AGC_CAP")) { uint32_t v = (uint32_t)strtoul(e, nullptr, 10); if (v >= 64 && v <= (1u << 20)) first_cap = v; }
    c.ag_cand_cap = p.use_affine_gap ? first_cap : 0;
    a.single_agc_cap = p.use_affine_gap ? 4096 : 0;
    a.max_k_paired = (int32_t)p.max_k; a.max_k_single = (int32_t)(p.max_k / 2);

    {   // affine-gap kernel variant: limits go up to MAX_K - 1 on this path (gapless-clipped reads, IntersectingPairedEndAligner.cpp:2577)
        int need = ag_max_positions((int)p.max_read_len - (int)ctx->ix.seed_len, 126);
        int need2 = ag_max_positions((int)p.max_read_len, 126);
        if (need2 > need) need = need2;
        ctx->p_ag_variant = need <= 192 ? 3 : need <= 256 ? 4 : need <= 384 ? 6 : 0;
        if (getenv("SNAPGPU_AG_LDS")) ctx->p_ag_variant = 0;
    }
    a.scfg.ag_lds = !sc.ag_buffers ? 0u : (ctx->p_ag_variant == 3 ? ag_lds_bytes_reg(sc.RL, 3) : ag_lds_bytes(sc.RL));
    sc.ag_lds = a.scfg.ag_lds;

    paired_lay_out(a, false);
    // The reference doubles its affine-gap candidate buffers when they overflow and aligns the pair again
    // (PairedAligner.cpp:727-779).  Here: pairs that overflow the first pass's buffers are redone by a second launch of a
    // few waves whose buffers are 32x larger; only what still does not fit is reported.
    PairedArgs &big = ctx->pargs_big;
    big = a;
    big.pcfg.ag_cand_cap = a.pcfg.ag_cand_cap * 8;
    big.single_agc_cap = a.single_agc_cap * 32;
    paired_lay_out(big, false);

    LdsLayout SL = lds_layout(sc.RL, sc.num_weight_lists, sc.kmax, sc.ag_lds);
    PairedLds PL = paired_lds_layout(SL.total, sc.RL, c.max_seeds);
    ctx->p_lds_per_wave = PL.total;
    if ((size_t)4 * PL.total > 160 * 1024) return fail(ctx, SNAPGPU_E_UNSUPPORTED, "per-pair LDS state exceeds 40 KiB per wave");
    int waves_per_cu = 4 * SNAPGPU_PAIRED_WAVES_PER_SIMD(ctx->p_ag_variant);           // (paired_args.h: what k_align_paired is compiled for)
    if (const char *e = getenv("SNAPGPU_PAIRED_WAVES_PER_CU")) { int v = atoi(e); if (v >= 4 && v <= 16) waves_per_cu = v & ~3; }
    while (waves_per_cu > 4 && (size_t)waves_per_cu * PL.total > 160 * 1024) waves_per_cu -= 4;
    size_t free_b = 0, total_b = 0;
    HIPCHK(ctx, hipMemGetInfo(&free_b, &total_b), SNAPGPU_E_NODEVICE);
    uint32_t slots = (uint32_t)ctx->num_cus * (uint32_t)waves_per_cu;
    while (slots > 64 && (size_t)slots * a.stride > free_b / 2) slots /= 2;                        // never take more than half of what is free
    slots &= ~3u;
    if ((size_t)slots * a.stride > free_b / 2) return fail(ctx, SNAPGPU_E_NOMEM, "not enough device memory for the paired-end candidate pools (lower -H / -mcp or set SNAPGPU_PAIRED_POOL)");
    ctx->p_wave_slots = slots;
    if (getenv("SNAPGPU_VERBOSE")) fprintf(stderr, "snapgpu: paired-end context: %u wave slots (%d per CU asked) x %.2f MB of pools, %.1f GB free before; second-pass slabs %.1f MB each; LDS %u B per wave\n", slots, waves_per_cu, a.stride / 1e6, free_b / 1e9, big.stride / 1e6, (unsigned)PL.total);
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_pscratch, (size_t)slots * a.stride), SNAPGPU_E_NOMEM);
    // only the single-end head tables must start zeroed
    for (uint32_t w = 0; w < slots; w++)
        HIPCHK(ctx, hipMemsetAsync(ctx->d_pscratch + (size_t)w * a.stride, 0, (size_t)sc.ht_size * 2, ctx->stream), SNAPGPU_E_NODEVICE);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream), SNAPGPU_E_NODEVICE);
    a.scratch = ctx->d_pscratch;
    ctx->p_big_slots = 256;               // one wave per flagged pair in all but the worst batches (each such pair is seconds of serial work)
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_pscratch_big, (size_t)ctx->p_big_slots * big.stride), SNAPGPU_E_NOMEM);
    for (uint32_t w = 0; w < ctx->p_big_slots; w++)
        HIPCHK(ctx, hipMemsetAsync(ctx->d_pscratch_big + (size_t)w * big.stride, 0, (size_t)sc.ht_size * 2, ctx->stream), SNAPGPU_E_NODEVICE);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream), SNAPGPU_E_NODEVICE);
    big.scratch = ctx->d_pscratch_big;
    if (ctx->d_pexact_persist) { (void)hipFree(ctx->d_pexact_persist); ctx->d_pexact_persist = nullptr; }
    if (ctx->d_help) { (void)hipFree(ctx->d_help); ctx->d_help = nullptr; }
    if (ctx->d_help_spec) { (void)hipFree(ctx->d_help_spec); ctx->d_help_spec = nullptr; }
    // Phase-4 lists at least this long are offered to idle waves once a wave has run out of pairs (on-demand publication, paired_dev.h).
    // On by default at 64 candidates since round 3: measured on the bench batch (256 Mb / 500 k pairs, profiles/r03a) 8.56 -> 5.66 s per
    // launch with one context, same bytes, 500 k pairs equal to the reference.  SNAPGPU_PAIRED_HELP_MIN=<n> changes it, 0 turns it off.
    ctx->help_min = 64;
    if (const char *e = getenv("SNAPGPU_PAIRED_HELP_MIN")) ctx->help_min = (uint32_t)strtoul(e, nullptr, 10);
    if (p.use_affine_gap && ctx->help_min != 0) {
        ctx->n_help = 32;
        ctx->help_spec_cap = big.pcfg.ag_cand_cap;
        ctx->help_bytes = 64 + (size_t)ctx->n_help * sizeof(PEHelpSlot);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_help, ctx->help_bytes), SNAPGPU_E_NOMEM);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_help_spec, (size_t)ctx->n_help * ctx->help_spec_cap * sizeof(PEHelpSpec)), SNAPGPU_E_NOMEM);
    }
    if (p.use_affine_gap) {             // four traceback-array images per wave: every wave (192-position variant) or 64 replay waves
        // (SNAPGPU_PAIRED_ALWAYS_EXACT=1: the exact kernel as the main pass, as on the single-end path.  Not the default: on the 256 Mb /
        //  500 k-pair bench batch it stops with a memory access fault that smaller batches and genomes do not show -- profiles/r02g.)
        ctx->p_always_exact = ctx->p_ag_variant == 3 && getenv("SNAPGPU_PAIRED_ALWAYS_EXACT") != nullptr && atoi(getenv("SNAPGPU_PAIRED_ALWAYS_EXACT")) != 0;
        ctx->pexact_slots = ctx->p_always_exact ? (ctx->p_wave_slots > ctx->p_big_slots ? ctx->p_wave_slots : ctx->p_big_slots) : 64;
        ctx->pexact_persist_stride = 4 * (uint64_t)((ag_scratch_bytes(sc.RL) + 255) & ~(size_t)255);
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_pexact_persist, (size_t)ctx->pexact_slots * ctx->pexact_persist_stride), SNAPGPU_E_NOMEM);
        HIPCHK(ctx, hipMemsetAsync(ctx->d_pexact_persist, 0, (size_t)ctx->pexact_slots * ctx->pexact_persist_stride, ctx->stream), SNAPGPU_E_NODEVICE);
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream), SNAPGPU_E_NODEVICE);
    }
    ctx->heavy_first = getenv("SNAPGPU_PAIRED_HEAVY_FIRST") != nullptr && atoi(getenv("SNAPGPU_PAIRED_HEAVY_FIRST")) != 0;
    if (const char *e = getenv("SNAPGPU_PAIRED_REPLAY_BESIDE")) ctx->replay_beside = atoi(e) != 0 ? 1 : 0;
    if (!ctx->replay_stream && ctx->replay_beside == 1) {      // (only a context that asked for the replay beside the main pass has the second stream: see snapgpu_hw_queues)
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->replay_stream, hipStreamNonBlocking), SNAPGPU_E_NODEVICE);
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming), SNAPGPU_E_NODEVICE);
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming), SNAPGPU_E_NODEVICE);
    }
    ctx->paired = true;
    return setup_paired_secondary(ctx);
}

struct PairedSecOut {          // device buffers of a call that wants secondary results
    void *secondary; uint32_t stride; void *n_secondary;
    void *single_secondary; uint32_t single_stride; void *n_single_secondary;
};

// the exact twin of the context's paired-end kernel variant (paired_k.hip)
static void launch_paired_exact(int variant, const PairedArgs *x, uint32_t blocks, size_t lds, hipStream_t s) {
    switch (variant) {
    case 3:  snapgpu_launch_paired_exact_3(x, blocks, lds, s); break;
    case 4:  snapgpu_launch_paired_exact_4(x, blocks, lds, s); break;
    case 6:  snapgpu_launch_paired_exact_6(x, blocks, lds, s); break;
    default: snapgpu_launch_paired_exact_0(x, blocks, lds, s); break;
    }
}

// Paired-end calls in flight on each device, and the part of the chip one of them asks for (round 6, profiles/r06q - r06w).  The paired kernel
// is a persistent grid: one that asks for every wave slot of the chip while other feeders' kernels are resident has blocks PENDING until
// those kernels end.  The kernel-trace timeline of three feeders showed the chip with one kernel running 27 % of the time and replays running
// alone; with each grid sized to a share (`slots / calls in flight`, at most three shares) every feeder's kernel is resident at once: 357 - 466 k
// -> 586 - 604 k reads/s paired, 187 -> 267 k configs[4] (256 Mb, three feeders).  Half of that was the hardware queues (snapgpu_hw_queues): a
// pending block holds back whatever shares ITS queue.  With one stream per context on a queue of its own, a launch asks for 1.5 shares: the
// pending half fills the slots a feeder leaves idle while its replay (a few waves, 0.5 - 2 s) runs -- against exact shares +4.5 % / +5.5 % paired
// and +7.5 % configs[4] at 256 Mb / 3 100 Mb (profiles/r06w; 2 shares and the whole chip: better at one genome size, worse at the other).
// Single-end launches keep whole-chip grids: their calls are one short kernel each, and smaller grids measured slower (8 of 24 waves per CU each:
// 14.96 -> 12.86 M reads/s, profiles/r06s).  SNAPGPU_PAIRED_GRID_SHARE=<n> fixes the divisor, SNAPGPU_PAIRED_GRID_OVER=<x> the shares asked
// for (SHARE=1: every launch asks for the whole chip, as before).  Calls on a caller's stream return before their kernels end and so count only
// while they are being enqueued.
struct PairedGate {                      // per device
    std::mutex mu; int inflight = 0, peak = 0; std::chrono::steady_clock::time_point peak_at;
};
static PairedGate g_paired_gate[64];
struct PairedInFlight {
    PairedGate *g = nullptr; int share = 1;
    explicit PairedInFlight(const snapgpu_ctx *ctx) {
        if (!ctx || ctx->device < 0 || ctx->device >= 64) return;
        g = &g_paired_gate[ctx->device];
        const auto now = std::chrono::steady_clock::now();
        std::lock_guard<std::mutex> lk(g->mu);
        const int cur = ++g->inflight;
        // feeders that start together (a barrier, the first batches of a run) would see 1, 2, 3 ... calls in flight: the concurrency of the
        // last two seconds stands in for what the count is about to become
        const bool fresh = g->peak > 0 && std::chrono::duration<double>(now - g->peak_at).count() < 2.0;
        share = fresh && g->peak > cur ? g->peak : cur;
        if (cur >= g->peak || !fresh) { g->peak = cur; g->peak_at = now; }
    }
    ~PairedInFlight() {
        if (!g) return;
        std::lock_guard<std::mutex> lk(g->mu);
        if (g->inflight >= g->peak) { g->peak = g->inflight; g->peak_at = std::chrono::steady_clock::now(); }
        --g->inflight;
    }
    PairedInFlight(const PairedInFlight &) = delete; PairedInFlight &operator=(const PairedInFlight &) = delete;
};
static uint32_t paired_grid_share(const snapgpu_ctx *ctx, uint32_t all_blocks) {
    int share = ctx->paired_share;
    if (share > 3) share = 3;
    if (const char *e = getenv("SNAPGPU_PAIRED_GRID_SHARE")) { int v = atoi(e); if (v >= 1 && v <= 16) share = v; }
    if (share < 1) share = 1;
    double over = 1.5;                  // shares asked for (above)
    if (const char *e = getenv("SNAPGPU_PAIRED_GRID_OVER")) { double v = atof(e); if (v >= 0.25 && v <= 4.0) over = v; }
    double f = over / (double)share; if (f > 1.0) f = 1.0;
    const uint32_t b = (uint32_t)((double)all_blocks * f + 0.999);
    return b ? (b > all_blocks ? all_blocks : b) : 1;
}

static int launch_paired(snapgpu_ctx *ctx, uint32_t n, const void *d_bases, const void *d_quals, const void *d_offsets,
                         void *d_primary, void *d_first_alt, hipStream_t s, const PairedSecOut *so = nullptr)
{
    PairedArgs a = so ? ctx->pargs_sec : ctx->pargs;
    a.ix = ctx->ix;                    // (the device-native tables may have been (re)built since snapgpu_enable_paired)
    if (so) {
        a.secondary = (snapgpu_paired_result *)so->secondary; a.sec_out_stride = so->stride; a.n_secondary = (uint32_t *)so->n_secondary;
        a.single_secondary = (snapgpu_single_result *)so->single_secondary; a.ssec_out_stride = so->single_stride;
        a.n_single_secondary = (uint32_t *)so->n_single_secondary;
    }
    a.bases = (const uint8_t *)d_bases; a.quals = (const uint8_t *)d_quals; a.offsets = (const uint64_t *)d_offsets;
    a.n_pairs = n; a.primary = (snapgpu_paired_result *)d_primary; a.first_alt = (snapgpu_paired_result *)d_first_alt;
    a.work_counter = ctx->d_work; a.counters = ctx->d_counters;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_work, 0, 4, s), SNAPGPU_E_LAUNCH);
    if (ctx->flag_list_cap < n) {
        if (ctx->d_flag_list) (void)hipFree(ctx->d_flag_list);
        ctx->d_flag_list = nullptr; ctx->flag_list_cap = 0;
        size_t cap = (size_t)n + n / 4 + 1024;
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_flag_list, cap * 8), SNAPGPU_E_NOMEM);     // (second half: launch_paired's list of pairs for the exact kernel beside the main pass)
        ctx->flag_list_cap = cap;
    }
    uint32_t blocks = paired_grid_share(ctx, (so ? ctx->p_sec_slots : ctx->p_wave_slots) / 4);
    uint32_t need = (n + 3) / 4; if (blocks > need) blocks = need;
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s), SNAPGPU_E_LAUNCH);
    const size_t lds = (size_t)4 * ctx->p_lds_per_wave;
    // (the secondary-results kernels have fewer slots of their own: they keep the fast pass + replay scheme)
    const bool p_always = ctx->p_always_exact && ctx->d_pexact_persist != nullptr && !so;
    auto launch = [&](const PairedArgs &x0, uint32_t nblocks) {
        if (p_always) {                        // one exact pass (plus the large-buffer pass for pairs that overflowed), nothing to replay
            PairedArgs x = x0;
            x.persist = ctx->d_pexact_persist; x.persist_stride = ctx->pexact_persist_stride;
            snapgpu_launch_paired_exact_3(&x, nblocks, lds, s);
            return;
        }
        const PairedArgs &x = x0;
        if (so) {                              // secondary results: the 192-position register variant, or the LDS form for everything longer
            if (ctx->p_ag_variant == 3) snapgpu_launch_paired_sec_3(&x, nblocks, lds, s);
            else snapgpu_launch_paired_sec_0(&x, nblocks, lds, s);
            return;
        }
        switch (ctx->p_ag_variant) {           // one translation unit per affine-gap variant (paired_k.hip), compiled in parallel
        case 3:  snapgpu_launch_paired_3(&x, nblocks, lds, s); break;
        case 4:  snapgpu_launch_paired_4(&x, nblocks, lds, s); break;
        case 6:  snapgpu_launch_paired_6(&x, nblocks, lds, s); break;
        default: snapgpu_launch_paired_0(&x, nblocks, lds, s); break;
        }
    };
    a.is_replay = 0;
    if (ctx->heavy_first && !so) {         // experimental (order.h): heaviest pairs first, through the remap list
        const int orc = launch_unit_order(ctx, d_bases, d_offsets, n, 2, a.pcfg.max_big_hits, s);
        if (orc) return orc;
        a.remap = ctx->d_order; a.n_remap = ctx->d_whist + 33;
    }
    auto with_help = [&](PairedArgs &x) -> hipError_t {       // fresh help state for the launch that is about to start
        x.help = nullptr; x.n_help = 0; x.help_spec = nullptr; x.help_spec_cap = 0; x.help_done = nullptr; x.help_min = 0xffffffffu; x.help_eager = 0;
        if (!ctx->d_help || so) return hipSuccess;
        x.help = (PEHelpSlot *)(ctx->d_help + 64); x.n_help = ctx->n_help; x.help_spec = ctx->d_help_spec; x.help_spec_cap = ctx->help_spec_cap;
        x.help_done = getenv("SNAPGPU_PAIRED_HELP_NOHELPERS") ? nullptr : (uint32_t *)ctx->d_help; x.help_min = ctx->help_min;      // (debug: owner-only speculation)
        x.help_eager = (getenv("SNAPGPU_PAIRED_HELP_EAGER") != nullptr && atoi(getenv("SNAPGPU_PAIRED_HELP_EAGER")) != 0) || x.help_done == nullptr;
        return hipMemsetAsync(ctx->d_help, 0, ctx->help_bytes, s);
    };
    HIPCHK(ctx, with_help(a), SNAPGPU_E_LAUNCH);
    PairedArgs b = so ? ctx->pargs_sec_big : ctx->pargs_big;       // the second / third pass: a few waves over large slabs
    b.ix = ctx->ix;
    b.secondary = a.secondary; b.sec_out_stride = a.sec_out_stride; b.n_secondary = a.n_secondary;
    b.single_secondary = a.single_secondary; b.ssec_out_stride = a.ssec_out_stride; b.n_single_secondary = a.n_single_secondary;
    b.bases = a.bases; b.quals = a.quals; b.offsets = a.offsets; b.n_pairs = n; b.primary = a.primary; b.first_alt = a.first_alt;
    b.counters = a.counters; b.is_replay = 1; b.rq = nullptr; b.rq_list = nullptr; b.rq_mode = 0;
    // The exact replay BESIDE the main pass (PairedArgs::rq), SNAPGPU_PAIRED_REPLAY_BESIDE=1: flagged pairs are redone while the main pass
    // is still running instead of in a launch of their own after it (6 pairs, 1.8 s of a 5.6 s call on the bench batch, profiles/r03z).
    // OFF by default, for two measured reasons (profiles/r03p): (1) the pairs that get flagged are the heaviest ones, which the main pass
    // finishes last, so little of the replay overlaps it -- one context 5.92 -> 5.53 s per 500 k pairs, not the 1.8 s; (2) a kernel that
    // waits for another kernel is only safe while the two sit in different hardware queues: with three feeders (six streams over the
    // runtime's four queues) a waiting exact kernel ended up in front of another context's main kernel and the launches advanced only by
    // the list's watchdog.  Never used with more than one feeder, whatever the variable says.
    bool beside = !so && !p_always && ctx->d_pexact_persist != nullptr && ctx->replay_stream != nullptr && !getenv("SNAPGPU_NO_EXACT_REPLAY") &&
                  ctx->replay_beside == 1 && ctx->feeders->load() <= 1;
    a.rq = nullptr; a.rq_list = nullptr; a.rq_mode = 0;
    a.dbg_flag_every = b.dbg_flag_every = 0;
    if (const char *e = getenv("SNAPGPU_DEBUG_PAIRED_FLAG_EVERY")) a.dbg_flag_every = b.dbg_flag_every = (uint32_t)strtoul(e, nullptr, 10);
    PairedArgs xr = b;
    uint32_t xr_blocks = 0;
    if (beside) {
        a.rq = ctx->d_work + 8; a.rq_list = ctx->d_flag_list + ctx->flag_list_cap; a.rq_mode = 1;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_work + 8, 0, 16, s), SNAPGPU_E_LAUNCH);
        HIPCHK(ctx, hipMemsetAsync(a.rq_list, 0xff, (size_t)n * 4, s), SNAPGPU_E_LAUNCH);
        xr.rq = a.rq; xr.rq_list = a.rq_list; xr.rq_mode = 2;
        xr.work_counter = ctx->d_work + 3; xr.remap = nullptr; xr.n_remap = nullptr;
        xr.persist = ctx->d_pexact_persist; xr.persist_stride = ctx->pexact_persist_stride;
        xr.help = nullptr; xr.n_help = 0; xr.help_spec = nullptr; xr.help_spec_cap = 0; xr.help_done = nullptr; xr.help_min = 0xffffffffu; xr.help_eager = 0;
        uint32_t slots = ctx->p_big_slots; if (slots > ctx->pexact_slots) slots = ctx->pexact_slots;
        xr_blocks = slots / 4;
        // both kernels resident from the start: the exact kernel's blocks have the footprint of the main kernel's (same launch bounds, same
        // LDS), so the main grid leaves them their wave slots
        const uint32_t all_blocks = ctx->p_wave_slots / 4;
        if (blocks + xr_blocks > all_blocks && all_blocks > 2 * xr_blocks) blocks = all_blocks - xr_blocks;
    }
    auto launch_beside = [&](hipStream_t st) {
        launch_paired_exact(ctx->p_ag_variant, &xr, xr_blocks, lds, st);
    };
#ifndef SNAPGPU_WAVE_EMU
    if (beside) {                      // fork: the exact kernel goes first, on its own stream, behind everything `s` has been given so far
        HIPCHK(ctx, hipEventRecord(ctx->ev_fork, s), SNAPGPU_E_LAUNCH);
        HIPCHK(ctx, hipStreamWaitEvent(ctx->replay_stream, ctx->ev_fork, 0), SNAPGPU_E_LAUNCH);
        launch_beside(ctx->replay_stream);
        HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
        HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->replay_stream), SNAPGPU_E_LAUNCH);
    }
#endif
    launch(a, blocks);
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    if (beside) {
#ifdef SNAPGPU_WAVE_EMU
        launch_beside(s);              // (the emulator runs a kernel to its end at the launch: the list is complete when this one starts)
#else
        HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0), SNAPGPU_E_LAUNCH);        // join
#endif
    }
    {   // second pass over the pairs the first flagged (usually none: the launch then ends at once)
        uint32_t *d_count = ctx->d_work + 2, *d_work2 = ctx->d_work + 1;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_work + 1, 0, 8, s), SNAPGPU_E_LAUNCH);
        snapgpu_launch_collect_flagged((snapgpu_paired_result *)d_primary, n, ctx->d_flag_list, d_count, 0, s);
        b.work_counter = d_work2; b.remap = ctx->d_flag_list; b.n_remap = d_count;
        HIPCHK(ctx, with_help(b), SNAPGPU_E_LAUNCH);
        launch(b, (so ? ctx->p_sec_big_slots : ctx->p_big_slots) / 4);
        HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
        // third pass: pairs whose banded affine-gap traceback left the band are redone the way a newly constructed reference aligner
        // would do them (paired_dev.h: EXACT), in the large slabs of the second pass
        if (!p_always && ctx->d_pexact_persist && !getenv("SNAPGPU_NO_EXACT_REPLAY")) {
            uint32_t *d_count3 = ctx->d_work + 4, *d_work3 = ctx->d_work + 3;
            HIPCHK(ctx, hipMemsetAsync(ctx->d_work + 3, 0, 8, s), SNAPGPU_E_LAUNCH);
            snapgpu_launch_collect_flagged((snapgpu_paired_result *)d_primary, n, ctx->d_flag_list, d_count3, 1, s);
            PairedArgs x = b;
            x.work_counter = d_work3; x.remap = ctx->d_flag_list; x.n_remap = d_count3;
            x.persist = ctx->d_pexact_persist; x.persist_stride = ctx->pexact_persist_stride;
            x.help = nullptr; x.n_help = 0; x.help_done = nullptr; x.help_min = 0xffffffffu;
            uint32_t slots = so ? ctx->p_sec_big_slots : ctx->p_big_slots;
            if (slots > ctx->pexact_slots) slots = ctx->pexact_slots;
            if (so) { if (ctx->p_ag_variant == 3) snapgpu_launch_paired_sec_exact_3(&x, slots / 4, lds, s); else snapgpu_launch_paired_sec_exact_0(&x, slots / 4, lds, s); }
            else launch_paired_exact(ctx->p_ag_variant, &x, slots / 4, lds, s);
        }
    }
    HIPCHK(ctx, hipGetLastError(), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s), SNAPGPU_E_LAUNCH);
    return SNAPGPU_OK;
}

extern "C" int snapgpu_align_paired_device(snapgpu_ctx *ctx, uint32_t n_pairs, const void *d_bases, const void *d_quals,
                                           const void *d_offsets, void *d_primary, void *d_first_alt, void *stream)
{
    if (!ctx || !d_bases || !d_quals || !d_offsets || !d_primary) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_align_paired_device: null argument");
    if (!ctx->paired) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_enable_paired has not been called on this context");
    if (n_pairs == 0) return SNAPGPU_OK;
    const PairedInFlight in_flight(ctx); ctx->paired_share = in_flight.share;      // (paired_grid_share: this call's launches ask for their share of the chip)
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    int rc = launch_paired(ctx, n_pairs, d_bases, d_quals, d_offsets, d_primary, d_first_alt, s);
    if (rc) return rc;
    if (!stream) return finish_timing(ctx);
    return SNAPGPU_OK;
}

extern "C" int snapgpu_align_paired(snapgpu_ctx *ctx, uint32_t n_pairs, const char *bases, const char *quals,
                                    const uint64_t *offsets, snapgpu_paired_result *primary, snapgpu_paired_result *first_alt)
{
    if (!ctx || !bases || !quals || !offsets || !primary) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_align_paired: null argument");
    if (!ctx->paired) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_enable_paired has not been called on this context");
    if (n_pairs == 0) return SNAPGPU_OK;
    const size_t nr = (size_t)2 * n_pairs;
    for (size_t i = 0; i < nr; i++) {
        if (offsets[i + 1] < offsets[i]) return fail(ctx, SNAPGPU_E_INVALID, "offsets must be non-decreasing");
        if (offsets[i + 1] - offsets[i] > ctx->params.max_read_len)
            return fail(ctx, SNAPGPU_E_INVALID, "read longer than max_read_len given at snapgpu_create (IntersectingPairedEndAligner.cpp:361-365)");
    }
    const PairedInFlight in_flight(ctx); ctx->paired_share = in_flight.share;      // (paired_grid_share: this call's launches ask for their share of the chip)
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    size_t nb = (size_t)offsets[nr];
    int rc;
    if ((rc = ensure_stage(ctx, 0, nb + 16)) || (rc = ensure_stage(ctx, 1, nb + 16)) || (rc = ensure_stage(ctx, 2, (nr + 1) * 8)) ||
        (rc = ensure_stage(ctx, 3, (size_t)n_pairs * sizeof(snapgpu_paired_result))) ||
        (rc = ensure_stage(ctx, 4, (size_t)n_pairs * sizeof(snapgpu_paired_result)))) return rc;
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage[0], bases, nb, hipMemcpyHostToDevice, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage[1], quals, nb, hipMemcpyHostToDevice, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage[2], offsets, (nr + 1) * 8, hipMemcpyHostToDevice, s), SNAPGPU_E_LAUNCH);
    rc = launch_paired(ctx, n_pairs, ctx->d_stage[0], ctx->d_stage[1], ctx->d_stage[2], ctx->d_stage[3], first_alt ? ctx->d_stage[4] : nullptr, s);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(primary, ctx->d_stage[3], (size_t)n_pairs * sizeof(snapgpu_paired_result), hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    if (first_alt) HIPCHK(ctx, hipMemcpyAsync(first_alt, ctx->d_stage[4], (size_t)n_pairs * sizeof(snapgpu_paired_result), hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    rc = finish_timing(ctx);
    if (rc) return rc;
    for (uint32_t i = 0; i < n_pairs; i++)
        if (primary[i].flags & SNAPGPU_PAIR_POOL_OVERFLOW)
            return fail(ctx, SNAPGPU_E_UNSUPPORTED, "a read pair needed more candidate entries than the per-wave pools hold (the reference would grow its buffers or ask for -mcp); its result is flagged");
    return SNAPGPU_OK;
}

extern "C" int snapgpu_align_paired_secondary_device(snapgpu_ctx *ctx, uint32_t n_pairs, const void *d_bases, const void *d_quals,
                                                     const void *d_offsets, void *d_primary, void *d_first_alt,
                                                     void *d_secondary, uint32_t secondary_stride, void *d_n_secondary,
                                                     void *d_single_secondary, uint32_t single_stride, void *d_n_single_secondary, void *stream)
{
    if (!ctx || !d_bases || !d_quals || !d_offsets || !d_primary || !d_n_secondary || !d_n_single_secondary ||
        (secondary_stride && !d_secondary) || (single_stride && !d_single_secondary))
        return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_align_paired_secondary_device: null argument");
    if (!ctx->paired_sec) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_enable_paired and snapgpu_enable_secondary must both have been called on this context");
    if (n_pairs == 0) return SNAPGPU_OK;
    const PairedInFlight in_flight(ctx); ctx->paired_share = in_flight.share;      // (paired_grid_share: this call's launches ask for their share of the chip)
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    PairedSecOut so{d_secondary, secondary_stride, d_n_secondary, d_single_secondary, single_stride, d_n_single_secondary};
    int rc = launch_paired(ctx, n_pairs, d_bases, d_quals, d_offsets, d_primary, d_first_alt, s, &so);
    if (rc) return rc;
    if (!stream) return finish_timing(ctx);
    return SNAPGPU_OK;
}

static int ensure_psec_stage(snapgpu_ctx *ctx, int which, size_t bytes) {
    if (ctx->psec_stage_cap[which] >= bytes) return 0;
    if (ctx->d_psec_stage[which]) (void)hipFree(ctx->d_psec_stage[which]);
    ctx->d_psec_stage[which] = nullptr; ctx->psec_stage_cap[which] = 0;
    size_t cap = bytes + bytes / 4 + 4096;
    HIPCHK(ctx, hipMalloc(&ctx->d_psec_stage[which], cap), SNAPGPU_E_NOMEM);
    ctx->psec_stage_cap[which] = cap;
    return 0;
}

extern "C" int snapgpu_align_paired_secondary(snapgpu_ctx *ctx, uint32_t n_pairs, const char *bases, const char *quals, const uint64_t *offsets,
                                              snapgpu_paired_result *primary, snapgpu_paired_result *first_alt,
                                              snapgpu_paired_result *secondary, uint32_t secondary_stride, uint32_t *n_secondary,
                                              snapgpu_single_result *single_secondary, uint32_t single_stride, uint32_t *n_single_secondary)
{
    if (!ctx || !bases || !quals || !offsets || !primary || !n_secondary || !n_single_secondary ||
        (secondary_stride && !secondary) || (single_stride && !single_secondary))
        return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_align_paired_secondary: null argument");
    if (!ctx->paired_sec) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_enable_paired and snapgpu_enable_secondary must both have been called on this context");
    if (n_pairs == 0) return SNAPGPU_OK;
    const size_t nr = (size_t)2 * n_pairs;
    for (size_t i = 0; i < nr; i++) {
        if (offsets[i + 1] < offsets[i]) return fail(ctx, SNAPGPU_E_INVALID, "offsets must be non-decreasing");
        if (offsets[i + 1] - offsets[i] > ctx->params.max_read_len)
            return fail(ctx, SNAPGPU_E_INVALID, "read longer than max_read_len given at snapgpu_create (IntersectingPairedEndAligner.cpp:361-365)");
    }
    const PairedInFlight in_flight(ctx); ctx->paired_share = in_flight.share;      // (paired_grid_share: this call's launches ask for their share of the chip)
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    size_t nb = (size_t)offsets[nr];
    const size_t sec_bytes = (size_t)n_pairs * secondary_stride * sizeof(snapgpu_paired_result);
    const size_t ssec_bytes = (size_t)n_pairs * single_stride * sizeof(snapgpu_single_result);
    int rc;
    if ((rc = ensure_stage(ctx, 0, nb + 16)) || (rc = ensure_stage(ctx, 1, nb + 16)) || (rc = ensure_stage(ctx, 2, (nr + 1) * 8)) ||
        (rc = ensure_stage(ctx, 3, (size_t)n_pairs * sizeof(snapgpu_paired_result))) ||
        (rc = ensure_stage(ctx, 4, (size_t)n_pairs * sizeof(snapgpu_paired_result))) ||
        (rc = ensure_psec_stage(ctx, 0, sec_bytes + 16)) || (rc = ensure_psec_stage(ctx, 1, (size_t)n_pairs * 4)) ||
        (rc = ensure_psec_stage(ctx, 2, ssec_bytes + 16)) || (rc = ensure_psec_stage(ctx, 3, nr * 4))) return rc;
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage[0], bases, nb, hipMemcpyHostToDevice, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage[1], quals, nb, hipMemcpyHostToDevice, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage[2], offsets, (nr + 1) * 8, hipMemcpyHostToDevice, s), SNAPGPU_E_LAUNCH);
    if (sec_bytes) HIPCHK(ctx, hipMemsetAsync(ctx->d_psec_stage[0], 0, sec_bytes, s), SNAPGPU_E_LAUNCH);
    if (ssec_bytes) HIPCHK(ctx, hipMemsetAsync(ctx->d_psec_stage[2], 0, ssec_bytes, s), SNAPGPU_E_LAUNCH);
    PairedSecOut so{ctx->d_psec_stage[0], secondary_stride, ctx->d_psec_stage[1], ctx->d_psec_stage[2], single_stride, ctx->d_psec_stage[3]};
    rc = launch_paired(ctx, n_pairs, ctx->d_stage[0], ctx->d_stage[1], ctx->d_stage[2], ctx->d_stage[3], first_alt ? ctx->d_stage[4] : nullptr, s, &so);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(primary, ctx->d_stage[3], (size_t)n_pairs * sizeof(snapgpu_paired_result), hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    if (first_alt) HIPCHK(ctx, hipMemcpyAsync(first_alt, ctx->d_stage[4], (size_t)n_pairs * sizeof(snapgpu_paired_result), hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    if (sec_bytes) HIPCHK(ctx, hipMemcpyAsync(secondary, ctx->d_psec_stage[0], sec_bytes, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(n_secondary, ctx->d_psec_stage[1], (size_t)n_pairs * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    if (ssec_bytes) HIPCHK(ctx, hipMemcpyAsync(single_secondary, ctx->d_psec_stage[2], ssec_bytes, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipMemcpyAsync(n_single_secondary, ctx->d_psec_stage[3], nr * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    rc = finish_timing(ctx);
    if (rc) return rc;
    bool truncated = false;
    for (uint32_t i = 0; i < n_pairs; i++) {
        if (primary[i].flags & SNAPGPU_PAIR_POOL_OVERFLOW)
            return fail(ctx, SNAPGPU_E_UNSUPPORTED, "a read pair needed more candidate entries than the per-wave pools hold (the reference would grow its buffers or ask for -mcp); its result is flagged");
        if (n_secondary[i] > secondary_stride || (uint64_t)n_single_secondary[2 * (size_t)i] + n_single_secondary[2 * (size_t)i + 1] > single_stride) truncated = true;
    }
    return truncated ? SNAPGPU_W_SECONDARY_TRUNCATED : SNAPGPU_OK;
}

extern "C" int snapgpu_get_counters(snapgpu_ctx *ctx, snapgpu_counters *out, int reset) {
    if (!ctx || !out) return SNAPGPU_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->d_counters, sizeof(*out), hipMemcpyDeviceToHost, ctx->stream), SNAPGPU_E_LAUNCH);
    if (reset) HIPCHK(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(*out), ctx->stream), SNAPGPU_E_LAUNCH);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream), SNAPGPU_E_LAUNCH);
    return SNAPGPU_OK;
}

// Diagnostics of the last single-end launch of a context created under SNAPGPU_PHASE_TIMERS=1 (not part of the drop-in surface):
// out[0 .. 64): reads by floor(log2(wave cycles)); then per wave slot: first-read clock, out-of-reads clock, (worst read's cycles << 24 | its
// affine-gap calls).  n_words = 64 + 3 * *n_slots.  Returns SNAPGPU_E_INVALID when the context has no such data.
extern "C" int snapgpu_debug_launch_profile(snapgpu_ctx *ctx, uint64_t *out, uint64_t cap_words, uint32_t *n_slots) {
    if (!ctx || !out || !n_slots || !ctx->d_dbg) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_debug_launch_profile: no profile (SNAPGPU_PHASE_TIMERS=1 at snapgpu_create)");
    const size_t words = 64 + 3 * (size_t)ctx->n_wave_slots;
    if (cap_words < words) return fail(ctx, SNAPGPU_E_INVALID, "snapgpu_debug_launch_profile: buffer too small");
    HIPCHK(ctx, hipSetDevice(ctx->device), SNAPGPU_E_NODEVICE);
    HIPCHK(ctx, hipMemcpy(out, ctx->d_dbg, words * 8, hipMemcpyDeviceToHost), SNAPGPU_E_NODEVICE);
    *n_slots = ctx->n_wave_slots;
    return SNAPGPU_OK;
}

extern "C" int snapgpu_kernel_time(snapgpu_ctx *ctx, double *total_ms, uint64_t *n_launches, int reset) {
    if (!ctx) return SNAPGPU_E_INVALID;
    if (total_ms) *total_ms = ctx->kernel_ms;
    if (n_launches) *n_launches = ctx->kernel_launches;
    if (reset) { ctx->kernel_ms = 0; ctx->kernel_launches = 0; }
    return SNAPGPU_OK;
}

// Host-side copies of the tables the kernels use, for bit-equality tests against the reference's.
extern "C" int snapgpu_debug_tables(snapgpu_ctx *ctx, double *phred256, double *indel, uint32_t n_indel,
                                    double *perfect, uint32_t n_perfect, double *seed_prob, double *mapq_thresholds72,
                                    uint32_t *wrapped33)
{
    if (!ctx) return SNAPGPU_E_INVALID;
    if (phred256) memcpy(phred256, ctx->h_tab.phred, sizeof(double) * 256);
    if (indel) memcpy(indel, ctx->h_tab.indel, sizeof(double) * (n_indel < N_INDEL_PROB ? n_indel : N_INDEL_PROB));
    if (perfect) memcpy(perfect, ctx->h_tab.perfect, sizeof(double) * (n_perfect < N_PERFECT_PROB ? n_perfect : N_PERFECT_PROB));
    if (seed_prob) *seed_prob = ctx->h_tab.seed_prob;
    if (mapq_thresholds72) memcpy(mapq_thresholds72, ctx->h_tab.mapq_threshold, sizeof(double) * 72);
    if (wrapped33) memcpy(wrapped33, ctx->h_tab.wrapped_seed, sizeof(uint32_t) * 33);
    return SNAPGPU_OK;
}
