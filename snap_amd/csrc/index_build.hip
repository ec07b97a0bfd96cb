// index_build.hip -- host side of the GPU index builder (include/snapgpu.h: snapgpu_index_build*; kernels in index_build.h).
//
// Restates (does not copy):
//   GenomeIndex::runIndexer             SNAPLib/GenomeIndex.cpp:126-506   option defaults, key size, location size
//   ReadFASTAGenome, IsContigALT        SNAPLib/FASTA.cpp:36-409          FASTA -> genome image
//   Genome::saveToFile                  SNAPLib/Genome.cpp:203-260        `Genome`
//   GenomeIndex::BuildIndexToDirectory  SNAPLib/GenomeIndex.cpp:527-1022  tables, `OverflowTable`, `GenomeIndex`
//   GenomeIndex::allocateHashTables     SNAPLib/GenomeIndex.cpp:1026-1110 table sizes
//   SNAPHashTable::saveToFile           SNAPLib/HashTable.cpp:199-262     `GenomeIndexHash`
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c index_build.hip
#include <hip/hip_runtime.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/stat.h>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/snapgpu.h"
#include "index_build.h"

namespace {

thread_local std::string g_ib_error;

struct IBContig {
    std::string name;
    uint64_t begin = 0;
    bool is_alt = false;
    int original_number = 0;
    uint64_t proj_begin = 0; bool proj_rc = false; std::string proj_cigar;      // proj_cigar empty = "*"
};

}  // namespace

struct snapgpu_built_index {
    int device = 0;
    uint32_t seed_len = 0, key_bytes = 0, chromosome_padding = 0, n_tables = 0;
    uint64_t n_bases = 0;
    uint32_t genome_pad = 1024;
    uint8_t *d_genome_padded = nullptr;          // genome_pad + n_bases + genome_pad bytes
    unsigned long long *d_hash = nullptr; uint64_t hash_bytes = 0;
    uint32_t *d_overflow = nullptr; uint64_t overflow_words = 0;
    std::vector<uint64_t> table_offset, table_size, table_used;     // byte offset into the blob, slots, used slots
    std::vector<IBContig> contigs;
    std::vector<uint64_t> contig_begin, proj_begin; std::vector<uint8_t> proj_rc; std::vector<uint32_t> cigar_start, cigar_ops;
    uint64_t first_alt = ~0ull >> 2;
    snapgpu_index_build_stats stats{};
};

static int ib_fail(int code, const std::string &msg);
#define IBCHK(call, code) do { hipError_t _e = (call); if (_e != hipSuccess) return ib_fail(code, std::string(#call) + ": " + hipGetErrorString(_e)); } while (0)

extern "C" const char *snapgpu_last_error(const snapgpu_ctx *ctx);
// (snapgpu.hip keeps the thread's last message; the index builder appends to the same channel through this hook)
extern "C" void snapgpu_set_last_error(const char *msg);
static int ib_fail(int code, const std::string &msg) { g_ib_error = msg; snapgpu_set_last_error(msg.c_str()); return code; }

extern "C" void snapgpu_default_index_build_params(snapgpu_index_build_params *bp)
{
    if (!bp) return;
    memset(bp, 0, sizeof(*bp));
    bp->seed_len = 20;              // (the reference's indexer defaults to 24, GenomeIndex.cpp:46; the north star's index is -s 20)
    bp->slack = 0.3;                // DEFAULT_SLACK, GenomeIndex.cpp:47
    bp->key_bytes = 0;
    bp->chromosome_padding = 2000;  // DEFAULT_PADDING, GenomeIndex.cpp:48
    bp->space_terminates_name = 1;  // GenomeIndex.cpp:143
    bp->name_terminators = nullptr;
    bp->auto_alt = 1;               // GenomeIndex.cpp:156
    bp->max_alt_contig_size = -1;   // GenomeIndex.cpp:151
}

extern "C" void snapgpu_built_index_destroy(snapgpu_built_index *bi)
{
    if (!bi) return;
    (void)hipSetDevice(bi->device);
    if (bi->d_genome_padded) (void)hipFree(bi->d_genome_padded);
    if (bi->d_hash) (void)hipFree(bi->d_hash);
    if (bi->d_overflow) (void)hipFree(bi->d_overflow);
    delete bi;
}

namespace {

struct DevFree {                     // frees what is still allocated when a build leaves early
    std::vector<void *> ptrs;
    ~DevFree() { for (void *p : ptrs) if (p) (void)hipFree(p); }
    template <typename T> hipError_t alloc(T **p, size_t bytes) {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
        if (e == hipSuccess) { ptrs.push_back(q); *p = (T *)q; }
        return e;
    }
    void release(void *p) { for (auto &q : ptrs) if (q == p) { (void)hipFree(q); q = nullptr; } }
    void disown(void *p) { for (auto &q : ptrs) if (q == p) q = nullptr; }
};

// exclusive scan of in[0 .. n) into out (may alias in); *total = the sum.  partial: scratch of >= ceil(n / 1024) + 1 words
int ib_exclusive_scan(const uint32_t *in, uint64_t n, uint32_t *out, uint32_t *partial, uint32_t *d_total, uint32_t grid, hipStream_t s)
{
    const uint32_t n_chunks = (uint32_t)((n + IB_TILE - 1) / IB_TILE);
    if (n_chunks == 0) { IBCHK(hipMemsetAsync(d_total, 0, 4, s), SNAPGPU_E_LAUNCH); return SNAPGPU_OK; }
    hipLaunchKernelGGL(k_ib_scan_sums, dim3(grid), dim3(256), 0, s, in, n, partial, n_chunks);
    hipLaunchKernelGGL(k_ib_scan_partials, dim3(1), dim3(64), 0, s, partial, n_chunks, d_total);
    hipLaunchKernelGGL(k_ib_scan_apply, dim3(grid), dim3(256), 0, s, in, n, (const uint32_t *)partial, n_chunks, out);
    IBCHK(hipGetLastError(), SNAPGPU_E_LAUNCH);
    return SNAPGPU_OK;
}

}  // namespace

// The device part: genome image (already on the device inside bi) -> hash blob + overflow table.
static int ib_build_on_device(snapgpu_built_index *bi, double slack)
{
    const uint32_t L = bi->seed_len, key_bits = bi->key_bytes * 8;
    const uint32_t entry_bytes = 4 + bi->key_bytes;             // one 4-byte value, then the key (HashTable.h:148-156)
    const uint64_t n_bases = bi->n_bases;
    // locations [0, nBases - seedLen - 1): the last chunk of the reference's scan ends there (GenomeIndex.cpp:667-670, 1455)
    const uint64_t n_locs = n_bases > (uint64_t)L + 1 ? n_bases - L - 1 : 0;
    if (n_bases >= 0xFFFFFFF0ull) return ib_fail(SNAPGPU_E_UNSUPPORTED, "genome too big for 4-byte genome locations (GenomeIndex.cpp:571)");
    const uint32_t n_tables = 1u << ((L - bi->key_bytes * 4) * 2);
    bi->n_tables = n_tables;
    hipDeviceProp_t prop;
    IBCHK(hipGetDeviceProperties(&prop, bi->device), SNAPGPU_E_NODEVICE);
    const uint32_t grid = (uint32_t)prop.multiProcessorCount * 8;
    hipStream_t s = nullptr;
    hipEvent_t ev[5];
    for (auto &e : ev) IBCHK(hipEventCreate(&e), SNAPGPU_E_NODEVICE);
    struct EvFree { hipEvent_t *e; ~EvFree() { for (int i = 0; i < 5; i++) (void)hipEventDestroy(e[i]); } } evfree{ev};
    DevFree mem;
    const uint8_t *d_genome = bi->d_genome_padded + bi->genome_pad;

    // ---- 1. seeds
    uint64_t *d_keys_a = nullptr, *d_keys_b = nullptr; uint32_t *d_vals_a = nullptr, *d_vals_b = nullptr;
    IBCHK(mem.alloc(&d_keys_a, n_locs * 8), SNAPGPU_E_NOMEM);
    IBCHK(hipEventRecord(ev[0], s), SNAPGPU_E_LAUNCH);
    if (n_locs) hipLaunchKernelGGL(k_ib_keys, dim3(grid), dim3(256), 0, s, d_genome, n_locs, L, d_keys_a);
    IBCHK(hipGetLastError(), SNAPGPU_E_LAUNCH);
    IBCHK(hipEventRecord(ev[1], s), SNAPGPU_E_LAUNCH);

    // ---- 2. sort (seed, location) by seed; the first pass drops the locations without a seed
    const uint32_t n_tiles0 = (uint32_t)((n_locs + IB_TILE - 1) / IB_TILE);
    uint32_t *d_hist = nullptr, *d_partial = nullptr, *d_total = nullptr;
    IBCHK(mem.alloc(&d_hist, (size_t)64 * (n_tiles0 ? n_tiles0 : 1) * 4), SNAPGPU_E_NOMEM);
    // (the partial-sum scratch also serves the scans over the sorted elements further down)
    IBCHK(mem.alloc(&d_partial, ((size_t)64 * (n_tiles0 ? n_tiles0 : 1) / IB_TILE + n_locs / IB_TILE + 16) * 4), SNAPGPU_E_NOMEM);
    IBCHK(mem.alloc(&d_total, 64), SNAPGPU_E_NOMEM);
    uint64_t m = n_locs;                                   // elements alive (all of them, the non-seeds included, until the first scatter)
    const uint32_t key_total_bits = 2 * L;
    bool first = true;
    uint64_t *kin = d_keys_a, *kout = nullptr; uint32_t *vin = nullptr, *vout = nullptr;
    for (uint32_t shift = 0; shift < key_total_bits && n_locs; shift += IB_BITS) {
        const uint32_t n_tiles = (uint32_t)((m + IB_TILE - 1) / IB_TILE);
        if (n_tiles == 0) break;
        hipLaunchKernelGGL(k_ib_hist, dim3(grid), dim3(256), 0, s, (const uint64_t *)kin, m, shift, d_hist, n_tiles);
        int rc = ib_exclusive_scan(d_hist, (uint64_t)64 * n_tiles, d_hist, d_partial, d_total, grid, s);
        if (rc) return rc;
        if (first) {                                       // how many locations carry a seed: sizes the sorted arrays
            uint32_t valid = 0;
            IBCHK(hipMemcpyAsync(&valid, d_total, 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
            IBCHK(hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
            bi->stats.n_seed_locations = valid;
            IBCHK(mem.alloc(&d_keys_b, (size_t)valid * 8), SNAPGPU_E_NOMEM);
            IBCHK(mem.alloc(&d_vals_a, (size_t)valid * 4), SNAPGPU_E_NOMEM);
            IBCHK(mem.alloc(&d_vals_b, (size_t)valid * 4), SNAPGPU_E_NOMEM);
            kout = d_keys_b; vout = d_vals_b;
        }
        hipLaunchKernelGGL(k_ib_scatter, dim3(grid), dim3(256), 0, s, (const uint64_t *)kin, (const uint32_t *)vin, m, shift,
                           (const uint32_t *)d_hist, n_tiles, kout, vout);
        IBCHK(hipGetLastError(), SNAPGPU_E_LAUNCH);
        if (first) {
            first = false;
            m = bi->stats.n_seed_locations;
            // d_keys_a held n_locs keys; from now on it only has to hold m <= n_locs of them
            kin = d_keys_b; vin = d_vals_b; kout = d_keys_a; vout = d_vals_a;
        } else {
            std::swap(kin, kout); std::swap(vin, vout);
        }
    }
    if (first) m = 0;                                      // (no pass ran: no locations at all)
    const uint64_t *d_keys = kin; const uint32_t *d_vals = vin;            // sorted
    IBCHK(hipEventRecord(ev[2], s), SNAPGPU_E_LAUNCH);
    mem.release(d_hist); d_hist = nullptr;
    mem.release((void *)kout); mem.release((void *)vout);                 // the other halves of the double buffers

    // ---- 3. runs of equal seeds
    uint32_t *d_head = nullptr, *d_before = nullptr, *d_run_start = nullptr, *d_need = nullptr, *d_ovf_off = nullptr;
    uint32_t n_runs = 0, ovf_words = 0;
    unsigned long long ovf_words64 = 0;
    if (m) {
        IBCHK(mem.alloc(&d_head, (size_t)m * 4), SNAPGPU_E_NOMEM);
        IBCHK(mem.alloc(&d_before, (size_t)m * 4), SNAPGPU_E_NOMEM);
        hipLaunchKernelGGL(k_ib_heads, dim3(grid), dim3(256), 0, s, d_keys, m, d_head);
        int rc = ib_exclusive_scan(d_head, m, d_before, d_partial, d_total, grid, s);
        if (rc) return rc;
        IBCHK(hipMemcpyAsync(&n_runs, d_total, 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
        IBCHK(hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
        IBCHK(mem.alloc(&d_run_start, ((size_t)n_runs + 1) * 4), SNAPGPU_E_NOMEM);
        IBCHK(mem.alloc(&d_need, (size_t)n_runs * 4), SNAPGPU_E_NOMEM);
        IBCHK(mem.alloc(&d_ovf_off, (size_t)n_runs * 4), SNAPGPU_E_NOMEM);
        hipLaunchKernelGGL(k_ib_run_starts, dim3(grid), dim3(256), 0, s, (const uint32_t *)d_head, (const uint32_t *)d_before, m, n_runs, d_run_start);
        hipLaunchKernelGGL(k_ib_run_need, dim3(grid), dim3(256), 0, s, (const uint32_t *)d_run_start, n_runs, d_need);
        // (the scan of `need` reuses the partial-sum scratch: n_runs <= m)
        rc = ib_exclusive_scan(d_need, n_runs, d_ovf_off, d_partial, d_total, grid, s);
        if (rc) return rc;
        IBCHK(hipMemcpyAsync(&ovf_words, d_total, 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
        // the same total in 64 bits (d_total has 64 bytes: the second half is free): a wrapped 32-bit scan must not pass the check below
        unsigned long long *d_sum64 = (unsigned long long *)(d_total + 8);
        IBCHK(hipMemsetAsync(d_sum64, 0, 8, s), SNAPGPU_E_LAUNCH);
        hipLaunchKernelGGL(k_ib_sum64, dim3(grid), dim3(256), 0, s, (const uint32_t *)d_need, n_runs, d_sum64);
        IBCHK(hipMemcpyAsync(&ovf_words64, d_sum64, 8, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
        IBCHK(hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    }
    // value = nBases + overflow index must stay below the two reserved values (GenomeIndex.cpp:777)
    if (ovf_words64 + n_bases >= 0xFFFFFFFFull - 15 || ovf_words64 != ovf_words) return ib_fail(SNAPGPU_E_UNSUPPORTED, "not enough 32-bit address space for genome + overflow table (GenomeIndex.cpp:777): larger seed or location size needed");
    bi->stats.n_distinct_seeds = n_runs; bi->overflow_words = ovf_words; bi->stats.overflow_table_size = ovf_words;
    IBCHK(hipEventRecord(ev[3], s), SNAPGPU_E_LAUNCH);

    // ---- 4. tables: sizes from the exact distinct-seed counts (allocateHashTables, GenomeIndex.cpp:1084-1100 with bias = count * nTables / nBases)
    std::vector<uint32_t> first_run((size_t)n_tables + 1, 0);
    if (m) {
        uint32_t *d_first = nullptr;
        IBCHK(mem.alloc(&d_first, ((size_t)n_tables + 1) * 4), SNAPGPU_E_NOMEM);
        hipLaunchKernelGGL(k_ib_table_bounds, dim3((n_tables + 1 + 255) / 256), dim3(256), 0, s, d_keys, (const uint32_t *)d_run_start, n_runs, key_bits, n_tables, d_first);
        IBCHK(hipMemcpyAsync(first_run.data(), d_first, ((size_t)n_tables + 1) * 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
        IBCHK(hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    }
    bi->table_offset.assign(n_tables, 0); bi->table_size.assign(n_tables, 0); bi->table_used.assign(n_tables, 0);
    std::vector<uint64_t> slot0(n_tables);
    const size_t hash_table_size = (size_t)((double)n_bases * (slack + 1.0) / n_tables);
    uint64_t total_slots = 0;
    for (uint32_t t = 0; t < n_tables; t++) {
        const uint64_t count = (uint64_t)first_run[t + 1] - first_run[t];
        const double bias = ((double)count * n_tables) / (double)n_bases;
        unsigned biased = (unsigned)(hash_table_size * bias);
        if (biased < 100) biased = 100;
        if ((uint64_t)biased <= count) biased = (unsigned)(count + count / 8 + 8);       // (cannot happen with slack > 0; never build a full table)
        bi->table_size[t] = biased; bi->table_used[t] = count;
        slot0[t] = total_slots; bi->table_offset[t] = total_slots * entry_bytes;
        total_slots += biased;
    }
    bi->hash_bytes = total_slots * entry_bytes; bi->stats.hash_table_slots = total_slots; bi->stats.hash_blob_bytes = bi->hash_bytes;
    IBCHK(hipMalloc((void **)&bi->d_hash, (size_t)bi->hash_bytes + 64), SNAPGPU_E_NOMEM);
    IBCHK(hipMalloc((void **)&bi->d_overflow, ((size_t)ovf_words + 4) * 4), SNAPGPU_E_NOMEM);
    IBCHK(hipMemsetAsync(bi->d_overflow, 0, ((size_t)ovf_words + 4) * 4, s), SNAPGPU_E_LAUNCH);
    if (entry_bytes == 8) hipLaunchKernelGGL(k_ib_fill_empty, dim3(grid), dim3(256), 0, s, bi->d_hash, total_slots + 8);
    else hipLaunchKernelGGL(k_ib_fill_empty_wide, dim3(grid), dim3(256), 0, s, (uint32_t *)bi->d_hash, (total_slots * entry_bytes + 63) / 4, entry_bytes);
    if (m) {
        uint64_t *d_slot0 = nullptr, *d_tsize = nullptr; uint32_t *d_fail = nullptr;
        IBCHK(mem.alloc(&d_slot0, (size_t)n_tables * 8), SNAPGPU_E_NOMEM);
        IBCHK(mem.alloc(&d_tsize, (size_t)n_tables * 8), SNAPGPU_E_NOMEM);
        IBCHK(mem.alloc(&d_fail, 64), SNAPGPU_E_NOMEM);
        IBCHK(hipMemcpyAsync(d_slot0, slot0.data(), (size_t)n_tables * 8, hipMemcpyHostToDevice, s), SNAPGPU_E_LAUNCH);
        IBCHK(hipMemcpyAsync(d_tsize, bi->table_size.data(), (size_t)n_tables * 8, hipMemcpyHostToDevice, s), SNAPGPU_E_LAUNCH);
        IBCHK(hipMemsetAsync(d_fail, 0, 4, s), SNAPGPU_E_LAUNCH);
        hipLaunchKernelGGL(k_ib_fill_overflow, dim3(grid), dim3(256), 0, s, d_vals, (const uint32_t *)d_head, (const uint32_t *)d_before, m,
                           (const uint32_t *)d_run_start, (const uint32_t *)d_ovf_off, bi->d_overflow);
        if (entry_bytes == 8) {
            hipLaunchKernelGGL(k_ib_insert, dim3(grid), dim3(256), 0, s, d_keys, d_vals, (const uint32_t *)d_run_start, (const uint32_t *)d_ovf_off, n_runs,
                               key_bits, (uint32_t)n_bases, bi->d_hash, (const uint64_t *)d_slot0, (const uint64_t *)d_tsize, d_fail);
        } else {
            uint32_t *d_claim = nullptr;
            const size_t claim_bytes = (size_t)(total_slots / 32 + 1) * 4;
            IBCHK(mem.alloc(&d_claim, claim_bytes), SNAPGPU_E_NOMEM);
            IBCHK(hipMemsetAsync(d_claim, 0, claim_bytes, s), SNAPGPU_E_LAUNCH);
            hipLaunchKernelGGL(k_ib_insert_wide, dim3(grid), dim3(256), 0, s, d_keys, d_vals, (const uint32_t *)d_run_start, (const uint32_t *)d_ovf_off,
                               n_runs, key_bits, (uint32_t)n_bases, (uint8_t *)bi->d_hash, entry_bytes, d_claim, (const uint64_t *)d_slot0,
                               (const uint64_t *)d_tsize, d_fail);
        }
        IBCHK(hipGetLastError(), SNAPGPU_E_LAUNCH);
        uint32_t failed = 0;
        IBCHK(hipMemcpyAsync(&failed, d_fail, 4, hipMemcpyDeviceToHost, s), SNAPGPU_E_LAUNCH);
        IBCHK(hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
        if (failed) return ib_fail(SNAPGPU_E_LAUNCH, "a hash table filled up during the build (GenomeIndex.cpp:1601: increase slack)");
    }
    IBCHK(hipEventRecord(ev[4], s), SNAPGPU_E_LAUNCH);
    IBCHK(hipStreamSynchronize(s), SNAPGPU_E_LAUNCH);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev[0], ev[1]); bi->stats.ms_keys = ms;
    (void)hipEventElapsedTime(&ms, ev[1], ev[2]); bi->stats.ms_sort = ms;
    (void)hipEventElapsedTime(&ms, ev[2], ev[3]); bi->stats.ms_runs = ms;
    (void)hipEventElapsedTime(&ms, ev[3], ev[4]); bi->stats.ms_tables = ms;
    (void)hipEventElapsedTime(&ms, ev[0], ev[4]); bi->stats.ms_total_device = ms;
    uint64_t repeated = 0;
    {   // seeds with more than one occurrence = runs that needed overflow space: overflow words = repeated + their locations
        // (kept cheap: derived on the host from what is already known)
        repeated = 0;
        if (ovf_words) {
            // count of runs with len > 1 = ovf_words - (locations in such runs); locations in such runs = m - unique runs;
            // unique runs = n_runs - repeated  =>  ovf_words = repeated + m - (n_runs - repeated)  =>  repeated = (ovf_words - m + n_runs) / 2
            repeated = ((uint64_t)ovf_words + n_runs - m) / 2;
        }
    }
    bi->stats.n_repeated_seeds = repeated;
    bi->stats.n_bases = n_bases;
    return SNAPGPU_OK;
}

static int ib_check_shape(const snapgpu_index_build_params *bp, uint32_t *key_bytes)
{
    if (bp->seed_len < 8 || bp->seed_len > 32) return ib_fail(SNAPGPU_E_INVALID, "seed length must be between 8 and 32 (GenomeIndex.cpp:429)");
    uint32_t kb = bp->key_bytes;
    if (kb == 0) { kb = (bp->seed_len + 2) / 4 - 1; if (kb < 2) kb = 2; }                                  // GenomeIndex.cpp:437
    if (bp->seed_len * 2 < kb * 8) return ib_fail(SNAPGPU_E_INVALID, "the seed must be big enough to fill the key (GenomeIndex.cpp:452)");
    if (bp->seed_len * 2 - kb * 8 > 16) return ib_fail(SNAPGPU_E_INVALID, "more than 4^8 hash tables: bigger key size or smaller seed (GenomeIndex.cpp:458)");
    if (kb < 2 || kb > 8) return ib_fail(SNAPGPU_E_INVALID, "key size must be between 2 and 8 bytes (GenomeIndex.cpp:437, HashTable.h:148)");
    // a seed of 32 T's is the value this builder marks "no seed at this location" with (index_build.h: IB_INVALID_KEY)
    if (bp->seed_len == 32) return ib_fail(SNAPGPU_E_UNSUPPORTED, "the GPU index builder takes seeds up to 31 bases; use the reference's indexer for -s 32");
    if (bp->seed_len < 20) {
        // the reference picks 5-byte locations below seed 20 unless told otherwise (GenomeIndex.cpp:442-449); this builder always writes 4
    }
    if (!(bp->slack > 0)) return ib_fail(SNAPGPU_E_INVALID, "slack must be positive (GenomeIndex.cpp:1040)");
    if (bp->chromosome_padding == 0) return ib_fail(SNAPGPU_E_INVALID, "chromosome padding must be at least one (GenomeIndex.cpp:216)");
    *key_bytes = kb;
    return SNAPGPU_OK;
}

static void ib_finish_contigs(snapgpu_built_index *bi)
{
    const size_t n = bi->contigs.size();
    bi->contig_begin.resize(n); bi->proj_begin.resize(n); bi->proj_rc.resize(n); bi->cigar_start.assign(n + 1, 0); bi->cigar_ops.clear();
    bi->first_alt = ~0ull >> 2;
    for (size_t i = 0; i < n; i++) {
        const IBContig &c = bi->contigs[i];
        bi->contig_begin[i] = c.begin; bi->proj_begin[i] = c.proj_begin; bi->proj_rc[i] = c.proj_rc ? 1 : 0;
        if (c.is_alt && c.begin < bi->first_alt) bi->first_alt = c.begin;
        const char *p = c.proj_cigar.c_str();
        while (*p) {                                                    // repeated sscanf("%d%c") as Genome.cpp:389-403 does on load
            int count = 0, used = 0; char act = 0;
            if (sscanf(p, "%d%c%n", &count, &act, &used) != 2) break;
            bi->cigar_ops.push_back(((uint32_t)count << 8) | (uint32_t)(uint8_t)act);
            p += used;
        }
        bi->cigar_start[i + 1] = (uint32_t)bi->cigar_ops.size();
    }
    if (bi->cigar_ops.empty()) bi->cigar_ops.push_back(0);
}

extern "C" int snapgpu_index_build(const snapgpu_genome_view *g, const snapgpu_index_build_params *bp, int device, snapgpu_built_index **out)
{
    if (!g || !bp || !out || !g->bases) return ib_fail(SNAPGPU_E_INVALID, "snapgpu_index_build: null argument");
    *out = nullptr;
    uint32_t kb = 0;
    int rc = ib_check_shape(bp, &kb);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ib_fail(SNAPGPU_E_NODEVICE, "no HIP device available (the index builder has no CPU path)");
    if (device < 0 || device >= ndev) return ib_fail(SNAPGPU_E_NODEVICE, "device index out of range");
    IBCHK(hipSetDevice(device), SNAPGPU_E_NODEVICE);
    snapgpu_built_index *bi = new snapgpu_built_index();
    bi->device = device; bi->seed_len = bp->seed_len; bi->key_bytes = kb; bi->chromosome_padding = bp->chromosome_padding; bi->n_bases = g->n_bases;
    for (uint32_t i = 0; i < g->n_contigs; i++) {
        IBContig c;
        c.name = g->contig_name ? g->contig_name[i] : ("contig" + std::to_string(i));
        c.begin = g->contig_begin[i]; c.is_alt = g->contig_is_alt && g->contig_is_alt[i];
        c.original_number = g->contig_original_number ? g->contig_original_number[i] : (int)i;
        c.proj_begin = g->contig_proj_begin ? g->contig_proj_begin[i] : 0; c.proj_rc = g->contig_proj_rc && g->contig_proj_rc[i];
        if (g->contig_proj_cigar && g->contig_proj_cigar[i] && strcmp(g->contig_proj_cigar[i], "*") != 0) c.proj_cigar = g->contig_proj_cigar[i];
        bi->contigs.push_back(c);
    }
    ib_finish_contigs(bi);
    const size_t total = (size_t)g->n_bases + 2 * (size_t)bi->genome_pad;
    hipError_t e = hipMalloc((void **)&bi->d_genome_padded, total + 256);
    if (e != hipSuccess) { snapgpu_built_index_destroy(bi); return ib_fail(SNAPGPU_E_NOMEM, std::string("hipMalloc(genome): ") + hipGetErrorString(e)); }
    if (hipMemset(bi->d_genome_padded, 'n', total + 256) != hipSuccess ||
        hipMemcpy(bi->d_genome_padded + bi->genome_pad, g->bases, (size_t)g->n_bases, hipMemcpyHostToDevice) != hipSuccess) {
        snapgpu_built_index_destroy(bi); return ib_fail(SNAPGPU_E_NODEVICE, "uploading the genome failed");
    }
    rc = ib_build_on_device(bi, bp->slack);
    if (rc) { snapgpu_built_index_destroy(bi); return rc; }
    *out = bi;
    return SNAPGPU_OK;
}

// ---------------------------------------------------------------- FASTA -> genome image (ReadFASTAGenome, FASTA.cpp:188-409)
namespace {

bool ib_is_alt(const std::string &name, int64_t size, const snapgpu_index_build_params *bp)            // IsContigALT, FASTA.cpp:36-70
{
    for (uint32_t i = 0; i < bp->n_non_alt_contig_names; i++) if (!strcasecmp(bp->non_alt_contig_names[i], name.c_str())) return false;
    if (size <= bp->max_alt_contig_size) return true;
    for (uint32_t i = 0; i < bp->n_alt_contig_names; i++) if (!strcasecmp(bp->alt_contig_names[i], name.c_str())) return true;
    const size_t n = name.size();
    if (bp->auto_alt && ((n > 4 && !strcasecmp(name.c_str() + n - 4, "_alt")) ||
                         (n > 3 && (name[0] == 'H' || name[0] == 'h') && (name[1] == 'L' || name[1] == 'l') && (name[2] == 'A' || name[2] == 'a') && name[3] == '-')))
        return true;
    return false;
}

struct RawContig { std::string name; int number; size_t data_begin, data_len; bool alt; };

struct Liftover { std::string contig, proj_contig, cigar; unsigned flags, offset; bool mapped; };

bool ib_read_liftover(const char *path, std::vector<Liftover> &out, std::string &err)                  // GenomeIndex.cpp:318-420
{
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("unable to open ALT liftover file ") + path; return false; }
    char *line = nullptr; size_t cap = 0; ssize_t len;
    while ((len = getline(&line, &cap, f)) > 0) {
        if (line[0] == '@') continue;
        while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        std::vector<std::string> fld; const char *p = line;
        for (;;) { const char *t = strchr(p, '\t'); if (!t) { fld.emplace_back(p); break; } fld.emplace_back(p, t - p); p = t + 1; }
        if (fld.size() < 7) { err = std::string("invalid format for ALT liftover file ") + path + ": not tab separated"; free(line); fclose(f); return false; }
        Liftover l; l.contig = fld[0]; l.flags = (unsigned)strtoul(fld[1].c_str(), nullptr, 10); l.proj_contig = fld[2];
        l.offset = (unsigned)strtoul(fld[3].c_str(), nullptr, 10); l.cigar = fld[5]; l.mapped = fld[2].empty() || fld[2][0] != '*';
        out.push_back(l);
    }
    free(line); fclose(f);
    return true;
}

}  // namespace

extern "C" int snapgpu_index_build_from_fasta(const char *fasta_path, const snapgpu_index_build_params *bp, int device, snapgpu_built_index **out)
{
    if (!fasta_path || !bp || !out) return ib_fail(SNAPGPU_E_INVALID, "snapgpu_index_build_from_fasta: null argument");
    *out = nullptr;
    uint32_t kb = 0;
    int rc = ib_check_shape(bp, &kb);
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    FILE *f = fopen(fasta_path, "rb");
    if (!f) return ib_fail(SNAPGPU_E_INVALID, std::string("unable to open FASTA file '") + fasta_path + "'");
    // contig data is appended to one buffer as it is read (upper-cased, anything but ACGTN turned into N); the genome image is laid out afterwards
    std::vector<uint8_t> data;
    {   struct stat st; if (fstat(fileno(f), &st) == 0 && st.st_size > 0) data.reserve((size_t)st.st_size); }
    std::vector<RawContig> raw;
    bool valid_char[256]; memset(valid_char, 0, sizeof(valid_char));
    for (const char *c = "ATCGNatcgn"; *c; c++) valid_char[(unsigned char)*c] = true;
    char *line = nullptr; size_t cap = 0; ssize_t len;
    bool in_contig = false;
    while ((len = getline(&line, &cap, f)) > 0) {
        if (line[0] == '>') {
            // name: cut at the -B characters, at blank / tab (-bSpace), at the end of line (FASTA.cpp:262-296)
            if (bp->name_terminators) for (const char *t = bp->name_terminators; *t; t++) { char *q = strchr(line + 1, *t); if (q) *q = 0; }
            if (bp->space_terminates_name) { char *q = strchr(line, ' '); if (q) *q = 0; q = strchr(line, '\t'); if (q) *q = 0; }
            { char *q = strchr(line, '\n'); if (q) *q = 0; q = strchr(line, '\r'); if (q) *q = 0; }
            RawContig c; c.name = line + 1; c.number = (int)raw.size(); c.data_begin = data.size(); c.data_len = 0; c.alt = false;
            raw.push_back(c);
            in_contig = true;
        } else {
            if (!in_contig) { free(line); fclose(f); return ib_fail(SNAPGPU_E_INVALID, "FASTA file doesn't begin with a contig name (FASTA.cpp:305)"); }
            char *q = strchr(line, '\n'); if (q) *q = 0;
            q = strchr(line, '\r'); if (q) *q = 0;
            const size_t n = strlen(line);
            const size_t at = data.size();
            data.resize(at + n);
            for (size_t i = 0; i < n; i++) {
                unsigned char ch = (unsigned char)line[i];
                if (ch >= 'a' && ch <= 'z') ch = (unsigned char)(ch - 'a' + 'A');                      // toupper, then the validity test (:327-341)
                data[at + i] = valid_char[ch] ? ch : (uint8_t)'N';
            }
            raw.back().data_len += n;
        }
    }
    free(line); fclose(f);
    if (raw.empty()) return ib_fail(SNAPGPU_E_INVALID, "the FASTA file was empty (FASTA.cpp:351)");
    for (auto &c : raw) c.alt = ib_is_alt(c.name, (int64_t)c.data_len, bp);
    std::vector<Liftover> lift;
    if (bp->alt_liftover_file) { std::string err; if (!ib_read_liftover(bp->alt_liftover_file, lift, err)) return ib_fail(SNAPGPU_E_INVALID, err); }

    // genome image: padding, then each regular contig in FASTA order, then each ALT contig, padding before every contig and at the end (:364-396)
    const uint32_t pad = bp->chromosome_padding;
    size_t total = pad;
    for (auto &c : raw) total += pad + c.data_len;
    std::vector<uint8_t> image(total, (uint8_t)'n');
    std::vector<IBContig> contigs;
    size_t pos = 0;
    for (int pass = 0; pass < 2; pass++) {
        for (auto &c : raw) {
            if ((pass == 1) != c.alt) continue;
            pos += pad;
            IBContig ic; ic.name = c.name; ic.begin = pos; ic.is_alt = c.alt; ic.original_number = c.number;
            memcpy(image.data() + pos, data.data() + c.data_begin, c.data_len);
            pos += c.data_len;
            contigs.push_back(ic);
        }
    }
    pos += pad;
    std::vector<uint8_t>().swap(data);
    // ALT liftover (Genome::markContigLiftover, Genome.cpp:146-167): the first line naming the contig wins; '*' targets are skipped (GenomeIndex.cpp:389)
    for (auto &ic : contigs) {
        if (!ic.is_alt) continue;
        for (auto &l : lift) {
            if (!l.mapped || l.contig != ic.name) continue;
            for (auto &pc : contigs) if (pc.name == l.proj_contig) { ic.proj_begin = pc.begin + l.offset - 1; break; }
            ic.proj_rc = (l.flags & 16) != 0; ic.proj_cigar = l.cigar;
            break;
        }
    }
    std::vector<uint64_t> begin; std::vector<const char *> names; std::vector<uint8_t> alt, prc; std::vector<int32_t> orig; std::vector<uint64_t> pbeg;
    std::vector<const char *> pcig;
    for (auto &ic : contigs) {
        begin.push_back(ic.begin); names.push_back(ic.name.c_str()); alt.push_back(ic.is_alt ? 1 : 0); orig.push_back(ic.original_number);
        pbeg.push_back(ic.proj_begin); prc.push_back(ic.proj_rc ? 1 : 0); pcig.push_back(ic.proj_cigar.empty() ? "*" : ic.proj_cigar.c_str());
    }
    snapgpu_genome_view g; memset(&g, 0, sizeof(g));
    g.bases = image.data(); g.n_bases = pos; g.n_contigs = (uint32_t)contigs.size();
    g.contig_begin = begin.data(); g.contig_name = names.data(); g.contig_is_alt = alt.data(); g.contig_original_number = orig.data();
    g.contig_proj_begin = pbeg.data(); g.contig_proj_rc = prc.data(); g.contig_proj_cigar = pcig.data();
    const double s_fasta = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    rc = snapgpu_index_build(&g, bp, device, out);
    if (rc == SNAPGPU_OK) (*out)->stats.s_fasta = s_fasta;
    return rc;
}

extern "C" int snapgpu_built_index_stats(const snapgpu_built_index *bi, snapgpu_index_build_stats *out)
{
    if (!bi || !out) return ib_fail(SNAPGPU_E_INVALID, "snapgpu_built_index_stats: null argument");
    *out = bi->stats;
    return SNAPGPU_OK;
}

extern "C" int snapgpu_built_index_view(const snapgpu_built_index *bi, snapgpu_index_view *v)
{
    if (!bi || !v) return ib_fail(SNAPGPU_E_INVALID, "snapgpu_built_index_view: null argument");
    memset(v, 0, sizeof(*v));
    v->seed_len = bi->seed_len; v->key_bytes = bi->key_bytes; v->n_hash_tables = bi->n_tables; v->large_hash_table = 0; v->location_size = 4;
    v->chromosome_padding = bi->chromosome_padding; v->overflow_table_size = bi->overflow_words;
    v->hash_blob = (const uint8_t *)bi->d_hash; v->hash_blob_bytes = bi->hash_bytes + 16;
    v->table_offset = bi->table_offset.data(); v->table_size = bi->table_size.data();
    v->overflow = bi->d_overflow; v->genome = bi->d_genome_padded + bi->genome_pad; v->n_bases = bi->n_bases; v->genome_pad = bi->genome_pad;
    v->contig_begin = bi->contig_begin.data(); v->n_contigs = (uint32_t)bi->contigs.size(); v->first_alt_location = bi->first_alt;
    v->on_device = 1;
    v->contig_proj_begin = bi->proj_begin.data(); v->contig_proj_rc = bi->proj_rc.data();
    v->contig_cigar_start = bi->cigar_start.data(); v->cigar_ops = bi->cigar_ops.data();
    return SNAPGPU_OK;
}

// ---------------------------------------------------------------- the four files
namespace {
bool ib_write_all(FILE *f, const void *p, size_t n) {
    const char *c = (const char *)p;
    while (n) { const size_t w = n < ((size_t)1 << 28) ? n : ((size_t)1 << 28); if (fwrite(c, 1, w, f) != w) return false; c += w; n -= w; }
    return true;
}
// device -> file through a bounded pinned-size host buffer
int ib_copy_out(FILE *f, const void *d_src, size_t bytes, std::vector<uint8_t> &buf) {
    const size_t chunk = buf.size();
    for (size_t off = 0; off < bytes; off += chunk) {
        const size_t n = bytes - off < chunk ? bytes - off : chunk;
        if (hipMemcpy(buf.data(), (const uint8_t *)d_src + off, n, hipMemcpyDeviceToHost) != hipSuccess) return ib_fail(SNAPGPU_E_NODEVICE, "device-to-host copy failed while saving the index");
        if (!ib_write_all(f, buf.data(), n)) return ib_fail(SNAPGPU_E_INVALID, std::string("write failed while saving the index: ") + strerror(errno));
    }
    return SNAPGPU_OK;
}
}  // namespace

extern "C" int snapgpu_built_index_save(const snapgpu_built_index *bi, const char *directory)
{
    if (!bi || !directory) return ib_fail(SNAPGPU_E_INVALID, "snapgpu_built_index_save: null argument");
    IBCHK(hipSetDevice(bi->device), SNAPGPU_E_NODEVICE);
    if (mkdir(directory, 0777) != 0 && errno != EEXIST) return ib_fail(SNAPGPU_E_INVALID, std::string("failed to create directory ") + directory);
    const std::string dir(directory);
    std::vector<uint8_t> buf((size_t)256 << 20);
    int rc;
    {   // Genome (Genome::saveToFile, Genome.cpp:203-260): blanks in names become '_'
        FILE *f = fopen((dir + "/Genome").c_str(), "wb");
        if (!f) return ib_fail(SNAPGPU_E_INVALID, "unable to open " + dir + "/Genome");
        fprintf(f, "%lld %d %d\n", (long long)bi->n_bases, (int)bi->contigs.size(), 1);
        for (const IBContig &c : bi->contigs) {
            std::string name = c.name; for (auto &ch : name) if (ch == ' ') ch = '_';
            const std::string cigar = c.proj_cigar.empty() ? "*" : c.proj_cigar;
            fprintf(f, "%lld %x %d %lld %x %d %d %s %s\n", (long long)c.begin, c.is_alt ? 1 : 0, c.original_number, (long long)c.proj_begin,
                    c.proj_rc ? 1 : 0, (int)name.size(), (int)cigar.size(), name.c_str(), cigar.c_str());
        }
        rc = ib_copy_out(f, bi->d_genome_padded + bi->genome_pad, (size_t)bi->n_bases, buf);
        fclose(f);
        if (rc) return rc;
    }
    size_t hash_file_bytes = 0;
    {   // GenomeIndexHash: per table magic, tableSize, usedElementCount, keySize, valueSize, valueCount, invalidValue, slots (HashTable.cpp:199-262)
        FILE *f = fopen((dir + "/GenomeIndexHash").c_str(), "wb");
        if (!f) return ib_fail(SNAPGPU_E_INVALID, "unable to open " + dir + "/GenomeIndexHash");
        for (uint32_t t = 0; t < bi->n_tables; t++) {
            const uint32_t magic = 0xb111b010u, ks = bi->key_bytes, vs = 4, vc = 1, inv = 0xffffffffu;
            const uint64_t size = bi->table_size[t], used = bi->table_used[t];
            if (!ib_write_all(f, &magic, 4) || !ib_write_all(f, &size, 8) || !ib_write_all(f, &used, 8) || !ib_write_all(f, &ks, 4) ||
                !ib_write_all(f, &vs, 4) || !ib_write_all(f, &vc, 4) || !ib_write_all(f, &inv, 4)) { fclose(f); return ib_fail(SNAPGPU_E_INVALID, "write failed (GenomeIndexHash)"); }
            rc = ib_copy_out(f, (const uint8_t *)bi->d_hash + bi->table_offset[t], (size_t)size * (4 + ks), buf);
            if (rc) { fclose(f); return rc; }
            hash_file_bytes += 36 + (size_t)size * (4 + ks);
        }
        fclose(f);
    }
    {   // OverflowTable
        FILE *f = fopen((dir + "/OverflowTable").c_str(), "wb");
        if (!f) return ib_fail(SNAPGPU_E_INVALID, "unable to open " + dir + "/OverflowTable");
        rc = ib_copy_out(f, bi->d_overflow, (size_t)bi->overflow_words * 4, buf);
        fclose(f);
        if (rc) return rc;
    }
    {   // GenomeIndex, written last: its presence says the directory is complete (GenomeIndex.cpp:1007-1008)
        FILE *f = fopen((dir + "/GenomeIndex").c_str(), "w");
        if (!f) return ib_fail(SNAPGPU_E_INVALID, "unable to open " + dir + "/GenomeIndex");
        fprintf(f, "%d %d %d %lld %d %d %d %lld %d %d", 7, 1, (int)bi->n_tables, (long long)bi->overflow_words, (int)bi->seed_len,
                (int)bi->chromosome_padding, (int)bi->key_bytes, (long long)hash_file_bytes, 1, 4);
        fclose(f);
    }
    return SNAPGPU_OK;
}
