// index_build.h -- device side of the GPU index builder.
//
// What it builds is what GenomeIndex::BuildIndexToDirectory builds (SNAPLib/GenomeIndex.cpp:527-1022): for every genome location whose
// seed_len bases are all ACGT, (seed -> location); a seed that occurs once keeps its location in its hash-table slot, a seed that occurs
// n > 1 times points at [n][loc_0 > loc_1 > ...] in the overflow table (:759-889).  The reference gets there with per-table locks, linked
// back-pointer lists and a qsort per list, in minutes to hours.  Here it is a sort:
//   1. k_ib_keys         every location's 2-bit packed seed (Seed.h:40-53), or "no seed here"; 64 locations per wavefront step from two
//                        coalesced loads and six ballots (the wave's 128 base codes as bit planes; lane i's seed is bits [i, i + L))
//   2. LSD radix sort    of (seed, location) by seed, 6 bits per pass (one bin per lane: counts, offsets and running ranks are wave
//                        operations), stable, so locations stay ascending inside a seed; the first pass drops the non-seeds
//   3. run detection     equal neighbours = one seed; scans give each seed its run, each run its overflow offset
//   4. k_ib_fill_overflow / k_ib_insert   overflow lists written back to front (descending), one 64-bit compare-and-swap per seed into the
//                        reference's own closed hash table (value32 | key32, quadratic-then-linear probe, HashTable.h:87-118) -- so the
//                        tables ARE the reference's format: its loader reads them, and the device-native buckets are built from them
//                        like from any other index.
// Slot placement differs from a reference build (it depends on insertion order there too); lookup results do not.
// Shape built here: 4-byte locations, small tables, any key size the reference accepts for the seed (GenomeIndex.cpp:437-460; -s 20 with its
// 4-byte keys is the north star's index and keeps the one-compare-and-swap insert; other key sizes claim slots in a bit array, below).
// Everything is wave-level: no block barriers, LDS only per wave.
#pragma once
#include "dev_common.h"
#include "probe.h"

#define IB_INVALID_KEY 0xFFFFFFFFFFFFFFFFull
#define IB_TILE 1024u            // elements per wavefront tile: 16 rounds of 64
#define IB_ROUNDS 16
#define IB_BITS 6                // radix bits per pass: one bin per lane
#define IB_EMPTY_SLOT 0x00000000FFFFFFFFull      // value = invalid (0xffffffff), key bytes cleared (HashTable.cpp:63-70)

static __device__ __forceinline__ uint32_t ib_wave_id()  { return (uint32_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6); }
static __device__ __forceinline__ uint32_t ib_n_waves()  { return (uint32_t)((gridDim.x * blockDim.x) >> 6); }

// inclusive prefix sum over the 64 lanes
static __device__ __forceinline__ uint32_t ib_wave_incl_scan(uint32_t v) {
    const int lane = lane_id();
    for (int o = 1; o < WAVE; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)v, o); if (lane >= o) v += t; }
    return v;
}

// ---------------------------------------------------------------- 1. seeds of all locations
// keys[loc] = the seed at loc (first base most significant, A0 G1 C2 T3: Seed.h:48, Tables.cpp:52-58) or IB_INVALID_KEY when one of its
// bases is not ACGT (GenomeIndex.cpp:1464-1470: such locations are not indexed; that includes every seed that reaches into the 'n'
// padding between contigs).  genome must be readable up to loc + 127 + seed_len (the context's genome_pad covers it).
__global__ __launch_bounds__(256) void k_ib_keys(const uint8_t *genome, uint64_t n_locs, uint32_t L, uint64_t *keys)
{
    const int lane = lane_id();
    const uint64_t mask_l = L >= 64 ? ~0ull : ((1ull << L) - 1ull);
    for (uint64_t base = (uint64_t)ib_wave_id() * 64; base < n_locs; base += (uint64_t)ib_n_waves() * 64) {
        const uint32_t c0 = base_value(genome[base + (uint64_t)lane]), c1 = base_value(genome[base + 64 + (uint64_t)lane]);
        const uint64_t lo0 = BALLOT(c0 & 1u), lo1 = BALLOT(c0 & 2u), lob = BALLOT(c0 > 3u);
        const uint64_t hi0 = BALLOT(c1 & 1u), hi1 = BALLOT(c1 & 2u), hib = BALLOT(c1 > 3u);
        // bits [lane, lane + L) of the 128-bit planes
        const uint64_t w0 = ((lo0 >> lane) | (lane ? (hi0 << (64 - lane)) : 0ull)) & mask_l;
        const uint64_t w1 = ((lo1 >> lane) | (lane ? (hi1 << (64 - lane)) : 0ull)) & mask_l;
        const uint64_t wb = ((lob >> lane) | (lane ? (hib << (64 - lane)) : 0ull)) & mask_l;
        // base i of the window lands at bits 2(L-1-i)+1 .. 2(L-1-i): reverse the planes within L bits, then interleave
        const uint64_t r0 = __brevll(w0) >> (64 - L), r1 = __brevll(w1) >> (64 - L);
        const uint64_t key = spread_bits(r0) | (spread_bits(r1) << 1);
        const uint64_t loc = base + (uint64_t)lane;
        if (loc < n_locs) keys[loc] = wb ? IB_INVALID_KEY : key;
    }
}

// ---------------------------------------------------------------- 2. radix sort, one pass = histogram, scan, scatter
// lanes with the same (valid) digit as this one
static __device__ __forceinline__ uint64_t ib_peers(uint32_t d, bool valid) {
    uint64_t p = BALLOT(valid);
#pragma unroll
    for (int b = 0; b < IB_BITS; b++) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = BALLOT(bit);
        p &= bit ? m : ~m;
    }
    return valid ? p : 0ull;
}

// hist[bin * n_tiles + tile] = elements of the tile whose digit is bin (bin-major, so ONE exclusive scan over the whole array yields the
// output offset of every (bin, tile)).  IB_INVALID_KEY elements are not counted: the first pass drops them.
__global__ __launch_bounds__(256) void k_ib_hist(const uint64_t *keys, uint64_t n, uint32_t shift, uint32_t *hist, uint32_t n_tiles)
{
    static __shared__ uint32_t sh[4][64];
    const int lane = lane_id();
    uint32_t *h = sh[(threadIdx.x >> 6) & 3];
    for (uint32_t tile = ib_wave_id(); tile < n_tiles; tile += ib_n_waves()) {
        h[lane] = 0;
        WAVE_SYNC();
        for (int r = 0; r < IB_ROUNDS; r++) {
            const uint64_t i = (uint64_t)tile * IB_TILE + (uint64_t)r * 64 + (uint64_t)lane;
            const uint64_t key = i < n ? keys[i] : IB_INVALID_KEY;
            const bool valid = key != IB_INVALID_KEY;
            const uint32_t d = (uint32_t)(key >> shift) & 63u;
            const uint64_t peers = ib_peers(d, valid);
            if (valid && (peers & ((1ull << lane) - 1ull)) == 0) h[d] += (uint32_t)__popcll(peers);      // the lowest peer speaks for all
            WAVE_SYNC();
        }
        hist[(size_t)lane * n_tiles + tile] = h[lane];
        WAVE_SYNC();
    }
}

// Stable scatter of one pass.  vals_in == NULL (first pass): the value of element i is i, the genome location.
__global__ __launch_bounds__(256) void k_ib_scatter(const uint64_t *keys_in, const uint32_t *vals_in, uint64_t n, uint32_t shift,
                                                    const uint32_t *offs, uint32_t n_tiles, uint64_t *keys_out, uint32_t *vals_out)
{
    static __shared__ uint32_t sh[4][64];
    const int lane = lane_id();
    uint32_t *run = sh[(threadIdx.x >> 6) & 3];
    for (uint32_t tile = ib_wave_id(); tile < n_tiles; tile += ib_n_waves()) {
        run[lane] = offs[(size_t)lane * n_tiles + tile];          // where the tile's elements of bin `lane` start in the output
        WAVE_SYNC();
        for (int r = 0; r < IB_ROUNDS; r++) {
            const uint64_t i = (uint64_t)tile * IB_TILE + (uint64_t)r * 64 + (uint64_t)lane;
            const uint64_t key = i < n ? keys_in[i] : IB_INVALID_KEY;
            const bool valid = key != IB_INVALID_KEY;
            const uint32_t val = valid ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;
            const uint32_t d = (uint32_t)(key >> shift) & 63u;
            const uint64_t peers = ib_peers(d, valid);
            const uint64_t below = peers & ((1ull << lane) - 1ull);
            const uint32_t base = valid ? run[d] : 0u;
            WAVE_SYNC();
            if (valid) {
                const uint32_t pos = base + (uint32_t)__popcll(below);
                keys_out[pos] = key; vals_out[pos] = val;
                if (below == 0) run[d] = base + (uint32_t)__popcll(peers);
            }
            WAVE_SYNC();
        }
    }
}

// ---------------------------------------------------------------- exclusive scan of a u32 array (three kernels; chunk = 1024 per wave)
__global__ __launch_bounds__(256) void k_ib_scan_sums(const uint32_t *in, uint64_t n, uint32_t *partial, uint32_t n_chunks)
{
    const int lane = lane_id();
    for (uint32_t c = ib_wave_id(); c < n_chunks; c += ib_n_waves()) {
        uint32_t s = 0;
        for (int r = 0; r < IB_ROUNDS; r++) {
            const uint64_t i = (uint64_t)c * IB_TILE + (uint64_t)r * 64 + (uint64_t)lane;
            s += i < n ? in[i] : 0u;
        }
        s = ib_wave_incl_scan(s);
        if (lane == 63) partial[c] = s;
    }
}
// one wavefront: partial[] -> its exclusive scan in place; total[0] = the sum
__global__ __launch_bounds__(64) void k_ib_scan_partials(uint32_t *partial, uint32_t n_chunks, uint32_t *total)
{
    const int lane = lane_id();
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += 64) {
        const uint32_t c = c0 + (uint32_t)lane;
        const uint32_t v = c < n_chunks ? partial[c] : 0u;
        const uint32_t inc = ib_wave_incl_scan(v);
        if (c < n_chunks) partial[c] = carry + inc - v;
        carry += (uint32_t)__shfl((int)inc, 63);
    }
    if (lane == 0) total[0] = carry;
}
__global__ __launch_bounds__(256) void k_ib_scan_apply(const uint32_t *in, uint64_t n, const uint32_t *partial, uint32_t n_chunks, uint32_t *out)
{
    const int lane = lane_id();
    for (uint32_t c = ib_wave_id(); c < n_chunks; c += ib_n_waves()) {
        uint32_t carry = partial[c];
        for (int r = 0; r < IB_ROUNDS; r++) {
            const uint64_t i = (uint64_t)c * IB_TILE + (uint64_t)r * 64 + (uint64_t)lane;
            const uint32_t v = i < n ? in[i] : 0u;
            const uint32_t inc = ib_wave_incl_scan(v);
            if (i < n) out[i] = carry + inc - v;
            carry += (uint32_t)__shfl((int)inc, 63);
        }
    }
}

// ---------------------------------------------------------------- 3. runs of equal seeds
// head[i] = 1 where a new seed starts in the sorted array
__global__ __launch_bounds__(256) void k_ib_heads(const uint64_t *keys, uint64_t m, uint32_t *head)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x)
        head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// rid[i] (exclusive scan of head, in place semantics: rid_ex[i] = heads before i) -> run_start[run of i] = i for heads; run_start[n_runs] = m
__global__ __launch_bounds__(256) void k_ib_run_starts(const uint32_t *head, const uint32_t *heads_before, uint64_t m, uint32_t n_runs, uint32_t *run_start)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= m; i += (uint64_t)gridDim.x * blockDim.x) {
        if (i == m) run_start[n_runs] = (uint32_t)m;
        else if (head[i]) run_start[heads_before[i]] = (uint32_t)i;
    }
}
// words of overflow table a run needs: 0 for a unique seed, 1 + n for n > 1 occurrences (GenomeIndex.cpp:770)
__global__ __launch_bounds__(256) void k_ib_run_need(const uint32_t *run_start, uint32_t n_runs, uint32_t *need)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_runs; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t len = run_start[r + 1] - run_start[r];
        need[r] = len > 1 ? len + 1 : 0u;
    }
}
// ---------------------------------------------------------------- 4. overflow table and hash tables
// element i of a repeated seed goes to its list, last occurrence first ("reverse sorted ... it's necessary for correct functioning", :763)
// The total of `need` in 64 bits: the 32-bit scan above wraps silently when a multi-Gb, highly repetitive genome asks for 2^32 or more
// overflow words, and the address-space check after it must see the true total (GenomeIndex.cpp:684 / :777 fail there).
__global__ __launch_bounds__(256) void k_ib_sum64(const uint32_t *need, uint32_t n_runs, unsigned long long *total)
{
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_runs; i += (uint64_t)gridDim.x * blockDim.x) acc += need[i];
    if (acc) atomicAdd(total, acc);          // (at most one atomic per thread of a persistent grid)
}

__global__ __launch_bounds__(256) void k_ib_fill_overflow(const uint32_t *vals, const uint32_t *head, const uint32_t *heads_before, uint64_t m,
                                                          const uint32_t *run_start, const uint32_t *ovf_off, uint32_t *overflow)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = heads_before[i] + head[i] - 1u;
        const uint32_t s = run_start[r], len = run_start[r + 1] - s;
        if (len > 1) {
            const uint32_t off = ovf_off[r], j = (uint32_t)i - s;
            overflow[off + len - j] = vals[i];
            if (j == 0) overflow[off] = len;
        }
    }
}
// first run whose seed is >= bound[t] (bound[t] = t << key_bits; bound[n_tables] = everything): run-index boundaries of the hash tables
__global__ __launch_bounds__(256) void k_ib_table_bounds(const uint64_t *keys, const uint32_t *run_start, uint32_t n_runs, uint32_t key_bits,
                                                         uint32_t n_tables, uint32_t *first_run)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tables) return;
    if (t == n_tables) { first_run[t] = n_runs; return; }
    const uint64_t bound = key_bits >= 64 ? 0ull : (uint64_t)t << key_bits;       // (64 key bits: one table, t = 0)
    uint32_t lo = 0, hi = n_runs;                      // first run with key >= bound
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (keys[run_start[mid]] < bound) lo = mid + 1; else hi = mid;
    }
    first_run[t] = lo;
}
__global__ __launch_bounds__(256) void k_ib_fill_empty(unsigned long long *slots, uint64_t n_slots)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (uint64_t)gridDim.x * blockDim.x) slots[i] = IB_EMPTY_SLOT;
}
// One seed per thread into the reference's closed hash table of its high bases: slot = murmur(key) % size, then +1, +4, +9, +16, then
// linear (HashTable.h:87-118, QUADRATIC_CHAINING_DEPTH 5); the slot is taken with one 64-bit compare-and-swap {value32 | key32 << 32}.
// A taken slot never becomes empty again, so "no empty slot before a key on its probe sequence" -- what the lookup relies on -- holds
// whatever order the threads arrive in.  fail[0] is set if a table is full (cannot happen with slack > 0).
__global__ __launch_bounds__(256) void k_ib_insert(const uint64_t *keys, const uint32_t *vals, const uint32_t *run_start, const uint32_t *ovf_off,
                                                   uint32_t n_runs, uint32_t key_bits, uint32_t n_bases32, unsigned long long *blob,
                                                   const uint64_t *table_slot0 /* first slot of table t in blob */, const uint64_t *table_size,
                                                   uint32_t *fail)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_runs; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = run_start[r], len = run_start[r + 1] - s;
        const uint64_t seed = keys[s];
        const uint32_t key = (uint32_t)(seed & ((1ull << key_bits) - 1ull));      // (this kernel: key_bits == 32)
        const uint32_t t = (uint32_t)(seed >> key_bits);
        const uint32_t value = len == 1 ? vals[s] : n_bases32 + ovf_off[r];          // < nBases: the location; else nBases + overflow index (:2173-2201)
        const uint64_t size = table_size[t];
        unsigned long long *slots = blob + table_slot0[t];
        const unsigned long long entry = (unsigned long long)value | ((unsigned long long)key << 32);
        uint64_t idx = murmur_finalizer((uint64_t)key) % size;
        uint64_t probes = 0;
        for (;;) {
            const unsigned long long old = atomicCAS(&slots[idx], (unsigned long long)IB_EMPTY_SLOT, entry);
            if (old == IB_EMPTY_SLOT) break;
            probes++;
            if (probes > size + 5) { atomicAdd(fail, 1u); break; }
            idx = probes < 5 ? (idx + probes * probes) % size : (idx + 1) % size;
        }
    }
}

// The same for entries that are not 8 bytes wide (key sizes other than 4: the reference's default for seeds above 21 is 5 or more,
// GenomeIndex.cpp:437): entries then straddle words, so a slot is claimed in a bit array (one bit per slot of the whole blob) and its bytes --
// value, then key_bytes of key, little endian (HashTable.h:148-156) -- are written by the one thread that claimed it.  The invariant is the
// same: a claimed slot stays claimed, so no key has an empty slot before it on its probe sequence.
__global__ __launch_bounds__(256) void k_ib_fill_empty_wide(uint32_t *words, uint64_t n_words, uint32_t entry_bytes)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t w = 0;
        for (uint32_t b = 0; b < 4; b++) if ((4 * i + b) % entry_bytes < 4) w |= 0xffu << (8 * b);      // value bytes 0xff, key bytes 0 (HashTable.cpp:63-70)
        words[i] = w;
    }
}
__global__ __launch_bounds__(256) void k_ib_insert_wide(const uint64_t *keys, const uint32_t *vals, const uint32_t *run_start, const uint32_t *ovf_off,
                                                        uint32_t n_runs, uint32_t key_bits, uint32_t n_bases32, uint8_t *blob, uint32_t entry_bytes,
                                                        uint32_t *claim, const uint64_t *table_slot0, const uint64_t *table_size, uint32_t *fail)
{
    const uint64_t key_mask = key_bits >= 64 ? ~0ull : ((1ull << key_bits) - 1ull);
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_runs; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = run_start[r], len = run_start[r + 1] - s;
        const uint64_t seed = keys[s];
        const uint64_t key = seed & key_mask;
        const uint32_t t = key_bits >= 64 ? 0u : (uint32_t)(seed >> key_bits);
        const uint32_t value = len == 1 ? vals[s] : n_bases32 + ovf_off[r];
        const uint64_t size = table_size[t], slot0 = table_slot0[t];
        uint64_t idx = murmur_finalizer(key) % size;
        uint64_t probes = 0;
        for (;;) {
            const uint64_t g = slot0 + idx;
            const uint32_t bit = 1u << (g & 31);
            if (!(atomicOr(&claim[g >> 5], bit) & bit)) {
                uint8_t *e = blob + g * entry_bytes;
                for (uint32_t b = 0; b < 4; b++) e[b] = (uint8_t)(value >> (8 * b));
                for (uint32_t b = 4; b < entry_bytes; b++) e[b] = (uint8_t)(key >> (8 * (b - 4)));
                break;
            }
            probes++;
            if (probes > size + 5) { atomicAdd(fail, 1u); break; }
            idx = probes < 5 ? (idx + probes * probes) % size : (idx + 1) % size;
        }
    }
}
