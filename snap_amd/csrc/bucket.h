// bucket.h -- device-native layout of the seed hash tables: one 64-byte line answers a probe.
//
// The reference's closed hash table (SNAPLib/HashTable.h:87-118) is walked slot by slot -- quadratic steps 1, 4, 9, 16, then
// linear -- until the key or an empty slot turns up: 1.6 slots for a key that is there, 7+ for one that is not, and with the default
// small tables the reverse-complement strand of a seed is almost always "not there" (SURVEY.md 8(d)).  Those walks are dependent
// 8-byte reads scattered over two or three lines.  This layout is built ON THE GPU when a context is created, from the reference's own
// slot arrays already in HBM (so every way an index gets there -- upload, adoption of broadcast blobs, replica sharing -- ends with the
// same tables), and changes nothing about WHAT a lookup returns (GenomeIndex::lookupSeed32, GenomeIndex.cpp:2096-2202: value word of
// the key or "absent"), only where the bytes sit:
//   * table t (the seed's high bases, as in the reference) = n_buckets[t] buckets of 64 bytes;
//   * bucket = 7 entries {key32, value32} + one control word {count, flags}; an absent entry has value 0xffffffff
//     (SNAPHashTable's own "unused" value, HashTable.h:148-156);
//   * home bucket of a key = mulhi(murmur(key), n_buckets[t]) (the reference's finalizer, HashTable.h:72-85);
//   * a full bucket sets flag bit 0 and its late-comers go to the next bucket (cyclically); a probe continues past a bucket only if
//     that flag is set.  n_buckets = slots / 5.5, i.e. ~4.2 entries per bucket at the reference's load factor: 5 % of the buckets spill.
// Supported shape: 4-byte values, keys of at most 4 bytes, small tables (the north star's -s 20 index; also -s 16..20); every other shape
// (5-8 byte keys, -large) keeps the reference's slot walk of probe.h.
#pragma once
#include "dev_common.h"

#define BUCKET_ENTRIES 7
#define BUCKET_BYTES 64
#define BUCKET_INVALID 0xffffffffu

static __host__ __device__ __forceinline__ uint64_t bucket_count_for(uint64_t table_slots) {
    uint64_t nb = (table_slots * 2 + 10) / 11;            // slots / 5.5, rounded up
    return nb < 1 ? 1 : nb;
}

static __device__ __forceinline__ uint64_t bucket_murmur(uint64_t key) {   // HashTable.h:72-85
    key ^= key >> 33; key *= 0xff51afd7ed558ccdull; key ^= key >> 33; key *= 0xc4ceb9fe1a85ec53ull; key ^= key >> 33;
    return key;
}
static __device__ __forceinline__ uint64_t bucket_home(uint64_t key, uint64_t n_buckets) {
    return __umul64hi(bucket_murmur(key), n_buckets);
}

// every entry absent, every control word zero
static __global__ void k_bucket_init(uint32_t *blob, uint64_t n_buckets_total)
{
    const uint64_t n_words = n_buckets_total * (BUCKET_BYTES / 4);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x)
        blob[i] = (i % (BUCKET_BYTES / 4)) >= 2 * BUCKET_ENTRIES ? 0u : BUCKET_INVALID;
}

// one reference table (slots of 8 bytes: value32, key32 -- entry layout HashTable.h:148-156) into its buckets
static __global__ void k_bucket_build(const uint32_t *slots, uint64_t n_slots, uint32_t key_mask, uint32_t *buckets, uint64_t n_buckets)
{
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = slots[2 * s];
        if (v == BUCKET_INVALID) continue;
        const uint32_t key = slots[2 * s + 1] & key_mask;
        uint64_t b = bucket_home(key, n_buckets);
        for (uint64_t tries = 0; tries <= n_buckets; tries++) {
            uint32_t *bk = buckets + b * (BUCKET_BYTES / 4);
            const uint32_t pos = atomicAdd(&bk[2 * BUCKET_ENTRIES], 1u);
            if (pos < BUCKET_ENTRIES) { bk[2 * pos] = key; bk[2 * pos + 1] = v; break; }
            atomicOr(&bk[2 * BUCKET_ENTRIES + 1], 1u);
            b = b + 1 == n_buckets ? 0 : b + 1;
        }
    }
}

// Eight independent probes per wavefront: lanes 8g .. 8g+7 serve probe g = (table, key), wave-uniform WITHIN the group; lane e of a group
// reads dword pair e of the bucket, so one probe costs one 64-byte line (plus one more per spilled bucket on the way).  Returns the value
// word (BUCKET_INVALID: absent) in every lane of the group; *lines = buckets read by the group.
static __device__ __forceinline__ uint32_t bucket_probe8(const uint8_t *blob, const uint64_t *bucket_offset, const uint64_t *n_buckets_t,
                                                         uint32_t table, uint32_t key, bool active, uint32_t *lines)
{
    const int lane = lane_id();
    const int g = lane >> 3, e = lane & 7;
    const uint64_t nb = active ? n_buckets_t[table] : 1;
    const uint8_t *base = blob + (active ? bucket_offset[table] : 0);
    uint64_t b = bucket_home(key, nb);
    bool done = !active;
    uint32_t val = BUCKET_INVALID, nl = 0;
    for (;;) {
        uint32_t ek = BUCKET_INVALID, ev = BUCKET_INVALID;
        if (!done) {
            const uint2 w = *(const uint2 *)(base + b * BUCKET_BYTES + (size_t)e * 8);
            ek = w.x; ev = w.y;
        }
        const bool match = !done && e < BUCKET_ENTRIES && ek == key && ev != BUCKET_INVALID;
        const unsigned long long m = BALLOT(match);
        const uint32_t gm = (uint32_t)(m >> (g * 8)) & 0xffu;
        const int src = g * 8 + (gm ? (int)__builtin_ctz(gm) : BUCKET_ENTRIES);
        const uint32_t got = (uint32_t)__shfl((int)ev, src);             // the matching entry's value, or the control word's flags
        if (!done) {
            nl++;
            if (gm) { val = got; done = true; }
            else if (!(got & 1u)) done = true;                          // bucket never spilled: the key is not in the table
            else b = b + 1 == nb ? 0 : b + 1;
        }
        if (!BALLOT(!done)) break;
    }
    *lines = nl;
    return val;
}
