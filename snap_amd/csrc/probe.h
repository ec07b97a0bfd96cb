// probe.h -- seed packing and GenomeIndex hash probe, wave-cooperative.
//
// Restates (does not copy) the behaviour of:
//   Seed::Seed / DoesTextRepresentASeed      SNAPLib/Seed.h:40-53, Seed.cpp:29
//   SNAPHashTable::hash / GetFirstValueForKey SNAPLib/HashTable.h:72-118
//   GenomeIndex::lookupSeed32                 SNAPLib/GenomeIndex.cpp:2096-2157
//   GenomeIndex::fillInLookedUpResults32      SNAPLib/GenomeIndex.cpp:2160-2202
//
// GPU mapping: the reference walks one hash slot at a time (quadratic steps 1,4,9,16 then
// linear).  Here each lane of a 32-lane half-wave owns one position of that probe sequence,
// so the first 32 positions of the walk are fetched with one load per lane and the stop
// position is found with a ballot.  After the four quadratic steps the walk is linear, i.e.
// adjacent 8/12-byte slots: lanes 5.. read consecutive addresses, which is what makes the
// (common) absent-key walk of the reverse-complement probe cheap.  Lanes 0-31 probe the
// forward key while lanes 32-63 probe the reverse-complement key (small tables) so both
// dependent HBM round trips overlap.
#pragma once
#include "dev_common.h"
#include "bucket.h"

struct SeedBits {
    uint64_t bases;      // 2 bits/base, first base most significant (Seed.h:48)
    uint64_t rc;         // reverse complement (Seed.h:49)
    bool     valid;      // all bases in ACGT
};

// spread the low 32 bits of x to the even bit positions of a 64-bit word
static __device__ __forceinline__ uint64_t spread_bits(uint64_t x) {
    x &= 0xffffffffull;
    x = (x | (x << 16)) & 0x0000ffff0000ffffull;
    x = (x | (x << 8))  & 0x00ff00ff00ff00ffull;
    x = (x | (x << 4))  & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x << 2))  & 0x3333333333333333ull;
    x = (x | (x << 1))  & 0x5555555555555555ull;
    return x;
}

// Pack seed_len (<= 32) bases starting at text[0] (LDS or global, uniform pointer).
// Lane i encodes base i; two ballots give the bit-planes, which are bit-reversed /
// interleaved on the scalar unit.
static __device__ __forceinline__ SeedBits pack_seed(const uint8_t *text, uint32_t seed_len) {
    int lane = lane_id();
    uint32_t enc = 0;
    if ((uint32_t)lane < seed_len) enc = base_value(text[lane]);
    uint64_t in_seed = seed_len >= 64 ? ~0ull : ((1ull << seed_len) - 1);
    uint64_t b0 = BALLOT(enc & 1) & in_seed;
    uint64_t b1 = BALLOT(enc & 2) & in_seed;
    uint64_t bad = BALLOT(enc > 3) & in_seed;
    SeedBits s;
    s.valid = (bad == 0);
    // reverse complement: base i (complemented) lands at bits 2i+1..2i
    s.rc = spread_bits(~b0 & in_seed) | (spread_bits(~b1 & in_seed) << 1);
    // forward: base i lands at bits 2(L-1-i)+1..2(L-1-i)  -> reverse the planes within L bits
    uint64_t r0 = __brevll(b0) >> (64 - seed_len);
    uint64_t r1 = __brevll(b1) >> (64 - seed_len);
    s.bases = spread_bits(r0) | (spread_bits(r1) << 1);
    return s;
}

static __device__ __forceinline__ uint64_t murmur_finalizer(uint64_t key) {   // HashTable.h:72-85
    key ^= key >> 33;
    key *= 0xff51afd7ed558ccdull;
    key ^= key >> 33;
    key *= 0xc4ceb9fe1a85ec53ull;
    key ^= key >> 33;
    return key;
}

struct HitList {
    int64_t         n_hits;     // nHits as the reference reports it
    const uint32_t *hits;       // overflow list (n_hits > 1) -- descending genome locations
    uint32_t        singleton;  // the hit when n_hits == 1
    uint32_t        slots;      // hash slots examined (for the roofline byte model)
};

static __device__ __forceinline__ uint32_t load_u32_bytes(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// Both strands of one seed.  Wave-uniform inputs; results are wave-uniform.
// out[0] = forward, out[1] = reverse complement.
static __device__ __forceinline__ void lookup_seed(const DevIndex &ix, const SeedBits &seed, HitList out[2]) {
    const int lane = lane_id();
    const uint32_t key_bits = ix.key_bytes * 8;
    uint32_t sub_entry[2]; bool have[2]; uint32_t sl[2];
    if (ix.bucket_blob != nullptr) {
        // device-native layout (bucket.h): lanes 0-7 read the forward key's bucket, lanes 8-15 the reverse complement's -- one line each
        const int g = lane >> 3;
        const uint64_t bases_g = g == 0 ? seed.bases : seed.rc;
        const uint32_t key = (uint32_t)(bases_g & ((1ull << key_bits) - 1));
        const uint32_t table = (uint32_t)(bases_g >> key_bits);
        uint32_t lines = 0;
        const uint32_t v = bucket_probe8(ix.bucket_blob, ix.bucket_offset, ix.n_buckets, table, key, g < 2, &lines);
        for (int d = 0; d < 2; d++) {
            sub_entry[d] = (uint32_t)__builtin_amdgcn_readlane((int)v, 8 * d);
            have[d] = sub_entry[d] != BUCKET_INVALID;
            sl[d] = 8u * (uint32_t)__builtin_amdgcn_readlane((int)lines, 8 * d);      // 8 dword pairs = 64 bytes per bucket read
        }
    } else {
    const int half = lane >> 5;          // which probe this lane works on
    const int sub = lane & 31;           // position within the probe sequence window
    const uint32_t value_count = ix.large ? 2u : 1u;

    // Which 2-bit strings get probed (GenomeIndex.cpp:2105-2155).
    uint64_t probe_bases[2];
    bool looked_up_complement = false;
    int n_probes;
    if (ix.large) {
        looked_up_complement = seed.bases > seed.rc;      // isBiggerThanItsReverseComplement
        probe_bases[0] = looked_up_complement ? seed.rc : seed.bases;
        probe_bases[1] = probe_bases[0];
        n_probes = 1;
    } else {
        probe_bases[0] = seed.bases;
        probe_bases[1] = seed.rc;
        n_probes = 2;
    }

    const uint64_t my_bases = probe_bases[half];
    const uint64_t key_mask = key_bits >= 64 ? ~0ull : ((1ull << key_bits) - 1);
    const uint64_t key = my_bases & key_mask;                             // getLowBases
    const uint32_t table = key_bits >= 64 ? 0u : (uint32_t)(my_bases >> key_bits);   // getHighBases
    const uint64_t tsize = ix.table_size[table];
    const uint8_t *tbase = ix.hash_blob + ix.table_offset[table];
    const uint64_t h0 = murmur_finalizer(key) % tsize;
    const bool active = half < n_probes;
    const bool aligned4 = (ix.entry_bytes & 3) == 0;

    // Probe positions: n=0 -> +0; n=1..4 -> +1,+5,+14,+30 (cumulative squares); then +1 each.
    bool done = false;
    uint32_t found_val[2] = {0xffffffffu, 0xffffffffu};
    bool found = false;
    uint32_t slots_examined = 0;
    uint64_t base_n = 0;
    // nProbes > tableSize + QUADRATIC_CHAINING_DEPTH gives up (HashTable.h:99)
    const uint64_t max_n = tsize + 5;
    uint32_t v0 = 0, v1 = 0;
    while (true) {
        uint64_t n = base_n + (uint64_t)sub;
        uint64_t off = n <= 4 ? (n * (n + 1) * (2 * n + 1)) / 6 : 30 + (n - 4);
        uint64_t slot = h0 + off;
        if (slot >= tsize) slot %= tsize;
        bool stop = false, hit = false;
        if (active && n <= max_n) {
            const uint8_t *e = tbase + slot * ix.entry_bytes;
            uint64_t k = 0;
            if (aligned4) {
                const uint32_t *e32 = (const uint32_t *)e;
                v0 = e32[0];
                if (value_count == 2) v1 = e32[1];
                k = e32[value_count];
                if (ix.key_bytes > 4) k |= (uint64_t)e32[value_count + 1] << 32;
            } else {
                v0 = load_u32_bytes(e);
                if (value_count == 2) v1 = load_u32_bytes(e + 4);
                for (uint32_t b = 0; b < ix.key_bytes; b++) k |= (uint64_t)e[4 * value_count + b] << (8 * b);
            }
            k &= key_mask;
            bool key_eq = (k == key);
            bool invalid = (v0 == 0xffffffffu);       // doesEntryHaveInvalidValue: first value only
            // first slot: must match AND be valid; later slots: stop at match OR empty (HashTable.h:91,106)
            stop = (n == 0) ? (key_eq && !invalid) : (key_eq || invalid);
            hit = stop && !invalid;
        }
        uint64_t stop_mask = BALLOT(stop);
        // each half looks at its own 32 bits
        uint32_t my_mask = half ? (uint32_t)(stop_mask >> 32) : (uint32_t)stop_mask;
        // A half that is already finished (or inactive) must not block the other one: handled by
        // per-half state below.  All values derived from ballots are uniform within the half.
        bool half_done_now = (my_mask != 0);
        int first = half_done_now ? __ffs((int)my_mask) - 1 : 0;
        // shuffles are executed by every lane (no divergence around cross-lane ops)
        int src = (half << 5) + (first & 31);
        bool hit_f = __shfl((int)hit, src) != 0;
        uint32_t a = (uint32_t)__shfl((int)v0, src);
        uint32_t b = (uint32_t)__shfl((int)v1, src);
        if (!done) {
            if (half_done_now) {
                found = hit_f;
                found_val[0] = a; found_val[1] = b;
                slots_examined += (uint32_t)first + 1;
                done = true;
            } else {
                slots_examined += 32;
                if (base_n + 32 > max_n) done = true;   // walked the whole table
            }
        }
        bool all_done = __all(done || !active);
        if (all_done) break;
        base_n += 32;
    }

    // Hand each half's answer to the whole wave.
    uint32_t fv[2][2]; bool fnd[2];
    for (int p = 0; p < 2; p++) {
        int src = p << 5;
        fnd[p] = __shfl((int)found, src) != 0;
        fv[p][0] = first_u32((uint32_t)__shfl((int)found_val[0], src));
        fv[p][1] = first_u32((uint32_t)__shfl((int)found_val[1], src));
        sl[p] = first_u32((uint32_t)__shfl((int)slots_examined, src));
    }

    // Map probe results to (forward, rc) sub-entries (GenomeIndex.cpp:2130-2153).
    if (ix.large) {
        have[0] = have[1] = fnd[0];
        sub_entry[0] = looked_up_complement ? fv[0][1] : fv[0][0];
        sub_entry[1] = looked_up_complement ? fv[0][0] : fv[0][1];
        sl[1] = 0;
    } else {
        have[0] = fnd[0]; have[1] = fnd[1];
        sub_entry[0] = fv[0][0]; sub_entry[1] = fv[1][0];
    }
    }   // (reference slot walk)

    const uint32_t n_bases32 = (uint32_t)ix.n_bases;
    for (int d = 0; d < 2; d++) {
        HitList r; r.n_hits = 0; r.hits = nullptr; r.singleton = 0; r.slots = sl[d];
        if (have[d]) {
            uint32_t v = sub_entry[d];
            if ((uint64_t)v < ix.n_bases) {             // singleton
                r.n_hits = 1; r.singleton = v;
            } else if (v == 0xfffffffeu) {              // other strand only (large tables)
                r.n_hits = 0;
            } else {
                uint32_t ofs = v - n_bases32;
                uint32_t cnt = first_u32(ix.overflow[ofs]);
                r.n_hits = (int64_t)(int32_t)cnt;       // `int hitCount = overflowTable32[...]`
                r.hits = ix.overflow + ofs + 1;
            }
        }
        out[d] = r;
    }
    if (ix.large && seed.bases == seed.rc) {            // isOwnReverseComplement: same hits both ways
        out[1] = out[0];
        out[1].slots = 0;
    }
}
