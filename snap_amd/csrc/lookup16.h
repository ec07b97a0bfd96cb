// lookup16.h -- GenomeIndex::lookupSeed32 for a batch of seeds, sixteen probes per wavefront pass (the stand-alone index-probe kernel).
//
// What a lookup returns is the reference's (SNAPLib/GenomeIndex.cpp:2096-2202: for the seed and for its reverse complement, the hit
// count and the descending hit list).  This kernel is about how many independent 64-byte lines one wavefront keeps in flight -- the
// probe is bound by memory-level parallelism, not by bytes:
//   * EIGHT seeds per pass, both strands: sixteen probe groups of four lanes.  The pass's 8 x 20 bases arrive as one coalesced load of
//     40 dwords; every lane turns its four bases into one byte of the packed seed (and one byte of the reverse complement's), and a
//     group collects its five bytes with lane shuffles: no per-seed ballots, no scalar bit-plane work.
//   * a group reads its bucket (bucket.h: 64 bytes = 7 {key, value} entries + control word) as four 16-byte loads: sixteen lines in
//     flight per wave where k_lookup_seeds has eight;
//   * the overflow header is not a round trip of its own: the group reads [count | first 15 hits] as one 64-byte access, so a seed with
//     at most 15 hits in a direction is finished after TWO dependent accesses (bucket, list head); longer lists continue 16 hits per
//     group per step, all groups side by side.
//   * work counters are kept per lane and reduced once per wave, not once per pass.
// Shape: seed length 20 (five whole dwords per seed), bucket tables present; everything else stays with k_lookup_seeds (snapgpu.hip).
#pragma once
#include "dev_common.h"
#include "probe.h"

struct __attribute__((packed, aligned(4))) LkWords4 { uint32_t a, b, c, d; };      // four consecutive words at a 4-byte aligned address

#ifndef LK_DEPTH
#define LK_DEPTH 4              // 16-byte list loads per lane in flight in the long-list loop (LK_DEPTH * 16 hits per group per round)
#endif
#ifndef LK_MINBLOCKS
#define LK_MINBLOCKS 8          // 64 VGPRs, eight blocks per CU resident: 6.50 G lookups/s against 6.31 G at 82 VGPRs / five blocks (profiles/r04j)
#endif
__global__ __launch_bounds__(256, LK_MINBLOCKS) void k_lookup_seeds20(
DevIndex ix, uint32_t n, const uint8_t *seeds, long long *n_hits, uint32_t *hits,
                                                        uint32_t max_hits_out, unsigned long long *counters)
{
    const int lane = lane_id();
    const int q = lane >> 2, e = lane & 3;                 // probe group, lane inside it
    const int sidx = q >> 1, dir = q & 1;                  // which of the pass's eight seeds, which strand
    const uint32_t n_bases32 = (uint32_t)ix.n_bases;
    const uint64_t ovf_last = ix.overflow_size ? ix.overflow_size - 1 : 0;
    unsigned long long c_lookups = 0, c_lines = 0, c_hits = 0, c_lists = 0;
    uint32_t sink = 0;
    // (a fixed stride per wave: the host sizes the grid to what is resident at once -- hipOccupancyMaxActiveBlocksPerMultiprocessor -- so that
    //  every wave starts at the beginning and does an equal share.  Round 3 launched 32 waves per CU where the registers let 20 .. 24 run, and
    //  the launch lasted as long as the quarter that started late; handing passes out with an atomic counter instead was measured TWICE AS
    //  SLOW (profiles/r04i: 344 k same-address device-scope atomics in 2 ms).)
    const uint32_t wave = (uint32_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t n_waves = (uint32_t)((gridDim.x * blockDim.x) >> 6);
    {
      for (uint32_t base = wave * 8; base < n; base += n_waves * 8) {
        const uint32_t n_here = n - base < 8 ? n - base : 8;
        // ---- the pass's text: 5 dwords per seed
        uint32_t w = 0x4e4e4e4eu;                                              // 'NNNN'
        if ((uint32_t)lane < 5 * n_here) w = ((const uint32_t *)(seeds + (size_t)base * 20))[lane];
        const uint32_t c0 = base_value((uint8_t)w), c1 = base_value((uint8_t)(w >> 8)), c2 = base_value((uint8_t)(w >> 16)), c3 = base_value((uint8_t)(w >> 24));
        const uint32_t bad = (c0 | c1 | c2 | c3) > 3u ? 1u : 0u;
        // first base most significant (Seed.h:48); the reverse complement reads the bases backwards, complemented (A0 G1 C2 T3: 3 - code)
        const uint32_t fw = (c0 << 6) | (c1 << 4) | (c2 << 2) | c3;
        const uint32_t rc = ((c3 ^ 3u) << 6) | ((c2 ^ 3u) << 4) | ((c1 ^ 3u) << 2) | (c0 ^ 3u);
        const uint32_t packed = (fw & 0xffu) | ((rc & 0xffu) << 8) | (bad << 16);
        uint64_t bits = 0; uint32_t any_bad = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint32_t v = (uint32_t)__shfl((int)packed, 5 * sidx + (dir ? 4 - i : i));
            bits = (bits << 8) | (uint64_t)(dir ? ((v >> 8) & 0xffu) : (v & 0xffu));
            any_bad |= v >> 16;
        }
        const uint32_t seed_i = base + (uint32_t)sidx;
        const bool in_range = seed_i < n;
        const bool active = in_range && !any_bad;
        const uint32_t key = (uint32_t)bits, table = (uint32_t)(bits >> 32);   // low 16 bases = key, high 4 bases = table (key size 4)

        // ---- the bucket: four lanes x 16 bytes; a full bucket sends the probe on to the next one (bucket.h)
        const uint64_t nb = active ? ix.n_buckets[table] : 1;
        const uint8_t *tb = ix.bucket_blob + (active ? ix.bucket_offset[table] : 0);
        uint64_t b = bucket_home(key, nb);
        bool done = !active;
        uint32_t val = BUCKET_INVALID, nl = 0;
        for (;;) {
            uint32_t k0 = BUCKET_INVALID, v0 = BUCKET_INVALID, k1 = BUCKET_INVALID, v1 = BUCKET_INVALID;
            if (!done) {
                const uint4 x = *(const uint4 *)(tb + b * BUCKET_BYTES + (size_t)e * 16);
                k0 = x.x; v0 = x.y; k1 = x.z; v1 = x.w;
            }
            const bool m0 = !done && k0 == key && v0 != BUCKET_INVALID;                        // entries 2e (0, 2, 4, 6)
            const bool m1 = !done && e < 3 && k1 == key && v1 != BUCKET_INVALID;               // entries 2e + 1 (1, 3, 5); lane 3's second pair is the control word
            const uint32_t cand = m0 ? v0 : v1;
            const unsigned long long m = BALLOT(m0 || m1);
            const uint32_t gm = (uint32_t)(m >> (q * 4)) & 0xfu;
            const uint32_t got = (uint32_t)__shfl((int)cand, q * 4 + (gm ? (int)__builtin_ctz(gm) : 0));
            const uint32_t flags = (uint32_t)__shfl((int)v1, q * 4 + 3);
            if (!done) {
                nl++;
                if (gm) { val = got; done = true; }
                else if (!(flags & 1u)) done = true;
                else b = b + 1 == nb ? 0 : b + 1;
            }
            if (!BALLOT(!done)) break;
        }
        // ---- decode (GenomeIndex.cpp:2160-2202): absent / one location / overflow list
        long long nh = active ? 0 : -1;
        uint32_t ofs = 0; bool is_list = false;
        if (active && val != BUCKET_INVALID) {
            if ((uint64_t)val < ix.n_bases) nh = 1;
            else if (val != 0xfffffffeu) { ofs = val - n_bases32; is_list = true; }
        }
        // ---- list head: [count | 15 hits] as one access of the group
        uint32_t cnt = 0;
        uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
        if (is_list) {
            const uint64_t at = (uint64_t)ofs + 4u * (uint32_t)e;
            if (at + 3 <= ovf_last) {
                const LkWords4 x = *(const LkWords4 *)(ix.overflow + at);
                h0 = x.a; h1 = x.b; h2 = x.c; h3 = x.d;
            } else {                                                                        // (only the table's very last lists get here)
                if (at <= ovf_last) h0 = ix.overflow[at];
                if (at + 1 <= ovf_last) h1 = ix.overflow[at + 1];
                if (at + 2 <= ovf_last) h2 = ix.overflow[at + 2];
            }
        }
        cnt = (uint32_t)__shfl((int)h0, q * 4);
        if (is_list) nh = (long long)(int32_t)cnt;
        if (in_range && e == 0) n_hits[2 * (size_t)seed_i + (uint32_t)dir] = nh;
        const long long lim = nh < (long long)max_hits_out ? nh : (long long)max_hits_out;      // hits to consume, as BaseAligner does
        uint32_t *dst = hits ? hits + (2 * (size_t)seed_i + (uint32_t)dir) * max_hits_out : nullptr;
        if (active && e == 0) {
            c_lines += nl; if (dir == 0) c_lookups++;
            if (nh > 1) c_lists++;
            if (lim > 0) c_hits += (unsigned long long)lim;
        }
        if (nh == 1 && active) { if (dst) { if (e == 0) dst[0] = val; } else sink ^= val; }
        if (is_list && lim > 0) {
            // words 1 .. 15 of the head are hits 0 .. 14
            const long long j0 = 4 * e - 1;
            const uint32_t hv[4] = {h0, h1, h2, h3};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const long long j = j0 + c;
                if (j >= 0 && j < lim) { if (dst) dst[j] = hv[c]; else sink ^= hv[c]; }
            }
        }
        // ---- the rest of the long lists: 16 hits per group per step, FOUR steps' loads issued before the first is consumed.  A pass lasts as
        // long as its longest list (-h 300: 19 steps for a popular seed, and with sixteen lookups per pass most passes hold one), and a step
        // that waits for its own load before the next is issued pays the full memory latency 19 times over (round 3: 4.2 G lookups/s, 1.0 TB/s
        // at the memory side).  The addresses are known up front, so the loads of a whole 64-hit stretch go out together.
        long long next = 15;
        while (BALLOT(is_list && next < lim)) {
            LkWords4 xs[LK_DEPTH];
            bool wide[LK_DEPTH];
#pragma unroll
            for (int c4 = 0; c4 < LK_DEPTH; c4++) {
                const long long j = next + 16 * c4 + 4 * e;
                const uint64_t at = (uint64_t)ofs + 1u + (uint64_t)j;
                wide[c4] = is_list && j + 3 < lim && at + 3 <= ovf_last;
                xs[c4] = LkWords4{0u, 0u, 0u, 0u};
                if (wide[c4]) xs[c4] = *(const LkWords4 *)(ix.overflow + at);
            }
#pragma unroll
            for (int c4 = 0; c4 < LK_DEPTH; c4++) {
                const long long j = next + 16 * c4 + 4 * e;
                const uint64_t at = (uint64_t)ofs + 1u + (uint64_t)j;
                if (is_list && j < lim) {
                    if (wide[c4]) {
                        const LkWords4 x = xs[c4];
                        if (dst) { dst[j] = x.a; dst[j + 1] = x.b; dst[j + 2] = x.c; dst[j + 3] = x.d; } else sink ^= x.a ^ x.b ^ x.c ^ x.d;
                    } else {
                        for (int c = 0; c < 4 && j + c < lim; c++) { const uint32_t hvv = ix.overflow[at + (uint64_t)c]; if (dst) dst[j + c] = hvv; else sink ^= hvv; }
                    }
                }
            }
            next += 16 * LK_DEPTH;
        }
      }
    }
    if (sink == 0xDEADBEEFu && n == 0xFFFFFFFFu) n_hits[0] = (long long)sink;       // keeps the hit loads alive when nothing is stored
    if (counters) {
        // one reduction per wave
        auto wsum = [&](unsigned long long v) -> unsigned long long {
            for (int o = 32; o >= 1; o >>= 1) {
                const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o);
                v += ((unsigned long long)hi << 32) | lo;
            }
            return v;
        };
        c_lookups = wsum(c_lookups); c_lines = wsum(c_lines); c_hits = wsum(c_hits); c_lists = wsum(c_lists);
        if (lane == 0) {
            atomicAdd(&counters[1], c_lookups); atomicAdd(&counters[2], 8ull * c_lines);      // (8 slots of 8 bytes = one bucket line)
            atomicAdd(&counters[3], c_hits); atomicAdd(&counters[4], c_lists);
        }
    }
}
