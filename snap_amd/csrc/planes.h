// planes.h -- the genome as bit planes (second half: the "2-bit genome"), and Landau-Vishkin's mismatch bitmaps
// computed from planes.
//
// The reference keeps one byte per base (SNAPLib/Genome.h:450) and its Landau-Vishkin compares bytes, eight at a time
// (LandauVishkin.h:377-407 countPerfectMatch).  The device copy built here holds, for every 64 bases, three 64-bit words:
//   plane 0 / plane 1   the two bits of the base code (A0 G1 C2 T3, Tables.cpp:52-58)
//   plane N             the base is not ACGT; then plane 0 tells the two such bytes a SNAP genome holds apart: 'N' (0) and the 'n' of
//                       the padding between contigs (1) -- a read's 'N' equals the former and not the latter, as bytes do (and a read's
//                       'n', should one ever arrive, the latter)
// i.e. 3 bits per base, 24 bytes per 64 bases, interleaved so that the window of a candidate (WIN_PAD + read + WIN_PAD bases) is one
// contiguous 200-byte read instead of 416 bytes.  A read gets four planes per direction (code bits, 'N', "some other byte": never equal to
// anything in a genome).  "P(i) != T(d + i)" for 64 consecutive i is then a handful of 64-bit operations IN ONE LANE, so every lane builds
// the bitmap of its own diagonal and all 2k + 1 diagonals are built at once, where the byte form needs the whole wave (64 byte compares +
// a ballot) per diagonal and word.  What Landau-Vishkin computes from the bitmaps is unchanged (lv.h).
// The byte genome stays (affine gap, clipping and the SAM side read bytes): this is a shadow, built on the device at context creation.
#pragma once
#include "dev_common.h"

#define PLANE_WORDS_PER_BLOCK 3
// blocks of 64 bases staged per candidate window: up to 63 bits of misalignment + pad + read + pad, + 1 so that x[w + 1] exists
static __host__ __device__ __forceinline__ uint32_t text_plane_blocks(uint32_t RL, uint32_t win_pad) { return (63 + 2 * win_pad + RL + 63) / 64 + 1; }

// words per plane of a read: ceil(RL / 64) + 1 (the extra word lets a 64-bit window start anywhere: plane_bits_fwd)
static __host__ __device__ __forceinline__ uint32_t read_plane_words(uint32_t RL) { return (RL + 63) / 64 + 1; }

// one wave per 64 bytes of the padded genome
static __global__ __launch_bounds__(256) void k_genome_planes(const uint8_t *bytes, uint64_t n_bytes, uint64_t n_blocks, unsigned long long *planes)
{
    const int lane = lane_id();
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t b = wave; b < n_blocks; b += n_waves) {
        const uint64_t i = b * 64 + (uint64_t)lane;
        const uint8_t c = i < n_bytes ? bytes[i] : (uint8_t)'n';
        const uint32_t v = base_value(c);
        const bool non = v > 3u;
        const unsigned long long p0 = BALLOT(non ? c == 'n' : (v & 1u)), p1 = BALLOT(!non && (v & 2u)), pn = BALLOT(non);
        if (lane == 0) { planes[b * 3] = p0; planes[b * 3 + 1] = p1; planes[b * 3 + 2] = pn; }
    }
}

// 64 bits of a bit string held in words x[0 ..]: bit i of the result = bit (start + i) of the string (start may be negative or run past
// the words the caller cares about: such bits are whatever, and the callers mask them)
template <typename W>
static __device__ __forceinline__ unsigned long long plane_bits_fwd(const W *x, int start, int n_words) {
    const int w = start >> 6, o = start & 63;
    const unsigned long long lo = (w >= 0 && w < n_words) ? x[w] : 0ull, hi = (w + 1 >= 0 && w + 1 < n_words) ? x[w + 1] : 0ull;
    return o ? (lo >> o) | (hi << (64 - o)) : lo;
}
// bit i of the result = bit (start - i) of the string
template <typename W>
static __device__ __forceinline__ unsigned long long plane_bits_bwd(const W *x, int start, int n_words) {
    return __brevll(plane_bits_fwd(x, start - 63, n_words));
}

// What lv_compute needs to build its bitmaps from planes: P(i) = pattern bit (p_org + st * i), T(j) = text bit (t_org + st * j).
struct LvPlanes {
    const LDS_AS unsigned long long *p0, *p1, *pn, *po;   // pattern: code bits (non-ACGT: bit 0 = the byte is 'n'), 'N' or 'n', any other byte
    const LDS_AS unsigned long long *t0, *t1, *tn;        // text: code bits ('n' : bit 0 set), not ACGT
    int p_org, t_org, st, p_words, t_words;
};

// bits i = 64 w .. 64 w + 63 of diagonal d's mismatch bitmap: set where i >= end, d + i < 0, or P(i) != T(d + i)  (lv.h: build_mask)
static __device__ __forceinline__ unsigned long long lv_plane_mask_word(const LvPlanes &pl, int d, int w, int end) {
    const int i0 = 64 * w;
    unsigned long long a0, a1, an, ao, b0, b1, bn;
    if (pl.st > 0) {
        const int ps = pl.p_org + i0, ts = pl.t_org + d + i0;
        a0 = plane_bits_fwd(pl.p0, ps, pl.p_words); a1 = plane_bits_fwd(pl.p1, ps, pl.p_words);
        an = plane_bits_fwd(pl.pn, ps, pl.p_words); ao = plane_bits_fwd(pl.po, ps, pl.p_words);
        b0 = plane_bits_fwd(pl.t0, ts, pl.t_words); b1 = plane_bits_fwd(pl.t1, ts, pl.t_words); bn = plane_bits_fwd(pl.tn, ts, pl.t_words);
    } else {
        const int ps = pl.p_org - i0, ts = pl.t_org - d - i0;
        a0 = plane_bits_bwd(pl.p0, ps, pl.p_words); a1 = plane_bits_bwd(pl.p1, ps, pl.p_words);
        an = plane_bits_bwd(pl.pn, ps, pl.p_words); ao = plane_bits_bwd(pl.po, ps, pl.p_words);
        b0 = plane_bits_bwd(pl.t0, ts, pl.t_words); b1 = plane_bits_bwd(pl.t1, ts, pl.t_words); bn = plane_bits_bwd(pl.tn, ts, pl.t_words);
    }
    // equal bytes: both ACGT with the same code, or 'N' against 'N' / 'n' against 'n'
    const unsigned long long eq = (~(an | ao | bn) & ~((a0 ^ b0) | (a1 ^ b1))) | (an & bn & ~(a0 ^ b0));
    unsigned long long mm = ~eq;
    // i >= end
    if (end <= i0) mm = ~0ull;
    else if (end < i0 + 64) mm |= ~0ull << (end - i0);
    // d + i < 0  <=>  i < -d
    if (-d > i0) mm |= (-d >= i0 + 64) ? ~0ull : ((1ull << (-d - i0)) - 1ull);
    return mm;
}
