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
// The prepared form: what a call has to do once so that a diagonal's bitmap costs a few operations per word.
//   pw[plane][m]  the pattern, oriented: bit i of word m = plane bit of P(64 m + i)
//   sw[plane][m]  the text, oriented and shifted by K = the call's limit: bit j of the string = plane bit of T(j - K), so that diagonal d's
//                 window for positions 64 w .. 64 w + 63 is the 64 bits starting at bit K + d + 64 w -- the same shift for every w of a lane
// n_pw = ceil(pattern_len / 64), n_sw = ceil((pattern_len + 2 K) / 64) + 1 words per plane; `work` holds 4 n_pw + 3 n_sw words.
static __host__ __device__ __forceinline__ uint32_t lv_plane_work_words(uint32_t RL) { const uint32_t n_pw = (RL + 63) / 64, n_sw = (RL + 62 + 63) / 64 + 2; return 4 * n_pw + 3 * n_sw; }

struct LvPlanes {
    const LDS_AS unsigned long long *p0, *p1, *pn, *po;   // pattern: code bits (non-ACGT: bit 0 = the byte is 'n'), 'N' or 'n', any other byte
    const LDS_AS unsigned long long *t0, *t1, *tn;        // text: code bits ('n' : bit 0 set), not ACGT
    int p_org, t_org, st, p_words, t_words;
    LDS_AS unsigned long long *work;                      // lv_plane_work_words(RL) words of LDS for the prepared form
    bool plain;                                           // the pattern is all ACGT: its 'N' / other planes are empty
};

// once per call (every lane takes words of the job): the oriented pattern and the oriented, shifted text
static __device__ __forceinline__ void lv_planes_prepare(const LvPlanes &pl, int K, int pattern_len, int n_pw, int n_sw) {
    const int lane = lane_id();
    LDS_AS unsigned long long *pw = pl.work, *sw = pl.work + 4 * n_pw;
    const int n_p = (pl.plain ? 2 : 4) * n_pw, n_jobs = n_p + 3 * n_sw;
    for (int j = lane; j < n_jobs; j += WAVE) {
        if (j < n_p) {
            const int plane = j / n_pw, m = j - plane * n_pw;
            const LDS_AS unsigned long long *x = plane == 0 ? pl.p0 : plane == 1 ? pl.p1 : plane == 2 ? pl.pn : pl.po;
            pw[plane * n_pw + m] = pl.st > 0 ? plane_bits_fwd(x, pl.p_org + 64 * m, pl.p_words) : plane_bits_bwd(x, pl.p_org - 64 * m, pl.p_words);
        } else {
            const int jj = j - n_p, plane = jj / n_sw, m = jj - plane * n_sw;
            const LDS_AS unsigned long long *x = plane == 0 ? pl.t0 : plane == 1 ? pl.t1 : pl.tn;
            sw[plane * n_sw + m] = pl.st > 0 ? plane_bits_fwd(x, pl.t_org - K + 64 * m, pl.t_words) : plane_bits_bwd(x, pl.t_org + K - 64 * m, pl.t_words);
        }
    }
    (void)pattern_len;
}
// diagonal d's bitmap word w from the prepared form (same value as lv_plane_mask_word)
static __device__ __forceinline__ unsigned long long lv_planes_word(const LvPlanes &pl, int K, int d, int w, int end, int n_pw, int n_sw) {
    const LDS_AS unsigned long long *pw = pl.work, *sw = pl.work + 4 * n_pw;
    const int sh = K + d, base = (sh >> 6) + w, s = sh & 63;
    unsigned long long b[3];
#pragma unroll
    for (int plane = 0; plane < 3; plane++) {
        const unsigned long long lo = sw[plane * n_sw + base], hi = sw[plane * n_sw + base + 1];
        b[plane] = (lo >> s) | ((hi << 1) << (63 - s));
    }
    const unsigned long long a0 = pw[w], a1 = pw[n_pw + w];
    unsigned long long eq = ~b[2] & ~((a0 ^ b[0]) | (a1 ^ b[1]));
    if (!pl.plain) {
        const unsigned long long an = pw[2 * n_pw + w], ao = pw[3 * n_pw + w];
        eq = (eq & ~(an | ao)) | (an & b[2] & ~(a0 ^ b[0]));
    }
    unsigned long long mm = ~eq;
    const int i0 = 64 * w;
    if (end <= i0) mm = ~0ull;
    else if (end < i0 + 64) mm |= ~0ull << (end - i0);
    if (-d > i0) mm |= (-d >= i0 + 64) ? ~0ull : ((1ull << (-d - i0)) - 1ull);
    return mm;
}

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
