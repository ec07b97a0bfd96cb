// ag_resolve.h -- the exact answer of ONE affine-gap call whose banded traceback leaves its band, without an image of the object's
// traceback array kept across the calls.
//
// What such a call reads outside its band is a byte of AffineGapVectorized's backtraceAction array (AffineGapVectorized.h:1374, read at
// :740-788): the action byte that the LATEST EARLIER call of the same object wrote at that address, or 0 if none has written there since
// the object was constructed.  What a call writes follows from its own arguments alone (rows 0 .. rows evaluated, the vectors of its
// band's segments, at row * numVec * numSeg * 8 + vector * 8 + element), and the forward pass that produces the bytes does not look at
// the array.  So, given the list of the object's earlier calls:
//   1. fill the image with "unknown" where this call can look, run the call in the exact form (its forward pass overwrites what it
//      evaluates) with a traceback that stops at the first unknown cell it reads outside the band and says which one;
//   2. walk the earlier calls from the latest back; for each whose band geometry covers the address, run its forward pass into a second
//      image and look whether it did write the cell (rows after an early end of the row loop are not written);
//   3. put the byte (or 0) into the first image and run the call again -- until its traceback gets through.
// Most calls need 1-12 cells, each costs about two forward passes; a traceback that has left its band can also stay out for as long as the
// earlier calls' bytes lead it (seen: hundreds of cells), hence the generous limit.  Against that the image-keeping scheme writes every
// call's cells to HBM (~57 KB per read, needed by 5 reads in a million) and replays whole reads / pairs in a pass of their own.
// The forward passes here go through the LDS form (ag.h) whatever form the kernel's hot path uses: in the exact layout all forms write
// the same bytes at the same addresses.
//
// Status: used by snapgpu_affine_gap_sequence's resolving mode (test entry; tests/test_gpu_parity.py) -- the aligners still keep images
// and replay ("Next").
#pragma once
#include "ag.h"

#define AG_RESOLVE_MAX_STEPS 4096       // a traceback that has left its band can stay out for as long as earlier calls' bytes lead it (seen: > 64 steps)

// does the forward pass of a call with these arguments write the cell at `at` (if its row loop gets that far)?
static __device__ __forceinline__ bool ag_call_covers(bool banded, int pattern_len, int text_len, int w, uint32_t at, uint32_t image_bytes) {
    if (w > 126) w = 126;
    if (w < 0 || pattern_len <= 0 || text_len <= 0) return false;
    int num_vec, seg_len, num_seg;
    if (banded) {
        const int bw = (2 * w + 1) < pattern_len ? (2 * w + 1) : pattern_len;
        num_vec = (bw + 7) >> 3; seg_len = num_vec * 8; num_seg = (pattern_len + seg_len - 1) / seg_len;
    } else { num_vec = (pattern_len + 7) >> 3; seg_len = num_vec * 8; num_seg = 1; }
    const uint32_t row_cells = (uint32_t)(num_vec * num_seg) * 8u;
    const uint32_t row = at / row_cells, cell = at % row_cells;
    if (row >= (uint32_t)text_len || (uint64_t)(row + 1) * row_cells > image_bytes) return false;
    const int vi = (int)(cell >> 3), j = vi / num_vec, k = vi % num_vec;
    if (!banded) return true;
    const int i = (int)row;
    const int band_beg = i - w > 0 ? i - w : 0, band_end = i + w < pattern_len - 1 ? i + w : pattern_len - 1;
    if (j < band_beg / seg_len || j > band_end / seg_len) return false;
    int nk = num_vec; const int lim = band_end - j * seg_len + 1; if (lim < nk) nk = lim;
    return k < nk;
}

// One call as its caller holds it: first element and step of the pattern (and its qualities) and of the text.
struct AGProblem { const uint8_t *p, *q, *t; int pst, tst; int plen, tlen, w, score_init; bool banded, is_rc; int use_clip; };

// `prob(c)` returns call c of the object (0 <= c < n_earlier: the calls made before this one, oldest first).
// image / other: two buffers of image_bytes each; rows: the LDS form's H / H-1 / E rows (ag_lds_bytes(RL) of the LDS form).
// Returns false when the call did not get through within the step limit (the caller then falls back to flagging it).
template <typename ProbFn>
static __device__ __forceinline__ bool ag_resolve_call(int dir, const AGParams &prm, const AGProblem &x, int n_earlier, ProbFn prob,
                                                       int16_t *rows, uint8_t *image, uint8_t *other, uint32_t image_bytes, uint32_t RL,
                                                       const DevTables *tab, AGResult *out, uint32_t *steps_out)
{
    const int lane = lane_id();
    auto run = [&](const AGProblem &c, uint8_t *img, uint32_t *pending) -> AGResult {
        ByteSeq P{c.p, c.pst}, Q{c.q, c.pst}, T{c.t, c.tst};
        return pending ? ag_compute<true, true>(c.banded, dir, prm, P, Q, c.plen, T, c.tlen, c.w, c.score_init, c.is_rc, c.use_clip, rows, img, RL, tab, pending)
                       : ag_compute<true, false>(c.banded, dir, prm, P, Q, c.plen, T, c.tlen, c.w, c.score_init, c.is_rc, c.use_clip, rows, img, RL, tab);
    };
    // 1. everything this call can address is unknown
    {
        int nv, sl, ns;
        int ww = x.w > 126 ? 126 : (x.w < 0 ? 0 : x.w);
        if (x.banded) { const int bw = (2 * ww + 1) < x.plen ? (2 * ww + 1) : x.plen; nv = (bw + 7) >> 3; sl = nv * 8; ns = (x.plen + sl - 1) / sl; }
        else { nv = (x.plen + 7) >> 3; sl = nv * 8; ns = 1; }
        uint64_t ext = (uint64_t)(x.tlen > 0 ? x.tlen : 0) * (uint64_t)(nv * ns * 8);
        if (ext > image_bytes) ext = image_bytes;
        for (uint64_t b = (uint64_t)lane * 4; b < ext; b += (uint64_t)WAVE * 4) *(uint32_t *)(image + b) = 0xFFFFFFFFu;
        WAVE_SYNC(); __threadfence_block();
    }
    uint32_t steps = 0;
    for (int iter = 0; iter < AG_RESOLVE_MAX_STEPS; iter++) {
        uint32_t pending = AG_NO_CELL;
        const AGResult r = run(x, image, &pending);
        WAVE_SYNC(); __threadfence_block();
        pending = first_u32(pending);
        if (pending == AG_NO_CELL) { *out = r; *steps_out = steps; return true; }
        steps++;
        // 2. the latest earlier call that wrote the cell
        uint32_t value = 0;
        for (int c = n_earlier - 1; c >= 0; c--) {
            const AGProblem e = prob(c);
            if (!ag_call_covers(e.banded, e.plen, e.tlen, e.w, pending, image_bytes)) continue;
            if (lane == 0) other[pending] = (uint8_t)AG_CELL_UNKNOWN;
            WAVE_SYNC(); __threadfence_block();
            (void)run(e, other, nullptr);
            WAVE_SYNC(); __threadfence_block();
            const uint32_t b = first_u32((uint32_t)other[pending]);
            if (b != (uint32_t)AG_CELL_UNKNOWN) { value = b; break; }
        }
        // 3.
        if (lane == 0) image[pending] = (uint8_t)value;
        WAVE_SYNC(); __threadfence_block();
    }
    *steps_out = steps;
    return false;
}

// ---- the aligners' form: the object's calls of the read as 16-byte records, the read and the genome by pointer
struct AgCall { int64_t loc; int16_t org, plen, tlen; uint16_t lim_flags; };       // lim_flags: limit | direction << 8 | use_clip << 9 | banded << 10
struct AgCallCtx {                             // what turns a record into a problem again (by value: see adjust.h on why not a reference)
    const uint8_t *rd0, *rd1, *ql0, *ql1;      // the read and its reverse complement, qualities in the same orientations
    const uint8_t *genome;                     // base 0
    int read_len, st;                          // score_init of every call; +1: affineGap (forward half), -1: reverseAffineGap
};
static __device__ __forceinline__ AGProblem ag_call_problem(const AgCallCtx &c, int64_t loc, int org, int plen, int tlen, int lim, int dir, int use_clip, bool banded) {
    AGProblem x;
    x.p = (dir ? c.rd1 : c.rd0) + org; x.q = (dir ? c.ql1 : c.ql0) + org; x.t = c.genome + loc + org; x.pst = c.st; x.tst = c.st;
    x.plen = plen; x.tlen = tlen; x.w = lim; x.score_init = c.read_len; x.banded = banded; x.is_rc = dir != 0; x.use_clip = use_clip;
    return x;
}
// (not inlined: a rare path that must not take part in the register allocation of the kernel's hot loops)
static __device__ __attribute__((noinline)) bool ag_resolve_from_list(const AgCallCtx c, const AGParams prm, const AGProblem x, const AgCall *list, int n_earlier,
                                                                      int16_t *rows, uint8_t *image, uint8_t *other, uint32_t image_bytes, uint32_t RL,
                                                                      const DevTables *tab, AGResult *out)
{
    uint32_t steps = 0;
    return ag_resolve_call(c.st, prm, x, n_earlier, [&](int k) {
                               const int64_t l = (int64_t)first_u64((uint64_t)list[k].loc);
                               const uint32_t w0 = first_u32(((const uint32_t *)&list[k])[2]), w1 = first_u32(((const uint32_t *)&list[k])[3]);
                               const uint32_t lf = w1 >> 16;
                               return ag_call_problem(c, l, (int)(int16_t)(w0 & 0xffffu), (int)(int16_t)(w0 >> 16), (int)(int16_t)(w1 & 0xffffu), (int)(lf & 0xffu),
                                                      (int)((lf >> 8) & 1u), (int)((lf >> 9) & 1u), ((lf >> 10) & 1u) != 0);
                           }, rows, image, other, image_bytes, RL, tab, out, &steps);
}

