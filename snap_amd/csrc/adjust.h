// adjust.h -- AlignmentAdjuster::AdjustAlignment (SNAPLib/AlignmentAdjuster.cpp:33-190): the `-ae` step that finalizeSecondaryResults runs on the
// primary and on every secondary result BEFORE it filters them (BaseAligner.cpp:2444-2463, !ignoreAlignmentAdjustmentsForOm), one wavefront
// per result.
//
// What the reference does to a result: recompute its score as the edit distance of LandauVishkinWithCigar (computeEditDistanceNormalized,
// LandauVishkin.cpp:507-648, k = MAX_K - 1) between the read as it lies on the genome and the reference, and while that alignment starts with
// a deletion or an insertion move the location (and, for a deletion, clip the read's front) and try again; then settle how many bases hang
// over the end of the contig.  The Read here is the one AlignRead was given: no clipping of its own (the C ABI hands over the bases to align),
// so its state is one number, the additional front clipping the adjuster has set so far.  Restated literally, including what looks
// unintended in the reference:
//   * an RC result clips the END of the reverse-complemented read when a front clip is added (the Read is clipped at its own front,
//     AlignmentAdjuster.cpp:74-79), while the location still moves forward;
//   * the loop that settles the overhang passes `data` -- the start of the UNCLIPPED buffer -- where the first loop passed clippedData (:167);
//   * additionalFrontClipping is not reset between attempts: an attempt whose edit distance is beyond k returns before it is assigned
//     (LandauVishkin.cpp:525-527) and the loop goes on with the previous value.
#pragma once
#include "cigar_lv.h"
#include "../../include/snapgpu.h"

struct AdjustScratch {                         // per wave, HBM (snapgpu_enable_secondary with adjust_alignments)
    uint8_t  *pat;                             // RL bytes: the read as it lies on the genome
    uint8_t  *txt;                             // RL + LVC_MAX_K bytes: the reference window
    uint32_t *cells;                           // lvc_scratch_bytes()
    uint32_t *ops; int ops_cap;
};
static __host__ __device__ __forceinline__ uint32_t adjust_ops_cap() { return 1024u; }
static __host__ __device__ __forceinline__ size_t adjust_scratch_bytes(uint32_t RL) {
    return (size_t)((RL + 255) & ~255u) + (size_t)((RL + LVC_MAX_K + 255) & ~255u) + (size_t)((lvc_scratch_bytes() + 255u) & ~255u) + (size_t)adjust_ops_cap() * 4u;
}
static __device__ __forceinline__ AdjustScratch adjust_scratch_at(uint8_t *base, uint32_t RL) {
    AdjustScratch s;
    s.pat = base; base += (RL + 255) & ~255u;
    s.txt = base; base += (RL + LVC_MAX_K + 255) & ~255u;
    s.cells = (uint32_t *)base; base += (lvc_scratch_bytes() + 255u) & ~255u;
    s.ops = (uint32_t *)base; s.ops_cap = (int)adjust_ops_cap();
    return s;
}

struct AdjustOut { int status; long long location; int score; int clipping; };
// what the adjuster needs of the index, by value (a reference to the kernel's DevIndex handed to a function that is not inlined would pin
// the caller's whole aligner object in scratch memory)
struct AdjustIx { const uint8_t *genome; const uint64_t *contig_begin; uint64_t n_bases; uint32_t n_contigs, chromosome_padding, genome_pad; };
static __device__ __forceinline__ AdjustIx adjust_ix(const DevIndex &ix) {
    AdjustIx a; a.genome = ix.genome; a.contig_begin = ix.contig_begin; a.n_bases = ix.n_bases; a.n_contigs = ix.n_contigs;
    a.chromosome_padding = ix.chromosome_padding; a.genome_pad = ix.genome_pad;
    return a;
}

static __device__ __forceinline__ int adjust_contig_at(const AdjustIx &ix, long long loc) {        // Genome::getContigAtLocation (Genome.cpp:574-594); -1 = NULL
    int lo = 0, hi = (int)ix.n_contigs - 1, c = -1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if ((long long)first_u64(ix.contig_begin[mid]) <= loc) { c = mid; lo = mid + 1; } else hi = mid - 1;
    }
    return c;
}
static __device__ __forceinline__ long long adjust_contig_end(const AdjustIx &ix, int c) {         // beginningLocation + length (the length includes the padding)
    return c == (int)ix.n_contigs - 1 ? (long long)ix.n_bases : (long long)first_u64(ix.contig_begin[c + 1]);
}

// fwd / rc: the read and its reverse complement (U bytes each, LDS or HBM).  in: status, direction, location, score of the result.
// (noinline: only the secondary-result kernels call it, once per result, and it should not take part in their register allocation)
static __device__ __noinline__ AdjustOut adjust_alignment(const AdjustIx ix, const uint8_t *fwd, const uint8_t *rc, int U, int status, int direction,
                                                          long long location, int score, long long invalid_location, AdjustScratch sc)
{
    const int lane = lane_id();
    AdjustOut o; o.status = status; o.location = location; o.score = score; o.clipping = 0;
    if (status == SNAPGPU_NotFound) return o;                                                       // :38-43
    const long long nb = (long long)ix.n_bases, pad = (long long)ix.chromosome_padding;
    int front = 0, cum = 0, afc = 0, net_indel = 0;
    unsigned n_adj = 0;
    long long data_len = 0, extra_after = 0, ref_loc = 0, real_end = 0;
    const uint8_t *data = direction == 1 ? rc : fwd;                                                // dataBuffer (:69-83)
    // one computeEditDistanceNormalized(reference, plen + MAX_K, pattern, plen, MAX_K - 1, ..., &afc, &net_indel) -> the score it returns
    auto normalized = [&](const uint8_t *pattern, int plen) -> int {
        const long long readable = nb + (long long)ix.genome_pad - ref_loc;
        for (int i = lane; i < plen; i += WAVE) sc.pat[i] = pattern[i];
        for (int i = lane; i < plen + LVC_MAX_K; i += WAVE) sc.txt[i] = i < readable ? ix.genome[ref_loc + i] : (uint8_t)0;
        WAVE_SYNC(); __threadfence_block();
        const LVCResult r = lvc_compute(sc.pat, plen, sc.txt, plen + LVC_MAX_K, LVC_MAX_K - 1, true, sc.cells, sc.ops, sc.ops_cap);
        WAVE_SYNC(); __threadfence_block();
        net_indel = r.net_indel;
        if (r.score < 0) return r.score;                                                            // LandauVishkin.cpp:525-527 (afc keeps its value)
        if (r.n_ops > 0) {                                                                          // :607-622
            const uint32_t op0 = first_u32(sc.ops[0]);
            if ((op0 & 0xfu) == LVC_OP_D) { afc = (int)(op0 >> 4); if (afc != 0) return 0; }
            else if ((op0 & 0xfu) == LVC_OP_I) afc = -(int)(op0 >> 4);
            else afc = 0;
        } else afc = 0;
        return r.score;
    };
    for (;;) {
        data_len = U - front;                                                                       // read->getDataLength()
        const uint8_t *clipped = direction == 1 ? data /* &data[fullLength - dataLength - frontClipped] */ : fwd + front;
        // getContigForRead(result->location, read->getDataLength(), &extraBasesClippedBefore)  (Genome.cpp:734-758)
        long long extra_before = 0;
        int contig = adjust_contig_at(ix, o.location);
        if (contig < 0 || o.location + data_len > adjust_contig_end(ix, contig)) {
            contig = contig + 1;                                                                    // getNextContigAfterLocation
            if (contig >= (int)ix.n_contigs) contig = (int)ix.n_contigs - 1;
            extra_before = (long long)first_u64(ix.contig_begin[contig]) - o.location;
        }
        ref_loc = o.location + extra_before;                                                        // :94-97
        clipped += extra_before; data_len -= extra_before;
        real_end = adjust_contig_end(ix, contig) - pad;
        extra_after = ref_loc + data_len > real_end ? ref_loc + data_len - real_end : 0;            // :100-108
        {   // getSubstring(genomeLocation, dataLength - extraBasesClippedAfter) == NULL (:110-116; Genome.h:339-367)
            const long long need = data_len - extra_after;
            bool ok;
            if (ref_loc < 0 || ref_loc > nb || ref_loc + need > nb + 1000) ok = false;
            else if (need <= pad && first_u32(ix.genome[ref_loc]) != 'n') ok = true;
            else if (need == 0) ok = true;
            else { const int c2 = adjust_contig_at(ix, ref_loc); ok = c2 >= 0 && adjust_contig_end(ix, c2) > ref_loc + need; }
            if (!ok || need < 0) { o.status = SNAPGPU_NotFound; o.location = invalid_location; return o; }
        }
        o.score = normalized(clipped, (int)(data_len - extra_after));                               // :119-131
        if (afc == 0) break;
        n_adj++;
        const int co = adjust_contig_at(ix, o.location), cn = adjust_contig_at(ix, o.location + afc);                    // :139-151
        if (cn < 0 || cn != co || o.location + afc > adjust_contig_end(ix, co) - pad || n_adj > (unsigned)(U - front)) {
            o.status = SNAPGPU_NotFound; o.location = invalid_location; return o;
        }
        cum += afc;
        front = cum > 0 ? cum : 0;                                                                  // read->setAdditionalFrontClipping(__max(0, cumulative))
        o.clipping = front;
        o.location += afc;
    }
    // the overhang at the end of the contig (:165-187)
    long long nw = o.location + data_len + net_indel - real_end; if (nw < 0) nw = 0;
    for (long long pass = 0; pass < data_len; pass++) {
        if (nw == extra_after) return o;
        extra_after = nw;
        o.score = normalized(data, (int)(data_len - extra_after));                                  // (`data`, not clippedData: see the header)
        nw = o.location + data_len + net_indel - real_end; if (nw < 0) nw = 0;
    }
    return o;
}
