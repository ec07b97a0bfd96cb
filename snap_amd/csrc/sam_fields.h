// sam_fields.h -- from an alignment result to the computed fields of its SAM record (FLAG, RNAME index, POS, MAPQ, CIGAR, NM),
// one wavefront per read.
//
// Restates, for the primary result of a single-end read:
//   SimpleReadWriter::writeReads   SNAPLib/ReadWriter.cpp:170-330  (the retry loop around a leading indel: move the alignment or
//                                  soft-clip the read and format again; give up across a contig boundary)
//   SAMFormat::writeRead           SNAPLib/SAM.cpp:1898-2112 (Landau-Vishkin cigar) and :2115-2352 (affine-gap cigar)
//   SAMFormat::createSAMLine       SNAPLib/SAM.cpp:1424-1572  (orientation, clipping bookkeeping, contig and position)
//   SAMFormat::computeCigarString  SNAPLib/SAM.cpp:2595-2674 / :2678-2766 (soft clips around the cigar)
//   Genome::getContigForRead       SNAPLib/Genome.cpp:734-758
// over cigar_lv.h / cigar_ag.h.  Text formatting (names, sequence, tags) stays with the caller.
#pragma once
#include "dev_common.h"
#include "cigar_lv.h"
#include "cigar_ag.h"
#include "../../include/snapgpu.h"

#define SAMF_UNMAPPED 0x4                    // SAM_UNMAPPED, SAM_REVERSE_COMPLEMENT (SAM.h)
#define SAMF_RC 0x10
#define SAMF_OP_S 4u                         // BAM code of 'S'

struct SamFieldsOut {
    int flag, contig, mapq, n_ops, nm, stale;
    long long pos;                           // 1-based position in the contig, 0 when unmapped
    // what SAMFormat::fillMateInfo needs from this read (paired-end writer)
    long long final_loc;                     // the location the record was written at (-1: unmapped)
    int final_dir, bases_clipped_before, ref_span, data_len;
};

static __device__ __forceinline__ int samf_contig_at(const DevIndex &ix, long long loc) {      // Genome::getContigAtLocation, Genome.cpp:574-594
    int lo = 0, hi = (int)ix.n_contigs - 1, c = -1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if ((long long)first_u64(ix.contig_begin[mid]) <= loc) { c = mid; lo = mid + 1; } else hi = mid - 1;
    }
    return c;
}
static __device__ __forceinline__ long long samf_contig_end(const DevIndex &ix, int c) {       // beginningLocation + length (length includes the padding)
    return c == (int)ix.n_contigs - 1 ? (long long)ix.n_bases : (long long)first_u64(ix.contig_begin[c + 1]);
}

// `oriented`: 2 * U bytes of per-wave HBM scratch (the read and its qualities as SAM prints them: reverse-complemented for an RC hit)
static __device__ __forceinline__ SamFieldsOut sam_fields_single_item(
    const DevIndex &ix, const AGCParams &agp, bool use_affine_gap, bool use_m,
    const uint8_t *bases, const uint8_t *quals, int U, int F0, int D0, const snapgpu_single_result &res,
    uint8_t *lds, uint32_t RL, uint8_t *oriented, uint32_t *lv_cells, uint8_t *ag_scratch, uint32_t *ops, int ops_cap, bool paired = false,
    const SamfPre *pre = nullptr)
{
    const int lane = lane_id();
    SamFieldsOut o; o.flag = 0; o.contig = -1; o.mapq = 0; o.n_ops = -1; o.nm = -1; o.stale = 0; o.pos = 0;
    o.final_loc = -1; o.final_dir = 0; o.bases_clipped_before = 0; o.ref_span = 0; o.data_len = 0;
    // the Read's clipping state (Read.h:508-553): front = F0 + addF, dataLength = D0 - addF - addB
    int addF = res.clipping_for_read_adjustment, addB = 0;                                       // ReadWriter.cpp:225
    int status = res.status, direction = res.direction;
    long long location = status == SNAPGPU_NotFound ? -1 : res.location;                        // :185-189 (InvalidGenomeLocation)
    long long final_loc = location;                                                             // :228
    const bool ag_branch = use_affine_gap && (res.used_affine_gap_scoring != 0 || res.score > 0);   // :232
    int cum = 0, n_adj = 0;
    int oriented_dir = -1;

    for (int attempt = 0; attempt < 2 * (int)RL + 8; attempt++) {
        const int front = F0 + addF, dlen = D0 - addF - addB;
        // ---------------- createSAMLine (SAM.cpp:1424-1572)
        long long loc = final_loc;
        if (status == SNAPGPU_NotFound) loc = -1;
        int dir = loc < 0 ? 0 : direction;
        int clipped_len = dlen;
        int bcb, bca;
        if (dir == 1) { bcb = U - clipped_len - front; bca = front; }
        else { bcb = front; bca = U - clipped_len - bcb; }
        if (ag_branch || paired) {                                                              // soft clipping from seed extension (:1541-1546; the
            bcb += res.bases_clipped_before; bca += res.bases_clipped_after;                    //  Landau-Vishkin writeRead passes none, ReadWriter.cpp:276)
            clipped_len -= res.bases_clipped_before + res.bases_clipped_after;
        }
        int flag = res.supplementary ? 0x800 : 0, contig = -1, mapq = 0;                         // SAM_SUPPLEMENTARY (:1481-1483); a caller writing a
        long long pos = 0, extra = 0;                                                           //  secondary result ORs 0x100 into the flag itself (:1477-1479)
        if (loc >= 0) {
            if (dir == 1) flag |= SAMF_RC;
            // getContigForRead(genomeLocation, read->getDataLength(), &extra)  (Genome.cpp:734-758)
            contig = samf_contig_at(ix, loc);
            if (contig < 0 || loc + dlen > samf_contig_end(ix, contig)) {
                contig = contig + 1;                                                            // getNextContigAfterLocation
                if (contig >= (int)ix.n_contigs) contig = (int)ix.n_contigs - 1;
                extra = (long long)first_u64(ix.contig_begin[contig]) - loc;
            }
            pos = loc + extra - (long long)first_u64(ix.contig_begin[contig]) + 1;
            mapq = res.mapq < 0 ? 0 : (res.mapq > 70 ? 70 : res.mapq);
        } else flag |= SAMF_UNMAPPED;
        int afc = 0, nm = -1, n_ops = -1;
        bool star = true;
        long long clip_before = 0, clip_after = 0;
        if (ag_branch && !paired && extra != 0) afc = (int)extra;                               // SAM.cpp:2193-2196 (writePairs passes extra on, :1654)
        else if (loc >= 0) {
            if (oriented_dir != dir) {                                                          // the read as SAM prints it (:1520-1538)
                for (int i = lane; i < U; i += WAVE) {
                    if (dir == 1) { oriented[U - 1 - i] = rc_base(bases[i]); oriented[U + U - 1 - i] = quals[i]; }
                    else { oriented[i] = bases[i]; oriented[U + i] = quals[i]; }
                }
                WAVE_SYNC();
                oriented_dir = dir;
            }
            const uint8_t *cd = oriented + bcb, *cq = oriented + U + bcb;
            long long extra_after = 0;
            int ed;
            int tail = 0;
            if (ag_branch) {
                // (the first attempt's row loop may have been run ahead of time: cigar_ag.h, SamfPre -- for this orientation and clipping only)
                const SamfPre *use_pre = (pre != nullptr && attempt == 0 && (int)first_u32((uint32_t)pre->dir) == dir && (int)first_u32((uint32_t)pre->bcb) == bcb) ? pre : nullptr;
                const CigarAGItemOut r = cigar_ag_item(ix, agp, cd, cq, clipped_len, res.score, extra, loc, use_m, lds, RL, ag_scratch, ops, ops_cap, use_pre);
                ed = r.edit_distance; afc = r.add_front_clipping; extra_after = r.extra_after; tail = r.tail_ins; n_ops = r.n_ops;
                if (r.stale) o.stale = 1;
            } else {
                const CigarItemOut r = cigar_lv_item(ix, cd, clipped_len, extra, loc, use_m, lds, lds + ((RL + 15) & ~15u), lv_cells, ops, ops_cap);
                ed = r.edit_distance; afc = r.add_front_clipping; extra_after = r.extra_after; n_ops = r.n_ops;
            }
            WAVE_SYNC();
            if (afc == 0 || (n_ops < 0)) {                                                      // computeCigarString (:2621-2674 / :2709-2766)
                afc = n_ops < 0 ? 0 : afc;
                nm = n_ops < 0 ? 0 : ed;                                                        // the "*" of computeCigar leaves editDistance 0 (:2403)
                if (n_ops >= 0 && ed >= 0) {
                    star = false;
                    if (ag_branch) bca += tail;                                                 // :2725
                    clip_before = bcb + extra; clip_after = bca + extra_after;
                }
            }
        }
        if (afc == 0) {                                                                         // the record is complete
            o.flag = flag; o.contig = loc >= 0 ? contig : -1; o.pos = pos; o.mapq = mapq; o.nm = nm;
            o.final_loc = loc; o.final_dir = dir; o.bases_clipped_before = bcb; o.data_len = dlen;
            if (loc < 0) { o.n_ops = -1; return o; }
            if (star) { o.n_ops = -1; return o; }
            // soft clips around the ops: shift right by one when there is a leading clip
            int n = n_ops;
            if (clip_before > 0) {
                if (n + 1 > ops_cap) { o.n_ops = -1; o.nm = -2; return o; }
                for (int i = n - 1; i >= 0; i--) { const uint32_t v = first_u32(ops[i]); WAVE_SYNC(); if (lane == 0) ops[i + 1] = v; WAVE_SYNC(); }
                if (lane == 0) ops[0] = ((uint32_t)clip_before << 4) | SAMF_OP_S;
                n++;
            }
            if (clip_after > 0) {
                if (n + 1 > ops_cap) { o.n_ops = -1; o.nm = -2; return o; }
                if (lane == 0) ops[n] = ((uint32_t)clip_after << 4) | SAMF_OP_S;
                n++;
            }
            WAVE_SYNC();
            o.n_ops = n;
            {   // getRefSpanFromCigar (SAM.cpp:2769-2800): the first op counts unless it is S / H, every later op unless it is I
                int span = 0;
                for (int i = 0; i < n; i++) {
                    const uint32_t v = first_u32(ops[i]);
                    const uint32_t code = v & 15u;
                    if (i == 0 ? (code != SAMF_OP_S && code != 5u) : (code != LVC_OP_I)) span += (int)(v >> 4);
                }
                o.ref_span = span;
            }
            return o;
        }
        // ---------------- the caller's reaction to a leading indel (ReadWriter.cpp:240-274 / :282-311)
        n_adj++;
        if (paired) {                                                                           // SAMFormat::writePairs, SAM.cpp:1660-1684 / :1694-1713
            const int co = samf_contig_at(ix, final_loc), cn = samf_contig_at(ix, final_loc + afc);
            if (cn != co || cn < 0 || final_loc + afc > samf_contig_end(ix, co) - (long long)ix.chromosome_padding || n_adj > 2 * (int)RL) {
                status = SNAPGPU_NotFound; location = -1; direction = 0; final_loc = -1; continue;
            }
            if (ag_branch) {
                if (afc < 0) { cum += afc; if (direction == 0) addF = -cum; else addB = -cum; }
                else final_loc += afc;
            } else {
                if (afc > 0) { cum += afc; addF = cum; }
                final_loc += afc;
            }
            continue;
        }
        const int c_orig = status == SNAPGPU_NotFound ? -1 : samf_contig_at(ix, location);
        const int c_new = status == SNAPGPU_NotFound ? -1 : samf_contig_at(ix, location + afc);
        const int c_lim = ag_branch ? c_new : c_orig;
        bool give_up = c_new < 0 || c_new != c_orig || n_adj > dlen;
        if (!give_up) give_up = final_loc + afc > samf_contig_end(ix, c_lim) - (long long)ix.chromosome_padding;
        if (give_up) { status = SNAPGPU_NotFound; location = -1; direction = 0; final_loc = -1; continue; }
        if (ag_branch) {
            if (afc < 0) { cum += afc; if (direction == 0) addF = -cum; else addB = -cum; }      // insertion: soft-clip (:263-269)
            else final_loc = location + afc;                                                    // deletion (:271)
        } else {
            if (afc > 0) { cum += afc; addF = cum; }                                            // :305-308
            final_loc += afc;                                                                   // :309
        }
    }
    o.flag = SAMF_UNMAPPED; o.n_ops = -1; o.nm = -1;
    return o;
}

// SAMFormat::fillMateInfo (SAM.cpp:1308-1421) for one read of a pair, from the two reads' own records.
// rnext: -1 "*", -2 "=", otherwise a contig index.
struct SamMateOut { int flag, contig, rnext; long long pos, pnext, tlen; };

static __device__ __forceinline__ SamMateOut sam_fill_mate_info(const DevIndex &ix, const SamFieldsOut &me, const SamFieldsOut &mate, bool first_in_pair, bool aligned_as_pair)
{
    SamMateOut m; m.flag = me.flag | 0x1 | (first_in_pair ? 0x40 : 0x80); m.contig = me.contig; m.pos = me.pos; m.rnext = -1; m.pnext = 0; m.tlen = 0;
    auto contig_for_read = [&](long long loc, int data_len, long long *extra) -> int {          // Genome::getContigForRead, Genome.cpp:734-758
        int c = samf_contig_at(ix, loc);
        *extra = 0;
        if (c < 0 || loc + data_len > samf_contig_end(ix, c)) {
            c = c + 1; if (c >= (int)ix.n_contigs) c = (int)ix.n_contigs - 1;
            *extra = (long long)first_u64(ix.contig_begin[c]) - loc;
        }
        return c;
    };
    long long mate_loc = mate.final_loc, mate_extra = 0;
    bool rnext_eq = false;
    if (mate_loc >= 0) {
        const int mc = contig_for_read(mate_loc, mate.data_len, &mate_extra);
        mate_loc += mate_extra;
        m.rnext = mc; m.pnext = mate_loc - (long long)first_u64(ix.contig_begin[mc]) + 1;
        if (mate.final_dir == 1) m.flag |= 0x20;
        if (me.final_loc < 0) { m.contig = mc; rnext_eq = true; m.pos = m.pnext; }                // :1347-1356
    } else {
        m.flag |= 0x8;
        rnext_eq = true; m.pnext = m.pos;                                                       // :1358-1365
    }
    if (me.final_loc >= 0 && mate.final_loc >= 0) {
        if (aligned_as_pair) m.flag |= 0x2;
        long long extra = 0;
        const int c = contig_for_read(me.final_loc, me.data_len, &extra);
        const long long loc = me.final_loc + extra;
        const long long my_start = loc - me.bases_clipped_before - extra, my_end = loc + me.ref_span;
        const long long mate_start = mate_loc - mate.bases_clipped_before - mate_extra, mate_end = mate_loc + mate.ref_span;
        m.contig = c;
        if (my_start < mate_start) {
            if (me.final_dir == 0) m.tlen = mate.final_dir == 1 ? mate_end - my_start : mate_start - my_start;
            else m.tlen = mate.final_dir == 0 ? mate_start - my_end : mate_end - my_end;
        } else {
            if (me.final_dir == 1) m.tlen = mate.final_dir == 0 ? -(my_end - mate_start) : -(my_end - mate_end);
            else m.tlen = mate.final_dir == 0 ? -(my_start - mate_start) : -(my_start - mate_end);
        }
    }
    if (rnext_eq || (m.rnext >= 0 && m.rnext == m.contig)) m.rnext = -2;                          // :1418-1420 (pointer equality of the names)
    return m;
}
