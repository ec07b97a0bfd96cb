/*
 * snapgpu.h -- C ABI of the MI355X-native seed-and-extend hot path.
 *
 * This is the drop-in boundary: everything SNAP's per-thread aligner objects do
 * between "here is a Read" and "here is a SingleAlignmentResult" is behind these
 * entry points.  Plain pointers and sizes only; no C++ or torch types cross the line.
 *
 * Reference interfaces each entry point replaces (paths relative to the reference tree):
 *
 *   snapgpu_create / snapgpu_destroy
 *       GenomeIndex::loadFromDirectory          SNAPLib/GenomeIndex.cpp:1839-2093 (index blobs -> memory)
 *       BaseAligner::BaseAligner                SNAPLib/BaseAligner.cpp:50-265   (per-thread scratch + options)
 *   snapgpu_lookup_seeds
 *       GenomeIndex::lookupSeed32               SNAPLib/GenomeIndex.cpp:2096-2157
 *       GenomeIndex::fillInLookedUpResults32    SNAPLib/GenomeIndex.cpp:2160-2202
 *       SNAPHashTable::GetFirstValueForKey      SNAPLib/HashTable.h:87-118
 *       Seed::Seed                              SNAPLib/Seed.h:40-53
 *   snapgpu_landau_vishkin
 *       LandauVishkin<1|-1>::computeEditDistance SNAPLib/LandauVishkin.h:100-351
 *   snapgpu_compute_cigar_lv
 *       SAMFormat::computeCigar (LV variant)    SNAPLib/SAM.cpp:2354-2467
 *       LandauVishkinWithCigar::computeEditDistanceNormalized / computeEditDistance  SNAPLib/LandauVishkin.cpp:507-648 / 141-505
 *   snapgpu_adjust_alignments
 *       AlignmentAdjuster::AdjustAlignment      SNAPLib/AlignmentAdjuster.cpp:33-190 (the `-ae` step, BaseAligner.cpp:2444-2463)
 *   snapgpu_sam_fields_paired
 *       SAMFormat::writePairs / fillMateInfo    SNAPLib/SAM.cpp:1575-1895 / 1308-1421; SimpleReadWriter::writePairs SNAPLib/ReadWriter.cpp:345-520
 *   snapgpu_sam_fields_single
 *       SimpleReadWriter::writeReads            SNAPLib/ReadWriter.cpp:170-330
 *       SAMFormat::writeRead / createSAMLine / computeCigarString  SNAPLib/SAM.cpp:1898-2352 / 1424-1572 / 2595-2766
 *   snapgpu_compute_cigar_ag
 *       SAMFormat::computeCigar (affine-gap variant)  SNAPLib/SAM.cpp:2470-2588
 *       AffineGapVectorizedWithCigar::computeGlobalScoreNormalized / Banded / computeGlobalScore  SNAPLib/AffineGapVectorized.cpp:1043 / 520 / 159
 *   snapgpu_affine_gap
 *       AffineGapVectorized<1|-1>::computeScore / computeScoreBanded
 *                                               SNAPLib/AffineGapVectorized.h:821-1339 / 256-819
 *   snapgpu_align_single
 *       BaseAligner::AlignRead                  SNAPLib/BaseAligner.cpp:273-763 (decl BaseAligner.h:76-90)
 *       called from SingleAlignerContext::runIterationThreadImpl SNAPLib/SingleAligner.cpp:250
 *
 * Error convention: every function returns 0 on success or a negative SNAPGPU_E_* code;
 * snapgpu_last_error() returns a human readable message for the last failure on that
 * context (or the last global failure when ctx is NULL).  Nothing here ever falls back to
 * a CPU implementation: if no gfx950 device is usable, snapgpu_create fails.
 */
#ifndef SNAPGPU_H
#define SNAPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNAPGPU_ABI_VERSION 4

/* error codes */
#define SNAPGPU_OK              0
#define SNAPGPU_E_INVALID      -1   /* bad argument                                      */
#define SNAPGPU_E_NODEVICE     -2   /* no usable HIP device / HIP runtime failure        */
#define SNAPGPU_E_UNSUPPORTED  -3   /* index/options outside what this build implements  */
#define SNAPGPU_E_NOMEM        -4
#define SNAPGPU_E_LAUNCH       -5   /* kernel launch or execution failed                 */
#define SNAPGPU_W_SECONDARY_TRUNCATED 1 /* some read has more secondary results than the caller's stride: see snapgpu_align_single_secondary */

/* mirrors enum AlignmentResult, SNAPLib/AlignmentResult.h:33 */
enum { SNAPGPU_NotFound = 0, SNAPGPU_SingleHit = 1, SNAPGPU_MultipleHits = 2 };

/* sentinels, SNAPLib/LandauVishkin.h:14-15, BaseAligner.h:141, GenomeIndex.cpp:511-517 */
#define SNAPGPU_ScoreAboveLimit     (-1)
#define SNAPGPU_TooBigScoreValue    65536
#define SNAPGPU_UnusedScoreValue    0xffff
#define SNAPGPU_InvalidGenomeLocation32 0xffffffffLL
#define SNAPGPU_MAX_K               127          /* LandauVishkin.h:11 */

/*
 * Read-only view of a loaded SNAP index (the four files of a SNAP index directory,
 * SURVEY.md Appendix B).  Pointers may be host or device pointers; `on_device` says which.
 * The bytes are exactly what the reference's index builder wrote, so probe sequences and
 * results are identical to GenomeIndex::lookupSeed32.
 */
typedef struct snapgpu_index_view {
    /* GenomeIndex text header (GenomeIndex.cpp:1879) */
    uint32_t seed_len;
    uint32_t key_bytes;            /* hashTableKeySize                                   */
    uint32_t n_hash_tables;
    uint32_t large_hash_table;     /* 1 = one entry holds fwd+rc values (valueCount 2)   */
    uint32_t location_size;        /* 4: the blobs of a view hold 32-bit values.  (snapgpu_create_from_directory also takes index
                                      directories written with 5 .. 8-byte locations -- what the indexer picks for seeds shorter than
                                      20, GenomeIndex.cpp:446-453 -- and narrows their tables on load where every value fits 32 bits.) */
    uint32_t chromosome_padding;
    uint64_t overflow_table_size;  /* in 32-bit words                                    */

    /* concatenated hash tables: for table t, slots are at
     * hash_blob + table_offset[t], table_size[t] slots of entry_bytes each
     * (entry = value_count x 4-byte value, then key_bytes of key; HashTable.h:148-156)  */
    const uint8_t  *hash_blob;
    uint64_t        hash_blob_bytes;
    const uint64_t *table_offset;   /* [n_hash_tables] byte offsets into hash_blob (always host) */
    const uint64_t *table_size;     /* [n_hash_tables] slot counts (always host)                  */

    const uint32_t *overflow;       /* [overflow_table_size]                              */

    /* genome: n_bases bytes, upper-case ACGTN with 'n' padding (Genome.h:450).  The
     * pointer addresses base 0; genome_pad bytes of 'n' are readable before and after.  */
    const uint8_t  *genome;
    uint64_t        n_bases;
    uint32_t        genome_pad;     /* >= 1000 (Genome::N_PADDING)                        */

    /* contig table (Genome.cpp:203-229): beginning location of each contig, ascending    */
    const uint64_t *contig_begin;   /* [n_contigs] (always host)                          */
    uint32_t        n_contigs;
    uint64_t        first_alt_location; /* genomeLocationOfFirstALTContig; >= n_bases if none */

    uint32_t on_device;             /* 0: blobs are host memory, copy them; 1: device ptrs */

    /* ALT-to-primary projection of each contig (Genome::Contig::projBeginningLocation / isProjRC / projCigarOps, Genome.h:386-400,
     * written by an index built with -altLiftoverFile; Genome.cpp:226, 362-392).  Always host pointers; NULL = no projection data
     * (every contig then projects to location 0 with no CIGAR, which is what such an index file says).                          */
    const uint64_t *contig_proj_begin;   /* [n_contigs]                                                  */
    const uint8_t  *contig_proj_rc;      /* [n_contigs]                                                  */
    const uint32_t *contig_cigar_start;  /* [n_contigs + 1] offsets into cigar_ops                       */
    const uint32_t *cigar_ops;           /* (count << 8) | action, action in "MIDSH"                     */
} snapgpu_index_view;

/* Aligner options: the BaseAligner constructor arguments (BaseAligner.h:47-72) with the
 * defaults of AlignerOptions::AlignerOptions (AlignerOptions.cpp:39-117).                */
typedef struct snapgpu_params {
    uint32_t max_hits;             /* -h, 300                                            */
    uint32_t max_k;                /* -d, maxDist                                        */
    uint32_t num_seeds;            /* -n, 25 for single-end (0 => use seed_coverage)     */
    double   seed_coverage;        /* -sc                                                */
    uint32_t min_weight_to_check;  /* 1                                                  */
    uint32_t extra_search_depth;   /* -D, 1                                              */
    uint32_t use_affine_gap;       /* 1 unless -G-                                       */
    uint32_t match_reward;         /* 1                                                  */
    uint32_t sub_penalty;          /* 4                                                  */
    uint32_t gap_open_penalty;     /* 6                                                  */
    uint32_t gap_extend_penalty;   /* 1                                                  */
    uint32_t five_prime_end_bonus; /* 10                                                 */
    uint32_t three_prime_end_bonus;/* 7                                                  */
    uint32_t alt_awareness;        /* 1                                                  */
    uint32_t emit_alt_alignments;  /* 0                                                  */
    int32_t  max_score_gap_to_prefer_non_alt; /* 64                                      */
    uint32_t max_read_len;         /* upper bound on read length in any batch (<= 1000)  */
} snapgpu_params;

/* POD mirror of SingleAlignmentResult (SNAPLib/AlignmentResult.h:48-76). */
typedef struct snapgpu_single_result {
    int32_t  status;               /* SNAPGPU_NotFound / SingleHit / MultipleHits        */
    int32_t  direction;            /* 0 FORWARD, 1 RC (directions.h:22-29)               */
    int64_t  location;             /* InvalidGenomeLocation32 when not found             */
    int64_t  orig_location;
    int32_t  score;
    int32_t  score_prior_to_clipping;
    int32_t  mapq;
    int32_t  clipping_for_read_adjustment;
    int32_t  used_affine_gap_scoring;
    int32_t  bases_clipped_before;
    int32_t  bases_clipped_after;
    int32_t  ag_score;
    int32_t  supplementary;
    int32_t  seed_offset;
    double   match_probability;
    double   probability_all_candidates;
    uint32_t popular_seeds_skipped;
    uint32_t reserved;             /* not in the reference.  Bits 0-29: banded affine-gap traceback steps that left the computed
                                      band while scoring this read; non-zero means the reference's own result for this read can
                                      depend on what its aligner object scored before (stale traceback cells,
                                      AffineGapVectorized.h:743).  Bit 30: such a step happened in a call that was not its
                                      object's first for this read, where even a newly constructed aligner reads what its own
                                      earlier calls left; bit 31: this record is the exact pass's answer (the read was redone
                                      with the reference's traceback arrays kept across its calls).  DESIGN.md section 2 / 14. */
} snapgpu_single_result;

/* Paired-end options: PairedAlignerOptions (SNAPLib/PairedAligner.cpp:227-242) and the paired
 * branch of AlignerOptions::AlignerOptions (AlignerOptions.cpp:103-110), i.e. what
 * PairedAlignerContext passes to the IntersectingPairedEndAligner / ChimericPairedEndAligner
 * constructors (PairedAligner.cpp:589-625).  The single-end aligner inside the chimeric
 * fallback uses the context's snapgpu_params with max_k/2 (ChimericPairedEndAligner.cpp:81). */
typedef struct snapgpu_paired_params {
    int32_t  min_spacing;                 /* -s, 0                                        */
    uint32_t max_spacing;                 /* -s, 1000                                     */
    uint32_t force_spacing;               /* -fs, 0                                       */
    uint32_t max_big_hits;                /* -H, intersectingAlignerMaxHits, 4000         */
    uint32_t max_candidate_pool_size;     /* -mcp, 1000000                                */
    uint32_t num_seeds;                   /* -n for the intersecting aligner, 8           */
    double   seed_coverage;               /* -sc, 0                                       */
    uint32_t max_k_for_indels;            /* -i, 40                                       */
    uint32_t min_read_length;             /* -mrl, 50                                     */
    uint32_t use_soft_clipping;           /* 1                                            */
    int32_t  flatten_mapq_at_or_below;    /* 3                                            */
    int32_t  min_score_realignment;       /* 3                                            */
    int32_t  min_score_gap_realignment_alt; /* 3                                          */
    int32_t  min_ag_score_improvement;    /* 24                                           */
    uint32_t enable_hamming_scoring_base_aligner; /* 1                                    */
    uint32_t max_single_seeds;            /* maxSeedsSingleEnd, 25 (PairedAligner.cpp:57) */
} snapgpu_paired_params;

/* POD mirror of PairedAlignmentResult (SNAPLib/AlignmentResult.h:87-128).                */
typedef struct snapgpu_paired_result {
    int32_t  status[2];
    int32_t  direction[2];
    int64_t  location[2];
    int64_t  orig_location[2];
    int32_t  score[2];
    int32_t  score_prior_to_clipping[2];
    int32_t  mapq[2];
    int32_t  clipping_for_read_adjustment[2];
    int32_t  used_affine_gap_scoring[2];
    int32_t  bases_clipped_before[2];
    int32_t  bases_clipped_after[2];
    int32_t  ag_score[2];
    int32_t  supplementary[2];
    int32_t  seed_offset[2];
    int32_t  lv_indels[2];
    double   match_probability[2];
    double   probability_all_pairs;
    uint32_t popular_seeds_skipped[2];
    int32_t  used_gapless_clipping[2];
    int32_t  ref_span[2];
    int32_t  liftover[2];
    int32_t  aligned_as_pair;
    int32_t  ag_forced_single_aligner_call;
    uint32_t reserved;                    /* stale-traceback flag count, as in snapgpu_single_result */
    uint32_t flags;                       /* SNAPGPU_PAIR_* bits                          */
} snapgpu_paired_result;

#define SNAPGPU_PAIR_EXACT_REPLAY    4u   /* not an error: the pair's banded affine-gap traceback left the band in a call whose answer can depend on
                                             the object's earlier calls for the same pair, so the exact pass redid (or must redo) it -- informational */
#define SNAPGPU_PAIR_POOL_OVERFLOW   1u   /* candidate pool / affine-gap candidate buffer too small: result invalid */
#define SNAPGPU_PAIR_REF_BUFFER_DEPENDENT 2u /* (-om only) not an error.  The Hamming retry of the chimeric fallback produced more single-end
                                                secondary candidates than the 32-entry buffer PairedAligner.cpp:566 starts with.  The reference
                                                ignores that AlignRead's failure (ChimericPairedEndAligner.cpp:339 never assigns its return value),
                                                so ITS answer for this pair -- an unaligned read plus a truncated, unfiltered list -- depends on how
                                                far earlier pairs on the same thread had grown the buffer.  This library completes the call. */

/* per-call work counters (what BaseAligner exposes through getNHashTableLookups() etc.,
 * BaseAligner.h:106-111), used for the algorithmic-bytes roofline model.                */
typedef struct snapgpu_counters {
    uint64_t n_reads;
    uint64_t n_hash_table_lookups;      /* seed lookups (both strands count as one)      */
    uint64_t n_hash_slots_probed;       /* hash-table slots examined                     */
    uint64_t n_hits_consumed;           /* overflow/singleton hits fed to the candidate table */
    uint64_t n_overflow_lists;          /* lookups that dereferenced the overflow table  */
    uint64_t n_lv_locations;            /* nLocationsScoredWithLandauVishkin             */
    uint64_t n_ag_locations;            /* nLocationsScoredWithAffineGap                 */
    uint64_t n_lv_ref_bytes;            /* reference bytes the LV calls were entitled to */
    /* wave-cycles (shader clock) summed over all waves, for profile breakdowns */
    uint64_t cycles_lookup;             /* lookup_seed: hash probes + overflow header             */
    uint64_t cycles_hits;               /* candidate-table updates for the hits of a lookup        */
    uint64_t cycles_lv;                 /* Landau-Vishkin (both halves)                            */
    uint64_t cycles_ag;                 /* affine gap (both halves)                                */
    uint64_t cycles_total;              /* whole AlignRead                                         */
    /* paired-end path only.  [0]: wave-cycles of the chimeric fallback's single-end aligner.  [1]: Phase-4 help waits that were given
     * up by a watchdog (0 in a healthy run; bits 0-15 owner waits for answers, 16-31 owner waits for helpers to leave, 32+ idle waves).
     * [2]: lists published << 32 | speculative answers the ordered walks used (SNAPGPU_PAIRED_HELP_MIN; both 0 with the help off);
     * after a watchdog event -- top nibble set -- what the watchdog saw instead. */
    uint64_t reserved[3];
} snapgpu_counters;

typedef struct snapgpu_ctx snapgpu_ctx;

int         snapgpu_abi_version(void);
const char *snapgpu_last_error(const snapgpu_ctx *ctx);

/* Fill *p with the reference's single-end defaults (AlignerOptions.cpp:39-117).          */
void        snapgpu_default_params(snapgpu_params *p);

/* Create a context on HIP device `device`: uploads (or adopts, when idx->on_device) the
 * index blobs, precomputes the probability tables on the host with host libm
 * (LandauVishkin.cpp:716-763) and uploads them, and reserves per-wave scratch.           */
int  snapgpu_create(const snapgpu_index_view *idx, const snapgpu_params *p, int device, snapgpu_ctx **out);
void snapgpu_destroy(snapgpu_ctx *ctx);

/* Same as snapgpu_create, but reads the four files of a SNAP index directory itself
 * (GenomeIndex::loadFromDirectory, GenomeIndex.cpp:1839-2093; Genome::loadFromFile, Genome.cpp:277;
 * SNAPHashTable::loadCommon, HashTable.cpp:98-175) and uploads them.  This is what a C/C++ host
 * (e.g. the AlignerExtension shim of INTEGRATION.md) calls.                                      */
int  snapgpu_create_from_directory(const char *index_dir, const snapgpu_params *p, int device, snapgpu_ctx **out);

/*
 * Several contexts over ONE index: the reference runs one aligner object per thread over the shared, read-only g_index
 * (SNAPLib/ParallelTask.h:128-138, SingleAligner.cpp:145-173, AlignerContext.cpp:253-266).  Here a context is the unit of concurrency: calls on
 * one context are serial (one launch in flight, shared per-wave scratch), different contexts run concurrently on their own streams.
 *   snapgpu_device_count       visible HIP devices (0 without a GPU)
 *   snapgpu_create_replica     another context with src's options over src's index, on `device`:
 *                                share_index != 0  device must be src's: the new context ADOPTS src's index blobs in HBM (a second feeder
 *                                                  thread on the same GPU; destroy it before src)
 *                                share_index == 0  same-size blobs are allocated on `device` and left unfilled for snapgpu_broadcast_index
 *                              (snapgpu_enable_paired / _secondary are per context: call them on the replica as on src)
 *   snapgpu_create_replica_with_params   the same, with options of its own (`p` NULL: src's): one resident index serving several
 *                              option sets at once -- the reference's equivalent is a second `snap-aligner` command line of a comma-
 *                              separated run over the index it keeps loaded (SNAPLib/CommandProcessor.cpp: the index cache, -d per run)
 *   snapgpu_broadcast_index    fills the blobs of ctxs[1 .. n) from ctxs[0] (the one that read the index) with one RCCL broadcast per blob
 *                              over xGMI -- the only collective of the whole path; reads never cross GPUs.  n == 1 is a no-op.
 *                              SNAPGPU_E_UNSUPPORTED when librccl cannot be loaded (the caller may then load the directory per device).
 */
int  snapgpu_device_count(void);
int  snapgpu_create_replica(const snapgpu_ctx *src, int device, int share_index, snapgpu_ctx **out);
int  snapgpu_create_replica_with_params(const snapgpu_ctx *src, int device, int share_index, const snapgpu_params *p, snapgpu_ctx **out);
int  snapgpu_broadcast_index(snapgpu_ctx **ctxs, int n);

/* Device pointers of the context's index blobs, so that a caller that owns the
 * collective (RCCL broadcast of the index) can fill them in place.
 * Each pointer may be NULL if the caller does not want it.                               */
int  snapgpu_index_device_ptrs(snapgpu_ctx *ctx, void **hash_blob, void **overflow, void **genome_with_pad);

/*
 * Seed lookup, GenomeIndex::lookupSeed32 for n seeds.
 *   seeds:     n * seed_len bytes of ACGT text (host pointer)
 *   n_hits:    [2n] out; n_hits[2i] forward, n_hits[2i+1] reverse-complement hit counts;
 *              -1 for a seed whose text is not a seed (Seed::DoesTextRepresentASeed)
 *   hits:      [2n * max_hits_out] out; row 2i+dir holds the first min(n_hits,max_hits_out)
 *              locations in stored (descending) order
 */
int  snapgpu_lookup_seeds(snapgpu_ctx *ctx, uint32_t n, const char *seeds,
                          int64_t *n_hits, uint32_t *hits, uint32_t max_hits_out);

/* Device-pointer form: d_seeds (n * seed_len bytes), d_n_hits ([2n] int64) and d_hits ([2n * max_hits_out] uint32, or NULL: hit counts
 * only -- the lists are then read, up to max_hits_out entries each as BaseAligner consumes them, but not stored) are device pointers.
 * The launch is timed (snapgpu_kernel_time) and counted (snapgpu_get_counters: lookups, slots examined, hits, overflow lists), which
 * makes it the stand-alone index-probe kernel whose HBM roofline bench.py reports.  `stream` as in snapgpu_align_single_device. */
int  snapgpu_lookup_seeds_device(snapgpu_ctx *ctx, uint32_t n, const void *d_seeds, void *d_n_hits, void *d_hits,
                                 uint32_t max_hits_out, void *stream);

/*
 * Batched LandauVishkin<dir>::computeEditDistance.  Problem i:
 *   text    = texts + text_off[i], text_len[i] bytes readable in the direction of travel
 *             (for dir -1 the pointer addresses one past the first compared byte, as in the reference)
 *   pattern = patterns + pat_off[i], quality = quals + pat_off[i], pattern length pat_len[i]
 *   k[i]    = edit distance limit
 * Outputs (each [n]): score (ScoreAboveLimit if > k), match_probability, net_indel,
 * total_indels, text_span.
 */
int  snapgpu_landau_vishkin(snapgpu_ctx *ctx, int dir, uint32_t n,
                            const char *texts, uint64_t texts_bytes, const uint32_t *text_off, const int32_t *text_len,
                            const char *patterns, const char *quals, uint64_t patterns_bytes,
                            const uint32_t *pat_off, const int32_t *pat_len, const int32_t *k,
                            int32_t *score, double *match_probability, int32_t *net_indel,
                            int32_t *total_indels, int32_t *text_span);

/*
 * -f and -x of the single-end aligner (AlignerOptions.cpp:571-574; BaseAligner::setStopOnFirstHit / setExplorePopularSeeds as
 * SingleAligner.cpp:179-180 calls them after constructing each BaseAligner):
 *   stop_on_first_hit      AlignRead stops at the first location it scores within maxK and reports it with status MultipleHits and
 *                          MAPQ 0 (BaseAligner.cpp:1490-1505)
 *   explore_popular_seeds  a seed with more than maxHits hits in a direction is not skipped: its first maxHits hits are applied
 *                          (BaseAligner.cpp:574, :625)
 * They apply to this context's single-end entry points (snapgpu_align_single*, with or without secondary results).  The paired-end
 * aligners never see them in the reference either -- PairedAligner.cpp sets neither on the BaseAligner inside ChimericPairedEndAligner, and
 * IntersectingPairedEndAligner's stopOnFirstHit stays false (IntersectingPairedEndAligner.cpp:66) -- so snapgpu_align_paired* ignores them.
 * Both 0 by default.  Returns SNAPGPU_OK.
 */
int  snapgpu_set_aligner_flags(snapgpu_ctx *ctx, int stop_on_first_hit, int explore_popular_seeds);

/*
 * AlignmentAdjuster::AdjustAlignment (SNAPLib/AlignmentAdjuster.cpp:33-190) for a batch of results: what finalizeSecondaryResults does to
 * the primary and to every secondary result before the -om filter when the aligner runs with -ae (BaseAligner.cpp:2444-2463), and what
 * snapgpu_enable_secondary's adjust_alignments = 1 runs inside the alignment kernels.  Result i belongs to read
 * data[off[i] .. off[i] + len[i]) (as given to AlignRead: forward strand, no clipping of its own).
 *   in:  results[i].status / direction / location / score
 *   out: results[i].score = the edit distance LandauVishkinWithCigar::computeEditDistanceNormalized finds (k = MAX_K - 1); location moved
 *        and clipping_for_read_adjustment set when that alignment began with an indel; status = NotFound, location =
 *        InvalidGenomeLocation32 when moving it would leave the contig (:139-148).  NotFound results are left alone.
 * Host pointers; the genome is the one resident on the device.  Returns SNAPGPU_OK or a negative error.
 * LIMITATION: the read is taken as one the reader has NOT clipped.  The reference settles a contig-end overhang on the start of the UNCLIPPED
 * buffer (AlignmentAdjuster.cpp:167), which differs from the clipped start for an RC result of a back-clipped read and for a forward result
 * of a front-clipped one; this interface is not given the unclipped bytes.  The callers that ship with the library (shim/, snapgpu-sam)
 * therefore refuse, under -ae, a read the reader actually clipped whose alignment reaches the end of its contig (run them with -C--).
 */
int  snapgpu_adjust_alignments(snapgpu_ctx *ctx, uint32_t n, const char *data, uint64_t data_bytes, const uint64_t *off, const int32_t *len,
                               snapgpu_single_result *results);

/*
 * The CIGAR of a written read (first step of result -> SAM record on the device):
 * SAMFormat::computeCigar, Landau-Vishkin variant (SNAPLib/SAM.cpp:2354-2467), over
 * LandauVishkinWithCigar::computeEditDistanceNormalized (SNAPLib/LandauVishkin.cpp:507-648, BAM_CIGAR_OPS format) -- what
 * SAMFormat::writeRead runs per record (SAM.cpp:1976-1983) and writePairs runs for a read aligned without affine gap (:1687).
 * Item i: the clipped read in REFERENCE orientation, data + off[i], len[i] bases, aligned at genome location loc[i], with
 * extra_before[i] leading bases to soft-clip because the alignment starts before its contig (createSAMLine's
 * extraBasesClippedBefore).  k = MAX_K - 1 as in the reference.
 * Outputs per item:
 *   ops[i * ops_stride ..]   BAM cigar ops (count << 4 | code; codes M0 I1 D2 =7 X8); use_m != 0: M instead of = / X
 *   n_ops[i]                 ops written; -1 = the "*" cigar (the read falls off its contig, SAM.cpp:2397-2408)
 *   edit_distance[i]         NM; -1 above the limit, -2 when ops_stride is too small (the reference's "cigarBuf too small")
 *   add_front_clipping[i]    > 0: the alignment starts with that many deleted reference bases -- no cigar is returned, the
 *                            caller moves the location and calls again (SAM.cpp:1660-1684); < 0: leading insertion of that
 *                            many bases (cigar returned; the caller soft-clips them and calls again)
 *   extra_clipped_after[i]   bases hanging off the end of the contig, to be soft-clipped (iterated as :2434-2460 does)
 * Host pointers; the genome is the one resident on the device.  Returns SNAPGPU_OK or a negative error.
 */
int  snapgpu_compute_cigar_lv(snapgpu_ctx *ctx, uint32_t n, const char *data, uint64_t data_bytes, const uint64_t *off,
                              const int32_t *len, const int64_t *loc, const int32_t *extra_before, int use_m,
                              uint32_t *ops, uint32_t ops_stride, int32_t *n_ops, int32_t *edit_distance,
                              int32_t *add_front_clipping, int64_t *extra_clipped_after);

/*
 * The CIGAR of a read that was scored with affine gap: SAMFormat::computeCigar, affine-gap variant (SNAPLib/SAM.cpp:2470-2588),
 * over AffineGapVectorizedWithCigar::computeGlobalScoreNormalized (SNAPLib/AffineGapVectorized.cpp:1043-1128: the banded global
 * alignment when patternLen >= 3 (2k + 1), the full one otherwise or when the band failed) -- what SAMFormat::writePairs /
 * writeReads run for a read with usedAffineGapScoring or score > 0 (SAM.cpp:1653, :2200).  Scoring parameters are the context's
 * (snapgpu_params: match / substitution / gap open / gap extend).  Inputs as snapgpu_compute_cigar_lv plus quals (the clipped
 * read's qualities, same offsets; indexed from the clipped read's first base even when extra_before[i] > 0, as the reference
 * does) and score[i] = the alignment's edit distance (SingleAlignmentResult::score), the k of the band.
 * Additional outputs:
 *   back_clipping_missed[i]          bases of a tail insertion that the caller soft-clips (computeCigarString, SAM.cpp:2725-2727)
 *   reference_history_dependent[i]   1: the banded traceback stepped through a cell this call did not evaluate; the reference
 *                                    reads there what an earlier call left in its object (AffineGapVectorized.cpp:811 over
 *                                    AffineGapVectorized.h:1441), this library reads 0
 */
int  snapgpu_compute_cigar_ag(snapgpu_ctx *ctx, uint32_t n, const char *data, const char *quals, uint64_t data_bytes,
                              const uint64_t *off, const int32_t *len, const int64_t *loc, const int32_t *extra_before,
                              const int32_t *score, int use_m, uint32_t *ops, uint32_t ops_stride, int32_t *n_ops,
                              int32_t *edit_distance, int32_t *add_front_clipping, int64_t *extra_clipped_after,
                              int32_t *back_clipping_missed, int32_t *reference_history_dependent);

/*
 * From alignment results to the computed fields of their SAM records, on the device: for the primary
 * result of each single-end read, what SimpleReadWriter::writeReads (SNAPLib/ReadWriter.cpp:170-330) and SAMFormat::writeRead
 * (SNAPLib/SAM.cpp:1898-2352; createSAMLine :1424-1572, computeCigarString :2595-2766) compute before they print: orientation,
 * clipping bookkeeping, contig and position (Genome::getContigForRead), the CIGAR through the Landau-Vishkin or the affine-gap
 * variant (use_affine_gap of the context && (usedAffineGapScoring || score > 0), ReadWriter.cpp:232), the retry loop around a
 * leading indel (move the alignment / soft-clip the read / give the read up across a contig boundary), the soft clips around the
 * cigar and NM.  Printing names, sequence and tags stays with the caller.
 * Inputs: the reads as they came from the file (bases, quals, offsets[n + 1]), Read::clip's outcome for each (front_clip[i] bases
 * clipped in front, data_len[i] bases kept: what BaseAligner::AlignRead was given), and its result.
 * Outputs per read: flag (0x4 unmapped, 0x10 reverse strand, 0x800 when the result says supplementary; a caller that writes a secondary
 * result ORs 0x100 in itself, as createSAMLine does from its argument), contig (index into the index's contig table, -1 unmapped), pos
 * (1-based, 0 unmapped), mapq, ops / n_ops (BAM cigar ops incl. S = 4; n_ops = -1: "*"), nm (NM:i, -1 unmapped),
 * reference_history_dependent (see snapgpu_compute_cigar_ag).
 */
int  snapgpu_sam_fields_single(snapgpu_ctx *ctx, uint32_t n, const char *bases, const char *quals, const uint64_t *offsets,
                               const int32_t *front_clip, const int32_t *data_len, const snapgpu_single_result *results, int use_m,
                               int32_t *flag, int32_t *contig, int64_t *pos, int32_t *mapq, uint32_t *ops, uint32_t ops_stride,
                               int32_t *n_ops, int32_t *nm, int32_t *reference_history_dependent);

/* Device-pointer form of snapgpu_sam_fields_single: every array already in HBM (reads, Read::clip's outcome, the results
 * snapgpu_align_single_device left there), outputs left in HBM.  max_read_len >= the longest read of the batch (it sizes the LDS rows and
 * the per-wave scratch, which this call still allocates and frees itself).  Synchronous on `stream` (NULL: the context's). */
int  snapgpu_sam_fields_single_device(snapgpu_ctx *ctx, uint32_t n, uint32_t max_read_len, const void *d_bases, const void *d_quals,
                                      const void *d_offsets, const void *d_front_clip, const void *d_data_len, const void *d_results, int use_m,
                                      void *d_flag, void *d_contig, void *d_pos, void *d_mapq, void *d_ops, uint32_t ops_stride,
                                      void *d_n_ops, void *d_nm, void *d_reference_history_dependent, void *stream);

/*
 * The single-end path of a SAM writer in ONE call: BaseAligner::AlignRead (SingleAligner.cpp:250) over the clipped reads, then what
 * SimpleReadWriter::writeReads computes for each read's primary result (snapgpu_sam_fields_single) -- with the batch uploaded once and
 * the results handed from the align kernel to the SAM-field kernel in HBM.  Host pointers.
 *   bases / quals / offsets   the UNCLIPPED reads; front_clip / data_len: Read::clip's outcome (the aligner sees
 *                             bases[offsets[i] + front_clip[i] .. + data_len[i]), the SAM-field kernel the whole read)
 *   skip[i] != 0              the read is not given to the aligner (too short, too many Ns: SingleAligner.cpp:211-232) and is written unaligned
 *   results / first_alt       [n] out, each may be NULL: the SingleAlignmentResults, for a caller that wants them (first_alt NULL also
 *                             means the first-ALT result is not computed into a caller-visible buffer)
 *   flag .. reference_history_dependent   as snapgpu_sam_fields_single (a cigar that does not fit ops_stride: n_ops -1, nm -2)
 * The context must be a plain single-end one (no snapgpu_enable_secondary / _paired).  With secondary results, -ae or ALT records to
 * write, use the two calls it replaces.
 */
int  snapgpu_align_sam_single(snapgpu_ctx *ctx, uint32_t n, const char *bases, const char *quals, const uint64_t *offsets,
                              const int32_t *front_clip, const int32_t *data_len, const uint8_t *skip, int use_m,
                              snapgpu_single_result *results, snapgpu_single_result *first_alt,
                              int32_t *flag, int32_t *contig, int64_t *pos, int32_t *mapq, uint32_t *ops, uint32_t ops_stride,
                              int32_t *n_ops, int32_t *nm, int32_t *reference_history_dependent);

/*
 * The paired-end writer: for the primary PairedAlignmentResult of each pair, the computed fields of BOTH SAM records -- what
 * SAMFormat::writePairs (SNAPLib/SAM.cpp:1575-1895: createSAMLine and the cigar with its leading-indel loop per mate, :1636-1715) and
 * SAMFormat::fillMateInfo (:1308-1421: pairing flags, RNEXT / PNEXT, the signed template length from the clipped starts and the cigars'
 * reference spans, the "unmapped mate takes its partner's RNAME / POS" rule) compute, and the order SimpleReadWriter::writePairs prints the
 * two records in (ReadWriter.cpp:453-461: by final location).  Reads 2 i and 2 i + 1 are read 0 and read 1 of pair i.
 * Per read: the outputs of snapgpu_sam_fields_single (flag now with 0x1 / 0x2 / 0x8 / 0x20 / 0x40 / 0x80) plus rnext (-1 "*", -2 "=",
 * else a contig index), pnext, tlen.  Per pair: first_written (0 / 1).
 */
int  snapgpu_sam_fields_paired(snapgpu_ctx *ctx, uint32_t n_pairs, const char *bases, const char *quals, const uint64_t *offsets,
                               const int32_t *front_clip, const int32_t *data_len, const snapgpu_paired_result *results, int use_m,
                               int32_t *flag, int32_t *contig, int64_t *pos, int32_t *mapq, uint32_t *ops, uint32_t ops_stride,
                               int32_t *n_ops, int32_t *nm, int32_t *rnext, int64_t *pnext, int64_t *tlen, int32_t *first_written,
                               int32_t *reference_history_dependent);

/*
 * Batched AffineGapVectorized<dir>::computeScore (banded[i] == 0) / computeScoreBanded
 * (banded[i] != 0).  Same string conventions as snapgpu_landau_vishkin; w[i] is the band
 * / edit limit, score_init[i] the initial score, is_rc[i] selects the end bonus.
 * Outputs (each [n]): ag_score (-1 when not better than score_init), text_offset,
 * pattern_offset, n_edits, match_probability.
 */
int  snapgpu_affine_gap(snapgpu_ctx *ctx, int dir, uint32_t n,
                        const char *texts, uint64_t texts_bytes, const uint32_t *text_off, const int32_t *text_len,
                        const char *patterns, const char *quals, uint64_t patterns_bytes,
                        const uint32_t *pat_off, const int32_t *pat_len,
                        const int32_t *w, const int32_t *score_init, const uint8_t *is_rc,
                        const uint8_t *banded, const uint8_t *use_clipping_optimizations,
                        int32_t *ag_score, int32_t *text_offset, int32_t *pattern_offset,
                        int32_t *n_edits, double *match_probability);

/*
 * The same problems as CALLS IN ORDER ON ONE OBJECT (test entry): problem i is the i-th computeScore / computeScoreBanded call of one
 * newly constructed AffineGapVectorized<dir>, whose backtraceAction array (AffineGapVectorized.h:1374) starts zeroed and keeps what
 * every call wrote -- so a banded traceback step outside the band of call i reads what calls 0 .. i-1 left there (:740-788), as it does
 * in the reference.  One wavefront, the exact form of the kernels (the form the replay passes run).  stale_steps[i] (may be NULL) =
 * how many such steps call i made.  Arguments otherwise as snapgpu_affine_gap.
 */
int  snapgpu_affine_gap_sequence(snapgpu_ctx *ctx, int dir, uint32_t n,
                                 const char *texts, uint64_t texts_bytes, const uint32_t *text_off, const int32_t *text_len,
                                 const char *patterns, const char *quals, uint64_t patterns_bytes,
                                 const uint32_t *pat_off, const int32_t *pat_len,
                                 const int32_t *w, const int32_t *score_init, const uint8_t *is_rc,
                                 const uint8_t *banded, const uint8_t *use_clipping_optimizations,
                                 int32_t *ag_score, int32_t *text_offset, int32_t *pattern_offset,
                                 int32_t *n_edits, double *match_probability, int32_t *stale_steps);

/*
 * BaseAligner::AlignRead for a batch of n reads.
 *   bases/quals: concatenated read bytes (upper-case ACGTN / Phred+33), host pointers
 *   offsets:     [n+1] byte offsets; read i is bases[offsets[i] .. offsets[i+1])
 *   primary:     [n] out
 *   first_alt:   [n] out or NULL (only status is meaningful unless emit_alt_alignments)
 * Reads are independent; results do not depend on batch composition or order.
 * Secondary alignments (-om) are not produced by this entry point (the reference default,
 * maxSecondaryAlignmentAdditionalEditDistance = -1, AlignerOptions.cpp:70); see
 * snapgpu_align_single_secondary.
 */
int  snapgpu_align_single(snapgpu_ctx *ctx, uint32_t n, const char *bases, const char *quals,
                          const uint64_t *offsets, snapgpu_single_result *primary,
                          snapgpu_single_result *first_alt);

/*
 * Same, but the caller has already placed bases/quals/offsets in device memory and wants
 * results left in device memory (all four are device pointers).  This is the form the
 * throughput metric is quoted on (inputs resident in HBM when the clock starts).
 * `stream` is a hipStream_t (NULL = the context's own stream); the call is asynchronous
 * with respect to the host when stream is non-NULL.
 */
int  snapgpu_align_single_device(snapgpu_ctx *ctx, uint32_t n, const void *d_bases, const void *d_quals,
                                 const void *d_offsets, void *d_primary, void *d_first_alt, void *stream);

/*
 * Secondary alignments: AlignRead's maxEditDistanceForSecondaryResults / secondaryResults / maxSecondaryResults arguments
 * (BaseAligner.h:76-90; SingleAligner.cpp:250) and BaseAligner's maxSecondaryAlignmentsPerContig (BaseAligner.h:62),
 * i.e. -om, -omax and -mpc (AlignerOptions.cpp:70-72).
 */
typedef struct snapgpu_secondary_params {
    int32_t  max_edit_distance;    /* -om  maxSecondaryAlignmentAdditionalEditDistance, 0 <= om <= extra_search_depth   */
    int32_t  max_per_contig;       /* -mpc maxSecondaryAlignmentsPerContig, -1 = no limit                                */
    int64_t  max_results;          /* -omax maxSecondaryAlignments, 0x7fffffff                                           */
    uint32_t adjust_alignments;    /* -ae (!ignoreAlignmentAdjustmentsForOm): AlignmentAdjuster::AdjustAlignment on the primary and on
                                      every secondary result before the -om filter (BaseAligner.cpp:2444-2463); single-end contexts only:
                                      with snapgpu_enable_paired on the same context the call returns SNAPGPU_E_UNSUPPORTED               */
} snapgpu_secondary_params;

/* Sizes the per-wavefront secondary-result lists of `ctx` (2 * seeds * max_hits entries each: they cannot overflow, so the
 * reference's "buffer too small -> caller doubles and re-calls" path, SingleAligner.cpp:250-263, has no counterpart). */
int  snapgpu_enable_secondary(snapgpu_ctx *ctx, const snapgpu_secondary_params *sp);

/*
 * BaseAligner::AlignRead with secondary results, including finalizeSecondaryResults (BaseAligner.cpp:2423-2553).
 *   secondary:        [n * secondary_stride] out; read i's results are secondary[i*secondary_stride .. + min(n_secondary[i], stride)),
 *                     in the reference's order.  Fields the reference leaves unset in a secondary result (probability of all
 *                     candidates, popular seeds skipped) are 0.
 *   n_secondary:      [n] out; the number of secondary results read i HAS.
 * Returns SNAPGPU_OK, or SNAPGPU_W_SECONDARY_TRUNCATED (> 0) when some n_secondary[i] > secondary_stride: everything else is
 * filled in, and the caller may call again with a larger stride (min(-omax, what n_secondary says) always suffices).
 * primary / first_alt are what snapgpu_align_single returns, except that -- like the reference, BaseAligner.cpp:1512 -- the
 * search does not stop early once the candidates' total probability reaches 4.9, so probability_all_candidates and mapq of
 * repeat reads can differ from a run without -om.
 */
int  snapgpu_align_single_secondary(snapgpu_ctx *ctx, uint32_t n, const char *bases, const char *quals,
                                    const uint64_t *offsets, snapgpu_single_result *primary, snapgpu_single_result *first_alt,
                                    snapgpu_single_result *secondary, uint32_t secondary_stride, uint32_t *n_secondary);
/* device-pointer form, as snapgpu_align_single_device */
int  snapgpu_align_single_secondary_device(snapgpu_ctx *ctx, uint32_t n, const void *d_bases, const void *d_quals,
                                           const void *d_offsets, void *d_primary, void *d_first_alt,
                                           void *d_secondary, uint32_t secondary_stride, void *d_n_secondary, void *stream);

/*
 * Paired-end path.  snapgpu_enable_paired builds, on an existing context, the per-wave state of
 *       IntersectingPairedEndAligner::IntersectingPairedEndAligner   SNAPLib/IntersectingPairedEndAligner.cpp:36-100
 *       ChimericPairedEndAligner::ChimericPairedEndAligner           SNAPLib/ChimericPairedEndAligner.cpp:42-98
 * as PairedAlignerContext::runIterationThreadImpl builds them (SNAPLib/PairedAligner.cpp:556-625); the context's
 * snapgpu_params supply the options both aligners share (maxHits, maxDist, affine-gap scores, ALT handling).
 * snapgpu_align_paired replaces
 *       ChimericPairedEndAligner::align                               SNAPLib/ChimericPairedEndAligner.cpp:126-448
 *         -> IntersectingPairedEndAligner::align                      SNAPLib/IntersectingPairedEndAligner.cpp:169-251
 *         -> BaseAligner::AlignRead / alignAffineGap (fallback)       SNAPLib/BaseAligner.cpp:273, 1537
 *       called from PairedAlignerContext::runIterationThreadImpl      SNAPLib/PairedAligner.cpp:727
 * for a batch of n_pairs pairs: offsets has 2*n_pairs+1 entries, read r of pair i is
 * bases[offsets[2i+r] .. offsets[2i+r+1]).  primary/first_alt: [n_pairs] out (first_alt may be NULL).
 * ALT alignments are lifted over to the primary assembly as the reference does (IntersectingPairedEndAligner.cpp:2866-2968) when the
 * index view carries the contigs' projection data.  Secondary alignments (-om): snapgpu_align_paired_secondary.
 * A pair whose candidate pools overflowed is flagged (SNAPGPU_PAIR_POOL_OVERFLOW) and the call returns
 * SNAPGPU_E_UNSUPPORTED after filling in every other pair.
 */
void snapgpu_default_paired_params(snapgpu_paired_params *pp);
int  snapgpu_enable_paired(snapgpu_ctx *ctx, const snapgpu_paired_params *pp);
int  snapgpu_align_paired(snapgpu_ctx *ctx, uint32_t n_pairs, const char *bases, const char *quals,
                          const uint64_t *offsets, snapgpu_paired_result *primary, snapgpu_paired_result *first_alt);
/* device-pointer form, as snapgpu_align_single_device */
int  snapgpu_align_paired_device(snapgpu_ctx *ctx, uint32_t n_pairs, const void *d_bases, const void *d_quals,
                                 const void *d_offsets, void *d_primary, void *d_first_alt, void *stream);

/*
 * ... with secondary results (both snapgpu_enable_paired and snapgpu_enable_secondary called on the context, in either order):
 * ChimericPairedEndAligner::align's maxEditDistanceForSecondaryResults / secondaryResults / singleEndSecondaryResults arguments
 * (ChimericPairedEndAligner.cpp:126-148), called as PairedAligner.cpp:727 calls it.
 *   secondary:          [n_pairs * secondary_stride] paired secondary results, in the reference's order (IntersectingPairedEndAligner.cpp:
 *                       999-1034, 1082-1124, final filtering 1289-1411); n_secondary[i] = how many pair i HAS
 *   single_secondary:   [n_pairs * single_stride] single-end secondary results of the chimeric fallback: read 0's
 *                       n_single_secondary[2i] results, then read 1's n_single_secondary[2i+1] (the layout of PairedAligner.cpp:872)
 * Fields the reference leaves unset in a secondary result read as 0.  Returns SNAPGPU_W_SECONDARY_TRUNCATED when a pair has more than
 * fits a stride (as snapgpu_align_single_secondary).  SNAPGPU_PAIR_REF_BUFFER_DEPENDENT marks the pairs on which the reference's own
 * answer is an accident of its buffer size.
 */
int  snapgpu_align_paired_secondary(snapgpu_ctx *ctx, uint32_t n_pairs, const char *bases, const char *quals, const uint64_t *offsets,
                                    snapgpu_paired_result *primary, snapgpu_paired_result *first_alt,
                                    snapgpu_paired_result *secondary, uint32_t secondary_stride, uint32_t *n_secondary,
                                    snapgpu_single_result *single_secondary, uint32_t single_stride, uint32_t *n_single_secondary);
int  snapgpu_align_paired_secondary_device(snapgpu_ctx *ctx, uint32_t n_pairs, const void *d_bases, const void *d_quals,
                                           const void *d_offsets, void *d_primary, void *d_first_alt,
                                           void *d_secondary, uint32_t secondary_stride, void *d_n_secondary,
                                           void *d_single_secondary, uint32_t single_stride, void *d_n_single_secondary, void *stream);

/* Counters accumulated by snapgpu_align_single* since the last reset (device -> host). */
int  snapgpu_get_counters(snapgpu_ctx *ctx, snapgpu_counters *out, int reset);

/* Total duration in milliseconds of the kernel launches since the last reset, measured with hipEvents on the launch stream
 * (bench.py roofline), and how many launches that covers: the align kernels and the SAM-side kernels (snapgpu_compute_cigar_*,
 * snapgpu_sam_fields_*) share the accumulator, so reset before the call you want to time. */
int  snapgpu_kernel_time(snapgpu_ctx *ctx, double *total_ms, uint64_t *n_launches, int reset);

/* ------------------------------------------------------------------------------------------------------------------------------
 * The index builder on the GPU.  Replaces, for 4-byte locations and small tables -- the index shape the north star
 * uses: `snap-aligner index <fasta> <dir> -s 8..31 [-keysize 2..8]`, in particular the default -s 20 and the reference's own default -s 24,
 *   GenomeIndex::runIndexer            SNAPLib/GenomeIndex.cpp:126-506   options, FASTA -> Genome
 *   ReadFASTAGenome                    SNAPLib/FASTA.cpp:188-409         contigs, 'n' padding, ALT contigs last, upper-casing
 *   GenomeIndex::BuildIndexToDirectory SNAPLib/GenomeIndex.cpp:527-1022  hash tables + overflow table, the four files
 * The result is an index in the REFERENCE'S format: the directory snapgpu_built_index_save writes is loaded by the reference's
 * GenomeIndex::loadFromDirectory (and by snapgpu_create_from_directory), its `Genome` file is byte-identical to the reference's, and every
 * seed lookup returns what it returns on an index the reference built (hit lists are sorted, so they are equal; slot placement inside a
 * table depends on insertion order in both builders).  Hash-table sizes follow the reference's formula with exact distinct-seed counts
 * (its -exact mode; its default estimates them with approximate counters).
 * The build itself (seeds of all locations, radix sort, overflow lists, closed-hash insertion) runs on the device: index_build.h.
 * Anything outside that shape (-large, -locationSize > 4, key sizes other than 4) returns SNAPGPU_E_UNSUPPORTED: use the reference's indexer.
 */
typedef struct snapgpu_built_index snapgpu_built_index;

typedef struct snapgpu_index_build_params {
    uint32_t seed_len;              /* -s, default 20 (DEFAULT_SEED_SIZE)                                              */
    double   slack;                 /* -h, default 0.3 (DEFAULT_SLACK): tables are sized distinct seeds x (1 + slack)  */
    uint32_t key_bytes;             /* -keysize; 0 = auto: max(2, (seed_len + 2) / 4 - 1) (GenomeIndex.cpp:437)         */
    uint32_t chromosome_padding;    /* -p, default 2000 (DEFAULT_PADDING)                                              */
    /* FASTA -> Genome (snapgpu_index_build_from_fasta only) */
    uint32_t space_terminates_name; /* -bSpace (default 1) / -bSpace-                                                  */
    const char *name_terminators;   /* -B<chars>, NULL = none                                                          */
    uint32_t auto_alt;              /* default 1; -AutoAlt- = 0: contigs named *_alt or HLA-* are ALT (FASTA.cpp:64)    */
    int64_t  max_alt_contig_size;   /* -maxAltContigSize, default -1                                                   */
    const char *const *alt_contig_names;     uint32_t n_alt_contig_names;       /* -altContigName / -altContigFile       */
    const char *const *non_alt_contig_names; uint32_t n_non_alt_contig_names;   /* -nonAltContigName / -nonAltContigFile */
    const char *alt_liftover_file;  /* -altLiftoverFile, NULL = none                                                   */
} snapgpu_index_build_params;

/* what one build did, for logs and the bench line */
typedef struct snapgpu_index_build_stats {
    uint64_t n_bases, n_seed_locations, n_distinct_seeds, n_repeated_seeds, overflow_table_size, hash_table_slots, hash_blob_bytes;
    double   ms_keys, ms_sort, ms_runs, ms_tables, ms_total_device;
    double   s_fasta;               /* host: reading the FASTA into the genome image                                   */
} snapgpu_index_build_stats;

void snapgpu_default_index_build_params(snapgpu_index_build_params *bp);

/* A genome as Genome::saveToFile describes it (host memory): n_bases bytes (contigs separated by chromosome_padding bytes of 'n', upper-case
 * ACGTN), and the contig table in genome order.  proj_* may be NULL (no liftover data). */
typedef struct snapgpu_genome_view {
    const uint8_t *bases; uint64_t n_bases;
    uint32_t n_contigs;
    const uint64_t *contig_begin; const char *const *contig_name; const uint8_t *contig_is_alt; const int32_t *contig_original_number;
    const uint64_t *contig_proj_begin; const uint8_t *contig_proj_rc; const char *const *contig_proj_cigar;
} snapgpu_genome_view;

int  snapgpu_index_build(const snapgpu_genome_view *genome, const snapgpu_index_build_params *bp, int device, snapgpu_built_index **out);
int  snapgpu_index_build_from_fasta(const char *fasta_path, const snapgpu_index_build_params *bp, int device, snapgpu_built_index **out);
/* The built index as a view with on_device = 1 (feed it to snapgpu_create: no copy, no files); valid until snapgpu_built_index_destroy. */
int  snapgpu_built_index_view(const snapgpu_built_index *bi, snapgpu_index_view *view);
/* The four files of a SNAP index directory (GenomeIndex, Genome, OverflowTable, GenomeIndexHash). */
int  snapgpu_built_index_save(const snapgpu_built_index *bi, const char *directory);
int  snapgpu_built_index_stats(const snapgpu_built_index *bi, snapgpu_index_build_stats *out);
void snapgpu_built_index_destroy(snapgpu_built_index *bi);

#ifdef __cplusplus
}
#endif
#endif /* SNAPGPU_H */
