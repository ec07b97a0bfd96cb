/*
 * snap_oracle.h -- TEST INFRASTRUCTURE ONLY (see snap_oracle.c).
 * Plain-C restatement of the scoring/lookup primitives on SNAP's single-end hot path.
 */
#ifndef SNAP_ORACLE_H
#define SNAP_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

void   oracle_init(void);
const double *oracle_phred_table(void);      /* [256]  */
const double *oracle_indel_table(void);      /* [10001] */
const double *oracle_perfect_table(void);    /* [1001] */
double oracle_seed_prob(int seed_len);
double oracle_seed_prob_pow(int seed_len);      /* the libm-pow twin: BaseAligner.cpp:907 */
int    oracle_compute_mapq(double p_all, double p_best, int score, int popular_seeds_skipped);
unsigned oracle_wrapped_next_seed(unsigned seed_len, unsigned wrap_count);

/* Seed::Seed + DoesTextRepresentASeed: returns 0 if the text is not a seed */
int    oracle_pack_seed(const char *text, unsigned seed_len, uint64_t *bases, uint64_t *rc);

typedef struct oracle_index {
    uint32_t seed_len, key_bytes, n_hash_tables, large;
    const uint8_t *hash_blob; const uint64_t *table_offset; const uint64_t *table_size;
    const uint32_t *overflow; uint64_t n_bases;
} oracle_index;

/* GenomeIndex::lookupSeed32: n_hits[2], hits[2] (pointers into the index or to singleton[]), slots probed[2] */
void   oracle_lookup_seed(const oracle_index *ix, uint64_t bases, uint64_t rc, int64_t n_hits[2],
                          const uint32_t *hits[2], uint32_t singleton[2], uint32_t slots[2]);

/* LandauVishkin<dir>::computeEditDistance.  For dir == -1, `text` addresses one past the first
 * compared byte, exactly like the reference.  Bytes outside [0,text_len) / [0,pattern_len) are never read. */
int    oracle_lv(int dir, const char *text, int text_len, const char *pattern, const char *quality,
                 int pattern_len, int k, double *match_probability, int *net_indel, int *total_indels,
                 int *text_span);

typedef struct oracle_ag_params { int match_reward, sub_penalty, gap_open, gap_extend, five_bonus, three_bonus; } oracle_ag_params;

/* AffineGapVectorized<dir>::computeScore (banded == 0) / computeScoreBanded (banded != 0).
 * *stale_reads (if non-NULL) counts traceback reads of cells this call never wrote (the reference
 * would read whatever an earlier call left there). */
void   oracle_ag_bind_objects(uint8_t *fwd, uint8_t *bwd, size_t cap_each);      /* see snap_oracle.c: the traceback arrays of ONE aligner's two objects */
void   oracle_ag_bound_objects(uint8_t **fwd, uint8_t **bwd, size_t *cap_each);
int    oracle_ag(int dir, int banded, const oracle_ag_params *prm, const char *text, int text_len,
                 const char *pattern, const char *quality, int pattern_len, int w, int score_init,
                 int is_rc, int use_clipping, int *text_offset, int *pattern_offset, int *n_edits,
                 double *match_probability, int *stale_reads);

typedef struct oracle_genome {          /* what Genome holds for the aligner and the CIGAR writer */
    const uint8_t *genome;              /* base 0; genome_pad readable bytes before and after */
    uint64_t n_bases;
    uint32_t genome_pad, chromosome_padding;
    const uint64_t *contig_begin;
    uint32_t n_contigs;
    uint64_t first_alt_location;
} oracle_genome;

#ifdef __cplusplus
}
#endif
/* -f / -x for the oracle_align_read* calls that follow (align_oracle.c) */
void oracle_set_aligner_flags(int stop_on_first_hit, int explore_popular_seeds);

#endif
