/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * A C-ABI driver written for this repo that links against the *unmodified* reference
 * sources where they lie (/root/reference/SNAPLib/*.cpp, compiled by oracle/Makefile into
 * oracle/_ref/).  It exposes the reference's own classes for the hot path so that tests
 * and bench.py's cpu_baseline leg can run "the real thing":
 *
 *   snapref_lookup_seeds   -> GenomeIndex::lookupSeed32            (GenomeIndex.cpp:2096)
 *   snapref_landau_vishkin -> LandauVishkin<+-1>::computeEditDistance (LandauVishkin.h:100)
 *   snapref_affine_gap     -> AffineGapVectorized<+-1>::computeScore[Banded] (AffineGapVectorized.h:821/256)
 *   snapref_align_single   -> BaseAligner::AlignRead               (BaseAligner.cpp:273), one
 *                             aligner object per thread exactly as SingleAligner.cpp:145-173 builds it
 *   snapref_align_paired   -> ChimericPairedEndAligner::align over IntersectingPairedEndAligner::align
 *                             (ChimericPairedEndAligner.cpp:126, IntersectingPairedEndAligner.cpp:169),
 *                             objects built as PairedAligner.cpp:556-625 builds them
 *
 * Nothing under snap_amd/ may link, import or call this file; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() do, and only as the checker.
 */
#include "stdafx.h"
#include "Compat.h"
#include "BaseAligner.h"
#include "GenomeIndex.h"
#include "Genome.h"
#include "SeedSequencer.h"
#include "LandauVishkin.h"
#include "AffineGapVectorized.h"
#include "AlignerOptions.h"
#include "BigAlloc.h"
#include "Read.h"
#include "Seed.h"
#include "mapq.h"
#include "IntersectingPairedEndAligner.h"
#include "ChimericPairedEndAligner.h"
#include "SAM.h"
#include "AlignmentAdjuster.h"

#include <pthread.h>
#include <string.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
#include <new>

#include "../include/snapgpu.h"

extern GenomeIndex *g_index;                 // SNAPLib/AlignerContext.cpp:58

extern "C" {

static bool g_inited = false;

int snapref_init(void)
{
    if (!g_inited) {
        InitializeSeedSequencers();                    // CommandProcessor.cpp:196
        initializeLVProbabilitiesToPhredPlus33();      // LandauVishkin.cpp:716 (also MAPQ tables)
        g_inited = true;
    }
    return 0;
}

void *snapref_load_index(const char *dir)
{
    snapref_init();
    char *d = strdup(dir);
    GenomeIndex *index = GenomeIndex::loadFromDirectory(d, false /*map*/, false /*prefetch*/);
    free(d);
    return index;
}

struct snapref_index_info {
    uint32_t seed_len;
    uint32_t location_size_is_64;
    uint64_t n_bases;
    uint32_t n_contigs;
    uint32_t chromosome_padding;
};

int snapref_index_info_get(void *vindex, snapref_index_info *out)
{
    GenomeIndex *index = (GenomeIndex *)vindex;
    const Genome *genome = index->getGenome();
    out->seed_len = index->getSeedLength();
    out->location_size_is_64 = index->doesGenomeIndexHave64BitLocations() ? 1 : 0;
    out->n_bases = (uint64_t)genome->getCountOfBases();
    out->n_contigs = genome->getNumContigs();
    out->chromosome_padding = genome->getChromosomePadding();
    return 0;
}

/* probability tables as the reference computed them (for bit-equality checks of ours) */
int snapref_tables(double *phred256, double *indel, uint32_t n_indel, double *perfect, uint32_t n_perfect)
{
    snapref_init();
    for (int i = 0; i < 256; i++) phred256[i] = lv_phredToProbability[i];
    for (uint32_t i = 0; i < n_indel; i++) indel[i] = lv_indelProbabilities[i];
    for (uint32_t i = 0; i < n_perfect; i++) perfect[i] = lv_perfectMatchProbability[i];
    return 0;
}

int snapref_compute_mapq(double pAll, double pBest, int score, int popularSeedsSkipped)
{
    return computeMAPQ(pAll, pBest, score, popularSeedsSkipped);
}

/* pow(1 - SNP_PROB, seedLen) exactly as BaseAligner.cpp:1314 evaluates it (int seedLen, C++98 overloads) */
double snapref_seed_prob(int seedLen)
{
    return pow(1 - SNP_PROB, seedLen);
}

unsigned snapref_wrapped_next_seed(unsigned seedLen, unsigned wrapCount)
{
    snapref_init();
    return GetWrappedNextSeedToTest(seedLen, wrapCount);
}

int snapref_lookup_seeds(void *vindex, uint32_t n, const char *seeds,
                         int64_t *n_hits, uint32_t *hits, uint32_t max_hits_out)
{
    GenomeIndex *index = (GenomeIndex *)vindex;
    const bool wide = index->doesGenomeIndexHave64BitLocations();         // lookupSeed / overflowTable64 (GenomeIndex.cpp:2205-2328)
    unsigned seedLen = index->getSeedLength();
    for (uint32_t i = 0; i < n; i++) {
        const char *text = seeds + (size_t)i * seedLen;
        if (!Seed::DoesTextRepresentASeed(text, seedLen)) {
            n_hits[2 * i] = n_hits[2 * i + 1] = -1;
            continue;
        }
        Seed seed(text, seedLen);
        if (wide) {
            _int64 nh64[2] = {0, 0};
            const GenomeLocation *h64[2] = {NULL, NULL};
            GenomeLocation single64[2];
            index->lookupSeed(seed, &nh64[0], &h64[0], &nh64[1], &h64[1], &single64[0], &single64[1]);
            for (int d = 0; d < 2; d++) {
                n_hits[2 * i + d] = nh64[d];
                _int64 lim = nh64[d] < (_int64)max_hits_out ? nh64[d] : (_int64)max_hits_out;
                for (_int64 j = 0; j < lim; j++) hits[(size_t)(2 * i + d) * max_hits_out + j] = (uint32_t)GenomeLocationAsInt64(h64[d][j]);
            }
            continue;
        }
        _int64 nh[2] = {0, 0};
        const unsigned *h[2] = {NULL, NULL};
        index->lookupSeed32(seed, &nh[0], &h[0], &nh[1], &h[1]);
        for (int d = 0; d < 2; d++) {
            n_hits[2 * i + d] = nh[d];
            _int64 lim = nh[d] < (_int64)max_hits_out ? nh[d] : (_int64)max_hits_out;
            for (_int64 j = 0; j < lim; j++) {
                hits[(size_t)(2 * i + d) * max_hits_out + j] = h[d][j];
            }
        }
    }
    return 0;
}

/*
 * The reference's string kernels use unaligned 8-byte loads that run a few bytes past
 * either string (harmless inside SNAP because of genome padding and read buffers).  To call
 * them on arbitrary test strings we copy each problem into buffers with slack on both sides.
 */
static const int SLACK = 64;

struct PaddedProblem {
    std::vector<char> tbuf, pbuf, qbuf;
    const char *text;
    const char *pattern;
    const char *quality;
    void set(int dir, const char *t, int tlen, const char *p, const char *q, int plen) {
        int tl = tlen > 0 ? tlen : 0;
        tbuf.assign(tl + 2 * SLACK, 'n');
        if (dir == 1) {
            memcpy(&tbuf[SLACK], t, tl);
            text = &tbuf[SLACK];
        } else {
            // bytes t[-tlen .. -1] are the text, travelled backwards
            memcpy(&tbuf[SLACK], t - tl, tl);
            text = &tbuf[SLACK] + tl;
        }
        pbuf.assign(plen + 2 * SLACK, 'N' + 1);   // a byte that matches nothing in the text slack
        memcpy(&pbuf[SLACK], p, plen);
        pattern = &pbuf[SLACK];
        qbuf.assign(plen + 2 * SLACK, '!');
        if (q) memcpy(&qbuf[SLACK], q, plen);
        quality = &qbuf[SLACK];
    }
};

int snapref_landau_vishkin(int dir, uint32_t n,
                           const char *texts, const uint32_t *text_off, const int32_t *text_len,
                           const char *patterns, const char *quals, const uint32_t *pat_off,
                           const int32_t *pat_len, const int32_t *k,
                           int32_t *score, double *match_probability, int32_t *net_indel,
                           int32_t *total_indels, int32_t *text_span)
{
    snapref_init();
    LandauVishkin<1> *fwd = new LandauVishkin<1>;
    LandauVishkin<-1> *bwd = new LandauVishkin<-1>;
    PaddedProblem pp;
    for (uint32_t i = 0; i < n; i++) {
        pp.set(dir, texts + text_off[i], text_len[i], patterns + pat_off[i], quals + pat_off[i], pat_len[i]);
        double prob = 0; int ni = 0, ti = 0, ts = 0; int s;
        if (dir == 1) {
            s = fwd->computeEditDistance(pp.text, text_len[i], pp.pattern, pp.quality, pat_len[i], k[i], &prob, &ni, &ti, &ts);
        } else {
            s = bwd->computeEditDistance(pp.text, text_len[i], pp.pattern, pp.quality, pat_len[i], k[i], &prob, &ni, &ti, &ts);
        }
        score[i] = s; match_probability[i] = prob; net_indel[i] = ni; total_indels[i] = ti; text_span[i] = ts;
    }
    delete fwd;
    delete bwd;
    return 0;
}

struct AGParams { int match, sub, open, ext, five, three; };

/* "Fresh object" mode (parity exclusion): with snapref_set_fresh_objects(1) every read / pair is aligned by aligner objects
 * that were constructed, just before the call, inside a ZERO-FILLED arena -- what a newly started reference thread would use for its first
 * read.  The reference's answer is then a function of the read alone (its banded affine gap can trace back through cells an earlier call
 * left in the object, AffineGapVectorized.h:740-788 over :1374; see DESIGN.md "Reference nondeterminism"), so every read can be compared,
 * none excluded.  BigAllocator::allocate is virtual (BigAlloc.h:95): the arena below hands out zeroed memory it can take back. */
class ZeroedArena : public BigAllocator {
public:
    ZeroedArena(size_t cap_) : BigAllocator(0, 16), cap(cap_ + 4096), used(0) { base = (char *)calloc(cap, 1); }
    ~ZeroedArena() { free(base); }
    virtual void *allocate(size_t amount) {
        size_t a = (amount + 63) & ~(size_t)63;
        if (used + a > cap) { fprintf(stderr, "ref_driver: ZeroedArena overflow (%zu + %zu > %zu)\n", used, a, cap); abort(); }
        void *r = base + used; used += a; return r;
    }
    void reset() { memset(base, 0, used); used = 0; }          // everything handed out since the last reset reads as zero again
private:
    char *base; size_t cap, used;
};
/* -f / -x for the aligners snapref_align_single* construct: applied as SingleAligner.cpp:179-180 applies them, after each construction */
static volatile int g_stop_on_first_hit = 0, g_explore_popular_seeds = 0;
void snapref_set_aligner_flags(int stop_on_first_hit, int explore_popular_seeds) { g_stop_on_first_hit = stop_on_first_hit; g_explore_popular_seeds = explore_popular_seeds; }
static inline BaseAligner *with_flags(BaseAligner *a) { a->setStopOnFirstHit(g_stop_on_first_hit != 0); a->setExplorePopularSeeds(g_explore_popular_seeds != 0); return a; }
static volatile int g_fresh_objects = 0;
void snapref_set_fresh_objects(int on) { g_fresh_objects = on; }
int snapref_get_fresh_objects(void) { return g_fresh_objects; }

int snapref_affine_gap(int dir, uint32_t n, const int32_t *agparams /* match,sub,open,ext,5',3' */,
                       const char *texts, const uint32_t *text_off, const int32_t *text_len,
                       const char *patterns, const char *quals, const uint32_t *pat_off, const int32_t *pat_len,
                       const int32_t *w, const int32_t *score_init, const uint8_t *is_rc,
                       const uint8_t *banded, const uint8_t *use_clip,
                       int32_t *ag_score, int32_t *text_offset, int32_t *pattern_offset,
                       int32_t *n_edits, double *match_probability)
{
    snapref_init();
    // 16-byte granularity exactly as SingleAligner.cpp:147 ("FIXME: Used larger allocation granularity for __m128i").
    // The n problems are calls IN ORDER on these two objects; with snapref_set_fresh_objects(1) the objects live in zero-filled memory,
    // so what an out-of-band traceback step of problem i reads is what problems 0 .. i-1 left there and nothing else.
    const size_t ag_reservation = AffineGapVectorized<1>::getBigAllocatorReservation() + AffineGapVectorized<-1>::getBigAllocatorReservation() + 4096;
    BigAllocator *alloc = g_fresh_objects ? (BigAllocator *)new ZeroedArena(ag_reservation + (1 << 16)) : new BigAllocator(ag_reservation, 16);
    AffineGapVectorized<1> *fwd = new (alloc) AffineGapVectorized<1>(agparams[0], agparams[1], agparams[2], agparams[3], agparams[4], agparams[5]);
    AffineGapVectorized<-1> *bwd = new (alloc) AffineGapVectorized<-1>(agparams[0], agparams[1], agparams[2], agparams[3], agparams[4], agparams[5]);
    PaddedProblem pp;
    for (uint32_t i = 0; i < n; i++) {
        pp.set(dir, texts + text_off[i], text_len[i], patterns + pat_off[i], quals + pat_off[i], pat_len[i]);
        double prob = 0; int to = 0, po = 0, ne = 0; int s;
        bool clip = use_clip && use_clip[i];
        bool lift = use_clip && use_clip[i] == 2;          // 2 = clipping optimisations with useAltLiftover
        if (dir == 1) {
            if (banded[i]) s = fwd->computeScoreBanded(pp.text, text_len[i], pp.pattern, pp.quality, pat_len[i], w[i], score_init[i], is_rc[i] != 0, &to, &po, &ne, &prob, clip, lift);
            else           s = fwd->computeScore(pp.text, text_len[i], pp.pattern, pp.quality, pat_len[i], w[i], score_init[i], is_rc[i] != 0, &to, &po, &ne, &prob, clip, lift);
        } else {
            if (banded[i]) s = bwd->computeScoreBanded(pp.text, text_len[i], pp.pattern, pp.quality, pat_len[i], w[i], score_init[i], is_rc[i] != 0, &to, &po, &ne, &prob, clip, lift);
            else           s = bwd->computeScore(pp.text, text_len[i], pp.pattern, pp.quality, pat_len[i], w[i], score_init[i], is_rc[i] != 0, &to, &po, &ne, &prob, clip, lift);
        }
        ag_score[i] = s; text_offset[i] = to; pattern_offset[i] = po; n_edits[i] = ne; match_probability[i] = prob;
    }
    fwd->~AffineGapVectorized();
    bwd->~AffineGapVectorized();
    delete alloc;
    return 0;
}

/* InvalidGenomeLocation is all ones in as many bytes as the loaded index's locations have (GenomeIndex.cpp: it is set when an index
 * loads); the C ABI has one sentinel, SNAPGPU_InvalidGenomeLocation32, whatever the files' location size. */
static inline int64_t abi_location(GenomeLocation l)
{
    return l == InvalidGenomeLocation ? (int64_t)SNAPGPU_InvalidGenomeLocation32 : (int64_t)GenomeLocationAsInt64(l);
}

static void fill_result(snapgpu_single_result *o, const SingleAlignmentResult *r)
{
    memset(o, 0, sizeof(*o));
    o->status = (int32_t)r->status;
    o->direction = (int32_t)r->direction;
    o->location = abi_location(r->location);
    o->orig_location = (int64_t)GenomeLocationAsInt64(r->origLocation);
    o->score = r->score;
    o->score_prior_to_clipping = r->scorePriorToClipping;
    o->mapq = r->mapq;
    o->clipping_for_read_adjustment = r->clippingForReadAdjustment;
    o->used_affine_gap_scoring = r->usedAffineGapScoring ? 1 : 0;
    o->bases_clipped_before = r->basesClippedBefore;
    o->bases_clipped_after = r->basesClippedAfter;
    o->ag_score = r->agScore;
    o->supplementary = r->supplementary ? 1 : 0;
    o->seed_offset = r->seedOffset;
    o->match_probability = r->matchProbability;
    o->probability_all_candidates = r->probabilityAllCandidates;
    o->popular_seeds_skipped = r->popularSeedsSkipped;
}

/* -ae (AlignerOptions.cpp:476): the aligners below are constructed with ignoreAlignmentAdjustmentsForOm = !g_adjust_alignments */
static volatile int g_adjust_alignments = 0;
void snapref_set_adjust_alignments(int on) { g_adjust_alignments = on; }
int snapref_get_adjust_alignments(void) { return g_adjust_alignments; }

struct AlignJob {
    GenomeIndex *index;
    const snapgpu_params *p;
    uint32_t n;
    const char *bases;
    const char *quals;
    const uint64_t *offsets;
    snapgpu_single_result *primary;
    snapgpu_single_result *first_alt;
    volatile _int64 next;      // shared work cursor
    uint32_t chunk;
    // per-job counters, summed under lock
    pthread_mutex_t lock;
    _int64 lookups, lv, ag;
};

static void *align_thread(void *arg)
{
    AlignJob *job = (AlignJob *)arg;
    const snapgpu_params *p = job->p;
    GenomeIndex *index = job->index;
    int maxReadSize = MAX_READ_LENGTH;

    // mirror of SingleAligner.cpp:145-173
    const bool fresh = g_fresh_objects != 0;
    const size_t reservation = BaseAligner::getBigAllocatorReservation(index, true, p->max_hits, maxReadSize, index->getSeedLength(),
                                                p->num_seeds, p->seed_coverage, -1, p->extra_search_depth) + 4096;
    ZeroedArena *arena = fresh ? new ZeroedArena(reservation + (1 << 20)) : NULL;
    BigAllocator *allocator = fresh ? (BigAllocator *)arena : new BigAllocator(reservation, 16);
#define NEW_SINGLE_ALIGNER() with_flags(new (allocator) BaseAligner( \
        index, p->max_hits, p->max_k, maxReadSize, p->num_seeds, p->seed_coverage, p->min_weight_to_check, \
        p->extra_search_depth, DisabledOptimizations(), p->use_affine_gap != 0, \
        g_adjust_alignments == 0 /* ignoreAlignmentAdjustmentsForOm: true by default (AlignerOptions.cpp:96), false with -ae */, \
        p->alt_awareness != 0, p->emit_alt_alignments != 0, p->max_score_gap_to_prefer_non_alt, \
        -1 /* maxSecondaryAlignmentsPerContig */, NULL, NULL, \
        p->match_reward, p->sub_penalty, p->gap_open_penalty, p->gap_extend_penalty, \
        p->five_prime_end_bonus, p->three_prime_end_bonus, NULL, allocator))
    BaseAligner *aligner = NEW_SINGLE_ALIGNER();
    _int64 acc_lookups = 0, acc_lv = 0, acc_ag = 0;

    // A read buffer with slack: the reference's LV over-reads a few bytes past the read.
    std::vector<char> bbuf(MAX_READ_LENGTH + 2 * SLACK, 0), qbuf(MAX_READ_LENGTH + 2 * SLACK, 0);

    for (;;) {
        _int64 begin = __sync_fetch_and_add(&job->next, (_int64)job->chunk);
        if (begin >= (_int64)job->n) break;
        _int64 end = begin + job->chunk;
        if (end > (_int64)job->n) end = job->n;
        for (_int64 i = begin; i < end; i++) {
            unsigned len = (unsigned)(job->offsets[i + 1] - job->offsets[i]);
            memcpy(&bbuf[SLACK], job->bases + job->offsets[i], len);
            memcpy(&qbuf[SLACK], job->quals + job->offsets[i], len);
            Read read;
            read.init("r", 1, &bbuf[SLACK], &qbuf[SLACK], len, NULL, 0);
            SingleAlignmentResult r, alt;
            memset(&r, 0, sizeof(r));
            memset(&alt, 0, sizeof(alt));
            alt.status = NotFound;
            _int64 nSecondary = 0;
            if (fresh) {                                    // a newly constructed aligner in zero-filled memory for every read
                acc_lookups += aligner->getNHashTableLookups(); acc_lv += aligner->getLocationsScoredWithLandauVishkin();
                acc_ag += aligner->getLocationsScoredWithAffineGap();
                aligner->~BaseAligner(); arena->reset(); aligner = NEW_SINGLE_ALIGNER();
            }
            // same call shape as SingleAligner.cpp:250 with the default -om (none)
            aligner->AlignRead(&read, &r, &alt, -1, 0, &nSecondary, 0, NULL, 0, NULL, NULL);
            fill_result(&job->primary[i], &r);
            if (job->first_alt) {
                if (alt.status == NotFound) {
                    memset(&job->first_alt[i], 0, sizeof(job->first_alt[i]));
                    job->first_alt[i].status = NotFound;
                } else {
                    fill_result(&job->first_alt[i], &alt);
                }
            }
        }
    }

    pthread_mutex_lock(&job->lock);
    job->lookups += acc_lookups + aligner->getNHashTableLookups();
    job->lv += acc_lv + aligner->getLocationsScoredWithLandauVishkin();
    job->ag += acc_ag + aligner->getLocationsScoredWithAffineGap();
    pthread_mutex_unlock(&job->lock);

    aligner->~BaseAligner();
    if (fresh) delete arena; else delete allocator;
#undef NEW_SINGLE_ALIGNER
    return NULL;
}

/* Returns 0; *seconds = wall time of the parallel align phase (index load excluded, as
 * AlignerContext.cpp:420 measures it).  counters3 = {lookups, LV locations, AG locations}. */
int snapref_align_single(void *vindex, const snapgpu_params *p, uint32_t n, const char *bases,
                         const char *quals, const uint64_t *offsets, int n_threads,
                         snapgpu_single_result *primary, snapgpu_single_result *first_alt,
                         int64_t *counters3, double *seconds)
{
    snapref_init();
    GenomeIndex *index = (GenomeIndex *)vindex;
    /* (an index with 5 .. 8-byte locations: the reference aligners take their 64-bit branches; results are locations either way) */
    AlignJob job;
    job.index = index; job.p = p; job.n = n; job.bases = bases; job.quals = quals; job.offsets = offsets;
    job.primary = primary; job.first_alt = first_alt; job.next = 0; job.chunk = 256;
    job.lookups = job.lv = job.ag = 0;
    pthread_mutex_init(&job.lock, NULL);
    if (n_threads < 1) n_threads = 1;

    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    std::vector<pthread_t> th(n_threads);
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, align_thread, &job);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds) *seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    if (counters3) { counters3[0] = job.lookups; counters3[1] = job.lv; counters3[2] = job.ag; }
    pthread_mutex_destroy(&job.lock);
    return 0;
}


// ---------------------------------------------------------------------------------------- CIGAR (SAM writer side)

/* LandauVishkinWithCigar::computeEditDistance, COMPACT_CIGAR_STRING format (LandauVishkin.cpp:141).  The caller places text and
 * pattern inside larger buffers: the reference compares 8 bytes at a time and looks past both ends. */
int snapref_lv_cigar(const char *text, int text_len, const char *pattern, int pattern_len, int k, int use_m,
                     char *cigar, int cigar_cap, int *text_used, int *net_indel)
{
    static LandauVishkinWithCigar *lvc = NULL;           /* one object, as one SAM writer thread has (not thread safe: tests only) */
    if (lvc == NULL) lvc = new LandauVishkinWithCigar();
    int used = 0;
    return lvc->computeEditDistance(text, text_len, pattern, pattern_len, k, cigar, cigar_cap, use_m != 0, COMPACT_CIGAR_STRING, &used, text_used, net_indel);
}

/* SAMFormat::computeCigar, affine-gap variant (SAM.cpp:2470-2588), BAM_CIGAR_OPS output, for a batch.  agparams = match, sub, open,
 * extend as given to AffineGapVectorizedWithCigar's constructor.  fresh_object != 0: a new, zero-filled object per item (the answer
 * is then a function of the item alone); otherwise one object serves the whole batch, as one SAM writer thread's does. */
int snapref_compute_cigar_ag(void *vindex, const int32_t *agparams, uint32_t n, const char *data, const char *quals, const uint64_t *off,
                             const int32_t *len, const int64_t *loc, const int32_t *extra_before, const int32_t *score, int use_m,
                             int fresh_object, uint32_t *ops, uint32_t ops_stride, int32_t *n_ops, int32_t *edit_distance,
                             int32_t *add_front_clipping, int64_t *extra_after, int32_t *tail_ins)
{
    GenomeIndex *index = (GenomeIndex *)vindex;
    const Genome *genome = index->getGenome();
    AffineGapVectorizedWithCigar *agc = NULL;
    void *mem = NULL;
    for (uint32_t i = 0; i < n; i++) {
        if (agc == NULL || fresh_object) {
            if (agc) { agc->~AffineGapVectorizedWithCigar(); free(mem); }
            mem = calloc(1, sizeof(AffineGapVectorizedWithCigar) + 64);
            void *al = (void *)(((uintptr_t)mem + 15) & ~(uintptr_t)15);
            agc = new (al) AffineGapVectorizedWithCigar(agparams[0], agparams[1], agparams[2], agparams[3]);
        }
        char *buf = (char *)(ops + (size_t)i * ops_stride);
        int used = 0, afc = 0, ed = 0, tail = 0;
        GenomeDistance after = 0;
        buf[0] = 0;
        SAMFormat::computeCigar(BAM_CIGAR_OPS, genome, agc, buf, (int)(ops_stride * 4), data + off[i], quals + off[i], (GenomeDistance)len[i], score[i], 0,
                                (GenomeDistance)extra_before[i], 0, &after, GenomeLocation(loc[i]), use_m != 0, &ed, &used, &afc, &tail);
        n_ops[i] = (used == 0 && buf[0] == '*') ? -1 : used / 4;
        edit_distance[i] = ed; add_front_clipping[i] = afc; extra_after[i] = (int64_t)after; tail_ins[i] = tail;
    }
    if (agc) { agc->~AffineGapVectorizedWithCigar(); free(mem); }
    return 0;
}

/* SAMFormat::computeCigar, Landau-Vishkin variant (SAM.cpp:2354-2467), BAM_CIGAR_OPS output, for a batch: item i is the clipped read
 * data[off[i] .. off[i] + len[i]) in reference orientation at genome location loc[i] with extra_before[i] bases to clip in front.
 * Outputs per item: ops[i * ops_stride ..] (count << 4 | code), n_ops (-1 for the "*" cigar), edit distance, addFrontClipping,
 * extraBasesClippedAfter. */
int snapref_compute_cigar_lv(void *vindex, uint32_t n, const char *data, const uint64_t *off, const int32_t *len, const int64_t *loc,
                             const int32_t *extra_before, int use_m, uint32_t *ops, uint32_t ops_stride, int32_t *n_ops,
                             int32_t *edit_distance, int32_t *add_front_clipping, int64_t *extra_after)
{
    GenomeIndex *index = (GenomeIndex *)vindex;
    const Genome *genome = index->getGenome();
    static LandauVishkinWithCigar *lvc = NULL;
    if (lvc == NULL) lvc = new LandauVishkinWithCigar();
    for (uint32_t i = 0; i < n; i++) {
        char *buf = (char *)(ops + (size_t)i * ops_stride);
        int used = 0, afc = 0, ed = 0;
        GenomeDistance after = 0;
        buf[0] = 0;
        SAMFormat::computeCigar(BAM_CIGAR_OPS, genome, lvc, buf, (int)(ops_stride * 4), data + off[i], (GenomeDistance)len[i], 0,
                                (GenomeDistance)extra_before[i], 0, &after, GenomeLocation(loc[i]), use_m != 0, &ed, &used, &afc);
        n_ops[i] = (used == 0 && buf[0] == '*') ? -1 : used / 4;
        edit_distance[i] = ed; add_front_clipping[i] = afc; extra_after[i] = (int64_t)after;
    }
    return 0;
}

/* AlignmentAdjuster::AdjustAlignment (AlignmentAdjuster.cpp:33-190) for a batch: result i is (status, direction, location, score) of read
 * data[off[i] .. off[i] + len[i]) -- a Read without clipping of its own, as AlignRead's caller in this driver makes them. */
int snapref_adjust_alignments(void *vindex, uint32_t n, const char *data, const uint64_t *off, const int32_t *len, snapgpu_single_result *results)
{
    GenomeIndex *index = (GenomeIndex *)vindex;
    AlignmentAdjuster adjuster(index->getGenome());
    std::vector<char> bbuf(MAX_READ_LENGTH + 2 * SLACK, 0), qbuf(MAX_READ_LENGTH + 2 * SLACK, 'I');
    for (uint32_t i = 0; i < n; i++) {
        memcpy(&bbuf[SLACK], data + off[i], (size_t)len[i]);
        Read read;
        read.init("r", 1, &bbuf[SLACK], &qbuf[SLACK], (unsigned)len[i], NULL, 0);
        SingleAlignmentResult r;
        memset(&r, 0, sizeof(r));
        r.status = (AlignmentResult)results[i].status;
        r.direction = (Direction)results[i].direction;
        r.location = GenomeLocation(results[i].location);
        r.score = results[i].score;
        r.clippingForReadAdjustment = 0;
        adjuster.AdjustAlignment(&read, &r);
        results[i].status = (int32_t)r.status;
        results[i].location = abi_location(r.location);
        results[i].score = r.score;
        results[i].clipping_for_read_adjustment = r.clippingForReadAdjustment;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------- secondary results (-om)

struct SecJob {
    GenomeIndex *index;
    const snapgpu_params *p;
    int om, mpc; _int64 omax;
    uint32_t n;
    const char *bases, *quals;
    const uint64_t *offsets;
    snapgpu_single_result *primary, *first_alt, *secondary;
    uint32_t stride;
    uint32_t *n_secondary;
    volatile _int64 next;
    uint32_t chunk;
};

static void *align_secondary_thread(void *arg)
{
    SecJob *job = (SecJob *)arg;
    const snapgpu_params *p = job->p;
    GenomeIndex *index = job->index;
    int maxReadSize = MAX_READ_LENGTH;
    // SingleAligner.cpp:145-173 with -om / -mpc set
    const bool fresh = g_fresh_objects != 0;
    const size_t reservation = BaseAligner::getBigAllocatorReservation(index, true, p->max_hits, maxReadSize, index->getSeedLength(),
                                                p->num_seeds, p->seed_coverage, job->mpc, p->extra_search_depth) + 4096;
    ZeroedArena *arena = fresh ? new ZeroedArena(reservation + (1 << 20)) : NULL;
    BigAllocator *allocator = fresh ? (BigAllocator *)arena : new BigAllocator(reservation, 16);
#define NEW_SEC_ALIGNER() with_flags(new (allocator) BaseAligner( \
        index, p->max_hits, p->max_k, maxReadSize, p->num_seeds, p->seed_coverage, p->min_weight_to_check, \
        p->extra_search_depth, DisabledOptimizations(), p->use_affine_gap != 0, \
        g_adjust_alignments == 0 /* ignoreAlignmentAdjustmentsForOm */, p->alt_awareness != 0, p->emit_alt_alignments != 0, \
        p->max_score_gap_to_prefer_non_alt, job->mpc, NULL, NULL, \
        p->match_reward, p->sub_penalty, p->gap_open_penalty, p->gap_extend_penalty, \
        p->five_prime_end_bonus, p->three_prime_end_bonus, NULL, allocator))
    BaseAligner *aligner = NEW_SEC_ALIGNER();
    std::vector<char> bbuf(MAX_READ_LENGTH + 2 * SLACK, 0), qbuf(MAX_READ_LENGTH + 2 * SLACK, 0);
    // the result buffer of SingleAligner.cpp:176-190: [0] primary, [1..] secondary; doubled when AlignRead says it is too small
    _int64 bufCount = 32;
    std::vector<SingleAlignmentResult> buf(bufCount);

    for (;;) {
        _int64 begin = __sync_fetch_and_add(&job->next, (_int64)job->chunk);
        if (begin >= (_int64)job->n) break;
        _int64 end = begin + job->chunk;
        if (end > (_int64)job->n) end = job->n;
        for (_int64 i = begin; i < end; i++) {
            unsigned len = (unsigned)(job->offsets[i + 1] - job->offsets[i]);
            memcpy(&bbuf[SLACK], job->bases + job->offsets[i], len);
            memcpy(&qbuf[SLACK], job->quals + job->offsets[i], len);
            Read read;
            read.init("r", 1, &bbuf[SLACK], &qbuf[SLACK], len, NULL, 0);
            SingleAlignmentResult alt;
            memset(&alt, 0, sizeof(alt));
            alt.status = NotFound;
            _int64 nSecondary = 0;
            for (;;) {
                memset(&buf[0], 0, sizeof(SingleAlignmentResult) * bufCount);
                if (fresh) { aligner->~BaseAligner(); arena->reset(); aligner = NEW_SEC_ALIGNER(); }
                if (aligner->AlignRead(&read, &buf[0], &alt, job->om, bufCount - 1, &nSecondary, job->omax, &buf[1], 0, NULL, NULL)) break;
                bufCount *= 2;                                  // SingleAligner.cpp:250-263
                buf.resize(bufCount);
            }
            fill_result(&job->primary[i], &buf[0]);
            if (job->first_alt) {
                if (alt.status == NotFound) {
                    memset(&job->first_alt[i], 0, sizeof(job->first_alt[i]));
                    job->first_alt[i].status = NotFound;
                } else {
                    fill_result(&job->first_alt[i], &alt);
                }
            }
            job->n_secondary[i] = (uint32_t)nSecondary;
            for (_int64 k = 0; k < nSecondary && k < (_int64)job->stride; k++) {
                snapgpu_single_result *o = &job->secondary[(size_t)i * job->stride + k];
                fill_result(o, &buf[1 + k]);
                o->probability_all_candidates = 0; o->popular_seeds_skipped = 0;   // never written for a secondary result (BaseAligner.cpp:2182-2199)
            }
        }
    }
    aligner->~BaseAligner();
    if (fresh) delete arena; else delete allocator;
#undef NEW_SEC_ALIGNER
    return NULL;
}

/* BaseAligner::AlignRead with -om `om`, -omax `omax`, -mpc `mpc` (-1 = none), called as SingleAligner.cpp:250 calls it.
 * secondary: [n * stride]; n_secondary[i] = count for read i (entries beyond stride are dropped). */
int snapref_align_single_secondary(void *vindex, const snapgpu_params *p, int om, int64_t omax, int mpc, uint32_t n,
                                   const char *bases, const char *quals, const uint64_t *offsets, int n_threads,
                                   snapgpu_single_result *primary, snapgpu_single_result *first_alt,
                                   snapgpu_single_result *secondary, uint32_t stride, uint32_t *n_secondary)
{
    snapref_init();
    GenomeIndex *index = (GenomeIndex *)vindex;
    /* (an index with 5 .. 8-byte locations: the reference aligners take their 64-bit branches; results are locations either way) */
    g_index = index;                                            // SingleAlignmentResult::compareByContigAndScore reads it (AlignmentResult.cpp:31)
    SecJob job;
    job.index = index; job.p = p; job.om = om; job.omax = omax; job.mpc = mpc; job.n = n; job.bases = bases; job.quals = quals;
    job.offsets = offsets; job.primary = primary; job.first_alt = first_alt; job.secondary = secondary; job.stride = stride;
    job.n_secondary = n_secondary; job.next = 0; job.chunk = 256;
    if (n_threads < 1) n_threads = 1;
    std::vector<pthread_t> th(n_threads);
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, align_secondary_thread, &job);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    return 0;
}

// ---------------------------------------------------------------------------------------- paired end

static void fill_paired(snapgpu_paired_result *o, const PairedAlignmentResult *r)
{
    memset(o, 0, sizeof(*o));
    for (int i = 0; i < 2; i++) {
        o->status[i] = (int32_t)r->status[i];
        o->direction[i] = (int32_t)r->direction[i];
        o->location[i] = abi_location(r->location[i]);
        o->orig_location[i] = (int64_t)GenomeLocationAsInt64(r->origLocation[i]);
        o->score[i] = r->score[i];
        o->score_prior_to_clipping[i] = r->scorePriorToClipping[i];
        o->mapq[i] = r->mapq[i];
        o->clipping_for_read_adjustment[i] = r->clippingForReadAdjustment[i];
        o->used_affine_gap_scoring[i] = r->usedAffineGapScoring[i] ? 1 : 0;
        o->bases_clipped_before[i] = r->basesClippedBefore[i];
        o->bases_clipped_after[i] = r->basesClippedAfter[i];
        o->ag_score[i] = r->agScore[i];
        o->supplementary[i] = r->supplementary[i] ? 1 : 0;
        o->seed_offset[i] = r->seedOffset[i];
        o->lv_indels[i] = r->lvIndels[i];
        o->match_probability[i] = r->matchProbability[i];
        o->popular_seeds_skipped[i] = r->popularSeedsSkipped[i];
        o->used_gapless_clipping[i] = r->usedGaplessClipping[i] ? 1 : 0;
        o->ref_span[i] = r->refSpan[i];
        o->liftover[i] = r->liftover[i] ? 1 : 0;
    }
    o->probability_all_pairs = r->probabilityAllPairs;
    o->aligned_as_pair = r->alignedAsPair ? 1 : 0;
    o->ag_forced_single_aligner_call = r->agForcedSingleAlignerCall ? 1 : 0;
}

struct PairedJob {
    GenomeIndex *index;
    const snapgpu_params *p;
    const snapgpu_paired_params *pp;
    int stage;                 // 0: ChimericPairedEndAligner::align; 1: IntersectingPairedEndAligner::align only
    uint32_t n;                // pairs
    const char *bases;
    const char *quals;
    const uint64_t *offsets;   // [2n+1]
    snapgpu_paired_result *primary;
    snapgpu_paired_result *first_alt;
    volatile _int64 next;
    uint32_t chunk;
    pthread_mutex_t lock;
    _int64 lv, ag;
    // secondary results (-om / -omax / -mpc); om = -1: none, as PairedAligner.cpp does by default
    int om, mpc; _int64 omax;
    snapgpu_paired_result *secondary; uint32_t sec_stride; uint32_t *n_secondary;                  // [n * sec_stride], [n]
    snapgpu_single_result *single_secondary; uint32_t ssec_stride; uint32_t *n_single_secondary;   // [n * ssec_stride], [2n]
};

static void *paired_thread(void *arg)
{
    PairedJob *job = (PairedJob *)arg;
    const snapgpu_params *p = job->p;
    const snapgpu_paired_params *pp = job->pp;
    GenomeIndex *index = job->index;
    int maxReadSize = MAX_READ_LENGTH;
    const int maxSecondaryAlignmentsPerContig = job->mpc;

    // mirror of PairedAligner.cpp:556-640
    size_t memoryPoolSize = IntersectingPairedEndAligner::getBigAllocatorReservation(index, pp->max_big_hits, maxReadSize, index->getSeedLength(),
        pp->num_seeds, pp->seed_coverage, MAX_K, p->extra_search_depth, pp->max_candidate_pool_size, maxSecondaryAlignmentsPerContig);
    memoryPoolSize += ChimericPairedEndAligner::getBigAllocatorReservation(index, maxReadSize, p->max_hits, index->getSeedLength(), pp->max_single_seeds,
        p->seed_coverage, MAX_K, p->extra_search_depth, pp->max_candidate_pool_size, maxSecondaryAlignmentsPerContig);
    _int64 maxPairedCand = p->use_affine_gap ? 4096 : 0, maxSingleCand = p->use_affine_gap ? 4096 : 0;
    memoryPoolSize += 4096;
    const bool fresh = g_fresh_objects != 0;
    ZeroedArena *arena = fresh ? new ZeroedArena(memoryPoolSize + (4 << 20)) : NULL;
    BigAllocator *allocator = fresh ? (BigAllocator *)arena : new BigAllocator(memoryPoolSize, 16);

#define NEW_INTERSECTING() new (allocator) IntersectingPairedEndAligner(index, maxReadSize, p->max_hits, p->max_k, \
        pp->max_k_for_indels, pp->num_seeds, pp->seed_coverage, pp->min_spacing, pp->max_spacing, pp->max_big_hits, p->extra_search_depth, \
        pp->max_candidate_pool_size, maxSecondaryAlignmentsPerContig, allocator, DisabledOptimizations(), p->use_affine_gap != 0, \
        true /* ignoreAlignmentAdjustmentForOm */, p->alt_awareness != 0, p->max_score_gap_to_prefer_non_alt, \
        p->match_reward, p->sub_penalty, p->gap_open_penalty, p->gap_extend_penalty, pp->use_soft_clipping != 0)
#define NEW_CHIMERIC() new (allocator) ChimericPairedEndAligner(index, maxReadSize, p->max_hits, p->max_k, pp->max_single_seeds, \
        p->seed_coverage, p->min_weight_to_check, pp->force_spacing != 0, p->extra_search_depth, DisabledOptimizations(), p->use_affine_gap != 0, \
        true, p->alt_awareness != 0, p->emit_alt_alignments != 0, intersectingAligner, pp->min_read_length, maxSecondaryAlignmentsPerContig, \
        p->max_score_gap_to_prefer_non_alt, pp->flatten_mapq_at_or_below, pp->use_soft_clipping != 0, p->match_reward, p->sub_penalty, \
        p->gap_open_penalty, p->gap_extend_penalty, p->five_prime_end_bonus, p->three_prime_end_bonus, pp->min_score_realignment, \
        pp->min_score_gap_realignment_alt, pp->min_ag_score_improvement, pp->enable_hamming_scoring_base_aligner != 0, allocator)
    IntersectingPairedEndAligner *intersectingAligner = NEW_INTERSECTING();
    ChimericPairedEndAligner *aligner = NEW_CHIMERIC();
    _int64 acc_lv = 0, acc_ag = 0;

    // PairedAligner.cpp:556-566, 600-640: results[0] is the primary, results + 1 the paired secondary buffer
    _int64 maxPairedSecondaryHits = job->om < 0 ? 0 : 32, maxSingleSecondaryHits = job->om < 0 ? 0 : 32;
    PairedAlignmentResult *results = (PairedAlignmentResult *)BigAlloc((maxPairedSecondaryHits + 1) * sizeof(PairedAlignmentResult));
    SingleAlignmentResult *singleSecondary = maxSingleSecondaryHits ? (SingleAlignmentResult *)BigAlloc(maxSingleSecondaryHits * sizeof(SingleAlignmentResult)) : NULL;
    PairedAlignmentResult *pairedCand = maxPairedCand ? (PairedAlignmentResult *)BigAlloc(maxPairedCand * sizeof(PairedAlignmentResult)) : NULL;
    SingleAlignmentResult *singleCand = maxSingleCand ? (SingleAlignmentResult *)BigAlloc(maxSingleCand * sizeof(SingleAlignmentResult)) : NULL;

    std::vector<char> bbuf[2], qbuf[2];
    for (int r = 0; r < 2; r++) { bbuf[r].assign(MAX_READ_LENGTH + 2 * SLACK, 0); qbuf[r].assign(MAX_READ_LENGTH + 2 * SLACK, 0); }

    for (;;) {
        _int64 begin = __sync_fetch_and_add(&job->next, (_int64)job->chunk);
        if (begin >= (_int64)job->n) break;
        _int64 end = begin + job->chunk;
        if (end > (_int64)job->n) end = job->n;
        for (_int64 i = begin; i < end; i++) {
            Read reads[2];
            for (int r = 0; r < 2; r++) {
                unsigned len = (unsigned)(job->offsets[2 * i + r + 1] - job->offsets[2 * i + r]);
                memcpy(&bbuf[r][SLACK], job->bases + job->offsets[2 * i + r], len);
                memcpy(&qbuf[r][SLACK], job->quals + job->offsets[2 * i + r], len);
                reads[r].init("r", 1, &bbuf[r][SLACK], &qbuf[r][SLACK], len, NULL, 0);
            }
            PairedAlignmentResult alt;
            memset(&alt, 0, sizeof(alt));
            _int64 nSecondary = 0, nPairedCand = 0, nSingleSecondary[2] = {0, 0}, nSingleCand[2] = {0, 0};
            for (;;) {
                bool ok;
                if (fresh) {                                // newly constructed aligners in zero-filled memory for every call
                    acc_lv += aligner->getLocationsScoredWithLandauVishkin(); acc_ag += aligner->getLocationsScoredWithAffineGap();
                    aligner->~ChimericPairedEndAligner(); intersectingAligner->~IntersectingPairedEndAligner();
                    arena->reset();
                    intersectingAligner = NEW_INTERSECTING(); aligner = NEW_CHIMERIC();
                }
                memset(results, 0, (maxPairedSecondaryHits + 1) * sizeof(PairedAlignmentResult));
                if (singleSecondary) memset(singleSecondary, 0, maxSingleSecondaryHits * sizeof(SingleAlignmentResult));
                if (job->stage == 1) {
                    ok = intersectingAligner->align(&reads[0], &reads[1], results, &alt, job->om, maxPairedSecondaryHits, &nSecondary, results + 1,
                        maxSingleSecondaryHits, job->omax, &nSingleSecondary[0], &nSingleSecondary[1], singleSecondary, maxPairedCand, &nPairedCand,
                        pairedCand, maxSingleCand, &nSingleCand[0], &nSingleCand[1], singleCand, (int)p->max_k);
                } else {
                    // same call shape as PairedAligner.cpp:727
                    ok = aligner->align(&reads[0], &reads[1], results, &alt, job->om, maxPairedSecondaryHits, &nSecondary, results + 1,
                        maxSingleSecondaryHits, job->omax, &nSingleSecondary[0], &nSingleSecondary[1], singleSecondary, maxPairedCand, &nPairedCand,
                        pairedCand, maxSingleCand, &nSingleCand[0], &nSingleCand[1], singleCand, (int)p->max_k);
                }
                if (ok) break;
                // PairedAligner.cpp:732-781: double whichever buffer overflowed and call again
                if (nSecondary > maxPairedSecondaryHits) {
                    BigDealloc(results); maxPairedSecondaryHits *= 2;
                    results = (PairedAlignmentResult *)BigAlloc((maxPairedSecondaryHits + 1) * sizeof(PairedAlignmentResult));
                } else if (nSingleSecondary[0] > maxSingleSecondaryHits) {
                    BigDealloc(singleSecondary); maxSingleSecondaryHits *= 2;
                    singleSecondary = (SingleAlignmentResult *)BigAlloc(maxSingleSecondaryHits * sizeof(SingleAlignmentResult));
                } else if (nPairedCand > maxPairedCand) {
                    BigDealloc(pairedCand); maxPairedCand *= 2;
                    pairedCand = (PairedAlignmentResult *)BigAlloc(maxPairedCand * sizeof(PairedAlignmentResult));
                } else if (nSingleCand[0] > maxSingleCand) {
                    BigDealloc(singleCand); maxSingleCand *= 2;
                    singleCand = (SingleAlignmentResult *)BigAlloc(maxSingleCand * sizeof(SingleAlignmentResult));
                } else {
                    break;
                }
            }
            fill_paired(&job->primary[i], &results[0]);
            if (job->first_alt) fill_paired(&job->first_alt[i], &alt);
            if (job->n_secondary) {
                job->n_secondary[i] = (uint32_t)nSecondary;
                for (_int64 k = 0; k < nSecondary && k < (_int64)job->sec_stride; k++)
                    fill_paired(&job->secondary[(size_t)i * job->sec_stride + k], &results[1 + k]);
                job->n_single_secondary[2 * i] = (uint32_t)nSingleSecondary[0];
                job->n_single_secondary[2 * i + 1] = (uint32_t)nSingleSecondary[1];
                for (_int64 k = 0; k < nSingleSecondary[0] + nSingleSecondary[1] && k < (_int64)job->ssec_stride; k++) {
                    snapgpu_single_result *o = &job->single_secondary[(size_t)i * job->ssec_stride + k];
                    fill_result(o, &singleSecondary[k]);
                    o->probability_all_candidates = 0; o->popular_seeds_skipped = 0;   // never written for a secondary result
                }
            }
        }
    }

    pthread_mutex_lock(&job->lock);
    job->lv += acc_lv + aligner->getLocationsScoredWithLandauVishkin();
    job->ag += acc_ag + aligner->getLocationsScoredWithAffineGap();
    pthread_mutex_unlock(&job->lock);

    if (pairedCand) BigDealloc(pairedCand);
    if (singleCand) BigDealloc(singleCand);
    BigDealloc(results);
    if (singleSecondary) BigDealloc(singleSecondary);
    aligner->~ChimericPairedEndAligner();
    intersectingAligner->~IntersectingPairedEndAligner();
    if (fresh) delete arena; else delete allocator;
#undef NEW_INTERSECTING
#undef NEW_CHIMERIC
    return NULL;
}

/* offsets: [2n+1]; read r of pair i is bases[offsets[2i+r] .. offsets[2i+r+1]).
 * counters2 = {LV locations, AG locations} (paired + single-end fallback).
 * Secondary results (om >= 0): secondary[i * sec_stride + k] for k < n_secondary[i] (paired), single_secondary[i * ssec_stride + k]:
 * read 0's n_single_secondary[2i] results, then read 1's n_single_secondary[2i+1] (the layout of PairedAligner.cpp:872).          */
static int run_paired(void *vindex, const snapgpu_params *p, const snapgpu_paired_params *pp, int stage, uint32_t n,
                      const char *bases, const char *quals, const uint64_t *offsets, int n_threads,
                      snapgpu_paired_result *primary, snapgpu_paired_result *first_alt, int64_t *counters2, double *seconds,
                      int om, int64_t omax, int mpc, snapgpu_paired_result *secondary, uint32_t sec_stride, uint32_t *n_secondary,
                      snapgpu_single_result *single_secondary, uint32_t ssec_stride, uint32_t *n_single_secondary)
{
    snapref_init();
    GenomeIndex *index = (GenomeIndex *)vindex;
    /* (an index with 5 .. 8-byte locations: the reference aligners take their 64-bit branches; results are locations either way) */
    g_index = index;                                   // AlignerContext.cpp:253 (used by compareByContigAndScore only)
    PairedJob job;
    job.index = index; job.p = p; job.pp = pp; job.stage = stage; job.n = n; job.bases = bases; job.quals = quals; job.offsets = offsets;
    job.primary = primary; job.first_alt = first_alt; job.next = 0; job.chunk = 64; job.lv = job.ag = 0;
    job.om = om; job.omax = omax; job.mpc = mpc; job.secondary = secondary; job.sec_stride = sec_stride; job.n_secondary = n_secondary;
    job.single_secondary = single_secondary; job.ssec_stride = ssec_stride; job.n_single_secondary = n_single_secondary;
    pthread_mutex_init(&job.lock, NULL);
    if (n_threads < 1) n_threads = 1;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    std::vector<pthread_t> th(n_threads);
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, paired_thread, &job);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds) *seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    if (counters2) { counters2[0] = job.lv; counters2[1] = job.ag; }
    pthread_mutex_destroy(&job.lock);
    return 0;
}

int snapref_align_paired(void *vindex, const snapgpu_params *p, const snapgpu_paired_params *pp, int stage, uint32_t n,
                         const char *bases, const char *quals, const uint64_t *offsets, int n_threads,
                         snapgpu_paired_result *primary, snapgpu_paired_result *first_alt,
                         int64_t *counters2, double *seconds)
{
    return run_paired(vindex, p, pp, stage, n, bases, quals, offsets, n_threads, primary, first_alt, counters2, seconds,
                      -1, 0x7fffffff, -1, NULL, 0, NULL, NULL, 0, NULL);
}

int snapref_align_paired_secondary(void *vindex, const snapgpu_params *p, const snapgpu_paired_params *pp, int stage, int om, int64_t omax, int mpc,
                                   uint32_t n, const char *bases, const char *quals, const uint64_t *offsets, int n_threads,
                                   snapgpu_paired_result *primary, snapgpu_paired_result *first_alt,
                                   snapgpu_paired_result *secondary, uint32_t sec_stride, uint32_t *n_secondary,
                                   snapgpu_single_result *single_secondary, uint32_t ssec_stride, uint32_t *n_single_secondary)
{
    return run_paired(vindex, p, pp, stage, n, bases, quals, offsets, n_threads, primary, first_alt, NULL, NULL,
                      om, omax, mpc, secondary, sec_stride, n_secondary, single_secondary, ssec_stride, n_single_secondary);
}


// The single-end aligner inside ChimericPairedEndAligner (ChimericPairedEndAligner.cpp:81-88), on its own, so that the
// host build of snap_amd/csrc/paired.h (oracle/paired_host.cpp) can plug the reference in for the fallback calls
// (ChimericPairedEndAligner.cpp:304-360): AlignRead with a per-call maxK, and the Hamming retry + alignAffineGap.
struct ChimericSingle {
    BigAllocator *allocator;
    BaseAligner *aligner;
    SingleAlignmentResult *cand;
    _int64 maxCand;
    std::vector<char> b, q;
};

void *snapref_chimeric_single_create2(void *vindex, const snapgpu_params *p, const snapgpu_paired_params *pp, int mpc);
void *snapref_chimeric_single_create(void *vindex, const snapgpu_params *p, const snapgpu_paired_params *pp)
{
    return snapref_chimeric_single_create2(vindex, p, pp, -1);
}
void *snapref_chimeric_single_create2(void *vindex, const snapgpu_params *p, const snapgpu_paired_params *pp, int mpc)
{
    snapref_init();
    GenomeIndex *index = (GenomeIndex *)vindex;
    g_index = index;                                   // SingleAlignmentResult::compareByContigAndScore reads it (-mpc)
    int maxReadSize = MAX_READ_LENGTH;
    ChimericSingle *c = new ChimericSingle();
    c->allocator = new BigAllocator(BaseAligner::getBigAllocatorReservation(index, true, p->max_hits, maxReadSize, index->getSeedLength(),
                                    pp->max_single_seeds, p->seed_coverage, mpc, p->extra_search_depth) + 4096, 16);
    c->aligner = new (c->allocator) BaseAligner(index, p->max_hits, p->max_k / 2, maxReadSize, pp->max_single_seeds, p->seed_coverage,
        p->min_weight_to_check, p->extra_search_depth, DisabledOptimizations(), p->use_affine_gap != 0, true, p->alt_awareness != 0,
        p->emit_alt_alignments != 0, p->max_score_gap_to_prefer_non_alt, mpc, NULL, NULL, p->match_reward, p->sub_penalty,
        p->gap_open_penalty, p->gap_extend_penalty, p->five_prime_end_bonus, p->three_prime_end_bonus, NULL, c->allocator);
    c->maxCand = p->use_affine_gap ? 4096 : 0;                        // PairedAligner.cpp:570-577: no candidate buffers without affine gap
    c->cand = c->maxCand ? (SingleAlignmentResult *)BigAlloc(c->maxCand * sizeof(SingleAlignmentResult)) : NULL;
    c->b.assign(MAX_READ_LENGTH + 2 * SLACK, 0);
    c->q.assign(MAX_READ_LENGTH + 2 * SLACK, 0);
    return c;
}

int snapref_chimeric_single_align2(void *h, int max_k, int hamming, const char *bases, const char *quals, uint32_t len,
                                   snapgpu_single_result *res, snapgpu_single_result *alt,
                                   int om, int64_t omax, snapgpu_single_result *sec_out, uint32_t sec_room, uint32_t *n_sec,
                                   uint32_t first_room, uint32_t *overflowed_first);
int snapref_chimeric_single_align(void *h, int max_k, int hamming, const char *bases, const char *quals, uint32_t len,
                                  snapgpu_single_result *res, snapgpu_single_result *alt)
{
    return snapref_chimeric_single_align2(h, max_k, hamming, bases, quals, len, res, alt, -1, 0x7fffffff, NULL, 0, NULL, 32, NULL);
}
/* ... with secondary results: AlignRead(read, ..., om, bufSize, &n, omax, buf, ...) as ChimericPairedEndAligner.cpp:310 calls it.
 * first_room: the buffer size of the first attempt (the caller's remaining room); *overflowed_first says whether that was too small. */
int snapref_chimeric_single_align2(void *h, int max_k, int hamming, const char *bases, const char *quals, uint32_t len,
                                   snapgpu_single_result *res, snapgpu_single_result *alt,
                                   int om, int64_t omax, snapgpu_single_result *sec_out, uint32_t sec_room, uint32_t *n_sec,
                                   uint32_t first_room, uint32_t *overflowed_first)
{
    ChimericSingle *c = (ChimericSingle *)h;
    memcpy(&c->b[SLACK], bases, len);
    memcpy(&c->q[SLACK], quals, len);
    Read read;
    read.init("r", 1, &c->b[SLACK], &c->q[SLACK], len, NULL, 0);
    SingleAlignmentResult r, a;
    memset(&r, 0, sizeof(r));
    memset(&a, 0, sizeof(a));
    a.status = NotFound;
    c->aligner->setMaxK(max_k);
    _int64 nSecondary = 0, nCand = 0;
    std::vector<SingleAlignmentResult> secbuf(om >= 0 ? (first_room > 64 ? first_room : 64) : 0);
    _int64 secSize = om >= 0 ? (_int64)first_room : 0;
    if (overflowed_first) *overflowed_first = 0;
    for (;;) {
        nCand = 0;
        if (!secbuf.empty()) memset(&secbuf[0], 0, secbuf.size() * sizeof(SingleAlignmentResult));
        bool ok = c->aligner->AlignRead(&read, &r, &a, om, secSize, &nSecondary, omax, secbuf.empty() ? NULL : &secbuf[0],
                                        c->maxCand, &nCand, c->cand, hamming != 0);
        if (ok) break;
        if (om >= 0 && !(c->cand != NULL && nCand > c->maxCand)) {       // the secondary buffer overflowed: the caller doubles it (PairedAligner.cpp:746-756)
            if (overflowed_first && secSize == (_int64)first_room) *overflowed_first = 1;
            secSize = secSize < 32 ? 64 : secSize * 2;
            if ((_int64)secbuf.size() < secSize) secbuf.resize(secSize);
            continue;
        }
        if (c->cand != NULL && nCand > c->maxCand) {
            BigDealloc(c->cand);
            c->maxCand *= 2;
            c->cand = (SingleAlignmentResult *)BigAlloc(c->maxCand * sizeof(SingleAlignmentResult));
        } else {
            break;
        }
    }
    if (hamming) {
        c->aligner->alignAffineGap(&read, &r, &a, nCand, c->cand);
    }
    fill_result(res, &r);
    if (a.status == NotFound) { memset(alt, 0, sizeof(*alt)); alt->status = NotFound; } else fill_result(alt, &a);
    if (n_sec) {
        *n_sec = (uint32_t)nSecondary;
        for (_int64 k = 0; k < nSecondary && k < (_int64)sec_room; k++) {
            fill_result(&sec_out[k], &secbuf[k]);
            sec_out[k].probability_all_candidates = 0; sec_out[k].popular_seeds_skipped = 0;
        }
    }
    return 0;
}

void snapref_chimeric_single_destroy(void *h)
{
    ChimericSingle *c = (ChimericSingle *)h;
    if (c->cand) BigDealloc(c->cand);
    c->aligner->~BaseAligner();
    delete c->allocator;
    delete c;
}

} // extern "C"
