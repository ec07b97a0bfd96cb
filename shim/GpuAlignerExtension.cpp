/*
 * GpuAlignerExtension.cpp -- the reference-side binding of INTEGRATION.md, for real.
 *
 * Compiled TOGETHER WITH the reference (it includes SNAPLib headers; it contains no reference
 * code) and linked against snap_amd/libsnapgpu.so, this gives `snap-aligner-gpu`: SNAP's own
 * command line, FASTQ readers, filters, SAM/BAM writers and statistics, with the per-thread
 * BaseAligner::AlignRead loop (SNAPLib/SingleAligner.cpp:197-330) replaced by batches sent
 * across the C ABI of include/snapgpu.h to the HIP kernels.  It plugs in through the
 * reference's own hook, AlignerExtension::runIterationThread (SNAPLib/AlignerContext.h:165,
 * called at SingleAligner.cpp:102), so no SNAPLib source is modified.
 *
 * Built by oracle/Makefile (target `ref`) into oracle/_ref/snap-aligner-gpu, because the
 * resulting binary contains the reference's objects.  Usage is SNAP's:
 *     snap-aligner-gpu single <index-dir> reads.fq -o out.sam [-d 8 ...]
 * Written in C++98 like the reference.
 */
#include "stdafx.h"
#include "Compat.h"
#include "AlignerContext.h"
#include "AlignerOptions.h"
#include "AlignerStats.h"
#include "SingleAligner.h"
#include "PairedAligner.h"
#include "GenomeIndex.h"
#include "SeedSequencer.h"
#include "CommandProcessor.h"
#include "Read.h"
#include "Error.h"
#include "exit.h"

#include <vector>
#include <new>
#include <pthread.h>
#include <string.h>

#include "../include/snapgpu.h"

extern const char *SNAP_VERSION;                                   // SNAPLib/CommandProcessor.cpp:39

static pthread_mutex_t g_gpuLock = PTHREAD_MUTEX_INITIALIZER;      // one context, calls serialised
static snapgpu_ctx *g_ctx = NULL;

static void toSnap(const snapgpu_single_result &g, SingleAlignmentResult *r)
{
    r->status = (AlignmentResult)g.status;
    r->direction = g.direction;
    r->location = GenomeLocation(g.location);
    r->origLocation = GenomeLocation(g.orig_location);
    r->score = g.score;
    r->scorePriorToClipping = g.score_prior_to_clipping;
    r->mapq = g.mapq;
    r->clippingForReadAdjustment = g.clipping_for_read_adjustment;
    r->usedAffineGapScoring = g.used_affine_gap_scoring != 0;
    r->basesClippedBefore = g.bases_clipped_before;
    r->basesClippedAfter = g.bases_clipped_after;
    r->agScore = g.ag_score;
    r->supplementary = g.supplementary != 0;
    r->seedOffset = g.seed_offset;
    r->matchProbability = g.match_probability;
    r->probabilityAllCandidates = g.probability_all_candidates;
    r->popularSeedsSkipped = g.popular_seeds_skipped;
    r->alignmentTimeInNanoseconds = 0;
}

class GpuAlignerExtension : public AlignerExtension {
public:
    GpuAlignerExtension() {}

    // The base copy() returns a plain AlignerExtension and is invoked once per worker thread
    // (AlignerContext.cpp:225): without this override the hook would never run in the workers.
    virtual AlignerExtension *copy() { return new GpuAlignerExtension(); }

    virtual bool runIterationThread(PairedReadSupplier *supplier, AlignerContext *c) { return false; }   // paired: reference path

    virtual bool runIterationThread(ReadSupplier *supplier, AlignerContext *c)
    {
        if (c->index == NULL) {
            return false;                                   // I/O-only mode (SingleAligner.cpp:106-131): leave it to SNAP
        }
        if (c->maxSecondaryAlignmentAdditionalEditDistance >= 0 || c->options->stopOnFirstHit || c->options->explorePopularSeeds ||
            !c->ignoreAlignmentAdjustmentForOm || c->index->doesGenomeIndexHave64BitLocations()) {
            WriteErrorMessage("snap-aligner-gpu: option outside what libsnapgpu implements (-om, -f, -x, -sa or a 64-bit index)\n");
            soft_exit(1);
        }
        ensureContext(c);

        // Reads are only valid until the supplier moves on, so each batch is copied.  ReadWithOwnMemory
        // points into its own body and has no copy-assignment: construct in place in raw storage.
        const unsigned BATCH = 16384;
        ReadWithOwnMemory *reads = (ReadWithOwnMemory *)BigAlloc((size_t)BATCH * sizeof(ReadWithOwnMemory));
        std::vector<char> bases, quals;
        std::vector<uint64_t> offs;
        std::vector<snapgpu_single_result> prim(BATCH), alt(BATCH);
        bool more = true;
        while (more) {
            unsigned n = 0;
            bases.clear(); quals.clear(); offs.clear(); offs.push_back(0);
            Read *read;
            while (n < BATCH) {
                read = supplier->getNextRead();
                if (NULL == read) { more = false; break; }
                c->stats->totalReads++;
                // SingleAligner.cpp:213-233: too short or too many Ns -> written unaligned, counted useless
                if (read->getDataLength() < c->minReadLength || read->countOfNs() > c->maxDist) {
                    if (!c->options->passFilter(read, NotFound, true, false)) {
                        c->stats->filtered++;
                    } else {
                        if (NULL != c->readWriter) {
                            SingleAlignmentResult result;
                            result.status = NotFound; result.location = InvalidGenomeLocation; result.mapq = 0;
                            result.direction = FORWARD; result.clippingForReadAdjustment = 0; result.usedAffineGapScoring = false;
                            result.basesClippedBefore = 0; result.basesClippedAfter = 0; result.supplementary = false;
                            c->readWriter->writeReads(c->readerContext, read, &result, 1, true, c->useAffineGap);
                        }
                        c->stats->uselessReads++;
                    }
                    continue;
                }
                new (&reads[n]) ReadWithOwnMemory(*read);
                bases.insert(bases.end(), read->getData(), read->getData() + read->getDataLength());
                quals.insert(quals.end(), read->getQuality(), read->getQuality() + read->getDataLength());
                offs.push_back((uint64_t)bases.size());
                n++;
            }
            if (0 == n) continue;

            pthread_mutex_lock(&g_gpuLock);
            int rc = snapgpu_align_single(g_ctx, n, &bases[0], &quals[0], &offs[0], &prim[0], &alt[0]);
            pthread_mutex_unlock(&g_gpuLock);
            if (rc != SNAPGPU_OK) {
                WriteErrorMessage("snapgpu_align_single failed (%d): %s\n", rc, snapgpu_last_error(g_ctx));
                soft_exit(1);
            }

            for (unsigned i = 0; i < n; i++) {
                SingleAlignmentResult result, altResult;
                toSnap(prim[i], &result);
                bool containsPrimary = true;
                if (NULL != c->readWriter) {
                    // SingleAligner.cpp:300-322 with no secondary results
                    _int64 nResults = 1;
                    if (!c->options->passFilter(&reads[i], result.status, false, false)) {
                        containsPrimary = false;
                        nResults = 0;
                    }
                    c->readWriter->writeReads(c->readerContext, &reads[i], &result, nResults, containsPrimary, c->useAffineGap);
                    if (c->altAwareness && alt[i].status != SNAPGPU_NotFound) {
                        toSnap(alt[i], &altResult);
                        if (c->options->passFilter(&reads[i], altResult.status, false, false)) {
                            c->readWriter->writeReads(c->readerContext, &reads[i], &altResult, 1, false, c->useAffineGap);
                        }
                    }
                }
                if (containsPrimary) {                          // SingleAlignerContext::updateStats, SingleAligner.cpp:354-374
                    if (result.status == SingleHit) c->stats->singleHits++;
                    else if (result.status == MultipleHits) c->stats->multiHits++;
                    else c->stats->notFound++;
                    if (result.status != NotFound) c->stats->mapqHistogram[result.mapq]++;
                } else {
                    c->stats->filtered++;
                }
                reads[i].dispose();
            }
        }
        BigDealloc(reads);
        snapgpu_counters counters;
        pthread_mutex_lock(&g_gpuLock);
        if (snapgpu_get_counters(g_ctx, &counters, 1) == SNAPGPU_OK) {
            c->stats->lvCalls += (_int64)counters.n_lv_locations;
            c->stats->affineGapCalls += (_int64)counters.n_ag_locations;
        }
        pthread_mutex_unlock(&g_gpuLock);
        return true;                                            // we ran the whole per-thread loop
    }

private:
    static void ensureContext(AlignerContext *c)
    {
        pthread_mutex_lock(&g_gpuLock);
        if (NULL == g_ctx) {
            snapgpu_params p;
            snapgpu_default_params(&p);
            p.max_hits = (uint32_t)c->maxHits;
            p.max_k = c->maxDist;
            p.num_seeds = c->numSeedsFromCommandLine;
            p.seed_coverage = c->seedCoverage;
            p.min_weight_to_check = c->minWeightToCheck;
            p.extra_search_depth = c->extraSearchDepth;
            p.use_affine_gap = c->useAffineGap ? 1 : 0;
            p.match_reward = c->matchReward;
            p.sub_penalty = c->subPenalty;
            p.gap_open_penalty = c->gapOpenPenalty;
            p.gap_extend_penalty = c->gapExtendPenalty;
            p.five_prime_end_bonus = c->fivePrimeEndBonus;
            p.three_prime_end_bonus = c->threePrimeEndBonus;
            p.alt_awareness = c->altAwareness ? 1 : 0;
            p.emit_alt_alignments = c->emitALTAlignments ? 1 : 0;
            p.max_score_gap_to_prefer_non_alt = c->maxScoreGapToPreferNonALTAlignment;
            p.max_read_len = 400;                               // per-wave buffers; raise for longer reads (<= MAX_READ_LENGTH)
            const char *env = getenv("SNAPGPU_MAX_READ_LEN");
            if (env) p.max_read_len = (uint32_t)atoi(env);
            int rc = snapgpu_create_from_directory(c->options->indexDir, &p, 0, &g_ctx);
            if (rc != SNAPGPU_OK) {
                WriteErrorMessage("snapgpu_create_from_directory(%s) failed (%d): %s\n", c->options->indexDir, rc, snapgpu_last_error(NULL));
                soft_exit(1);
            }
        }
        pthread_mutex_unlock(&g_gpuLock);
    }
};

// Mirror of ProcessNonDaemonCommands (SNAPLib/CommandProcessor.cpp:59-88) with the extension installed.
int main(int argc, const char **argv)
{
    if (argc < 2 || (strcmp(argv[1], "single") != 0)) {
        fprintf(stderr, "usage: snap-aligner-gpu single <index-dir> <reads> [SNAP options]\n"
                        "       (index / paired / daemon: use the reference's snap-aligner)\n");
        return 1;
    }
    InitializeSeedSequencers();                                     // CommandProcessor.cpp:196
    unsigned nArgsConsumed = 0;
    GpuAlignerExtension *extension = new GpuAlignerExtension();
    SingleAlignerContext single(extension);                         // SingleAligner.h:36
    single.runAlignment(argc - 1, argv + 1, SNAP_VERSION, &nArgsConsumed);
    if (g_ctx) snapgpu_destroy(g_ctx);
    return 0;
}
