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
 * The paired hook (AlignerContext.h:165, called at PairedAligner.cpp:503) does the same for
 * ChimericPairedEndAligner::align (PairedAligner.cpp:664-960) with snapgpu_align_paired.
 *
 * Built by oracle/Makefile (target `ref`) into oracle/_ref/snap-aligner-gpu, because the
 * resulting binary contains the reference's objects.  Usage is SNAP's:
 *     snap-aligner-gpu single <index-dir> reads.fq -o out.sam [-d 8 ...]
 *     snap-aligner-gpu paired <index-dir> r1.fq r2.fq -o out.sam [-d 8 ...]
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

// One context per (GPU, feeder): the counterpart of the reference's one aligner object per thread over one shared GenomeIndex
// (SNAPLib/ParallelTask.h:128-138, SNAPLib/SingleAligner.cpp:197).  The first context loads the index; every further GPU gets
// same-size blobs filled by snapgpu_broadcast_index (RCCL over xGMI); every GPU gets SNAPGPU_SHIM_FEEDERS (default 2) contexts that share
// its blobs, so that the batch of one worker thread is on the GPU while another's finishes.  Worker thread t uses context t mod N; only
// workers that share a context wait for each other (g_setupLock covers set-up alone).
struct GpuSlot { snapgpu_ctx *ctx; pthread_mutex_t lock; };
static pthread_mutex_t g_setupLock = PTHREAD_MUTEX_INITIALIZER;
static GpuSlot *g_slots = NULL;
static int g_nSlots = 0;
static volatile int g_nextWorker = 0;
static bool g_pairedEnabled = false;
static bool g_secondaryEnabled = false;

// the C ABI has one "no location" value; the reference's InvalidGenomeLocation is all ones in as many bytes as the loaded index's locations have
static inline GenomeLocation toSnapLocation(int64_t l) { return l == (int64_t)SNAPGPU_InvalidGenomeLocation32 ? InvalidGenomeLocation : GenomeLocation(l); }

static void toSnapPaired(const snapgpu_paired_result &g, PairedAlignmentResult *r)
{
    memset(r, 0, sizeof(*r));
    for (int i = 0; i < NUM_READS_PER_PAIR; i++) {
        r->status[i] = (AlignmentResult)g.status[i];
        r->direction[i] = g.direction[i];
        r->location[i] = toSnapLocation(g.location[i]);
        r->origLocation[i] = toSnapLocation(g.orig_location[i]);      // (an Invalid32 must become the index's own InvalidGenomeLocation with 5 .. 8-byte locations too)
        r->score[i] = g.score[i];
        r->scorePriorToClipping[i] = g.score_prior_to_clipping[i];
        r->mapq[i] = g.mapq[i];
        r->clippingForReadAdjustment[i] = g.clipping_for_read_adjustment[i];
        r->usedAffineGapScoring[i] = g.used_affine_gap_scoring[i] != 0;
        r->basesClippedBefore[i] = g.bases_clipped_before[i];
        r->basesClippedAfter[i] = g.bases_clipped_after[i];
        r->agScore[i] = g.ag_score[i];
        r->supplementary[i] = g.supplementary[i] != 0;
        r->seedOffset[i] = g.seed_offset[i];
        r->lvIndels[i] = g.lv_indels[i];
        r->matchProbability[i] = g.match_probability[i];
        r->popularSeedsSkipped[i] = g.popular_seeds_skipped[i];
        r->usedGaplessClipping[i] = g.used_gapless_clipping[i] != 0;
        r->refSpan[i] = g.ref_span[i];
        r->liftover[i] = false;
    }
    r->probabilityAllPairs = g.probability_all_pairs;
    r->alignedAsPair = g.aligned_as_pair != 0;
    r->agForcedSingleAlignerCall = g.ag_forced_single_aligner_call != 0;
}

static void toSnap(const snapgpu_single_result &g, SingleAlignmentResult *r)
{
    r->status = (AlignmentResult)g.status;
    r->direction = g.direction;
    r->location = toSnapLocation(g.location);
    r->origLocation = toSnapLocation(g.orig_location);
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
    GpuAlignerExtension() : slot_(NULL) {}

    // The base copy() returns a plain AlignerExtension and is invoked once per worker thread
    // (AlignerContext.cpp:225): without this override the hook would never run in the workers.
    virtual AlignerExtension *copy() { return new GpuAlignerExtension(); }

    virtual bool runIterationThread(PairedReadSupplier *supplier, AlignerContext *c)
    {
        if (c->index == NULL) {
            return false;                                   // I/O-only mode: leave it to SNAP
        }
        PairedAlignerOptions *po = (PairedAlignerOptions *)c->options;
        if (!c->ignoreAlignmentAdjustmentForOm || po->inferSpacing) {      // (-f / -x: accepted and without effect, as in the reference's paired-end aligners)
            WriteErrorMessage("snap-aligner-gpu: option outside what libsnapgpu implements for `paired` (-ae, -ins)\n");
            soft_exit(1);
        }
        ensureContext(c, 25);
        ensurePaired(c, po);
        const bool secondary = c->maxSecondaryAlignmentAdditionalEditDistance >= 0;      // -om
        if (secondary) ensureSecondary(c);
        GpuSlot *slot = mySlot();
        uint32_t secStride = 8, singleStride = 16;
        std::vector<snapgpu_paired_result> sec;
        std::vector<snapgpu_single_result> ssec;
        std::vector<uint32_t> nSec, nSingleSec;
        std::vector<PairedAlignmentResult> results;
        std::vector<SingleAlignmentResult> singles;

        const unsigned BATCH = 8192;                        // pairs
        ReadWithOwnMemory *reads = (ReadWithOwnMemory *)BigAlloc((size_t)2 * BATCH * sizeof(ReadWithOwnMemory));
        bool *useful = new bool[2 * BATCH];
        std::vector<char> bases, quals;
        std::vector<uint64_t> offs;
        std::vector<snapgpu_paired_result> prim(BATCH), alt(BATCH);
        _int64 nSingleResults[2] = {0, 0};
        bool more = true;
        while (more) {
            unsigned n = 0;
            bases.clear(); quals.clear(); offs.clear(); offs.push_back(0);
            Read *pr[NUM_READS_PER_PAIR];
            while (n < BATCH) {
                if (!supplier->getNextReadPair(&pr[0], &pr[1])) { more = false; break; }
                if (!c->options->ignoreMismatchedIDs) {
                    Read::checkIdMatch(pr[0], pr[1]);
                }
                c->stats->totalReads += 2;
                // PairedAligner.cpp:681-710: both reads too short / too many Ns -> written unaligned, counted useless
                bool useful0 = pr[0]->getDataLength() >= c->minReadLength && (int)pr[0]->countOfNs() <= (int)c->maxDist;
                bool useful1 = pr[1]->getDataLength() >= c->minReadLength && (int)pr[1]->countOfNs() <= (int)c->maxDist;
                if (!useful0 && !useful1) {
                    PairedAlignmentResult result;
                    memset(&result, 0, sizeof(result));
                    result.status[0] = result.status[1] = NotFound;
                    result.location[0] = result.location[1] = InvalidGenomeLocation;
                    bool pass0 = c->options->passFilter(pr[0], result.status[0], true, false);
                    bool pass1 = c->options->passFilter(pr[1], result.status[1], true, false);
                    bool pass = (c->options->filterFlags & AlignerOptions::FilterBothMatesMatch) ? (pass0 && pass1) : (pass0 || pass1);
                    if (pass) {
                        if (NULL != c->readWriter) {
                            _int64 noSingles[2] = {0, 0};
                            c->readWriter->writePairs(c->readerContext, pr, &result, 1, NULL, noSingles, true, c->useAffineGap);
                        }
                        c->stats->uselessReads += 2;
                    } else {
                        c->stats->filtered += 2;
                    }
                    continue;
                }
                for (int r = 0; r < NUM_READS_PER_PAIR; r++) {
                    new (&reads[2 * n + r]) ReadWithOwnMemory(*pr[r]);
                    bases.insert(bases.end(), pr[r]->getData(), pr[r]->getData() + pr[r]->getDataLength());
                    quals.insert(quals.end(), pr[r]->getQuality(), pr[r]->getQuality() + pr[r]->getDataLength());
                    offs.push_back((uint64_t)bases.size());
                }
                useful[2 * n] = useful0; useful[2 * n + 1] = useful1;
                n++;
            }
            if (0 == n) continue;
            if (bases.empty()) { bases.push_back(0); quals.push_back(0); }

            pthread_mutex_lock(&slot->lock);
            int rc;
            if (secondary) {
                // align() with secondary-result buffers; like PairedAligner.cpp:727-756, grow what was too small and call again
                for (;;) {
                    sec.resize((size_t)n * secStride); nSec.resize(n); ssec.resize((size_t)n * singleStride); nSingleSec.resize((size_t)2 * n);
                    rc = snapgpu_align_paired_secondary(slot->ctx, n, &bases[0], &quals[0], &offs[0], &prim[0], &alt[0], &sec[0], secStride, &nSec[0],
                                                        &ssec[0], singleStride, &nSingleSec[0]);
                    if (rc != SNAPGPU_W_SECONDARY_TRUNCATED) break;
                    for (unsigned i = 0; i < n; i++) {
                        if (nSec[i] > secStride) secStride = nSec[i];
                        if (nSingleSec[2 * i] + nSingleSec[2 * i + 1] > singleStride) singleStride = nSingleSec[2 * i] + nSingleSec[2 * i + 1];
                    }
                }
            } else {
                rc = snapgpu_align_paired(slot->ctx, n, &bases[0], &quals[0], &offs[0], &prim[0], &alt[0]);
            }
            pthread_mutex_unlock(&slot->lock);
            if (rc != SNAPGPU_OK) {
                WriteErrorMessage("snapgpu_align_paired failed (%d): %s\n", rc, snapgpu_last_error(slot->ctx));
                soft_exit(1);
            }

            for (unsigned i = 0; i < n; i++) {
                Read *two[NUM_READS_PER_PAIR] = {&reads[2 * i], &reads[2 * i + 1]};
                PairedAlignmentResult result, altResult;
                toSnapPaired(prim[i], &result);
                if (po->forceSpacing && isOneLocation(result.status[0]) != isOneLocation(result.status[1])) {         // PairedAligner.cpp:833-841
                    result.status[0] = result.status[1] = NotFound;
                    result.location[0] = result.location[1] = InvalidGenomeLocation;
                    result.usedAffineGapScoring[0] = result.usedAffineGapScoring[1] = false;
                    result.basesClippedBefore[0] = result.basesClippedBefore[1] = 0;
                    result.basesClippedAfter[0] = result.basesClippedAfter[1] = 0;
                    result.agScore[0] = result.agScore[1] = 0;
                }
                // PairedAligner.cpp:843-890: the primary and the paired secondary results go through the filter (the last one moves into
                // a hole), then the single-end secondary results of each read, then everything is written
                _int64 nSecondaryResults = secondary ? (_int64)nSec[i] : 0;
                results.resize((size_t)nSecondaryResults + 1);
                results[0] = result;
                for (_int64 k = 0; k < nSecondaryResults; k++) toSnapPaired(sec[(size_t)i * secStride + k], &results[1 + k]);
                bool firstIsPrimary = true;
                for (_int64 k = 0; k <= nSecondaryResults; k++) {
                    bool pass0 = c->options->passFilter(two[0], results[k].status[0], !useful[2 * i], k != 0 || !firstIsPrimary);
                    bool pass1 = c->options->passFilter(two[1], results[k].status[1], !useful[2 * i + 1], k != 0 || !firstIsPrimary);
                    bool pass = (c->options->filterFlags & AlignerOptions::FilterBothMatesMatch) ? (pass0 && pass1) : (pass0 || pass1);
                    if (!pass) {
                        results[k] = results[nSecondaryResults];
                        nSecondaryResults--;
                        if (0 == k) firstIsPrimary = false;
                        k--;
                    }
                }
                nSingleResults[0] = secondary ? (_int64)nSingleSec[2 * i] : 0;
                nSingleResults[1] = secondary ? (_int64)nSingleSec[2 * i + 1] : 0;
                singles.resize((size_t)(nSingleResults[0] + nSingleResults[1]) + 1);
                for (_int64 k = 0; k < nSingleResults[0] + nSingleResults[1]; k++) toSnap(ssec[(size_t)i * singleStride + k], &singles[k]);
                SingleAlignmentResult *singleResults[2] = {&singles[0], &singles[0] + nSingleResults[0]};
                for (int r = 0; r < NUM_READS_PER_PAIR; r++) {
                    for (_int64 k = 0; k < nSingleResults[r]; k++) {
                        if (!c->options->passFilter(two[r], singleResults[r][k].status, false, true)) {
                            singleResults[r][k] = singleResults[r][nSingleResults[r] - 1];
                            nSingleResults[r]--;
                            k--;
                        }
                    }
                }
                c->stats->extraAlignments += nSecondaryResults + (firstIsPrimary ? 0 : 1);
                if (NULL != c->readWriter) {
                    c->readWriter->writePairs(c->readerContext, two, &results[0], nSecondaryResults + 1, singleResults, nSingleResults, firstIsPrimary, c->useAffineGap);
                    if (c->emitALTAlignments && (alt[i].status[0] != SNAPGPU_NotFound || alt[i].status[1] != SNAPGPU_NotFound)) {
                        toSnapPaired(alt[i], &altResult);
                        c->readWriter->writePairs(c->readerContext, two, &altResult, 1, NULL, 0, true, c->useAffineGap);
                    }
                }
                if (firstIsPrimary) {                               // PairedAlignerContext::updateStats, PairedAligner.cpp:962-1000 (base counters)
                    for (int r = 0; r < NUM_READS_PER_PAIR; r++) {
                        if (useful[2 * i + r]) {
                            if (isOneLocation(result.status[r])) c->stats->singleHits++;
                            else if (result.status[r] == MultipleHits) c->stats->multiHits++;
                            else c->stats->notFound++;
                            if (result.status[r] != NotFound) c->stats->mapqHistogram[result.mapq[r]]++;
                        } else {
                            c->stats->uselessReads++;
                        }
                    }
                    if (result.direction[0] == result.direction[1]) c->stats->sameComplement++;
                    if (result.alignedAsPair) c->stats->alignedAsPairs += 2;
                } else {
                    c->stats->filtered += 2;
                }
                reads[2 * i].dispose();
                reads[2 * i + 1].dispose();
            }
        }
        BigDealloc(reads);
        delete[] useful;
        snapgpu_counters counters;          // (workers that share a context: whoever ends first takes what has accumulated; the sums are the same)
        pthread_mutex_lock(&slot->lock);
        if (snapgpu_get_counters(slot->ctx, &counters, 1) == SNAPGPU_OK) {
            c->stats->lvCalls += (_int64)counters.n_lv_locations;
            c->stats->affineGapCalls += (_int64)counters.n_ag_locations;
        }
        pthread_mutex_unlock(&slot->lock);
        return true;
    }

    virtual bool runIterationThread(ReadSupplier *supplier, AlignerContext *c)
    {
        if (c->index == NULL) {
            return false;                                   // I/O-only mode (SingleAligner.cpp:106-131): leave it to SNAP
        }
        // -ae (!ignoreAlignmentAdjustmentForOm): with -om the library adjusts primary and secondary results before its filter
        // (snapgpu_secondary_params::adjust_alignments); without, finalizeSecondaryResults has only the primary to adjust
        // (BaseAligner.cpp:2444-2452) and snapgpu_adjust_alignments does that to the batch
        const bool adjustPrimaries = !c->ignoreAlignmentAdjustmentForOm;
        ensureContext(c, c->numSeedsFromCommandLine);
        const bool secondary = c->maxSecondaryAlignmentAdditionalEditDistance >= 0;      // -om
        if (secondary) ensureSecondary(c);
        GpuSlot *slot = mySlot();
        uint32_t secStride = 8;
        if ((_int64)secStride > (_int64)c->maxSecondaryAlignments) secStride = (uint32_t)c->maxSecondaryAlignments;
        std::vector<snapgpu_single_result> sec;
        std::vector<uint32_t> nSec;
        std::vector<SingleAlignmentResult> results;

        // Reads are only valid until the supplier moves on, so each batch is copied.  ReadWithOwnMemory
        // points into its own body and has no copy-assignment: construct in place in raw storage.
        const unsigned BATCH = 16384;
        ReadWithOwnMemory *reads = (ReadWithOwnMemory *)BigAlloc((size_t)BATCH * sizeof(ReadWithOwnMemory));
        std::vector<char> bases, quals;
        std::vector<uint64_t> offs;
        std::vector<int32_t> lens;
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

            pthread_mutex_lock(&slot->lock);
            int rc;
            if (secondary) {
                // AlignRead with a secondary-result buffer; like SingleAligner.cpp:250-263, grow it and call again when it was too small
                for (;;) {
                    sec.resize((size_t)n * secStride); nSec.resize(n);
                    rc = snapgpu_align_single_secondary(slot->ctx, n, &bases[0], &quals[0], &offs[0], &prim[0], &alt[0], &sec[0], secStride, &nSec[0]);
                    if (rc != SNAPGPU_W_SECONDARY_TRUNCATED) break;
                    for (unsigned i = 0; i < n; i++) if (nSec[i] > secStride) secStride = nSec[i];
                }
                // -ae with -om: the adjuster ran inside the call (snapgpu_secondary_params::adjust_alignments) on the primary and on every
                // secondary result, with the same limitation as below: a quality-clipped read whose result reaches the end of its contig
                // is refused, not answered differently
                for (uint32_t i = 0; rc == SNAPGPU_OK && adjustPrimaries && i < n; i++) {
                    Read *rd = &reads[i];
                    if (rd->getDataLength() == rd->getUnclippedLength()) continue;
                    const uint32_t ns = nSec[i] < secStride ? nSec[i] : secStride;
                    for (uint32_t j = 0; j <= ns; j++) {
                        const snapgpu_single_result &r = j == 0 ? prim[i] : sec[(size_t)i * secStride + (j - 1)];
                        if (r.status == SNAPGPU_NotFound) continue;
                        const Genome::Contig *ct = c->index->getGenome()->getContigAtLocation(GenomeLocation(r.location));
                        if (ct != NULL && r.location + (int64_t)rd->getDataLength() + (int64_t)c->maxDist + 2 >
                                          GenomeLocationAsInt64(ct->beginningLocation) + ct->length - c->index->getGenome()->getChromosomePadding()) {
                            WriteErrorMessage("snapgpu shim: -ae: a quality-clipped read hangs over the end of its contig, which the adjuster does not reproduce (run with -C--)\n");
                            soft_exit(1);
                        }
                    }
                }
            } else {
                rc = snapgpu_align_single(slot->ctx, n, &bases[0], &quals[0], &offs[0], &prim[0], &alt[0]);
                if (rc == SNAPGPU_OK && adjustPrimaries) {
                    lens.resize(n);
                    for (unsigned i = 0; i < n; i++) lens[i] = (int32_t)(offs[i + 1] - offs[i]);
                    rc = snapgpu_adjust_alignments(slot->ctx, n, &bases[0], (uint64_t)bases.size(), &offs[0], &lens[0], &prim[0]);
                    // The device adjuster is restated for a Read the reader has not clipped (snap_amd/csrc/adjust.h): the reference settles a
                    // contig-end overhang on the UNCLIPPED buffer (AlignmentAdjuster.cpp:167), which only differs for a clipped read that
                    // hangs over the end of its contig.  Such a read is refused rather than answered differently (-C-- avoids it).
                    for (uint32_t i = 0; rc == SNAPGPU_OK && i < n; i++) {
                        Read *rd = &reads[i];
                        if (rd->getDataLength() == rd->getUnclippedLength() || prim[i].status == SNAPGPU_NotFound) continue;
                        const Genome::Contig *ct = c->index->getGenome()->getContigAtLocation(GenomeLocation(prim[i].location));
                        if (ct != NULL && prim[i].location + (int64_t)rd->getDataLength() + (int64_t)c->maxDist + 2 >
                                          GenomeLocationAsInt64(ct->beginningLocation) + ct->length - c->index->getGenome()->getChromosomePadding()) {
                            WriteErrorMessage("snapgpu shim: -ae: a quality-clipped read hangs over the end of its contig, which the adjuster does not reproduce (run with -C--)\n");
                            soft_exit(1);
                        }
                    }
                }
            }
            pthread_mutex_unlock(&slot->lock);
            if (rc != SNAPGPU_OK) {
                WriteErrorMessage("snapgpu_align_single failed (%d): %s\n", rc, snapgpu_last_error(slot->ctx));
                soft_exit(1);
            }

            for (unsigned i = 0; i < n; i++) {
                SingleAlignmentResult result, altResult;
                toSnap(prim[i], &result);
                bool containsPrimary = true;
                if (NULL != c->readWriter) {
                    // SingleAligner.cpp:293-318: drop what the filter rejects (the last result moves into the hole), write the rest
                    _int64 nSecondaryResults = secondary ? (_int64)nSec[i] : 0;
                    results.resize((size_t)nSecondaryResults + 1);
                    results[0] = result;
                    for (_int64 k = 0; k < nSecondaryResults; k++) toSnap(sec[(size_t)i * secStride + k], &results[1 + k]);
                    for (_int64 k = 0; k <= nSecondaryResults; k++) {
                        if (!c->options->passFilter(&reads[i], results[k].status, false, k != 0 || !containsPrimary)) {
                            if (k == 0) containsPrimary = false;
                            results[k] = results[nSecondaryResults];
                            nSecondaryResults--;
                            k--;
                        }
                    }
                    c->stats->extraAlignments += nSecondaryResults + (containsPrimary ? 0 : 1);
                    c->readWriter->writeReads(c->readerContext, &reads[i], &results[0], nSecondaryResults + 1, containsPrimary, c->useAffineGap);
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
        snapgpu_counters counters;          // (workers that share a context: whoever ends first takes what has accumulated; the sums are the same)
        pthread_mutex_lock(&slot->lock);
        if (snapgpu_get_counters(slot->ctx, &counters, 1) == SNAPGPU_OK) {
            c->stats->lvCalls += (_int64)counters.n_lv_locations;
            c->stats->affineGapCalls += (_int64)counters.n_ag_locations;
        }
        pthread_mutex_unlock(&slot->lock);
        return true;                                            // we ran the whole per-thread loop
    }

private:
    static void ensurePaired(AlignerContext *c, PairedAlignerOptions *po)
    {
        pthread_mutex_lock(&g_setupLock);
        if (!g_pairedEnabled) {
            snapgpu_paired_params pp;
            snapgpu_default_paired_params(&pp);
            pp.min_spacing = po->minSpacing; pp.max_spacing = po->maxSpacing; pp.force_spacing = po->forceSpacing ? 1 : 0;
            pp.max_big_hits = po->intersectingAlignerMaxHits; pp.max_candidate_pool_size = po->maxCandidatePoolSize;
            pp.num_seeds = c->numSeedsFromCommandLine; pp.seed_coverage = c->seedCoverage; pp.max_k_for_indels = c->maxDistForIndels;
            pp.min_read_length = c->minReadLength; pp.use_soft_clipping = c->options->useSoftClipping ? 1 : 0;
            pp.flatten_mapq_at_or_below = c->options->flattenMAPQAtOrBelow; pp.min_score_realignment = po->minScoreRealignment;
            pp.min_score_gap_realignment_alt = po->minScoreGapRealignmentALT; pp.min_ag_score_improvement = po->minAGScoreImprovement;
            pp.enable_hamming_scoring_base_aligner = po->enableHammingScoringBaseAligner ? 1 : 0; pp.max_single_seeds = po->maxSeedsSingleEnd;
            for (int i = 0; i < g_nSlots; i++) {
                int rc = snapgpu_enable_paired(g_slots[i].ctx, &pp);
                if (rc != SNAPGPU_OK) {
                    WriteErrorMessage("snapgpu_enable_paired failed (%d): %s\n", rc, snapgpu_last_error(g_slots[i].ctx));
                    soft_exit(1);
                }
            }
            g_pairedEnabled = true;
        }
        pthread_mutex_unlock(&g_setupLock);
    }

    // numSeeds: -n for the single-end aligner (the paired path hands its own -n to snapgpu_enable_paired)
    static void ensureSecondary(AlignerContext *c)
    {
        pthread_mutex_lock(&g_setupLock);
        if (!g_secondaryEnabled) {
            snapgpu_secondary_params sp;
            sp.max_edit_distance = c->maxSecondaryAlignmentAdditionalEditDistance;      // -om
            sp.max_per_contig = c->maxSecondaryAlignmentsPerContig;                     // -mpc
            sp.max_results = c->maxSecondaryAlignments;                                 // -omax
            sp.adjust_alignments = c->ignoreAlignmentAdjustmentForOm ? 0 : 1;           // -ae
            for (int i = 0; i < g_nSlots; i++) {
                int rc = snapgpu_enable_secondary(g_slots[i].ctx, &sp);
                if (rc != SNAPGPU_OK) {
                    WriteErrorMessage("snapgpu_enable_secondary failed (%d): %s\n", rc, snapgpu_last_error(g_slots[i].ctx));
                    soft_exit(1);
                }
            }
            g_secondaryEnabled = true;
        }
        pthread_mutex_unlock(&g_setupLock);
    }

    static void ensureContext(AlignerContext *c, unsigned numSeeds)
    {
        pthread_mutex_lock(&g_setupLock);
        if (NULL == g_slots) {
            snapgpu_params p;
            snapgpu_default_params(&p);
            p.max_hits = (uint32_t)c->maxHits;
            p.max_k = c->maxDist;
            p.num_seeds = numSeeds;
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
            // GPUs: all that are visible, but never more than there are worker threads to feed them (SNAPGPU_SHIM_GPUS=<n> for fewer);
            // feeders per GPU: SNAPGPU_SHIM_FEEDERS (default 2)
            int nDev = snapgpu_device_count();
            if (nDev < 1) nDev = 1;
            env = getenv("SNAPGPU_SHIM_GPUS");
            if (env && atoi(env) >= 1 && atoi(env) < nDev) nDev = atoi(env);
            if (c->options->numThreads >= 1 && nDev > c->options->numThreads) nDev = c->options->numThreads;      // (-t: a GPU nobody feeds would only cost an index copy)
            int nFeed = 2;
            env = getenv("SNAPGPU_SHIM_FEEDERS");
            if (env && atoi(env) >= 1 && atoi(env) <= 8) nFeed = atoi(env);
            GpuSlot *slots = new GpuSlot[(size_t)nDev * nFeed];
            snapgpu_ctx **first = new snapgpu_ctx *[nDev];
            int rc = snapgpu_create_from_directory(c->options->indexDir, &p, 0, &first[0]);
            if (rc != SNAPGPU_OK) {
                WriteErrorMessage("snapgpu_create_from_directory(%s) failed (%d): %s\n", c->options->indexDir, rc, snapgpu_last_error(NULL));
                soft_exit(1);
            }
            // Further GPUs: same-size blobs filled by one RCCL broadcast.  If that is not to be had (librccl cannot be loaded, a
            // communicator does not come up), each further GPU loads the directory itself; if even that fails, the run goes on with the
            // GPUs that did come up -- a multi-GPU host without a working RCCL must not be worse off than a single-GPU one.
            if (nDev > 1) {
                int made = 1;
                for (int d = 1; d < nDev; d++, made++) {
                    rc = snapgpu_create_replica(first[0], d, 0, &first[d]);
                    if (rc != SNAPGPU_OK) {
                        WriteErrorMessage("snapgpu_create_replica(device %d) failed (%d): %s -- continuing with %d GPU(s)\n", d, rc, snapgpu_last_error(first[0]), made);
                        break;
                    }
                }
                nDev = made;
            }
            if (nDev > 1) {
                rc = snapgpu_broadcast_index(first, nDev);
                if (rc != SNAPGPU_OK) {
                    WriteErrorMessage("snapgpu_broadcast_index over %d GPUs failed (%d): %s -- loading the index on each GPU instead\n", nDev, rc, snapgpu_last_error(first[0]));
                    int kept = 1;
                    for (int d = 1; d < nDev; d++) {
                        snapgpu_destroy(first[d]);                  // its blobs were never filled
                        first[d] = NULL;
                    }
                    for (int d = 1; d < nDev; d++) {
                        snapgpu_ctx *own = NULL;
                        rc = snapgpu_create_from_directory(c->options->indexDir, &p, d, &own);
                        if (rc != SNAPGPU_OK) {
                            WriteErrorMessage("snapgpu_create_from_directory on device %d failed (%d): %s -- continuing with %d GPU(s)\n", d, rc, snapgpu_last_error(NULL), kept);
                            break;
                        }
                        first[kept++] = own;
                    }
                    nDev = kept;
                }
            }
            int n = 0;
            for (int f = 0; f < nFeed; f++) {                  // slot order: GPU 0, GPU 1, ..., then the second feeder of each
                for (int d = 0; d < nDev; d++) {
                    snapgpu_ctx *ctx = first[d];
                    if (f > 0) {
                        rc = snapgpu_create_replica(first[d], d, 1, &ctx);
                        if (rc != SNAPGPU_OK) {
                            WriteErrorMessage("snapgpu_create_replica(feeder %d of device %d) failed (%d): %s\n", f, d, rc, snapgpu_last_error(first[d]));
                            soft_exit(1);
                        }
                    }
                    // -f / -x: SingleAligner.cpp:179-180 sets them on every BaseAligner it constructs; the paired-end contexts ignore them, as
                    // the reference's paired-end aligners do (include/snapgpu.h: snapgpu_set_aligner_flags)
                    snapgpu_set_aligner_flags(ctx, c->options->stopOnFirstHit ? 1 : 0, c->options->explorePopularSeeds ? 1 : 0);
                    slots[n].ctx = ctx;
                    pthread_mutex_init(&slots[n].lock, NULL);
                    n++;
                }
            }
            delete[] first;
            g_nSlots = n;
            g_slots = slots;
        }
        pthread_mutex_unlock(&g_setupLock);
    }

    // this worker thread's context (the extension object is one per worker: AlignerContext.cpp:226 calls copy())
    GpuSlot *mySlot()
    {
        if (NULL == slot_) {
            int t = __sync_fetch_and_add(&g_nextWorker, 1);
            slot_ = &g_slots[t % g_nSlots];
        }
        return slot_;
    }
    GpuSlot *slot_;
};

// Mirror of ProcessNonDaemonCommands (SNAPLib/CommandProcessor.cpp:59-88) with the extension installed.
int main(int argc, const char **argv)
{
    if (argc < 2 || (strcmp(argv[1], "single") != 0 && strcmp(argv[1], "paired") != 0)) {
        fprintf(stderr, "usage: snap-aligner-gpu single <index-dir> <reads> [SNAP options]\n"
                        "       snap-aligner-gpu paired <index-dir> <reads1> <reads2> [SNAP options]\n"
                        "       (index / daemon: use the reference's snap-aligner)\n");
        return 1;
    }
    InitializeSeedSequencers();                                     // CommandProcessor.cpp:196
    unsigned nArgsConsumed = 0;
    GpuAlignerExtension *extension = new GpuAlignerExtension();
    if (strcmp(argv[1], "single") == 0) {
        SingleAlignerContext single(extension);                     // SingleAligner.h:36
        single.runAlignment(argc - 1, argv + 1, SNAP_VERSION, &nArgsConsumed);
    } else {
        PairedAlignerContext paired(extension);                     // PairedAligner.h:42
        paired.runAlignment(argc - 1, argv + 1, SNAP_VERSION, &nArgsConsumed);
    }
    for (int i = g_nSlots - 1; i >= 0; i--) snapgpu_destroy(g_slots[i].ctx);       // (sharing contexts before the ones that own the blobs)
    return 0;
}
