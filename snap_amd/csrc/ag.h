// ag.h -- affine-gap scoring (AffineGapVectorized<+-1>::computeScore / computeScoreBanded).
// PLACEHOLDER: filled in by the affine-gap milestone; until then use_affine_gap must be 0
// (snapgpu_create rejects anything else), so ag_compute is never reached.
#pragma once
#include "dev_common.h"

static __host__ __device__ __forceinline__ size_t ag_scratch_bytes(uint32_t RL) { (void)RL; return 0; }

struct AGParams { int match_reward, sub_penalty, gap_open, gap_extend, five_bonus, three_bonus; };

struct AGResult {
    int    ag_score;         // -1 when no alignment scores above score_init
    int    text_offset;
    int    pattern_offset;
    int    n_edits;
    double match_probability;
};

template <typename PSeq, typename TSeq, typename QSeq>
static __device__ __forceinline__ AGResult ag_compute(
    bool banded, int dir, const AGParams &prm, const PSeq &P, const QSeq &Q, int pattern_len,
    const TSeq &T, int text_len, int w, int score_init, bool is_rc, bool use_clipping,
    uint8_t *scratch, uint32_t numvec_max, const DevTables *tab)
{
    AGResult r; r.ag_score = -1; r.text_offset = -1; r.pattern_offset = -1; r.n_edits = -1; r.match_probability = 0.0;
    return r;
}
