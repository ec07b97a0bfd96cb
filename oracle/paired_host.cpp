/*
 * paired_host.cpp -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Compiles the paired-end control flow of snap_amd/csrc/paired.h for the host (SNAP_PE_HOST) with
 * the CPU restatement's primitives (snap_oracle.c: seed lookup, Landau-Vishkin, affine gap) plugged
 * in where the device build uses the wavefront kernels, and -- for the chimeric fallback -- the
 * reference's own single-end aligner (ref_driver.cpp: snapref_chimeric_single_*).  That lets the CPU
 * test-suite diff the exact control flow the GPU executes against the reference's
 * IntersectingPairedEndAligner / ChimericPairedEndAligner without a GPU.  The product never links this.
 */
#define SNAP_PE_HOST 1
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../snap_amd/csrc/paired.h"
#include "snap_oracle.h"

extern "C" {
void *snapref_chimeric_single_create(void *vindex, const snapgpu_params *p, const snapgpu_paired_params *pp);
int   snapref_chimeric_single_align(void *h, int max_k, int hamming, const char *bases, const char *quals, uint32_t len,
                                    snapgpu_single_result *res, snapgpu_single_result *alt);
void  snapref_chimeric_single_destroy(void *h);
void *snapref_chimeric_single_create2(void *vindex, const snapgpu_params *p, const snapgpu_paired_params *pp, int mpc);
int   snapref_chimeric_single_align2(void *h, int max_k, int hamming, const char *bases, const char *quals, uint32_t len,
                                     snapgpu_single_result *res, snapgpu_single_result *alt,
                                     int om, int64_t omax, snapgpu_single_result *sec_out, uint32_t sec_room, uint32_t *n_sec,
                                     uint32_t first_room, uint32_t *overflowed_first);
}

extern "C" {
/* oracle_genome: snap_oracle.h */
int oracle_align_read_chimeric(const oracle_index *ix, const oracle_genome *g, const snapgpu_params *p, int max_k, int hamming,
                               const char *bases, const char *quals, int len, snapgpu_single_result *primary, snapgpu_single_result *first_alt,
                               const snapgpu_secondary_params *sp, snapgpu_single_result *secondary, uint32_t sec_room, uint32_t *n_secondary,
                               uint32_t *stale, uint32_t *raw_secondary);
}
static int g_use_restatement = 0;
extern "C" void pairedhost_use_restatement(int on) { g_use_restatement = on; }

struct HostPL {
    typedef HostPL *SelfPtr;
    const snapgpu_index_view *ix;
    oracle_index oix;
    oracle_ag_params agp;
    const double *phred_t, *indel_t, *perfect_t;
    double seed_prob_v;
    int seed_len;
    void *single;                         // reference single-end aligner (chimeric fallback), may be NULL for stage 1
    const uint8_t *read_b[2], *read_q[2];
    int read_l[2];
    uint32_t stale_single;

    template <class T> static T ld(const T &x) { return x; }
    template <class T> static void st(T &x, T v) { x = v; }
    static int i32(int v) { return v; }
    static double f64(double v) { return v; }
    static bool lane0() { return true; }
    static void sync() {}
    static uint64_t clock() { return 0; }
    static const bool FAST_HITSET = false;
    static const bool HELP = false;                     // (Phase-4 help slots are a device-side scheduling matter)
    static const bool ALWAYS_COUNT_STALE = true;        // (the host build keeps the conservative count: its tests exclude by it)
    template <class T> static void spec_st(T &x, T v) { x = v; }
    template <class T> static T spec_ld(const T &x) { return x; }
    static const bool SECONDARY = true;
    // (never called: the scalar definitions in paired.h are what the host runs)
    static void lds(const void *) {}
    struct HSCursor {};
    void hs_begin_walk(PELookup *, PEHitSetHdr *, int, uint32_t, HSCursor &) {}
    template <class R> void hint_indels(R *, uint32_t, uint32_t, int) {}
    bool hs_first(HSCursor &, int64_t *, uint32_t *) { return true; }
    bool hs_next_lower(HSCursor &, int64_t *, uint32_t *) { return false; }
    bool hs_next_le(HSCursor &, int64_t, int64_t *, uint32_t *) { return false; }
    uint32_t hs_best_possible(HSCursor &, uint32_t *) { return 0; }

    bool lookup(const uint8_t *text, PEHits out[2]) {
        uint64_t bases, rc;
        if (!oracle_pack_seed((const char *)text, (unsigned)seed_len, &bases, &rc)) return false;
        int64_t n[2]; const uint32_t *h[2]; uint32_t single_v[2], slots[2];
        oracle_lookup_seed(&oix, bases, rc, n, h, single_v, slots);
        for (int d = 0; d < 2; d++) { out[d].n_hits = n[d]; out[d].hits = n[d] > 1 ? h[d] : NULL; out[d].singleton = single_v[d]; }
        return true;
    }
    uint32_t wrapped_seed(uint32_t wrap) const { return oracle_wrapped_next_seed((unsigned)seed_len, wrap); }
    uint32_t count_n(const uint8_t *b, int len) const { uint32_t n = 0; for (int i = 0; i < len; i++) n += b[i] == 'N'; return n; }
    const uint8_t *window(int64_t loc, int) const { return ix->genome + loc; }
    bool is_alt(int64_t loc) const { return loc >= 0 && (uint64_t)loc >= ix->first_alt_location; }

    // Genome::getSubstring(loc, len) != NULL, SNAPLib/Genome.h:339-367
    bool substring_ok(int64_t loc, int64_t len) const {
        int64_t nb = (int64_t)ix->n_bases;
        if (loc > nb || loc + len > nb + 1000) return false;
        if (loc < -(int64_t)ix->genome_pad) return false;
        if (len <= (int64_t)ix->chromosome_padding && ix->genome[loc] != 'n') return true;
        if (len == 0) return true;
        int lo = 0, hi = (int)ix->n_contigs - 1, found = -1;
        while (lo <= hi) {
            int mid = (lo + hi) >> 1;
            if ((int64_t)ix->contig_begin[mid] <= loc) { found = mid; lo = mid + 1; } else hi = mid - 1;
        }
        if (found < 0) return false;
        int64_t cend = found == (int)ix->n_contigs - 1 ? nb : (int64_t)ix->contig_begin[found + 1];
        return cend > loc + len;
    }
    int mapq(double p_all, double p_best, int popular) const { return oracle_compute_mapq(p_all, p_best, 0, popular); }
    double seed_prob() const { return seed_prob_v; }
    double phred(uint8_t q) const { return phred_t[q]; }
    double indel(int n) const { return indel_t[n]; }
    double perfect(int n) const { return perfect_t[n]; }

    // The core addresses pattern/quality/text of a backward problem at the first byte compared and walks with
    // stride -1 (the device convention).  The oracle follows the reference: reversed pattern/quality arrays and a
    // text pointer one past the first compared byte.
    LVOut lv(int st, const uint8_t *P, const uint8_t *Q, int plen, const uint8_t *T, int tlen, int k) const {
        LVOut o; o.score = -1; o.mp = 0; o.net_indel = 0; o.total_indels = 0; o.text_span = 0;
        std::vector<char> pb(plen + 16), qb(plen + 16);
        for (int i = 0; i < plen; i++) { pb[i] = (char)P[i * st]; qb[i] = (char)Q[i * st]; }
        double mp = 1.0;
        int net = 0, tot = 0, span = 0;
        o.score = oracle_lv(st, (const char *)(st > 0 ? T : T + 1), tlen, &pb[0], &qb[0], plen, k, &mp, &net, &tot, &span);
        o.mp = mp; o.net_indel = net; o.total_indels = tot; o.text_span = span;
        return o;
    }
    AGOut ag(bool banded, int st, const uint8_t *P, const uint8_t *Q, int plen, const uint8_t *T, int tlen, int lim, int read_len,
             bool is_rc, int use_clip) const {
        AGOut o;
        std::vector<char> pb(plen + 16), qb(plen + 16);
        for (int i = 0; i < plen; i++) { pb[i] = (char)P[i * st]; qb[i] = (char)Q[i * st]; }
        int to = -1, po = -1, ne = -1, stale = 0;
        double mp = 1.0;
        o.ag_score = oracle_ag(st, banded ? 1 : 0, &agp, (const char *)(st > 0 ? T : T + 1), tlen, &pb[0], &qb[0], plen, lim, read_len,
                               is_rc ? 1 : 0, use_clip, &to, &po, &ne, &mp, &stale);
        o.text_offset = to; o.pattern_offset = po; o.n_edits = ne; o.mp = mp; o.stale = stale;
        return o;
    }
    void sort_candidates(const snapgpu_paired_result *c, uint32_t n, uint32_t *order) const {     // stable counting sort by pair score
        std::vector<uint32_t> base(1024, 0);
        for (uint32_t j = 0; j < n; j++) base[(c[j].reserved & 511) + 1]++;
        for (int k = 1; k < 1024; k++) base[k] += base[k - 1];
        for (uint32_t j = 0; j < n; j++) order[base[c[j].reserved & 511]++] = j;
    }
    // returns the number of secondary results the read has; the first min(that, sec_room) go to sec_out
    uint32_t align_single(int r, int max_k, bool hamming, snapgpu_single_result &res, snapgpu_single_result &alt,
                          bool want_secondary, snapgpu_single_result *sec_out, uint32_t sec_room, uint32_t room32) {
        if (use_restatement) {            // oracle/align_oracle.c instead of the compiled reference's BaseAligner
            uint32_t n_sec = 0, stale = 0, raw = 0;
            snapgpu_secondary_params sp; sp.max_edit_distance = om; sp.max_per_contig = mpc; sp.max_results = omax; sp.adjust_alignments = 0;
            oracle_align_read_chimeric(&oix, &og, &single_params, max_k, hamming ? 1 : 0, (const char *)read_b[r], (const char *)read_q[r], read_l[r],
                                       &res, &alt, (want_secondary && om >= 0) ? &sp : NULL, sec_out, sec_room, &n_sec, &stale, &raw);
            last_raw = raw;
            return n_sec;
        }
        uint32_t n_sec = 0, over = 0;
        // what the reference's caller would have had left of its initial 32-entry buffer (PairedAligner.cpp:566; ChimericPairedEndAligner.cpp:311)
        const uint32_t first_room = room32;
        snapref_chimeric_single_align2(single, max_k, hamming ? 1 : 0, (const char *)read_b[r], (const char *)read_q[r], (uint32_t)read_l[r], &res, &alt,
                                       want_secondary ? om : -1, omax, sec_out, sec_room, &n_sec, first_room, &over);
        last_raw = over ? first_room + 1 : 0;
        return n_sec;
    }
    // pre-filter count of the last align_single's secondary candidates, as far as the caller needs it: > room or not
    uint32_t single_raw_secondary() const { return last_raw; }
    int om; int64_t omax;
    uint32_t last_raw;
    // the restatement of the single-end aligner, for runs that must not depend on the compiled reference
    int use_restatement, mpc;
    oracle_genome og;
    snapgpu_params single_params;
};

static uint8_t rc_of(uint8_t c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }

static int host_align(const snapgpu_index_view *ix, void *ref_index, const snapgpu_params *p, const snapgpu_paired_params *pp,
                      int stage, uint32_t n, const char *bases, const char *quals, const uint64_t *offsets,
                      snapgpu_paired_result *primary, snapgpu_paired_result *first_alt, int64_t *counters3,
                      const snapgpu_secondary_params *sp, snapgpu_paired_result *secondary, uint32_t sec_stride, uint32_t *n_secondary,
                      snapgpu_single_result *single_secondary, uint32_t ssec_stride, uint32_t *n_single_secondary);

extern "C" int pairedhost_align(const snapgpu_index_view *ix, void *ref_index, const snapgpu_params *p, const snapgpu_paired_params *pp,
                                int stage, uint32_t n, const char *bases, const char *quals, const uint64_t *offsets,
                                snapgpu_paired_result *primary, snapgpu_paired_result *first_alt, int64_t *counters3)
{
    return host_align(ix, ref_index, p, pp, stage, n, bases, quals, offsets, primary, first_alt, counters3, NULL, NULL, 0, NULL, NULL, 0, NULL);
}

/* ... with secondary results (-om / -omax / -mpc): secondary[i * sec_stride + k], k < n_secondary[i]; single_secondary[i * ssec_stride + k]:
 * read 0's n_single_secondary[2i] results, then read 1's n_single_secondary[2i+1]. */
extern "C" int pairedhost_align_secondary(const snapgpu_index_view *ix, void *ref_index, const snapgpu_params *p, const snapgpu_paired_params *pp,
                                          const snapgpu_secondary_params *sp, int stage, uint32_t n, const char *bases, const char *quals,
                                          const uint64_t *offsets, snapgpu_paired_result *primary, snapgpu_paired_result *first_alt,
                                          snapgpu_paired_result *secondary, uint32_t sec_stride, uint32_t *n_secondary,
                                          snapgpu_single_result *single_secondary, uint32_t ssec_stride, uint32_t *n_single_secondary)
{
    return host_align(ix, ref_index, p, pp, stage, n, bases, quals, offsets, primary, first_alt, NULL, sp, secondary, sec_stride, n_secondary,
                      single_secondary, ssec_stride, n_single_secondary);
}

static int host_align(const snapgpu_index_view *ix, void *ref_index, const snapgpu_params *p, const snapgpu_paired_params *pp,
                      int stage, uint32_t n, const char *bases, const char *quals, const uint64_t *offsets,
                      snapgpu_paired_result *primary, snapgpu_paired_result *first_alt, int64_t *counters3,
                      const snapgpu_secondary_params *sp, snapgpu_paired_result *secondary, uint32_t sec_stride, uint32_t *n_secondary,
                      snapgpu_single_result *single_secondary, uint32_t ssec_stride, uint32_t *n_single_secondary)
{
    oracle_init();
    HostPL pl;
    pl.ix = ix;
    pl.oix.seed_len = ix->seed_len; pl.oix.key_bytes = ix->key_bytes; pl.oix.n_hash_tables = ix->n_hash_tables; pl.oix.large = ix->large_hash_table;
    pl.oix.hash_blob = ix->hash_blob; pl.oix.table_offset = ix->table_offset; pl.oix.table_size = ix->table_size; pl.oix.overflow = ix->overflow;
    pl.oix.n_bases = ix->n_bases;
    pl.agp.match_reward = (int)p->match_reward; pl.agp.sub_penalty = (int)p->sub_penalty; pl.agp.gap_open = (int)p->gap_open_penalty;
    pl.agp.gap_extend = (int)p->gap_extend_penalty; pl.agp.five_bonus = (int)p->five_prime_end_bonus; pl.agp.three_bonus = (int)p->three_prime_end_bonus;
    pl.phred_t = oracle_phred_table(); pl.indel_t = oracle_indel_table(); pl.perfect_t = oracle_perfect_table();
    pl.seed_len = (int)ix->seed_len;
    pl.seed_prob_v = oracle_seed_prob(pl.seed_len);
    pl.use_restatement = g_use_restatement;
    pl.single = (stage == 0 && ref_index && !pl.use_restatement) ? snapref_chimeric_single_create2(ref_index, p, pp, sp ? sp->max_per_contig : -1) : NULL;
    pl.mpc = sp ? sp->max_per_contig : -1;
    pl.og.genome = ix->genome; pl.og.n_bases = ix->n_bases; pl.og.genome_pad = ix->genome_pad; pl.og.chromosome_padding = ix->chromosome_padding;
    pl.og.contig_begin = ix->contig_begin; pl.og.n_contigs = ix->n_contigs; pl.og.first_alt_location = ix->first_alt_location;
    pl.single_params = *p; pl.single_params.max_k = p->max_k / 2; pl.single_params.num_seeds = pp->max_single_seeds;     // ChimericPairedEndAligner.cpp:81-88
    pl.om = sp ? sp->max_edit_distance : -1; pl.omax = sp ? sp->max_results : 0x7fffffff;

    PECfg cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.max_k = (int)p->max_k; cfg.extra_depth = (int)p->extra_search_depth; cfg.max_k_for_indels = (int)pp->max_k_for_indels;
    cfg.max_gap_alt = p->max_score_gap_to_prefer_non_alt; cfg.use_ag = p->use_affine_gap; cfg.alt_aware = p->alt_awareness;
    cfg.emit_alt = p->emit_alt_alignments; cfg.use_soft_clip = pp->use_soft_clipping; cfg.force_spacing = pp->force_spacing;
    cfg.match_reward = (int)p->match_reward; cfg.sub_penalty = (int)p->sub_penalty; cfg.gap_open = (int)p->gap_open_penalty;
    cfg.gap_extend = (int)p->gap_extend_penalty; cfg.five_bonus = (int)p->five_prime_end_bonus; cfg.three_bonus = (int)p->three_prime_end_bonus;
    cfg.min_spacing = pp->min_spacing; cfg.max_spacing = pp->max_spacing; cfg.max_big_hits = pp->max_big_hits;
    cfg.num_seeds = pp->num_seeds < PE_MAX_SEEDS ? pp->num_seeds : PE_MAX_SEEDS; cfg.seed_coverage = pp->seed_coverage;
    cfg.min_read_length = pp->min_read_length; cfg.flatten_mapq = pp->flatten_mapq_at_or_below; cfg.min_score_realign = pp->min_score_realignment;
    cfg.min_score_gap_realign_alt = pp->min_score_gap_realignment_alt; cfg.min_ag_improve = pp->min_ag_score_improvement;
    cfg.enable_hamming_base = pp->enable_hamming_scoring_base_aligner; cfg.seed_len = (int)ix->seed_len;
    uint32_t max_seeds = cfg.num_seeds ? cfg.num_seeds : PE_MAX_SEEDS;
    uint64_t pool = (uint64_t)pp->max_big_hits * max_seeds * 2;
    cfg.pool_size = (uint32_t)(pool < pp->max_candidate_pool_size ? pool : pp->max_candidate_pool_size);
    cfg.ag_cand_cap = p->use_affine_gap ? 65536 : 0;
    cfg.max_seeds = PE_MAX_SEEDS;
    cfg.om = sp ? sp->max_edit_distance : -1; cfg.mpc = sp ? sp->max_per_contig : -1; cfg.omax = sp ? sp->max_results : 0x7fffffff;
    cfg.sec_cap = sp ? 65536 : 0;
    std::vector<uint64_t> zero_pb(ix->n_contigs + 1, 0); std::vector<uint8_t> zero_rc(ix->n_contigs + 1, 0); std::vector<uint32_t> zero_cs(ix->n_contigs + 2, 0), zero_op(1, 0);
    cfg.proj.contig_begin = ix->contig_begin; cfg.proj.n_contigs = ix->n_contigs;
    cfg.proj.proj_begin = ix->contig_proj_begin ? ix->contig_proj_begin : &zero_pb[0];
    cfg.proj.proj_rc = ix->contig_proj_rc ? ix->contig_proj_rc : &zero_rc[0];
    cfg.proj.cigar_start = ix->contig_cigar_start ? ix->contig_cigar_start : &zero_cs[0];
    cfg.proj.cigar_ops = ix->cigar_ops ? ix->cigar_ops : &zero_op[0];

    PairedCore<HostPL> core(pl, cfg);
    std::vector<PELookup> lk(4 * cfg.max_seeds);
    std::vector<uint32_t> exhausted(4 * cfg.max_seeds), miss(cfg.max_seeds), seed_used(64);
    std::vector<PEHitSetHdr> hs(4);
    std::vector<int32_t> list_head(SNAPGPU_MAX_K + 2);
    std::vector<PECand> cand(cfg.pool_size);
    std::vector<PEMate> mate0(cfg.pool_size / 2 + 1), mate1(cfg.pool_size / 2 + 1);
    std::vector<PEAnchor> anchor(cfg.pool_size);
    std::vector<snapgpu_paired_result> agc(cfg.ag_cand_cap + 1);
    std::vector<uint32_t> agc_order(cfg.ag_cand_cap + 1);
    std::vector<snapgpu_paired_result> sec(cfg.sec_cap + 1);
    std::vector<uint32_t> sec_ord(cfg.sec_cap + 1), sec_key(2 * cfg.sec_cap + 1);
    PEShared sh;
    memset(&sh, 0, sizeof(sh));
    core.lk = &lk[0]; core.exhausted = &exhausted[0]; core.miss = &miss[0]; core.hs = &hs[0]; core.list_head = &list_head[0];
    core.seed_used = &seed_used[0]; core.sh = &sh; core.cand = &cand[0]; core.mate[0] = &mate0[0]; core.mate[1] = &mate1[0];
    core.anchor = &anchor[0]; core.agc = &agc[0]; core.agc_order = &agc_order[0];
    core.sec = &sec[0]; core.sec_ord = &sec_ord[0]; core.sec_key = &sec_key[0]; core.n_sec = 0;
    core.ssec_out = NULL; core.ssec_stride = 0; core.n_ssec[0] = core.n_ssec[1] = 0;

    const int PAD = 160;
    std::vector<uint8_t> buf[2][2], qbuf[2][2];
    for (uint32_t i = 0; i < n; i++) {
        for (int r = 0; r < 2; r++) {
            int len = (int)(offsets[2 * i + r + 1] - offsets[2 * i + r]);
            const uint8_t *b = (const uint8_t *)bases + offsets[2 * i + r], *q = (const uint8_t *)quals + offsets[2 * i + r];
            for (int d = 0; d < 2; d++) { buf[r][d].assign(len + 2 * PAD, 0); qbuf[r][d].assign(len + 2 * PAD, 0); }
            for (int j = 0; j < len; j++) {
                buf[r][0][PAD + j] = b[j]; qbuf[r][0][PAD + j] = q[j];
                buf[r][1][PAD + len - 1 - j] = rc_of(b[j]); qbuf[r][1][PAD + len - 1 - j] = q[j];
            }
            for (int d = 0; d < 2; d++) { core.rd[r][d] = &buf[r][d][PAD]; core.ql[r][d] = &qbuf[r][d][PAD]; }
            core.read_len[r] = len;
            pl.read_b[r] = b; pl.read_q[r] = q; pl.read_l[r] = len;
        }
        memset(&sh.res, 0, sizeof(sh.res));
        memset(&sh.alt, 0, sizeof(sh.alt));
        core.overflow = 0; core.stale = 0;
        core.n_sec = 0; core.n_ssec[0] = core.n_ssec[1] = 0;
        core.ssec_out = single_secondary ? single_secondary + (size_t)i * ssec_stride : NULL; core.ssec_stride = ssec_stride;
        if (stage == 1) {
            core.intersecting_align();
        } else {
            core.align_pair((int)p->max_k, (int)p->max_k / 2);
        }
        sh.res.flags = (core.overflow ? SNAPGPU_PAIR_POOL_OVERFLOW : 0) | (core.ref_dep ? SNAPGPU_PAIR_REF_BUFFER_DEPENDENT : 0);
        sh.res.reserved = core.stale;
        primary[i] = sh.res;
        first_alt[i] = sh.alt;
        if (n_secondary) {
            n_secondary[i] = core.n_sec;
            for (uint32_t k = 0; k < core.n_sec && k < sec_stride; k++) secondary[(size_t)i * sec_stride + k] = *core.secondary(k);
            n_single_secondary[2 * i] = core.n_ssec[0]; n_single_secondary[2 * i + 1] = core.n_ssec[1];
        }
    }
    if (counters3) { counters3[0] = (int64_t)sh.cnt.lv; counters3[1] = (int64_t)sh.cnt.ag; counters3[2] = (int64_t)sh.cnt.lookups; }
    if (pl.single) snapref_chimeric_single_destroy(pl.single);
    return 0;
}
