"""Index shapes other than the north star's (-s 20, small hash tables): longer seeds (key bytes 5 and 7), `-large` hash tables
(one entry serves both strands), 16 instead of 256 tables -- SNAPHashTable / lookupSeed32 take a different path for each
(GenomeIndex.cpp:2096-2202, HashTable.h:72-118).  Against the compiled reference on a fresh genome, work counters included.

Written in a round that had no GPU time left: verified on the wavefront emulator (tests/test_emu_kernels.py), not yet on
hardware -- hence the file name, which puts it at the end of the `-m gpu` run."""
import os

import numpy as np
import pytest

from snap_amd import abi, synth
from tests import util
from oracle import ref

SHAPES = [(24, False), (20, True), (22, True), (32, False)]
# indexes whose files carry 5 .. 8-byte locations -- what the indexer picks by itself for seeds shorter than 20 (GenomeIndex.cpp:446-453), or
# -locationSize: the reference then aligns through lookupSeed / overflowTable64 (GenomeIndex.cpp:2205-2328); here the tables are narrowed on load
WIDE_SHAPES = [(16, False, []), (18, True, []), (19, False, []), (20, False, ["-locationSize", "6"]), (20, True, ["-locationSize", "7"])]      # (-locationSize 8: the reference's own indexer fails with bad_alloc)


def align_and_compare(tmp, seed_len, large, n_reads, genome_bases=400_000, extra=(), from_directory=False):
    from snap_amd.aligner import BaseAligner
    from snap_amd.index import GenomeIndex
    g = synth.make_genome(91, genome_bases, n_contigs=2, repeat_frac=0.4, max_copies=100, n_run_frac=0.002)
    fa = os.path.join(tmp, "ref.fa"); synth.write_fasta(fa, g)
    d = os.path.join(tmp, "idx")
    ref.build_index(fa, d, seed_len, threads=max(1, min(8, os.cpu_count() or 1)), large=large, extra=list(extra))
    ix = GenomeIndex.load_from_directory(d)
    assert ix.seed_len == seed_len and ix.large == large
    ri = ref.RefIndex(d)
    if extra or seed_len < 20:
        assert int(open(os.path.join(d, "GenomeIndex")).read().split()[9]) > 4          # the files do carry wide locations
        # the index probe on its own: genome seeds and absent ones, counts and lists, both strands
        rng = np.random.default_rng(seed_len)
        cat = np.concatenate([c[1] for c in g])
        starts = rng.integers(0, cat.size - seed_len, size=3000)
        seeds = np.stack([cat[s:s + seed_len] for s in starts] + [synth._ACGT[rng.integers(0, 4, size=seed_len)] for _ in range(300)])
        e_n, e_h = ri.lookup_seeds(seeds, max_hits_out=64)
        a = BaseAligner.from_directory(d, abi.default_params(max_read_len=160, max_k=8)) if from_directory else BaseAligner(ix, abi.default_params(max_read_len=160, max_k=8))
        try:
            g_n, g_h = a.lookupSeed32(seeds, max_hits_out=64)
        finally:
            a.close()
        assert (e_n == g_n).all()
        lim = np.minimum(np.maximum(e_n, 0), 64)
        for i in np.nonzero(lim.max(axis=1) > 0)[0]:
            for dd in range(2):
                assert (e_h[i, dd, :lim[i, dd]] == g_h[i, dd, :lim[i, dd]]).all(), (i, dd)
    p = abi.default_params(max_read_len=160, max_k=8)
    rd = synth.make_reads(5, g, n_reads, 150, sub=0.015, ins=0.002, dele=0.002, n_frac=0.0005)
    with ref.fresh_objects():           # the reference's answer as a function of the read alone: every read is compared
        pr, ar, cr, _ = ri.align_single(p, rd["bases"], rd["quals"], rd["offsets"], threads=max(1, os.cpu_count() or 1))
    a = BaseAligner.from_directory(d, p) if from_directory else BaseAligner(ix, p)
    try:
        pg, ag = a.AlignRead(rd["bases"], rd["quals"], rd["offsets"])
        c = a.counters()
    finally:
        a.close()
    assert (pr["status"] != 0).sum() > 0.9 * n_reads
    problems = util.compare_results(pr, pg)
    assert not problems, problems
    assert [c["n_hash_table_lookups"], c["n_lv_locations"], c["n_ag_locations"]] == [cr["lookups"], cr["lv"], cr["ag"]]


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
@pytest.mark.parametrize("seed_len,large", SHAPES)
def test_index_shapes_vs_live_reference(tmp_path, seed_len, large):
    align_and_compare(str(tmp_path), seed_len, large, 8000, genome_bases=1_000_000)


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
@pytest.mark.parametrize("seed_len,large,extra", WIDE_SHAPES)
def test_wide_location_indexes_vs_live_reference(tmp_path, seed_len, large, extra):
    align_and_compare(str(tmp_path), seed_len, large, 6000, genome_bases=1_000_000, extra=extra, from_directory=seed_len == 18)   # (18: through snapgpu_create_from_directory's narrowing)


def paired_over_wide_index(tmp, seed_len, large, extra, n_pairs):
    from snap_amd.aligner import ChimericPairedEndAligner
    from snap_amd.index import GenomeIndex
    from tests.pairs_util import hard_pairs, compare_paired
    contigs = synth.make_genome(171, 1_000_000, n_contigs=3, repeat_frac=0.35, max_copies=200, n_run_frac=0.002)
    fa = os.path.join(tmp, "g.fa"); synth.write_fasta(fa, contigs)
    d = os.path.join(tmp, "idx")
    ref.build_index(fa, d, seed_len=seed_len, threads=max(1, min(16, os.cpu_count() or 1)), large=large, extra=list(extra))
    assert int(open(os.path.join(d, "GenomeIndex")).read().split()[9]) > 4
    pr = hard_pairs(13, contigs, n_pairs, 150, insert_mean=400, insert_max=1000)
    p = abi.default_params(max_k=8, max_read_len=160); pp = abi.default_paired_params(max_spacing=1000)
    with ref.fresh_objects():
        rp, ra, rcnt, _ = ref.RefIndex(d).align_paired(p, pp, pr["bases"], pr["quals"], pr["offsets"], threads=max(1, os.cpu_count() or 1), stage=0)
    a = ChimericPairedEndAligner(GenomeIndex.load_from_directory(d), p, pp)
    try:
        gp, ga = a.align(pr["bases"], pr["quals"], pr["offsets"]); c = a.counters()
    finally:
        a.close()
    assert not compare_paired(rp, gp, verbose=3).any()
    assert (c["n_lv_locations"], c["n_ag_locations"]) == (rcnt["lv"], rcnt["ag"])


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
@pytest.mark.parametrize("seed_len,large,extra", [(16, False, []), (18, True, [])])
def test_paired_over_wide_location_index_vs_live_reference(tmp_path, seed_len, large, extra):
    paired_over_wide_index(str(tmp_path), seed_len, large, extra, 2000)


# ---- option sets beyond the ones the fixtures pin (-h, -n, -sc, -D, -d, scoring parameters, end bonuses, ALT gap), against the live reference
OPTION_SETS = [dict(max_hits=16), dict(max_hits=2000, max_k=12), dict(num_seeds=5), dict(num_seeds=0, seed_coverage=2.0),
               dict(num_seeds=0, seed_coverage=0.5, max_k=20), dict(min_weight_to_check=2), dict(extra_search_depth=0),
               dict(extra_search_depth=3, max_k=10), dict(max_k=4), dict(max_k=30), dict(max_score_gap_to_prefer_non_alt=0),
               dict(max_score_gap_to_prefer_non_alt=8, emit_alt_alignments=1),
               dict(match_reward=2, sub_penalty=3, gap_open_penalty=5, gap_extend_penalty=2), dict(five_prime_end_bonus=0, three_prime_end_bonus=0),
               dict(five_prime_end_bonus=20, three_prime_end_bonus=3, max_k=15), dict(use_affine_gap=0, max_hits=50, num_seeds=40)]


def option_workload(tmp, n_reads):
    from snap_amd.index import GenomeIndex
    g = synth.make_genome(311, 500_000, n_contigs=3, repeat_frac=0.45, max_copies=300, n_run_frac=0.002)
    rng = np.random.default_rng(3)
    alt = g[0][1][30_000:42_000].copy(); m = rng.random(alt.size) < 0.01; alt[m] = synth._ACGT[rng.integers(0, 4, size=int(m.sum()))]
    g.append(("alt1", alt))
    fa = os.path.join(tmp, "ref.fa"); synth.write_fasta(fa, g)
    d = os.path.join(tmp, "idx")
    ref.build_index(fa, d, 20, threads=max(1, min(8, os.cpu_count() or 1)), extra=["-altContigName", "alt1"])
    return GenomeIndex.load_from_directory(d), ref.RefIndex(d), synth.make_reads(5, g, n_reads, 150, sub=0.02, ins=0.004, dele=0.004, n_frac=0.001)


def check_option_set(ix, ri, rd, kw):
    from snap_amd.aligner import BaseAligner
    p = abi.default_params(max_read_len=160, **kw)
    with ref.fresh_objects():
        pr, ar, cr, _ = ri.align_single(p, rd["bases"], rd["quals"], rd["offsets"], threads=max(1, os.cpu_count() or 1))
    a = BaseAligner(ix, p)
    try:
        pg, ag = a.AlignRead(rd["bases"], rd["quals"], rd["offsets"]); c = a.counters()
    finally:
        a.close()
    problems = util.compare_results(pr, pg)
    assert (ar["status"] == ag["status"]).all()
    sel = ar["status"] != 0
    problems += util.compare_results(ar[sel], ag[sel], "firstALT")
    assert not problems, (kw, problems)
    assert [c["n_hash_table_lookups"], c["n_lv_locations"], c["n_ag_locations"]] == [cr["lookups"], cr["lv"], cr["ag"]], kw


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
def test_option_sets_vs_live_reference(tmp_path):
    ix, ri, rd = option_workload(str(tmp_path), 3000)
    for kw in OPTION_SETS:
        check_option_set(ix, ri, rd, kw)


# ---- paired-end option sets beyond the fixtures (-s, -n, -H as max_big_hits, -i, soft clipping off, ...), against the live reference
PAIRED_OPTION_SETS = [({}, dict(min_spacing=200, max_spacing=500)), ({}, dict(num_seeds=4)), ({}, dict(max_big_hits=200)), (dict(max_k=15), dict(max_k_for_indels=10)),
                      ({}, dict(use_soft_clipping=0)), (dict(use_affine_gap=0), dict(num_seeds=16)), (dict(max_hits=50), {}), ({}, dict(num_seeds=0, seed_coverage=1.5)),
                      (dict(extra_search_depth=3), dict(max_single_seeds=10))]


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
def test_paired_option_sets_vs_live_reference(tmp_path):
    """(Run once on the wavefront emulator with 160 pairs per set: all nine identical.)"""
    from snap_amd.aligner import ChimericPairedEndAligner
    from snap_amd.index import GenomeIndex
    from tests.pairs_util import compare_paired, hard_pairs
    d = str(tmp_path)
    g = synth.make_genome(411, 400_000, n_contigs=3, repeat_frac=0.3, max_copies=60, repeat_len=(150, 1500), n_run_frac=0.002)
    synth.write_fasta(d + "/ref.fa", g)
    ref.build_index(d + "/ref.fa", d + "/idx", 20, threads=max(1, min(8, os.cpu_count() or 1)))
    ix = GenomeIndex.load_from_directory(d + "/idx"); ri = ref.RefIndex(d + "/idx")
    pr = hard_pairs(31, g, 600, 150, insert_mean=380)
    for kw, pkw in PAIRED_OPTION_SETS:
        p = abi.default_params(max_read_len=160, **kw); pp = abi.default_paired_params(**pkw)
        with ref.fresh_objects():
            prim, alt, cnt, _ = ri.align_paired(p, pp, pr["bases"], pr["quals"], pr["offsets"], threads=max(1, os.cpu_count() or 1), stage=0)
        a = ChimericPairedEndAligner(ix, p, pp)
        try:
            got, galt = a.align(pr["bases"], pr["quals"], pr["offsets"])
        finally:
            a.close()
        bad = compare_paired(prim, got, verbose=3)
        assert not bad.any(), (kw, pkw, int(bad.sum()))


# ---- read lengths at the boundaries of the affine-gap kernel variants (64 / 192 / 256 / 384 striped positions) and of max_read_len
@pytest.mark.gpu
@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
def test_read_length_boundaries_and_paired_index_shapes_vs_live_reference(tmp_path):
    """(Run once on the wavefront emulator with 150 reads per length / 200 pairs per shape: all identical.)"""
    from snap_amd.aligner import BaseAligner, ChimericPairedEndAligner
    from snap_amd.index import GenomeIndex
    from tests.pairs_util import compare_paired, hard_pairs
    g = synth.make_genome(511, 400_000, n_contigs=3, repeat_frac=0.3, max_copies=60, repeat_len=(150, 1500), n_run_frac=0.002)
    fa = str(tmp_path / "ref.fa"); synth.write_fasta(fa, g)
    threads = max(1, min(8, os.cpu_count() or 1))
    ref.build_index(fa, str(tmp_path / "idx20"), 20, threads=threads)
    ix = GenomeIndex.load_from_directory(str(tmp_path / "idx20")); ri = ref.RefIndex(str(tmp_path / "idx20"))
    for L, mk, mrl in ((50, 4, 64), (63, 6, 64), (64, 6, 64), (65, 8, 128), (191, 12, 192), (192, 12, 192), (193, 12, 256), (300, 20, 320), (400, 27, 400)):
        rd = synth.make_reads(100 + L, g, 1500, L, sub=0.02, ins=0.004, dele=0.004, n_frac=0.001)
        p = abi.default_params(max_k=mk, max_read_len=mrl)
        with ref.fresh_objects():
            pr, ar, cr, _ = ri.align_single(p, rd["bases"], rd["quals"], rd["offsets"], threads=max(1, os.cpu_count() or 1))
        a = BaseAligner(ix, p)
        try:
            pg, ag = a.AlignRead(rd["bases"], rd["quals"], rd["offsets"])
        finally:
            a.close()
        problems = util.compare_results(pr, pg)
        assert not problems, (L, problems)
    for seed_len, large in ((24, False), (22, True)):                      # the paired-end hit sets over other key sizes / table layouts
        dd = str(tmp_path / ("idx%d%d" % (seed_len, large)))
        ref.build_index(fa, dd, seed_len, threads=threads, large=large)
        ixp = GenomeIndex.load_from_directory(dd); rip = ref.RefIndex(dd)
        prs = hard_pairs(33, g, 600, 150, insert_mean=380)
        p = abi.default_params(max_read_len=160); pp = abi.default_paired_params()
        with ref.fresh_objects():
            prim, alt, cnt, _ = rip.align_paired(p, pp, prs["bases"], prs["quals"], prs["offsets"], threads=max(1, os.cpu_count() or 1), stage=0)
        a = ChimericPairedEndAligner(ixp, p, pp)
        try:
            got, galt = a.align(prs["bases"], prs["quals"], prs["offsets"])
        finally:
            a.close()
        bad = compare_paired(prim, got, verbose=3)
        assert not bad.any(), (seed_len, large, int(bad.sum()))
