"""-f (stopOnFirstHit) and -x (explorePopularSeeds) of the single-end aligner: BaseAligner::setStopOnFirstHit / setExplorePopularSeeds as
SingleAligner.cpp:179-180 applies them (BaseAligner.cpp:574, :625, :1490-1505) against the compiled reference running with the same flags,
on a repeat-rich genome (seeds with far more than -h hits, so that -x changes what is found) -- every read, every field."""
import os

import numpy as np
import pytest

from snap_amd import abi, synth
from tests import util

FLAG_SETS = [(True, False), (False, True), (True, True)]


def check_flags_vs_live_reference(tmp_path, n_reads=6000, genome_bases=1_500_000, with_secondary=True):
    from oracle import ref
    from snap_amd.aligner import BaseAligner
    from snap_amd.index import GenomeIndex
    g = synth.make_genome(4711, genome_bases, n_contigs=2, repeat_frac=0.5, max_copies=900, repeat_len=(150, 1200), max_divergence=0.03)
    fa = str(tmp_path / "ref.fa"); synth.write_fasta(fa, g)
    ref.build_index(fa, str(tmp_path / "idx"), 20, threads=max(1, os.cpu_count() or 1))
    ix = GenomeIndex.load_from_directory(str(tmp_path / "idx"))
    ri = ref.RefIndex(str(tmp_path / "idx"))
    p = abi.default_params(max_k=8, max_read_len=160)
    p.max_hits = 40                                     # (-h 40: plenty of seeds of this genome are more popular than that)
    rd = synth.make_reads(4712, g, n_reads, 120, sub=0.02, ins=0.001, dele=0.001)
    threads = os.cpu_count() or 1
    with ref.fresh_objects():
        base, _, _, _ = ri.align_single(p, rd["bases"], rd["quals"], rd["offsets"], threads=threads)
    changed = {}
    for f, x in FLAG_SETS:
        with ref.fresh_objects(), ref.aligner_flags(stop_on_first_hit=f, explore_popular_seeds=x):
            pr, ar, _, _ = ri.align_single(p, rd["bases"], rd["quals"], rd["offsets"], threads=threads)
            if with_secondary:
                prs, ars, srs, nrs = ri.align_single_secondary(p, 1, rd["bases"][:1500], rd["quals"][:1500], rd["offsets"][:1501], threads=threads)
        a = BaseAligner(ix, p)
        try:
            a.set_flags(stop_on_first_hit=f, explore_popular_seeds=x)
            pg, ag = a.AlignRead(rd["bases"], rd["quals"], rd["offsets"])
            assert not util.compare_results(pr, pg), (f, x)
            assert not util.compare_results(ar, ag, what="first_alt"), (f, x)
            if with_secondary:
                a.enable_secondary(1)
                pgs, ags, sgs, ngs = a.AlignReadSecondary(rd["bases"][:1500], rd["quals"][:1500], rd["offsets"][:1501])
                assert not util.compare_results(prs, pgs), (f, x, "with -om 1")
                assert not util.compare_secondary(srs, nrs, sgs, ngs, np.zeros(len(nrs), bool)), (f, x, "with -om 1")
        finally:
            a.close()
        changed[(f, x)] = int(sum(1 for k in ("status", "location", "score", "mapq") if (pr[k] != base[k]).any()))
        if f:
            found = pr["status"] != 0
            assert (pr["mapq"][found] == 0).all() and (pr["status"][found] == 2).all()      # :1500-1501
    assert all(v > 0 for v in changed.values()), changed     # (the flags do change answers on this workload: the test is not vacuous)
    return changed


@pytest.mark.gpu
def test_stop_on_first_hit_and_explore_popular_seeds_vs_live_reference(tmp_path):
    check_flags_vs_live_reference(tmp_path)
