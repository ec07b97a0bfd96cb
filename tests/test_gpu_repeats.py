"""Repeat-heavy parity on the MI355X: the one parity bug of the project's life -- a 78-way tie between equally good
candidates of BaseAligner::alignAffineGap, decided by one ulp of pow(1 - SNP_PROB, seedLen) (BaseAligner.cpp:907 against :1314) -- sat
undetected for four rounds because no fixture had LONG lists of EQUAL candidates.  scripts/emu_paired_hits_check.py found it on the
emulator; these are its four configurations (a genome of high-copy repeats; both hit-window sizes; a narrow spacing; long deletions) as
`-m gpu` tests against the live reference at >= 2 000 pairs each, a configuration built to produce the ties themselves (EXACT copies of a
repeat, one mate unalignable, the other with a garbage tail: the chimeric fallback's Hamming retry hands alignAffineGap dozens of
candidates with the same score and the same affine-gap score, compareByScore at BaseAligner.cpp:1719), and the single-end twin: reads
out of the same repeat families through BaseAligner::AlignRead.  Every unit, every field, no exclusions; the expectation is the reference
with newly constructed aligner objects (oracle/ref.py: fresh_objects), i.e. a function of the unit alone."""
import numpy as np
import pytest

from snap_amd import abi, synth
from snap_amd.index import GenomeIndex
from tests import util
from tests.pairs_util import compare_paired
from oracle import ref

pytestmark = pytest.mark.gpu


def _genome(exact):
    if exact:       # EXACT copies (divergence 0): every copy of a family scores the same, so candidates tie in score, agScore and probability
        return synth.make_genome(29, 1_200_000, n_contigs=2, repeat_frac=0.7, max_copies=400, repeat_len=(400, 1200), max_divergence=0.0)
    return synth.make_genome(23, 1_500_000, n_contigs=2, repeat_frac=0.75, max_copies=900, repeat_len=(300, 1500), max_divergence=0.02)


@pytest.fixture(scope="module")
def repeat_bed(tmp_path_factory):
    if not ref.available():
        pytest.skip("oracle/_ref did not travel to this box")
    beds = {}

    def get(exact):
        if exact not in beds:
            d = str(tmp_path_factory.mktemp("rep%d" % int(exact)))
            g = _genome(exact)
            synth.write_fasta(d + "/g.fa", g)
            ref.build_index(d + "/g.fa", d + "/idx", 20, threads=16)
            beds[exact] = (g, GenomeIndex.load_from_directory(d + "/idx"), ref.RefIndex(d + "/idx"))
        return beds[exact]
    return get


CONFIGS = {
    "n8": (dict(max_k=8), {}, {}),                                                             # 16 staged hits per lookup
    "coverage": (dict(max_k=8), dict(num_seeds=0, seed_coverage=4.0), {}),                     # 30 lookups per set, 8 staged hits each
    "narrow": (dict(max_k=8), dict(min_spacing=50, max_spacing=350, num_seeds=12), {}),        # many getNextHitLessThanOrEqualTo jumps
    "indels": (dict(max_k=12), dict(num_seeds=16), dict(long_indel_frac=0.8, long_indel_max=30)),   # seed-hinted indel limits (Phase 2a)
}


@pytest.mark.parametrize("tag", list(CONFIGS))
def test_paired_long_hit_lists_vs_reference_live(repeat_bed, tag, n=2000):
    from snap_amd.aligner import ChimericPairedEndAligner
    g, gi, rix = repeat_bed(False)
    kw, pkw, mkw = CONFIGS[tag]
    pairs = synth.make_pairs(7 + len(tag), g, n, 150, **mkw)
    params, pparams = abi.default_params(max_read_len=160, **kw), abi.default_paired_params(**pkw)
    with ref.fresh_objects():
        exp, _, rcnt, _ = rix.align_paired(params, pparams, pairs["bases"], pairs["quals"], pairs["offsets"], threads=32, stage=0)
    a = ChimericPairedEndAligner(gi, params, pparams)
    try:
        a.counters(reset=True)
        got, _ = a.align(pairs["bases"], pairs["quals"], pairs["offsets"])
        c = a.counters()
    finally:
        a.close()
    assert not compare_paired(exp, got, verbose=3).any()
    assert (c["n_lv_locations"], c["n_ag_locations"]) == (rcnt["lv"], rcnt["ag"])
    assert c["n_hits_consumed"] / (2 * n) > 200          # the lists ARE long


def _tie_pairs(seed, g, n, L=150):
    """Pairs that reach BaseAligner::alignAffineGap with many equal candidates: mate 0 out of an exact-copy repeat family with a garbage
    tail (soft clipping: Landau-Vishkin within maxK/2 fails, the Hamming retry places the head in EVERY copy), mate 1 random bases (the
    pair cannot be aligned as a pair, so ChimericPairedEndAligner falls back to the single-end aligner for both)."""
    pr = synth.make_pairs(seed, g, n, L, sub=0.0, ins=0.0, dele=0.0)
    rng = np.random.default_rng(seed + 1)
    b = pr["bases"]
    for i in range(n):
        k = int(rng.integers(25, 60))
        b[2 * i, L - k:] = synth._ACGT[rng.integers(0, 4, size=k, dtype=np.uint8)]
        if i % 4 != 3:          # (a quarter keep their mate: the intersecting aligner's own Hamming phase sees the same families)
            b[2 * i + 1] = synth._ACGT[rng.integers(0, 4, size=L, dtype=np.uint8)]
    return pr


def test_hamming_fallback_many_way_ties_vs_reference_live(repeat_bed, n=2000):
    from snap_amd.aligner import ChimericPairedEndAligner
    g, gi, rix = repeat_bed(True)
    pairs = _tie_pairs(41, g, n)
    params, pparams = abi.default_params(max_k=8, max_read_len=160), abi.default_paired_params()
    with ref.fresh_objects():
        exp, _, rcnt, _ = rix.align_paired(params, pparams, pairs["bases"], pairs["quals"], pairs["offsets"], threads=32, stage=0)
    a = ChimericPairedEndAligner(gi, params, pparams)
    try:
        a.counters(reset=True)
        got, _ = a.align(pairs["bases"], pairs["quals"], pairs["offsets"])
        c = a.counters()
    finally:
        a.close()
    assert not compare_paired(exp, got, verbose=3).any()
    assert (c["n_lv_locations"], c["n_ag_locations"]) == (rcnt["lv"], rcnt["ag"])
    # the workload does what it was built for: clipped single-end placements with MAPQ <= 3 (dozens of equally good copies) are common
    clipped = (got["status"][:, 0] != 0) & (got["bases_clipped_after"][:, 0] + got["bases_clipped_before"][:, 0] > 0)
    assert clipped.sum() >= n // 2
    assert (got["mapq"][clipped, 0] <= 3).mean() > 0.25          # ... and a good part of them had equally good placements in other copies


@pytest.mark.parametrize("exact", [False, True])
def test_single_end_repeat_families_vs_reference_live(repeat_bed, exact, n=20000):
    from snap_amd.aligner import BaseAligner
    g, gi, rix = repeat_bed(exact)
    rd = synth.make_reads(11 + int(exact), g, n, 150)
    offs = np.arange(n + 1, dtype=np.uint64) * 150
    params = abi.default_params(max_k=8, max_read_len=160)
    with ref.fresh_objects():
        exp = rix.align_single(params, rd["bases"].reshape(-1), rd["quals"].reshape(-1), offs, threads=32)[0]
    a = BaseAligner(gi, params)
    try:
        got, _ = a.AlignRead(rd["bases"].reshape(-1), rd["quals"].reshape(-1), offs)
    finally:
        a.close()
    assert not util.compare_results(exp, got)
    # repeat families: many reads have equally good placements (diverged copies: 18 % of the aligned reads on this genome; exact copies: more)
    assert (got["mapq"][got["status"] != 0] <= 3).mean() > (0.2 if exact else 0.1)
