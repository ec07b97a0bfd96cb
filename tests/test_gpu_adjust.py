"""`-ae`: AlignmentAdjuster::AdjustAlignment (SNAPLib/AlignmentAdjuster.cpp:33-190) on the device -- snapgpu_adjust_alignments item by
item, and inside BaseAligner::AlignRead's finalizeSecondaryResults (BaseAligner.cpp:2444-2463, snapgpu_enable_secondary with
adjust_alignments = 1) -- against tests/golden/adjust.npz (scripts/make_golden_adjust.py, the compiled reference) and against the
reference run live on a fresh genome.  Bit-exact, every item, every read."""
import os

import numpy as np
import pytest

from snap_amd import abi
from tests import util, adjust_util

pytestmark = pytest.mark.gpu
FIELDS = ("status", "location", "score", "clipping_for_read_adjustment")


@pytest.fixture(scope="module")
def golden_adjust():
    return np.load(os.path.join(util.GOLDEN, "adjust.npz"))


def _diff(exp, got):
    return [(f, int(i), int(exp[f][i]), int(got[f][i])) for f in FIELDS for i in np.nonzero(exp[f] != got[f])[0][:5]]


def test_adjust_alignments_vs_reference_fixture(golden_index, golden_adjust):
    from snap_amd.aligner import BaseAligner
    z = golden_adjust
    b = z["case_bases"]; n, L = b.shape
    a = BaseAligner(golden_index, abi.default_params(max_k=8, max_read_len=160))
    try:
        got = a.AdjustAlignments(b, np.arange(n, dtype=np.uint64) * L, np.full(n, L, np.int32), z["case_in"])
        again = a.AdjustAlignments(b, np.arange(n, dtype=np.uint64) * L, np.full(n, L, np.int32), got)       # adjusted results need no second adjustment ...
    finally:
        a.close()
    exp = z["case_out"]
    assert not _diff(exp, got)
    assert (exp["clipping_for_read_adjustment"] != 0).sum() > 300 and ((exp["status"] == 0) & (z["case_in"]["status"] != 0)).sum() > 100
    keep = (got["status"] != 0) & (got["clipping_for_read_adjustment"] == 0)                                 # (... where the Read carries no clipping into it)
    assert (again["location"][keep] == got["location"][keep]).all() and (again["score"][keep] == got["score"][keep]).all()


def test_secondary_with_adjustment_vs_reference_fixture(golden_index, golden_adjust, min_changed=50):
    from snap_amd.aligner import BaseAligner
    z = golden_adjust
    b, q = z["read_bases"], z["read_quals"]; n, L = b.shape
    offs = np.arange(n + 1, dtype=np.uint64) * L
    a = BaseAligner(golden_index, abi.default_params(max_k=10, max_read_len=160, extra_search_depth=2))
    try:
        a.enable_secondary(1, adjust_alignments=1)
        prim, alt, sec, nsec = a.AlignReadSecondary(b, q, offs, stride=2)
        a.enable_secondary(1)                                                  # and back: the same context without -ae
        plain, _, _, nsec0 = a.AlignReadSecondary(b, q, offs, stride=2)
    finally:
        a.close()
    problems = util.compare_results(z["primary"], prim, "primary")
    problems += util.compare_secondary(z["secondary"], z["nsec"], sec, nsec, np.zeros(n, bool))
    assert not problems, problems[:10]
    assert (prim["score"] != plain["score"]).sum() > min_changed               # the adjuster did something on this input
    assert (plain["clipping_for_read_adjustment"] == 0).all()


def test_adjustment_vs_reference_live(tmp_path):
    """a fresh genome of 30 short contigs: 20 000 made-up results through snapgpu_adjust_alignments, 4 000 reads through AlignRead with
    -om 2 -ae, against the reference on the box's host cores"""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref did not travel to this box")
    from snap_amd import synth
    from snap_amd.aligner import BaseAligner
    from snap_amd.index import GenomeIndex
    d = str(tmp_path)
    contigs = synth.make_genome(78, 400_000, n_contigs=30, repeat_frac=0.3, max_copies=30, repeat_len=(150, 600), max_divergence=0.03)
    synth.write_fasta(d + "/g.fa", contigs)
    ref.build_index(d + "/g.fa", d + "/idx", seed_len=20, threads=8)
    gi = GenomeIndex.load_from_directory(d + "/idx")
    rix = ref.RefIndex(d + "/idx")
    L = 100
    cb, cres = adjust_util.adjust_cases(11, contigs, [int(c.begin) for c in gi.contigs], 20000, L)
    off = np.arange(cb.shape[0], dtype=np.uint64) * L; length = np.full(cb.shape[0], L, np.int32)
    rb, rq = adjust_util.adjust_reads(12, contigs, 4000, L)
    roff = np.arange(rb.shape[0] + 1, dtype=np.uint64) * L
    p = abi.default_params(max_k=10, max_read_len=L + 10, extra_search_depth=2)
    exp = rix.adjust_alignments(cb, off, length, cres)
    with ref.fresh_objects(), ref.adjust_alignments():
        e_prim, _, e_sec, e_nsec = rix.align_single_secondary(p, 2, rb, rq, roff, threads=16)
    a = BaseAligner(gi, p)
    try:
        got = a.AdjustAlignments(cb, off, length, cres)
        a.enable_secondary(2, adjust_alignments=1)
        prim, _, sec, nsec = a.AlignReadSecondary(rb, rq, roff, stride=8)
    finally:
        a.close()
    assert not _diff(exp, got)
    problems = util.compare_results(e_prim, prim, "primary") + util.compare_secondary(e_sec, e_nsec, sec, nsec, np.zeros(rb.shape[0], bool))
    assert not problems, problems[:10]


def test_adjustment_is_single_end_only(golden_index):
    from snap_amd.aligner import BaseAligner, ChimericPairedEndAligner, SnapGpuError
    a = ChimericPairedEndAligner(golden_index, abi.default_params(max_k=8, max_read_len=160), abi.default_paired_params())
    try:
        with pytest.raises(SnapGpuError, match="paired"):
            a.enable_secondary(1, adjust_alignments=1)                         # IntersectingPairedEndAligner.cpp:1298-1320 is not built
    finally:
        a.close()
    a = BaseAligner(golden_index, abi.default_params(max_k=8, max_read_len=160))
    try:
        res = np.zeros(1, dtype=abi.RESULT_DTYPE); res["status"] = 1; res["location"] = 1 << 40
        with pytest.raises(SnapGpuError):
            a.AdjustAlignments(np.zeros(100, np.uint8) + 65, np.array([0], np.uint64), np.array([100], np.int32), res)     # location outside the genome
    finally:
        a.close()
