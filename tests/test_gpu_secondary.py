"""GPU parity of BaseAligner::AlignRead WITH secondary results (-om / -omax / -mpc; BaseAligner.cpp:2143-2299, 2423-2553) against
what the compiled reference returned for the same reads (tests/golden/secondary_reads.npz, scripts/make_golden_secondary.py).
Order of the secondary results matters (it is the order the reference writes them to SAM), so records are compared position by
position.  Bit-exact."""
import ast

import numpy as np
import pytest

from snap_amd import abi
from tests import util
from tests.util import compare_secondary

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden_secondary():
    import os
    return np.load(os.path.join(util.GOLDEN, "secondary_reads.npz"))


def _sets(z):
    return [(str(r[0]), ast.literal_eval(str(r[1])), int(r[2]), int(r[3]), int(r[4])) for r in z["sets"]]


@pytest.mark.parametrize("tag", ["100", "150"])
def test_secondary_results_vs_reference_fixture(golden_index, golden_reads, golden_secondary, tag):
    from snap_amd.aligner import BaseAligner
    z = golden_secondary
    b, q = golden_reads["b" + tag], golden_reads["q" + tag]
    n, L = b.shape
    offs = np.arange(n + 1, dtype=np.uint64) * L
    for name, kw, om, omax, mpc in _sets(z):
        a = BaseAligner(golden_index, abi.default_params(max_read_len=160, **kw))
        try:
            a.enable_secondary(om, max_results=omax, max_per_contig=mpc)
            prim, alt, sec, nsec = a.AlignReadSecondary(b, q, offs, stride=4)        # small stride: exercises the grow-and-recall path
        finally:
            a.close()
        key = "%s_%s_" % (name, tag)
        # every read, none excluded: the expectation is the fresh-object reference (util.with_fresh_overrides)
        e_prim, patched = util.with_fresh_overrides(z[key + "primary"], "sec_" + key + "primary")
        e_sec, p2 = util.with_fresh_overrides(z[key + "secondary"], "sec_" + key + "secondary")
        e_nsec, p3 = util.with_fresh_overrides(z[key + "nsec"], "sec_" + key + "nsec")
        moved = z[key + "unstable"].copy(); moved[patched] = True; moved[p2] = True; moved[p3] = True
        assert not (moved & (prim["reserved"] == 0)).any(), "reference-unstable read not flagged"
        problems = util.compare_results(e_prim, prim, "primary")
        problems += compare_secondary(e_sec, e_nsec, sec, nsec, np.zeros(n, bool))
        assert not problems, (name, problems)
        assert int(nsec.sum()) > 0


def test_secondary_is_what_the_plain_call_returns_when_nothing_qualifies(golden_index, golden_reads):
    """A read with one candidate has no secondary results, and then -om changes nothing (the 4.9 early-out cannot fire)."""
    from snap_amd.aligner import BaseAligner
    b, q = golden_reads["b100"][:600], golden_reads["q100"][:600]
    offs = np.arange(b.shape[0] + 1, dtype=np.uint64) * 100
    a = BaseAligner(golden_index, abi.default_params(max_k=8, max_read_len=160))
    try:
        plain, _ = a.AlignRead(b, q, offs)
        a.enable_secondary(1)
        prim, _, sec, nsec = a.AlignReadSecondary(b, q, offs)
        again, _ = a.AlignRead(b, q, offs)                    # the default kernel is untouched by enable_secondary
    finally:
        a.close()
    none = nsec == 0
    assert none.sum() > 300
    assert not util.compare_results(plain[none], prim[none], "primary of reads without secondary results")
    assert not util.compare_results(plain, again, "plain call after enable_secondary")


def test_secondary_argument_errors(golden_index):
    from snap_amd.aligner import BaseAligner, SnapGpuError
    a = BaseAligner(golden_index, abi.default_params(max_k=8, max_read_len=160))
    try:
        with pytest.raises(SnapGpuError):
            a.enable_secondary(2)                             # -om > -D (AlignerContext.cpp:784)
        with pytest.raises(SnapGpuError):
            a.enable_secondary(1, max_results=0)
        with pytest.raises(SnapGpuError):
            a.enable_secondary(1, max_per_contig=0)
        with pytest.raises(SnapGpuError):                     # not enabled
            a.AlignReadSecondary(np.zeros(100, np.uint8) + 65, np.zeros(100, np.uint8) + 70, np.array([0, 100], np.uint64))
    finally:
        a.close()
