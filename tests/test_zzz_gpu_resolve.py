"""SNAPGPU_SINGLE_RESOLVE=1: the single-end fast form that answers an affine-gap call whose traceback leaves its band ON THE SPOT, from the
list of the object's earlier calls of the read (snap_amd/csrc/ag_resolve.h; align_single.h: Aligner<.., RESOLVE>) -- no traceback images,
no replay pass.  The reads of the fixture that the default path sends to the exact replay must come out as the reference (aligner objects
newly constructed per read) answers them with the replay switched OFF.

Verified on the wavefront emulator (tests/test_emu_kernels.py), the resolver itself on hardware at the level of call sequences
(tests/test_gpu_parity.py::test_affine_gap_call_sequences_without_an_image), and -- round 4, profiles/r04a -- this instantiation on the
MI355X: the test passes, and the bench batch (1 M reads, three feeders and one) comes out bit-identical to the reference with nothing
flagged.  It stays opt-in: 1 468 B of scratch per lane make it 3x slower than the default kernels (profiles/r04a)."""
import os

import numpy as np
import pytest

from snap_amd import abi
from tests import util


def check_resolve_on_fixture(golden_index, golden_reads, monkeypatch, sets=(("default_d8", dict(max_k=8)), ("default_d27", dict(max_k=27))), n_reads=None):
    from snap_amd.aligner import BaseAligner
    z = golden_reads
    b, q = z["b100"][:n_reads], z["q100"][:n_reads]
    n, L = b.shape
    offs = np.arange(n + 1, dtype=np.uint64) * L
    n_flagged_without = 0
    for name, kw in sets:
        key = "%s_100_" % name
        exp, _ = util.with_fresh_overrides(z[key + "primary"], key + "primary")
        out = {}
        for mode, env in (("fast", {"SNAPGPU_SINGLE_HELP": "1", "SNAPGPU_NO_EXACT_REPLAY": "1"}),
                          ("resolve", {"SNAPGPU_SINGLE_RESOLVE": "1", "SNAPGPU_NO_EXACT_REPLAY": "1"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            a = BaseAligner(golden_index, abi.default_params(max_read_len=160, **kw))
            try:
                out[mode], _ = a.AlignRead(b, q, offs)
            finally:
                a.close()
            for k in env:
                monkeypatch.delenv(k)
        n_flagged_without += int(((out["fast"]["reserved"] >> 30) & 1).sum())
        assert not util.compare_results(exp[:n], out["resolve"]), name              # every read, the replay switched off
        assert (((out["resolve"]["reserved"] >> 30) & 1) == 0).all(), name           # nothing left for a replay
    return n_flagged_without


@pytest.mark.gpu
def test_calls_leaving_the_band_are_answered_in_place(golden_index, golden_reads, monkeypatch):
    assert check_resolve_on_fixture(golden_index, golden_reads, monkeypatch) >= 2       # (the fixture does hold reads the fast form alone gets flagged)
