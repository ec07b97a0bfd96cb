"""SURVEY.md section 8(f) rank 2, the 2-bit half: the bit-plane shadow of the genome and the plane Landau-Vishkin (planes.h,
single_planes_k.hip; SNAPGPU_LV_PLANES=1) in the driver's GPU suite -- until round 5 only the emulator ran them."""
import pytest

pytestmark = pytest.mark.gpu


def test_plane_landau_vishkin_instantiations_on_the_gpu(golden_index, golden_reads, monkeypatch):
    """The first 1 000 golden reads per length through the instantiations that carry the plane Landau-Vishkin (fast form + help, exact
    form): every field against the reference's fixture, and the same bytes as a context without planes."""
    import tests.test_emu_kernels as tek
    tek.test_emu_plane_landau_vishkin_instantiations(None, golden_index, golden_reads, monkeypatch)


def test_paired_end_over_the_plane_shadow(monkeypatch):
    """The paired-end kernel of a context created under SNAPGPU_LV_PLANES=1 (its Landau-Vishkin calls take the byte form; the shadow is
    built and carried): the golden pairs, every field."""
    import os
    import numpy as np
    import tests.test_gpu_paired as gp
    from tests import util
    from tests.pairs_util import compare_paired
    from tests.test_paired_host import OPTS
    monkeypatch.setenv("SNAPGPU_LV_PLANES", "1")
    name = list(OPTS)[0]
    kw, pkw = OPTS[name]
    z = np.load(os.path.join(util.GOLDEN, "paired_reads.npz"))
    n = 400
    o = z["o150"][:2 * n + 1]
    a = gp._aligner(util.load_golden_index("paired_index.npz"), kw, pkw)
    try:
        prim, _ = a.align(z["b150"].reshape(-1)[:int(o[-1])], z["q150"].reshape(-1)[:int(o[-1])], o)
    finally:
        a.close()
    key = "%s_150_s0" % name
    exp, _ = util.with_fresh_overrides(z[key + "_primary"], "pe_" + key + "_primary")
    assert not compare_paired(exp[:n], prim, verbose=3).any()
