"""GPU parity tests of the paired-end path (run with -m gpu on an MI355X), through the C ABI
(snapgpu_enable_paired / snapgpu_align_paired).  Expectations: the committed golden PairedAlignmentResults the compiled
reference produced (scripts/make_golden_paired.py) and, where oracle/_ref travelled to the box, the reference itself on
fresh seeded pairs.  Integer fields must be identical for every aligned read."""
import os

import numpy as np
import pytest

from snap_amd import abi, synth
from snap_amd.index import GenomeIndex
from tests import util
from tests.pairs_util import compare_paired, hard_pairs
from tests.test_paired_host import OPTS
from oracle import ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pindex():
    return util.load_golden_index("paired_index.npz")


@pytest.fixture(scope="module")
def golden_pairs():
    return np.load(os.path.join(util.GOLDEN, "paired_reads.npz"))


def _aligner(index, kw, pkw, max_read_len=160):
    from snap_amd.aligner import ChimericPairedEndAligner
    return ChimericPairedEndAligner(index, abi.default_params(max_read_len=max_read_len, **kw), abi.default_paired_params(**pkw))


@pytest.mark.parametrize("name", list(OPTS))
def test_align_paired_matches_reference_fixture(pindex, golden_pairs, name):
    kw, pkw = OPTS[name]
    a = _aligner(pindex, kw, pkw)
    z = golden_pairs
    for tag in ("150", "100"):
        key = "%s_%s_s0" % (name, tag)
        a.counters(reset=True)
        prim, alt = a.align(z["b" + tag], z["q" + tag], z["o" + tag])
        # every pair, none excluded: the expectation is what reference aligners newly constructed in zero-filled memory answer for
        # the pair (util.with_fresh_overrides); pairs whose answer moves with the reference objects' history must be flagged
        exp, patched = util.with_fresh_overrides(z[key + "_primary"], "pe_" + key + "_primary")
        bad = compare_paired(exp, prim, verbose=3)
        assert not bad.any()
        moved = z[key + "_unstable"] | compare_paired(z[key + "_primary"], exp, verbose=0)
        assert not (moved & (prim["reserved"] == 0)).any(), "reference-unstable pair not flagged"
        e_alt, _ = util.with_fresh_overrides(z[key + "_alt"], "pe_" + key + "_alt")
        assert (alt["status"] == e_alt["status"]).all()
        c = a.counters()
        assert [c["n_lv_locations"], c["n_ag_locations"]] == z[key + "_counters"].tolist()
    a.close()


def test_results_do_not_depend_on_batch_composition(pindex, golden_pairs):
    a = _aligner(pindex, dict(max_k=8), {})
    z = golden_pairs
    prim, _ = a.align(z["b150"], z["q150"], z["o150"])
    n = prim.size
    order = np.random.default_rng(3).permutation(n)[: n // 2]
    o = z["o150"].astype(np.int64)
    bb = np.concatenate([z["b150"][o[2 * i]:o[2 * i + 2]] for i in order])
    qq = np.concatenate([z["q150"][o[2 * i]:o[2 * i + 2]] for i in order])
    lens = np.concatenate([[o[2 * i + 1] - o[2 * i], o[2 * i + 2] - o[2 * i + 1]] for i in order])
    oo = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    sub, _ = a.align(bb, qq, oo)
    assert sub.tobytes() == prim[order].tobytes()
    a.close()


def test_exact_replay_beside_the_main_pass(pindex, golden_pairs, monkeypatch, n=None):
    """launch_paired's exact kernel on its own stream, fed by the main kernel while it runs (PairedArgs::rq): with every fifth pair flagged
    by the test hook, the records equal those of the launch-after-launch scheme byte for byte, equal the fixture, and the internal marker
    does not leave the call.  (Also run twice per aligner: the list, its counters and the images are per call.)"""
    z = golden_pairs
    key = "default_d8_150_s0"
    n = n or (z["o150"].size - 1) // 2
    o = z["o150"][:2 * n + 1]
    b, q = z["b150"].reshape(-1)[:int(o[-1])], z["q150"].reshape(-1)[:int(o[-1])]
    from tests.test_paired_host import OPTS
    kw, pkw = OPTS[key[:-len("_150_s0")]]
    monkeypatch.setenv("SNAPGPU_DEBUG_PAIRED_FLAG_EVERY", "5")
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SNAPGPU_PAIRED_REPLAY_BESIDE", mode)
        a = _aligner(pindex, kw, pkw)
        first, _ = a.align(b, q, o)
        again, _ = a.align(b, q, o)
        a.close()
        assert first.tobytes() == again.tobytes()
        out[mode] = first
    assert out["1"].tobytes() == out["0"].tobytes()
    got = out["1"]
    assert ((got["flags"] & 4) != 0).sum() >= (n + 4) // 5 and (got["flags"] & ~np.uint32(7)).max() == 0      # 4: SNAPGPU_PAIR_EXACT_REPLAY
    exp, _ = util.with_fresh_overrides(z[key + "_primary"], "pe_" + key + "_primary")
    assert not compare_paired(exp[:n], got, verbose=3).any()


@pytest.mark.parametrize("maxk,L,npairs", [(8, 150, 4000), (20, 250, 1500), (27, 150, 1500), (27, 420, 600)])   # 420 bp: the AGC = 0 kernel variant
def test_align_paired_vs_reference_live(tmp_path, maxk, L, npairs):
    """C3 / C5-shaped inputs on a repeat-rich 3 Mb genome, diffed against the reference run on the box's host cores."""
    if not ref.available():
        pytest.skip("oracle/_ref did not travel to this box")
    d = str(tmp_path)
    contigs = synth.make_genome(21 + L, 3_000_000, n_contigs=3, repeat_frac=0.35, max_copies=400, n_run_frac=0.002)
    synth.write_fasta(d + "/g.fa", contigs)
    ref.build_index(d + "/g.fa", d + "/idx", seed_len=20, threads=16)
    rix = ref.RefIndex(d + "/idx")
    gi = GenomeIndex.load_from_directory(d + "/idx")
    pr = hard_pairs(5 + maxk, contigs, npairs, L, insert_mean=400 if L < 200 else (600 if L < 400 else 900), insert_max=1000 if L < 400 else 1400)
    p = abi.default_params(max_k=maxk, max_read_len=L + 10)
    pp = abi.default_paired_params(max_spacing=1000 if L < 400 else 1500)
    with ref.fresh_objects():           # newly constructed reference aligners per pair: the answer is a function of the pair alone
        rp, ra, rcnt, _ = rix.align_paired(p, pp, pr["bases"], pr["quals"], pr["offsets"], threads=16, stage=0)
    from snap_amd.aligner import ChimericPairedEndAligner
    a = ChimericPairedEndAligner(gi, p, pp)
    gp, ga = a.align(pr["bases"], pr["quals"], pr["offsets"])
    bad = compare_paired(rp, gp, verbose=3)                          # every pair, no exclusion
    assert not bad.any()
    c = a.counters()
    assert (c["n_lv_locations"], c["n_ag_locations"]) == (rcnt["lv"], rcnt["ag"])
    a.close()


def test_paired_error_behaviour(pindex, golden_index):
    from snap_amd.aligner import BaseAligner, ChimericPairedEndAligner, SnapGpuError
    import ctypes as C
    # align before enable
    b = BaseAligner(pindex, abi.default_params(max_k=8, max_read_len=160))
    prim = np.zeros(1, dtype=abi.PAIRED_RESULT_DTYPE)
    bases = np.frombuffer(b"ACGT" * 50, dtype=np.uint8).copy(); offs = np.array([0, 100, 200], dtype=np.uint64)
    rc = b.lib.snapgpu_align_paired(b.handle, C.c_uint32(1), abi.ptr(bases), abi.ptr(bases), abi.ptr(offs), abi.ptr(prim), None)
    assert rc == -1 and b"snapgpu_enable_paired" in b.lib.snapgpu_last_error(b.handle)
    b.close()
    a = ChimericPairedEndAligner(golden_index, abi.default_params(max_k=8, max_read_len=160))          # (an ALT index without liftover data is fine)
    # reads longer than max_read_len are rejected before anything is launched
    long_b = np.frombuffer(b"A" * 400, dtype=np.uint8).copy()
    with pytest.raises(SnapGpuError, match="max_read_len"):
        a.align(long_b, long_b, np.array([0, 200, 400], dtype=np.uint64))
    # empty batch
    p0, a0 = a.align(np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert p0.size == 0
    a.close()


@pytest.mark.parametrize("name,kw", [("default_d8", dict(max_k=8)), ("default_d27", dict(max_k=27)), ("emitalt_d8", dict(max_k=8, emit_alt_alignments=1))])
def test_alt_liftover_matches_reference_fixture(name, kw):
    """ALT liftover (IntersectingPairedEndAligner.cpp:2866-2968) on an index built with -altLiftoverFile: see the CPU twin in
    tests/test_paired_host.py."""
    from snap_amd.aligner import ChimericPairedEndAligner
    ix = util.load_golden_index("paired_alt_index.npz")
    z = np.load(os.path.join(util.GOLDEN, "paired_alt_reads.npz"))
    a = ChimericPairedEndAligner(ix, abi.default_params(max_read_len=160, **kw), abi.default_paired_params())
    prim, alt = a.align(z["b"], z["q"], z["o"])
    a.close()
    key = "%s_s0" % name
    fields = ["status", "direction", "location", "score", "mapq", "used_affine_gap_scoring", "bases_clipped_before", "bases_clipped_after",
              "ag_score", "liftover", "aligned_as_pair"]
    exp, _ = util.with_fresh_overrides(z[key + "_primary"], "pealt_" + key + "_primary")
    bad = compare_paired(exp, prim, verbose=3, fields=fields)        # every pair, no exclusion
    assert not bad.any()
    e_alt, _ = util.with_fresh_overrides(z[key + "_alt"], "pealt_" + key + "_alt")
    assert (alt["status"] == e_alt["status"]).all()
    assert prim["liftover"].all(axis=1).sum() == z[key + "_primary"]["liftover"].all(axis=1).sum()


def test_alt_index_without_liftover_data(golden_index, golden_reads):
    """The single-end golden index has an ALT contig but no projection data: pairs built from its reads must still equal the
    reference (the liftover attempt projects to location 0, fails, and the ALT alignment is kept)."""
    if not ref.available():
        pytest.skip("oracle/_ref did not travel to this box")
    pytest.importorskip("snap_amd")
    # (needs the reference live: there is no committed paired fixture for this index)
    import tempfile
    from snap_amd.aligner import ChimericPairedEndAligner
    cs = []
    for i, c in enumerate(golden_index.contigs):
        end = golden_index.contigs[i + 1].begin if i + 1 < len(golden_index.contigs) else golden_index.n_bases
        g = golden_index.genome[c.begin:end]
        cs.append((c.name, g[g != ord("n")].copy()))
    d = tempfile.mkdtemp(prefix="altidx_", dir="/tmp")
    synth.write_fasta(d + "/ref.fa", cs)
    ref.build_index(d + "/ref.fa", d + "/idx", 20, threads=4, extra=["-altContigName", cs[-1][0]])
    gi = GenomeIndex.load_from_directory(d + "/idx")
    pr = hard_pairs(12, cs, 1500, 150, insert_mean=380)
    p = abi.default_params(max_k=8, max_read_len=160); pp = abi.default_paired_params()
    with ref.fresh_objects():
        rp, ra, _, _ = ref.RefIndex(d + "/idx").align_paired(p, pp, pr["bases"], pr["quals"], pr["offsets"], threads=8, stage=0)
    a = ChimericPairedEndAligner(gi, p, pp)
    gp, ga = a.align(pr["bases"], pr["quals"], pr["offsets"])
    a.close()
    bad = compare_paired(rp, gp, verbose=3)
    assert not bad.any()
    assert (ra["status"] == ga["status"]).all()


# ---------------------------------------------------------------------------------------- secondary results (-om / -omax / -mpc)

@pytest.mark.parametrize("tag", ["150", "100"])
def test_paired_secondary_results_vs_reference_fixture(tag):
    """ChimericPairedEndAligner::align with secondary results through the C ABI against tests/golden/paired_secondary.npz:
    primary, paired secondary results (order included) and the single-end secondary results of the chimeric fallback."""
    from snap_amd.aligner import ChimericPairedEndAligner
    from tests.pairs_util import compare_paired_secondary, load_paired_secondary_sets
    z = np.load(os.path.join(util.GOLDEN, "paired_secondary.npz"))
    gp = np.load(os.path.join(util.GOLDEN, "paired_reads.npz"))
    gi = util.load_golden_index("paired_index.npz")
    b, q, o = gp["b" + tag], gp["q" + tag], gp["o" + tag]
    if tag == "150":
        o = o[:1201]; b = b[:int(o[-1])]; q = q[:int(o[-1])]
    for name, kw, pkw, om, omax, mpc in load_paired_secondary_sets(z):
        key = "%s_%s_" % (name, tag)
        ref_t = tuple(util.with_fresh_overrides(z[key + k], "pesec_" + key + k)[0] for k in ("primary", "alt", "secondary", "nsec", "single_secondary", "nssec"))
        a = ChimericPairedEndAligner(gi, abi.default_params(max_read_len=160, **kw), abi.default_paired_params(**pkw))
        try:
            a.enable_secondary(om, max_results=omax, max_per_contig=mpc)
            got = a.align_secondary(b, q, o, stride=2, single_stride=4)          # small strides: exercises the grow-and-recall path
            plain, _ = a.align(b[:int(o[200])], q[:int(o[200])], o[:201])         # the default kernel still runs on the same context
        finally:
            a.close()
        # excluded: only SNAPGPU_PAIR_REF_BUFFER_DEPENDENT pairs -- a reference bug (the ignored return value of ChimericPairedEndAligner.cpp:339)
        # whose outcome is decided by how far earlier pairs had grown the caller's buffer, not by the pair (include/snapgpu.h)
        exclude = (got[0]["flags"] & 2) != 0
        assert int(exclude.sum()) <= 2 + got[0].size // 100, name
        assert not compare_paired(ref_t[0], got[0], verbose=3, exclude=exclude).any(), name
        problems = compare_paired_secondary(ref_t, got, exclude)
        assert not problems, (name, problems)
        assert int(got[3].sum()) > 0 and int(got[5].sum()) > 0
        assert not (plain["flags"] & 1).any()


def test_paired_secondary_needs_both_enabled():
    from snap_amd.aligner import ChimericPairedEndAligner, SnapGpuError
    gi = util.load_golden_index("paired_index.npz")
    a = ChimericPairedEndAligner(gi, abi.default_params(max_k=8, max_read_len=160), abi.default_paired_params())
    try:
        with pytest.raises(SnapGpuError):
            a.align_secondary(np.zeros(200, np.uint8) + 65, np.zeros(200, np.uint8) + 70, np.array([0, 100, 200], np.uint64))
    finally:
        a.close()


def test_hamming_fallback_ties_are_broken_by_the_last_bit_of_the_seed_probability():
    """A pair found in round 5 by scripts/emu_paired_hits_check.py (a genome of high-copy repeats, reads with long deletions): the chimeric
    fallback's Hamming retry keeps 78 candidates that BaseAligner::alignAffineGap rescores to the same score, and which of them wins is
    decided by the LAST BIT of the match probability -- where BaseAligner::scoreLocationWithAffineGap's `pow(1 - SNP_PROB, seedLen)`
    (BaseAligner.cpp:907: seedLen is the unsigned member there, so libm's pow) is one ulp below the `pow(double, int)` = powi of
    BaseAligner.cpp:1314.  tests/golden/hamming_tie_pair.npz holds the pair; the genome is regenerated from its seed, the expectation is the
    compiled reference with fresh aligner objects."""
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not on this box")
    import shutil
    import tempfile
    from snap_amd.aligner import ChimericPairedEndAligner
    from snap_amd.index import GenomeIndex
    z = np.load(os.path.join(util.GOLDEN, "hamming_tie_pair.npz"))
    d = tempfile.mkdtemp(prefix="hamtie")
    try:
        g = synth.make_genome(23, 1_500_000, n_contigs=2, repeat_frac=0.75, max_copies=900, repeat_len=(300, 1500), max_divergence=0.02)
        synth.write_fasta(d + "/g.fa", g)
        ref.build_index(d + "/g.fa", d + "/idx", 20, threads=8)
        params, pparams = abi.default_params(max_k=12, max_read_len=160), abi.default_paired_params(num_seeds=16)
        with ref.fresh_objects():
            exp = ref.RefIndex(d + "/idx").align_paired(params, pparams, z["b"], z["q"], z["o"], threads=1, stage=0)[0]
        assert exp["aligned_as_pair"][0] == 0 and exp["status"][0][1] == 2 and exp["location"][0][1] == 1429937        # the case is the case
        a = ChimericPairedEndAligner(GenomeIndex.load_from_directory(d + "/idx"), params, pparams)
        try:
            got, _ = a.align(z["b"], z["q"], z["o"])
        finally:
            a.close()
        assert not compare_paired(exp, got, verbose=3).any()
    finally:
        shutil.rmtree(d, ignore_errors=True)
