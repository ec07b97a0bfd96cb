"""The device kernels, executed on the host by the wavefront emulator (tests/emu/), against the same fixtures and with the
same test bodies as the `-m gpu` parity tests.

What this adds to the CPU suite: the *device* code -- lane-parallel probes, Landau-Vishkin, the DPP/bpermute affine-gap
rows, the candidate table, the whole `k_align_single` / `k_align_paired` control flow -- runs here without a GPU, one fiber
per lane, so a change to a kernel is checked for parity before any GPU time is spent on it.  What it does not replace: the
`-m gpu` tests (the real compiler, the real hardware), which remain the parity gate.

Test infrastructure only: the emulator library is built from snap_amd/csrc by tests/emu/build.py into tests/emu/_build and
is loaded here by swapping the handle inside snap_amd.aligner for the duration of this module; the product never sees it."""
import ctypes as C
import os
import shutil

import numpy as np
import pytest

from snap_amd import abi
from tests import util

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ needed to build the wavefront emulator")


@pytest.fixture(scope="module")
def emu():
    import snap_amd.aligner as al
    from tests.emu.build import build
    path = build()
    saved = (al._lib, al.LIB_PATH)
    os.environ.setdefault("SNAPGPU_EMU_CUS", "4")         # 4 "compute units": 32 waves in flight over the host threads
    al._lib, al.LIB_PATH = None, path
    try:
        lib = al.load_library()
        for f in ("emu_total_ops", "emu_partial_ops", "emu_inactive_reads"):
            getattr(lib, f).restype = C.c_ulonglong
        yield lib
    finally:
        al._lib, al.LIB_PATH = saved


@pytest.fixture(scope="module")
def emu_aligner(emu, golden_index):
    from snap_amd.aligner import BaseAligner
    a = BaseAligner(golden_index, abi.default_params(max_k=8, max_read_len=160))
    yield a
    a.close()


def test_emulated_library_exports_the_c_abi(emu):
    from snap_amd.aligner import EXPORTED_SYMBOLS
    for s in EXPORTED_SYMBOLS:
        assert hasattr(emu, s), s
    assert emu.snapgpu_abi_version() == 4


def test_emu_tables_and_seed_lookup(emu_aligner, golden_primitives):
    import tests.test_gpu_parity as gp
    gp.test_tables_match_restatement(emu_aligner)
    gp.test_lookup_seeds_vs_reference_fixture(emu_aligner, golden_primitives)
    gp.test_lookup_kernels_agree_counts_only_and_odd_batch_sizes(emu_aligner, golden_primitives)


def test_emu_landau_vishkin(emu_aligner, golden_primitives):
    import tests.test_gpu_parity as gp
    gp.test_lv_known_answers_and_fixture(emu_aligner, golden_primitives)


def test_emu_affine_gap(emu, emu_aligner, golden_index, golden_primitives):
    import tests.test_gpu_parity as gp
    gp.test_affine_gap_known_answers(golden_index)
    gp.test_affine_gap_vs_reference_fixture(emu_aligner, golden_primitives)
    gp.test_affine_gap_clipping_modes_vs_restatement(emu_aligner)


@pytest.mark.parametrize("name,kw", [("default_d8", dict(max_k=8)), ("lvonly_d8", dict(max_k=8, use_affine_gap=0))])
def test_emu_align_read_vs_reference_fixture(emu, golden_index, golden_reads, name, kw):
    """All 4 000 golden reads (100 and 150 bp) through the emulated k_align_single, every field against the reference's."""
    import tests.test_gpu_parity as gp
    partial0, inactive0 = emu.emu_partial_ops(), emu.emu_inactive_reads()
    gp.test_align_read_vs_reference_fixture(golden_index, golden_reads, name, kw)
    # the single-end kernel's control flow is wave-uniform: no cross-lane operation was ever resolved with part of a wave,
    # and no lane read a lane that was not taking part
    assert emu.emu_partial_ops() == partial0
    assert emu.emu_inactive_reads() == inactive0


def test_emu_plane_landau_vishkin_instantiations(emu, golden_index, golden_reads, monkeypatch):
    """SNAPGPU_LV_PLANES=1: the context builds the bit-plane shadow of the genome and its plain launches go to the instantiations that
    carry the plane Landau-Vishkin (single_planes_k.hip: fast form + help, exact form): the first 1 000 golden reads per length, every
    field, as the default kernels -- and the same bytes as a context without planes."""
    import tests.test_gpu_parity as gp
    from snap_amd.aligner import BaseAligner
    sub = {k: (golden_reads[k][:1000] if k[0] in "bq" else golden_reads[k]) for k in golden_reads.files}
    b, q = sub["b150"], sub["q150"]
    offs = np.arange(b.shape[0] + 1, dtype=np.uint64) * 150
    out = {}
    for planes in ("1", "0"):
        monkeypatch.setenv("SNAPGPU_LV_PLANES", planes)
        for help_on in ("1", "0"):                                            # help on: fast form + replay; off: the exact form as the one pass
            monkeypatch.setenv("SNAPGPU_SINGLE_HELP", help_on)
            a = BaseAligner(golden_index, abi.default_params(max_k=8, max_read_len=160))
            try:
                prim, alt = a.AlignRead(b, q, offs)
            finally:
                a.close()
            out[planes + help_on] = prim
    exp, _ = util.with_fresh_overrides(golden_reads["default_d8_150_primary"], "default_d8_150_primary")
    for k, prim in out.items():
        assert not util.compare_results(exp[:1000], prim), k
    mask = np.uint32(0x3fffffff)
    for f in out["11"].dtype.names:
        if f != "reserved":
            assert (out["11"][f] == out["01"][f]).all() and (out["10"][f] == out["00"][f]).all(), f
    assert ((out["11"]["reserved"] & mask) == (out["01"]["reserved"] & mask)).all()


def test_emu_ragged_and_degenerate_reads(emu, golden_index):
    import tests.test_gpu_parity as gp
    gp.test_ragged_and_degenerate_reads(golden_index)


def test_emu_secondary_results(emu, golden_index, golden_reads):
    """-om / -omax / -mpc: the first 250 reads of two option sets through k_align_single<., true>, record order included."""
    import tests.test_gpu_secondary as gs
    from snap_amd.aligner import BaseAligner
    z = np.load(os.path.join(util.GOLDEN, "secondary_reads.npz"))
    n = 250
    b, q = golden_reads["b100"][:n], golden_reads["q100"][:n]
    offs = np.arange(n + 1, dtype=np.uint64) * b.shape[1]
    for name, kw, om, omax, mpc in gs._sets(z)[:2]:
        a = BaseAligner(golden_index, abi.default_params(max_read_len=160, **kw))
        try:
            a.enable_secondary(om, max_results=omax, max_per_contig=mpc)
            prim, alt, sec, nsec = a.AlignReadSecondary(b, q, offs, stride=4)
        finally:
            a.close()
        key = "%s_100_" % name
        e_prim, _ = util.with_fresh_overrides(z[key + "primary"], "sec_" + key + "primary")
        e_sec, _ = util.with_fresh_overrides(z[key + "secondary"], "sec_" + key + "secondary")
        e_nsec, _ = util.with_fresh_overrides(z[key + "nsec"], "sec_" + key + "nsec")
        problems = util.compare_results(e_prim[:n], prim, "primary")           # every read, no exclusion
        problems += gs.compare_secondary(e_sec[:n], e_nsec[:n], sec, nsec, np.zeros(n, bool))
        assert not problems, (name, problems)


def test_emu_compute_cigar(emu, emu_aligner, golden_index):
    """SAMFormat::computeCigar (Landau-Vishkin variant) on the emulated device: every item of the reference fixture, both
    op alphabets; 1 500 fresh reads against the C restatement; argument errors."""
    import tests.test_zz_gpu_cigar as gc
    z = np.load(os.path.join(util.GOLDEN, "cigar_lv.npz"))
    for use_m in (0, 1):
        got = gc.check_against_fixture(emu_aligner, z, use_m)
        gc.cigar_properties(got, z["length"], z["extra_before"])
    gc.test_compute_cigar_argument_errors(emu_aligner, golden_index)


def test_emu_paired_end(emu):
    """ChimericPairedEndAligner::align over IntersectingPairedEndAligner::align on the emulated device (k_align_paired): the first
    200 golden pairs (2 x 150 bp) of the default option set, every field of every aligned read against the reference's."""
    import tests.test_gpu_paired as gp
    from tests.pairs_util import compare_paired
    from tests.test_paired_host import OPTS
    name = list(OPTS)[0]
    kw, pkw = OPTS[name]
    z = np.load(os.path.join(util.GOLDEN, "paired_reads.npz"))
    n = 200
    o = z["o150"][:2 * n + 1]
    a = gp._aligner(util.load_golden_index("paired_index.npz"), kw, pkw)
    try:
        prim, alt = a.align(z["b150"].reshape(-1)[:int(o[-1])], z["q150"].reshape(-1)[:int(o[-1])], o)
    finally:
        a.close()
    key = "%s_150_s0" % name
    exp, _ = util.with_fresh_overrides(z[key + "_primary"], "pe_" + key + "_primary")
    bad = compare_paired(exp[:n], prim, verbose=3)                             # every pair, no exclusion
    assert not bad.any()
    assert (alt["status"] == z[key + "_alt"]["status"][:n]).all()


def test_emu_exact_replay_beside_the_main_pass(emu, monkeypatch):
    """tests/test_gpu_paired.py::test_exact_replay_beside_the_main_pass on the emulated device (the exact kernel there starts when the main
    kernel has ended: the list protocol, the markers and the pass after it, not the concurrency), first 150 golden pairs."""
    import tests.test_gpu_paired as gp
    gp.test_exact_replay_beside_the_main_pass(util.load_golden_index("paired_index.npz"), np.load(os.path.join(util.GOLDEN, "paired_reads.npz")),
                                              monkeypatch, n=150)


def test_emu_paired_end_lv_only_hamming_retry_uses_affine_gap(emu):
    """`use_affine_gap = 0` with soft clipping: the Hamming retry of the chimeric fallback still calls alignAffineGap
    (ChimericPairedEndAligner.cpp:330-360), so the affine-gap LDS rows and traceback slab must exist in that configuration
    (AlignCfg::ag_buffers; round 2: with size 0 the staged text codes ran into the per-pair result).  First 300 pairs of the LV-only set."""
    import tests.test_gpu_paired as gp
    from tests.pairs_util import compare_paired
    from tests.test_paired_host import OPTS
    kw, pkw = OPTS["lvonly_d12"]
    z = np.load(os.path.join(util.GOLDEN, "paired_reads.npz"))
    n = 300
    o = z["o150"][:2 * n + 1]
    a = gp._aligner(util.load_golden_index("paired_index.npz"), kw, pkw)
    try:
        prim, _ = a.align(z["b150"].reshape(-1)[:int(o[-1])], z["q150"].reshape(-1)[:int(o[-1])], o)
    finally:
        a.close()
    exp, _ = util.with_fresh_overrides(z["lvonly_d12_150_s0_primary"], "pe_lvonly_d12_150_s0_primary")
    assert (exp["used_affine_gap_scoring"][:n] != 0).any()                        # the case is in the sample
    assert not compare_paired(exp[:n], prim, verbose=3).any()


def test_emu_feeders_share_one_index(emu):
    """BaseAligner.replica (snapgpu_create_replica, share_index): tests/test_gpu_multi_ctx.py on the emulated device."""
    import tests.test_gpu_multi_ctx as mc
    mc.test_paired_feeders_two_batches_in_flight()


def test_emu_paired_grid_share_is_scheduling_only(emu, monkeypatch):
    """snapgpu.hip: paired_grid_share on the emulated device (3 workgroups of wave slots): whatever part of them a launch asks for --
    SNAPGPU_PAIRED_GRID_SHARE 1 / 2 / 16 (never less than one workgroup), SNAPGPU_PAIRED_GRID_OVER 0.25 / 4 -- the results are the same bytes,
    and the fixture's."""
    from tests.pairs_util import compare_paired
    from snap_amd.aligner import ChimericPairedEndAligner
    z = np.load(os.path.join(util.GOLDEN, "paired_reads.npz"))
    n = 48
    o = z["o150"][:2 * n + 1]
    b, q = z["b150"].reshape(-1)[:int(o[-1])], z["q150"].reshape(-1)[:int(o[-1])]
    a = ChimericPairedEndAligner(util.load_golden_index("paired_index.npz"), abi.default_params(max_k=8, max_read_len=160), abi.default_paired_params())
    try:
        base = a.align(b, q, o)[0]
        for share, over in (("1", None), ("2", None), ("16", None), (None, "0.25"), ("2", "4")):
            for k_, v_ in (("SNAPGPU_PAIRED_GRID_SHARE", share), ("SNAPGPU_PAIRED_GRID_OVER", over)):
                if v_ is None:
                    monkeypatch.delenv(k_, raising=False)
                else:
                    monkeypatch.setenv(k_, v_)
            assert a.align(b, q, o)[0].tobytes() == base.tobytes(), (share, over)
    finally:
        a.close()
    exp, _ = util.with_fresh_overrides(z["default_d8_150_s0_primary"], "pe_default_d8_150_s0_primary")
    assert not compare_paired(exp[:n], base, verbose=3).any()


@pytest.mark.parametrize("eager", [False, True])
def test_emu_phase4_help_on_demand(emu, monkeypatch, eager):
    """The Phase-4 help slots (paired_dev.h), published on demand / eagerly, on pairs that have long candidate lists (a genome built of
    high-copy repeats): every pair equals the reference with fresh aligner objects, and -- eager mode -- speculative answers were used."""
    from oracle import ref
    if not os.path.exists(ref.LIB_PATH):
        pytest.skip("oracle/_ref not built")
    from snap_amd import synth
    from snap_amd.aligner import ChimericPairedEndAligner
    from snap_amd.index import GenomeIndex
    from tests.pairs_util import compare_paired
    import tempfile
    d = tempfile.mkdtemp(prefix="emuhelp")
    g = synth.make_genome(11, 1_200_000, n_contigs=2, repeat_frac=0.8, max_copies=1200, repeat_len=(400, 1200), max_divergence=0.012)
    synth.write_fasta(d + "/g.fa", g)
    ref.build_index(d + "/g.fa", d + "/idx", 20, threads=8)
    pairs = synth.make_pairs(5, g, 40, 150)
    params, pparams = abi.default_params(max_k=8, max_read_len=160), abi.default_paired_params()
    with ref.fresh_objects():
        exp = ref.RefIndex(d + "/idx").align_paired(params, pparams, pairs["bases"], pairs["quals"], pairs["offsets"], threads=4, stage=0)[0]
    monkeypatch.setenv("SNAPGPU_PAIRED_HELP_MIN", "16")
    if eager:
        monkeypatch.setenv("SNAPGPU_PAIRED_HELP_EAGER", "1")
    gi = GenomeIndex.load_from_directory(d + "/idx")
    a = ChimericPairedEndAligner(gi, params, pparams)
    try:
        a.counters(reset=True)
        got, _ = a.align(pairs["bases"], pairs["quals"], pairs["offsets"])
        c = a.counters()
        a.close()
        if eager:       # who scored a candidate leaves no trace: every byte (the informational flags too) as without help
            monkeypatch.setenv("SNAPGPU_PAIRED_HELP_MIN", "0")
            a = ChimericPairedEndAligner(gi, params, pparams)
            alone, _ = a.align(pairs["bases"], pairs["quals"], pairs["offsets"])
    finally:
        a.close()
        shutil.rmtree(d, ignore_errors=True)
    assert not compare_paired(exp, got, verbose=3).any()
    assert c["help_watchdog_events"] == 0
    if eager:
        assert c["help_lists_published"] > 0 and c["help_answers_used"] > 0
        assert got.tobytes() == alone.tobytes()


@pytest.mark.parametrize("seed_len,large", [(24, False), (22, True)])
def test_emu_other_index_shapes(emu, tmp_path, seed_len, large):
    """Key bytes 5, `-large` entries, 16 hash tables: the probe takes other paths than with the north star's -s 20 index.
    600 fresh reads against the compiled reference, work counters included."""
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built here")
    from tests.test_zy_gpu_index_shapes import align_and_compare
    align_and_compare(str(tmp_path), seed_len, large, 600)


@pytest.mark.parametrize("seed_len,large,extra,from_directory", [(16, False, [], False), (18, True, [], True), (20, False, ["-locationSize", "6"], True)])
def test_emu_wide_location_indexes(emu, tmp_path, seed_len, large, extra, from_directory):
    """Indexes whose files carry 5 .. 8-byte locations (seeds shorter than 20 get them by default): narrowed on load by the Python loader /
    by snapgpu_create_from_directory; the probe and 500 reads against the reference, which goes through lookupSeed / overflowTable64."""
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built here")
    from tests.test_zy_gpu_index_shapes import align_and_compare
    align_and_compare(str(tmp_path), seed_len, large, 500, extra=extra, from_directory=from_directory)


def test_emu_paired_over_wide_location_index(emu, tmp_path):
    """The paired-end path over an index with 5-byte locations (seed 16), 250 hard pairs against the reference."""
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built here")
    from tests.test_zy_gpu_index_shapes import paired_over_wide_index
    paired_over_wide_index(str(tmp_path), 16, False, [], 250)


def test_emu_affine_gap_call_sequences(emu, emu_aligner):
    """The exact (image-keeping) affine-gap forms against the reference's history-dependent answers: the first quarter of each call sequence
    of tests/golden/ag_sequence.npz (300 calls through the 192-position register form, 125 through the LDS form, both directions)."""
    import tests.test_gpu_parity as gp
    z = np.load(os.path.join(util.GOLDEN, "ag_sequence.npz"), allow_pickle=True)
    assert gp.check_affine_gap_call_sequences(emu_aligner, z, step=4) > 5


def test_emu_affine_gap_wide_bands(emu, golden_index):
    """ag_banded_win2 (w 13 .. 31) on the emulated device: 260 fuzzed problems against the restatement in the three batch instantiations,
    and the first half of the wide-band call sequence of the reference (exact form)."""
    import tests.test_gpu_parity as gp
    from snap_amd.aligner import BaseAligner
    a = BaseAligner(golden_index, abi.default_params(max_k=20, max_read_len=400))
    try:
        assert gp.check_affine_gap_wide_bands(a, n=260) > 300
    finally:
        a.close()
    gp.test_affine_gap_wide_band_call_sequences_vs_reference_fixture(golden_index, step=2)


def test_emu_compute_cigar_affine_gap(emu, emu_aligner):
    """SAMFormat::computeCigar, affine-gap variant (banded and full global alignment with traceback), on the emulated device: every
    third item of the reference fixture (tests/golden/cigar_ag.npz), both op alphabets."""
    import tests.test_zz_gpu_cigar as gc
    z = np.load(os.path.join(util.GOLDEN, "cigar_ag.npz"))
    for use_m in (0, 1):
        gc.check_ag_against_fixture(emu_aligner, z, use_m, step=3)


@pytest.mark.parametrize("tag", ["default", "lvonly_eqx", "clipfront"])
def test_emu_sam_fields(emu, golden_index, tag):
    """Results -> FLAG / RNAME / POS / MAPQ / CIGAR / NM on the emulated device (k_sam_fields: the writeReads retry loop, createSAMLine,
    both cigar variants, soft clips), against what the unmodified reference CLI printed: every third read of the fixture
    (clipfront: the CLI run with -C++, so that Read::clip also removed a leading run of '#')."""
    import tests.test_zz_gpu_cigar as gc
    z = np.load(os.path.join(util.GOLDEN, "sam_fields.npz"))
    gc.check_sam_fields_against_reference_cli(golden_index, z, tag, step=3)


@pytest.mark.parametrize("tag", ["default", "clipfront"])
def test_emu_align_sam_single(emu, golden_index, tag):
    """snapgpu_align_sam_single on the emulated device: results = the reference aligner's, fields = the reference CLI's, and both equal to
    the two calls it replaces (the first 500 reads of the fixture)."""
    import tests.test_zz_gpu_cigar as gc
    z = np.load(os.path.join(util.GOLDEN, "sam_fields.npz"))
    gc.check_align_sam_single_against_reference_cli(golden_index, z, tag, n=500)


@pytest.mark.parametrize("opts", [[], ["-G-", "-=", "-C++", "-b", "97"], ["-G-", "-ea", "-om", "1", "-omax", "4"], ["-ae"], ["-ae", "-om", "1"]])     # -b 97: nine batches, the last one short; -C++: '#' heads
                                                                                                                  # clipped too; -om / -ea: secondary and first-ALT records
def test_emu_native_fastq_to_sam(emu, tmp_path, opts):
    """FASTQ in, SAM out: the native host program (snap_amd/csrc/host/snapgpu_sam.cpp, linked against the emulated library) writes the
    same file as the unmodified reference CLI, every line but @PG; 600 reads (350 for the -ae sets) incl. ragged / '#'-clipped / N-rich / unalignable ones."""
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built here")
    from tests.emu.build import TOOL
    from tests.test_zz_gpu_native_sam import make_workload, run_and_compare
    n = 350 if "-ae" in opts else 600
    index_dir, fastq = make_workload(str(tmp_path), n, genome_bases=300_000)
    env = dict(os.environ, SNAPGPU_EMU_CUS="4")
    assert run_and_compare(TOOL, str(tmp_path), index_dir, fastq, opts, env=env, ref_opts=[o for o in opts if o not in ("-b", "97")]) > n


def test_emu_native_tool_pins_its_group_buffers(emu, tmp_path):
    """snapgpu-sam page-locks the buffers it hands to snapgpu_align_sam_single (hipHostRegister, looked up by name).  The emulated runtime
    checks the calls -- no overlapping registration, every unregister names a registered start -- and reports at exit: ranges were
    registered, none is left; and the file is still the reference CLI's (groups of two 61-read batches of ragged reads, two feeders)."""
    import subprocess
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built here")
    from tests.emu.build import TOOL
    from tests.test_zz_gpu_native_sam import make_workload, sam_lines
    d = str(tmp_path)
    index_dir, fastq = make_workload(d, 400, genome_bases=300_000)
    env = dict(os.environ, SNAPGPU_EMU_CUS="4", SNAPGPU_SAM_PIN_MIN="1", SNAPGPU_EMU_PIN_REPORT="1")
    r = subprocess.run([ref.CLI_PATH, "single", index_dir, fastq, "-o", os.path.join(d, "ref.sam"), "-t", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=600)
    assert r.returncode == 0
    r = subprocess.run([TOOL, "single", index_dir, fastq, "-o", os.path.join(d, "new.sam"), "-b", "61", "-g", "2", "-q", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       stdin=subprocess.DEVNULL, timeout=1800, env=env)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-2000:]
    assert sam_lines(os.path.join(d, "ref.sam"), False) == sam_lines(os.path.join(d, "new.sam"), False)
    line = [x for x in out.splitlines() if x.startswith("emu: hipHostRegister calls")]
    assert line, out[-2000:]
    calls, left = int(line[0].split("calls")[1].split(",")[0]), int(line[0].rsplit(" ", 1)[1])
    assert calls >= 8 and left == 0, line[0]
    r = subprocess.run([TOOL, "single", index_dir, fastq, "-o", os.path.join(d, "y.sam"), "-b", "200"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       stdin=subprocess.DEVNULL, timeout=1800, env=dict(env, SNAPGPU_SAM_PIN="0", SNAPGPU_LOAD_PIN="0"))       # (neither the tool's buffers nor the index loader's pieces)
    assert r.returncode == 0 and b"hipHostRegister calls" not in r.stdout


def test_emu_sam_fields_paired(emu):
    """The paired-end writer on the emulated device (k_sam_fields_paired: both mates' records + SAMFormat::fillMateInfo + print order):
    the first 400 pairs of the fixture parsed from the unmodified reference CLI's `paired` output, all 9 computed fields."""
    import tests.test_zz_gpu_cigar as gc
    z = np.load(os.path.join(util.GOLDEN, "sam_fields_paired.npz"))
    gc.check_sam_fields_paired_against_reference_cli(z, "default", n_pairs=400)


def test_emu_native_paired_fastq_to_sam(emu, tmp_path):
    """Two FASTQ files in, SAM out, paired end: the native host program on the emulated device writes the same file as `snap-aligner paired`,
    line for line and in the same order (every line but @PG); 200 hard pairs incl. '#'-clipped mates and pairs too short to align."""
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built here")
    from tests.emu.build import TOOL
    from tests.test_zz_gpu_native_sam import make_paired_workload, run_and_compare_paired
    index_dir, fq = make_paired_workload(str(tmp_path), 200, genome_bases=300_000)
    assert run_and_compare_paired(TOOL, str(tmp_path), index_dir, fq, [], env=dict(os.environ, SNAPGPU_EMU_CUS="4")) > 400


def test_emu_native_paired_secondary_and_alt_records(emu, tmp_path):
    """`snapgpu-sam paired -om 1` (further pair results + the mates' single-end secondary results as records) and `-ea` (the first ALT pair)
    on the emulated device: the reference CLI's file, line for line in its order."""
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built here")
    from tests.emu.build import TOOL
    from tests.test_zz_gpu_native_sam import make_paired_workload, make_paired_alt_workload, run_and_compare_paired
    env = dict(os.environ, SNAPGPU_EMU_CUS="4")
    d1, d2 = str(tmp_path / "om"), str(tmp_path / "ea")
    os.makedirs(d1); os.makedirs(d2)
    index_dir, fq = make_paired_workload(d1, 100, genome_bases=300_000)
    assert run_and_compare_paired(TOOL, d1, index_dir, fq, ["-om", "1"], env=env) > 200
    index_dir, fq = make_paired_alt_workload(d2, 80, genome_bases=300_000)
    assert run_and_compare_paired(TOOL, d2, index_dir, fq, ["-ea"], env=env) > 160


def test_emu_native_paired_records_after_a_reverse_complement_record_at_a_contig_start(emu, tmp_path):
    """`snapgpu-sam paired -om`: a read whose record is reverse-complement at POS 1 and has further records after it (the writer recomputes that
    record's Read state on the host instead of stopping): the reference CLI's file, line for line, and the case does occur in the workload."""
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built here")
    import subprocess
    from tests.emu.build import TOOL
    from tests.test_zz_gpu_native_sam import make_contig_start_workload, run_and_compare_paired
    env = dict(os.environ, SNAPGPU_EMU_CUS="4", SNAPGPU_SAM_VERBOSE="1")
    d = str(tmp_path)
    index_dir, fq = make_contig_start_workload(d, 40)
    assert run_and_compare_paired(TOOL, d, index_dir, fq, ["-om", "3", "-D", "3"], env=env) > 160
    r = subprocess.run([TOOL, "paired", index_dir, fq[0], fq[1], "-om", "3", "-D", "3", "-o", os.path.join(d, "again.sam")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    assert r.returncode == 0 and r.stdout.decode().count("recomputed on the host") >= 1


def test_emu_native_fastq_to_bam(emu, tmp_path):
    """`-o x.bam`: header, reference table and records of the native program's BAM (decompressed) equal the reference CLI's, single end
    (with secondary records) and paired end."""
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built here")
    from tests.emu.build import TOOL
    from tests.test_zz_gpu_native_sam import make_workload, make_paired_workload, run_and_compare_bam
    env = dict(os.environ, SNAPGPU_EMU_CUS="4")
    ds, dp = str(tmp_path / "s"), str(tmp_path / "p")
    os.makedirs(ds); os.makedirs(dp)
    index_dir, fastq = make_workload(ds, 400, genome_bases=300_000)
    assert run_and_compare_bam(TOOL, ds, "single", index_dir, [fastq], ["-om", "1", "-omax", "3"], env=env) > 400
    index_dir, fq = make_paired_workload(dp, 150, genome_bases=300_000)
    assert run_and_compare_bam(TOOL, dp, "paired", index_dir, fq, [], env=env) == 300


def test_emu_option_sets(emu, tmp_path):
    """Option sets no fixture pins (-h 16, seed coverage instead of -n, -D 3, other scoring parameters and end bonuses), 250 reads of a
    repeat-rich genome with an ALT contig against the compiled reference, work counters included."""
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built here")
    from tests.test_zy_gpu_index_shapes import OPTION_SETS, option_workload, check_option_set
    ix, ri, rd = option_workload(str(tmp_path), 250)
    for kw in (OPTION_SETS[0], OPTION_SETS[4], OPTION_SETS[7], OPTION_SETS[12], OPTION_SETS[14]):
        check_option_set(ix, ri, rd, kw)


def test_emu_sam_fields_device_pointer_form(emu, golden_index):
    """snapgpu_sam_fields_single_device (all arrays "in device memory", which on the emulated device is host memory): same answers as the
    reference CLI printed, first 600 reads of the fixture."""
    import tests.test_zz_gpu_cigar as gc
    from snap_amd.aligner import BaseAligner
    z = np.load(os.path.join(util.GOLDEN, "sam_fields.npz"))
    n = 600
    keep = []
    def to_dev(x):
        t = np.ascontiguousarray(x).copy(); keep.append(t)
        return (t, t.ctypes.data)
    a = BaseAligner(golden_index, abi.default_params(max_k=14, max_read_len=400))
    try:
        flag, contig, pos, mapq, ops, n_ops, nm, stale = gc.run_sam_fields_device_form(a, z, "default", n, to_dev, lambda t, like: t[0])
    finally:
        a.close()
    for k, v in (("flag", flag), ("contig", contig), ("pos", pos), ("mapq", mapq), ("nm", nm), ("n_ops", n_ops)):
        assert (v == z["default_" + k][:n]).all(), k
    for i in range(n):
        assert util.cigar_text(ops[i], n_ops[i]) == util.cigar_text(z["default_ops"][i], z["default_n_ops"][i]), i


def test_emu_index_builder_vs_reference(emu, tmp_path):
    """the GPU index builder's directory against the reference's own `snap-aligner index` on the same FASTA --
    same Genome file, same table sizes, same answers to every probed seed, same alignments (tests/index_build_util.py)."""
    from tests.index_build_util import compare_with_reference
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built")
    stats, _, d_gpu = compare_with_reference(tmp_path, lib=emu)
    assert stats["n_repeated_seeds"] > 0 and stats["overflow_table_size"] > 0
    # the index straight from HBM, without the files: build, adopt the view, align the reads of the file-based index
    from snap_amd.index import build_index, GenomeIndex
    from snap_amd.aligner import BaseAligner
    st2, built = build_index(os.path.join(str(tmp_path), "g.fa"), None, lib=emu, keep=True)
    assert st2["n_distinct_seeds"] == stats["n_distinct_seeds"]
    ix = GenomeIndex.load_from_directory(d_gpu)
    a_files = BaseAligner(ix, abi.default_params(max_k=8, max_read_len=112))
    a_view = BaseAligner.from_built_index(built, ix, abi.default_params(max_k=8, max_read_len=112))
    from snap_amd import synth
    contigs = [(c.name, ix.genome[c.begin:c.begin + 20000]) for c in ix.contigs[:3]]
    reads = synth.make_reads(3, contigs, 400, 100)
    p1, _ = a_files.AlignRead(reads["bases"], reads["quals"], reads["offsets"])
    p2, _ = a_view.AlignRead(reads["bases"], reads["quals"], reads["offsets"])
    assert not util.compare_results(p1, p2)
    a_files.close(); a_view.close(); built.close()


@pytest.mark.parametrize("seed_len,kw,extra", [(24, {}, []), (22, {}, []), (17, {}, ["-locationSize", "4"]), (26, {}, []), (31, {}, []),
                                               (20, dict(key_bytes=3), ["-keysize", "3"]), (12, {}, ["-locationSize", "4"])])
def test_emu_index_builder_key_sizes_vs_reference(emu, tmp_path, seed_len, kw, extra):
    """Key sizes other than 4 (GenomeIndex.cpp:437: the default for -s 24 is 5): entries that straddle words, claimed through the bit array
    (index_build.h: k_ib_insert_wide).  Same checks as above -- and the library's own lookup over the built directory."""
    from tests.index_build_util import compare_with_reference
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built")
    stats, d_ref, d_gpu = compare_with_reference(tmp_path, lib=emu, seed_len=seed_len, extra_ref=extra, n_reads=600, **kw)
    want = kw.get("key_bytes") or max(2, (seed_len + 2) // 4 - 1)
    assert int(open(os.path.join(d_gpu, "GenomeIndex")).read().split()[6]) == want
    from snap_amd.index import GenomeIndex
    from snap_amd.aligner import BaseAligner
    from snap_amd import synth
    ix = GenomeIndex.load_from_directory(d_gpu)
    contigs = [(c.name, ix.genome[c.begin:c.begin + 20000]) for c in ix.contigs[:3]]
    reads = synth.make_reads(3, contigs, 300, 100)
    params = abi.default_params(max_k=8, max_read_len=112)
    with ref.fresh_objects():
        exp = ref.RefIndex(d_ref).align_single(params, reads["bases"], reads["quals"], reads["offsets"], threads=4)[0]
    a = BaseAligner(ix, params)
    got, _ = a.AlignRead(reads["bases"], reads["quals"], reads["offsets"])
    a.close()
    assert not util.compare_results(exp, got), util.compare_results(exp, got)


def test_emu_single_end_help_for_heavy_reads(emu, monkeypatch, tmp_path):
    """se_help.h: a forced walk's remaining candidates published for idle waves (here: published eagerly, so that the path runs whatever
    the emulator's scheduling does), on reads out of diverged high-copy repeats.  Every read against the reference with fresh aligner
    objects; the work counters must equal the reference's and those of a run without the help."""
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built")
    from snap_amd import synth
    from snap_amd.aligner import BaseAligner
    from snap_amd.index import GenomeIndex
    d = str(tmp_path)
    g = synth.make_genome(13, 1_500_000, n_contigs=2, repeat_frac=0.85, max_copies=280, repeat_len=(400, 1500), max_divergence=0.03)
    synth.write_fasta(d + "/g.fa", g)
    ref.build_index(d + "/g.fa", d + "/idx", 20, threads=8)
    ix = GenomeIndex.load_from_directory(d + "/idx")
    reads = synth.make_reads(7, g, 160, 150)
    params = abi.default_params(max_k=8, max_read_len=160)
    with ref.fresh_objects():
        exp, _, rc, _ = ref.RefIndex(d + "/idx").align_single(params, reads["bases"], reads["quals"], reads["offsets"], threads=8)
    out = {}
    for mode, env in (("eager", {"SNAPGPU_SINGLE_HELP_EAGER": "1"}), ("off", {"SNAPGPU_SINGLE_HELP": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        a = BaseAligner(ix, params)
        a.counters(reset=True)
        got, _ = a.AlignRead(reads["bases"], reads["quals"], reads["offsets"])
        out[mode] = (got, a.counters())
        a.close()
        for k in env:
            monkeypatch.delenv(k)
        assert not util.compare_results(exp, got), mode
    ce, co = out["eager"][1], out["off"][1]
    assert ce["help_lists_published"] > 0 and ce["help_answers_used"] > 0 and ce["help_watchdog_events"] == 0
    assert co["help_lists_published"] == 0
    for k in ("n_hash_table_lookups", "n_hits_consumed", "n_lv_locations", "n_ag_locations", "n_lv_ref_bytes"):
        assert ce[k] == co[k], k
    assert [ce["n_hash_table_lookups"], ce["n_lv_locations"], ce["n_ag_locations"]] == [rc["lookups"], rc["lv"], rc["ag"]]


def test_emu_alignment_adjuster(emu, golden_index):
    """tests/test_gpu_adjust.py on the emulated device: the `-ae` adjuster item by item and inside finalizeSecondaryResults, against the
    golden fixture of the compiled reference."""
    import tests.test_gpu_adjust as ga
    z = dict(np.load(os.path.join(util.GOLDEN, "adjust.npz")))
    for k in ("read_bases", "read_quals", "primary", "secondary", "nsec"):
        z[k] = z[k][:500]                                                     # (the first 500 of the 1 200 reads; all 3 000 adjuster items)
    partial0, inactive0 = emu.emu_partial_ops(), emu.emu_inactive_reads()
    ga.test_adjust_alignments_vs_reference_fixture(golden_index, z)
    ga.test_secondary_with_adjustment_vs_reference_fixture(golden_index, z, min_changed=20)
    assert emu.emu_partial_ops() == partial0 and emu.emu_inactive_reads() == inactive0      # the adjuster's control flow is wave-uniform
    ga.test_adjustment_is_single_end_only(golden_index)


def test_emu_stop_on_first_hit_and_explore_popular_seeds(emu, tmp_path):
    """tests/test_gpu_flags.py on the emulated device: -f / -x against the compiled reference running with the same flags (a smaller workload)."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built here")
    import tests.test_gpu_flags as gf
    partial0 = emu.emu_partial_ops()
    gf.check_flags_vs_live_reference(tmp_path, n_reads=700, genome_bases=600_000, with_secondary=False)
    assert emu.emu_partial_ops() == partial0


def test_emu_native_fastq_readers_agree(emu, tmp_path):
    """snapgpu-sam's mapped, multi-threaded FASTQ reader (round 4) and its sequential reader write the same file whatever the shape of the
    input: CRLF line ends, no newline at the end, blank lines between records and gzip (both fall back to the sequential reader), -seqread,
    one or many parser threads, batches that end inside the file's chunks; and the mapped reader refuses what the sequential one refuses."""
    import gzip
    import subprocess
    from oracle import ref
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not built here")
    from tests.emu.build import TOOL
    from tests.test_zz_gpu_native_sam import make_workload, sam_lines
    d = str(tmp_path)
    index_dir, fastq = make_workload(d, 120, genome_bases=200_000)
    raw = open(fastq, "rb").read()
    recs = raw.split(b"\n")
    assert recs[-1] == b"" and (len(recs) - 1) % 4 == 0
    variants = {"plain": raw,
                "crlf": raw.replace(b"\n", b"\r\n"),
                "no_final_newline": raw[:-1],
                "blank_lines": b"\n".join(b"\n".join(recs[i:i + 4]) + (b"\n" if (i // 4) % 7 == 3 else b"") for i in range(0, len(recs) - 1, 4)) + b"\n"}
    env = dict(os.environ, SNAPGPU_EMU_CUS="4", SNAPGPU_SAM_VERBOSE="1")

    def run(path, opts, expect_reader):
        out = os.path.join(d, "o_%d.sam" % len(os.listdir(d)))
        r = subprocess.run([TOOL, "single", index_dir, path, "-o", out] + opts, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=1200, env=env)
        assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
        assert ("(%s reader" % expect_reader).encode() in r.stdout, r.stdout.decode(errors="replace")[-600:]
        return sam_lines(out)
    paths = {}
    for k, v in variants.items():
        paths[k] = os.path.join(d, k + ".fq"); open(paths[k], "wb").write(v)
    paths["gz"] = os.path.join(d, "plain.fq.gz")
    with gzip.open(paths["gz"], "wb") as f:
        f.write(raw)
    want = run(paths["plain"], ["-seqread"], "sequential")
    assert len(want) > 120
    assert run(paths["plain"], [], "mapped") == want
    assert run(paths["plain"], ["-tp", "1", "-b", "97"], "mapped") == want          # batches of 97 records cut the file's 1 MB chunks anywhere
    assert run(paths["crlf"], [], "mapped") == want
    assert run(paths["no_final_newline"], ["-b", "50"], "mapped") == want
    assert run(paths["blank_lines"], [], "sequential") == want                       # lines do not come in fours: the sequential reader takes over
    assert run(paths["gz"], [], "sequential") == want
    # a malformed record is refused by both readers (the mapped one names the way out)
    bad = os.path.join(d, "bad.fq")
    open(bad, "wb").write(raw.replace(b"\n+\n", b"\n-\n", 1))
    for opts in ([],):
        r = subprocess.run([TOOL, "single", index_dir, bad, "-o", os.path.join(d, "bad.sam")] + opts, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=600, env=env)
        assert r.returncode != 0 and b"'+' line" in r.stdout


def test_emu_hamming_fallback_tie_broken_by_the_seed_probability(emu):
    """tests/test_gpu_paired.py's regression for BaseAligner.cpp:907 (libm pow against powi: one ulp that decides a 78-way tie) on the emulated device."""
    import tests.test_gpu_paired as gp
    gp.test_hamming_fallback_ties_are_broken_by_the_last_bit_of_the_seed_probability()
