"""CPU tests of the paired-end control flow.  snap_amd/csrc/paired.h -- the code k_align_paired executes -- is compiled for
the host by oracle/Makefile (oracle/_ref/libpairedhost.so: primitives from the C restatement, the reference's single-end
aligner for the chimeric fallback) and compared with the committed golden PairedAlignmentResults the compiled reference
produced (scripts/make_golden_paired.py).  Nothing here touches the product library or a GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from snap_amd import abi, synth
from snap_amd.aligner import make_index_view
from snap_amd.index import GenomeIndex
from oracle import ref
from tests import util
from tests.pairs_util import compare_paired

HOSTLIB = os.path.join(util.ROOT, "oracle", "_ref", "libpairedhost.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(HOSTLIB) and ref.available()),
                                reason="oracle/_ref was not built (needs /root/reference: run __graft_entry__.build() in the container)")

OPTS = dict(default_d8=(dict(max_k=8), {}), default_d27=(dict(max_k=27), {}), lvonly_d12=(dict(max_k=12, use_affine_gap=0), {}),
            spacing_d8=(dict(max_k=8), dict(min_spacing=100, max_spacing=600, num_seeds=12)))


@pytest.fixture(scope="module")
def golden_pairs():
    return np.load(os.path.join(util.GOLDEN, "paired_reads.npz"))


@pytest.fixture(scope="module")
def ref_index(tmp_path_factory):
    """The golden index, rebuilt as a directory the reference can load (same seeded genome as make_golden_paired.py)."""
    d = str(tmp_path_factory.mktemp("pidx"))
    g = synth.make_genome(20260926, 240_000, n_contigs=3, repeat_frac=0.35, max_copies=40, repeat_len=(150, 1500), n_run_frac=0.003)
    synth.write_fasta(d + "/ref.fa", g)
    ref.build_index(d + "/ref.fa", d + "/idx", 20, threads=4)
    gi = GenomeIndex.load_from_directory(d + "/idx")
    golden = util.load_golden_index("paired_index.npz")
    assert (gi.genome_padded == golden.genome_padded).all() and (gi.contig_begin == golden.contig_begin).all()
    return ref.RefIndex(d + "/idx"), gi


def host_align(gi, rix, p, pp, bases, quals, offsets, stage):
    lib = C.CDLL(HOSTLIB)
    v, keep = make_index_view(gi)
    n = (offsets.size - 1) // 2
    prim = np.zeros(n, dtype=abi.PAIRED_RESULT_DTYPE)
    alt = np.zeros(n, dtype=abi.PAIRED_RESULT_DTYPE)
    cnt = np.zeros(3, dtype=np.int64)
    b = np.ascontiguousarray(bases).reshape(-1); q = np.ascontiguousarray(quals).reshape(-1)
    o = np.ascontiguousarray(offsets, dtype=np.uint64)
    rc = lib.pairedhost_align(C.byref(v), rix.handle, C.byref(p), C.byref(pp), C.c_int(stage), C.c_uint32(n), abi.ptr(b), abi.ptr(q),
                              abi.ptr(o), abi.ptr(prim), abi.ptr(alt), abi.ptr(cnt))
    assert rc == 0
    return prim, alt, cnt


@pytest.mark.parametrize("name", list(OPTS))
@pytest.mark.parametrize("tag", ["150", "100"])
def test_chimeric_align_matches_reference_fixture(golden_pairs, ref_index, name, tag):
    rix, gi = ref_index
    kw, pkw = OPTS[name]
    p = abi.default_params(max_read_len=160, **kw)
    pp = abi.default_paired_params(**pkw)
    z = golden_pairs
    prim, alt, cnt = host_align(gi, rix, p, pp, z["b" + tag], z["q" + tag], z["o" + tag], 0)
    key = "%s_%s_s0" % (name, tag)
    bad = compare_paired(z[key + "_primary"], prim, verbose=3, exclude=z[key + "_unstable"])
    assert not bad.any()
    assert (alt["status"] == z[key + "_alt"]["status"]).all()
    assert not prim["flags"].any()


@pytest.mark.parametrize("name", ["default_d8", "spacing_d8", "lvonly_d12"])
def test_intersecting_align_matches_reference_fixture(golden_pairs, ref_index, name):
    """IntersectingPairedEndAligner::align alone (no chimeric fallback): every field it defines, and its work counters."""
    rix, gi = ref_index
    kw, pkw = OPTS[name]
    p = abi.default_params(max_read_len=160, **kw)
    pp = abi.default_paired_params(**pkw)
    z = golden_pairs
    prim, alt, cnt = host_align(gi, rix, p, pp, z["b150"], z["q150"], z["o150"], 1)
    key = "%s_150_s1" % name
    fields = ["status", "direction", "location", "orig_location", "score", "mapq", "used_affine_gap_scoring", "bases_clipped_before",
              "bases_clipped_after", "ag_score", "seed_offset", "match_probability", "lv_indels", "used_gapless_clipping", "popular_seeds_skipped"]
    bad = compare_paired(z[key + "_primary"], prim, verbose=3, exclude=z[key + "_unstable"], fields=fields)
    assert not bad.any()
    found = (z[key + "_primary"]["status"] != 0).all(axis=1) & ~z[key + "_unstable"]
    assert (z[key + "_primary"]["probability_all_pairs"][found] == prim["probability_all_pairs"][found]).all()
    if not z[key + "_unstable"].any():
        assert [int(cnt[0]), int(cnt[1])] == z[key + "_counters"].tolist()      # LV / affine-gap locations scored


# ---------------------------------------------------------------------------------------- ALT liftover

@pytest.fixture(scope="module")
def alt_ref_index(tmp_path_factory):
    from tests.pairs_util import alt_liftover_genome
    d = str(tmp_path_factory.mktemp("paltidx"))
    g, sam, alt_args = alt_liftover_genome()
    synth.write_fasta(d + "/ref.fa", g)
    open(d + "/lift.sam", "w").write(sam)
    ref.build_index(d + "/ref.fa", d + "/idx", 20, threads=4, extra=alt_args + ["-altLiftoverFile", d + "/lift.sam"])
    gi = GenomeIndex.load_from_directory(d + "/idx")
    golden = util.load_golden_index("paired_alt_index.npz")
    assert (gi.genome_padded == golden.genome_padded).all()
    assert [(c.proj_begin, c.proj_rc, c.proj_cigar) for c in gi.contigs] == [(c.proj_begin, c.proj_rc, c.proj_cigar) for c in golden.contigs]
    return ref.RefIndex(d + "/idx"), gi


@pytest.mark.parametrize("name,kw", [("default_d8", dict(max_k=8)), ("default_d27", dict(max_k=27)), ("emitalt_d8", dict(max_k=8, emit_alt_alignments=1))])
@pytest.mark.parametrize("stage", [0, 1])
def test_alt_liftover_matches_reference_fixture(alt_ref_index, name, kw, stage):
    """Pairs drawn from two ALT contigs (one forward with indels in its projection CIGAR, one reverse-complemented behind a soft
    clip) of an index built with -altLiftoverFile: the best ALT alignment is projected onto the primary assembly and rescored
    there (IntersectingPairedEndAligner.cpp:2866-2968)."""
    rix, gi = alt_ref_index
    z = np.load(os.path.join(util.GOLDEN, "paired_alt_reads.npz"))
    p = abi.default_params(max_read_len=160, **kw)
    pp = abi.default_paired_params()
    prim, alt, cnt = host_align(gi, rix, p, pp, z["b"], z["q"], z["o"], stage)
    key = "%s_s%d" % (name, stage)
    fields = ["status", "direction", "location", "score", "mapq", "used_affine_gap_scoring", "bases_clipped_before", "bases_clipped_after",
              "ag_score", "liftover"] + (["aligned_as_pair"] if stage == 0 else [])
    bad = compare_paired(z[key + "_primary"], prim, verbose=3, exclude=z[key + "_unstable"], fields=fields)
    assert not bad.any()
    assert (alt["status"] == z[key + "_alt"]["status"]).all()
    assert z[key + "_primary"]["liftover"].all(axis=1).sum() > (100 if kw["max_k"] == 8 else 3)      # the path is exercised


# ---------------------------------------------------------------------------------------- secondary results (-om / -omax / -mpc)

def host_align_secondary(gi, rix, p, pp, sp, bases, quals, offsets, stage, stride, single_stride):
    lib = C.CDLL(HOSTLIB)
    v, keep = make_index_view(gi)
    n = (offsets.size - 1) // 2
    prim = np.zeros(n, dtype=abi.PAIRED_RESULT_DTYPE); alt = np.zeros(n, dtype=abi.PAIRED_RESULT_DTYPE)
    sec = np.zeros((n, stride), dtype=abi.PAIRED_RESULT_DTYPE); nsec = np.zeros(n, np.uint32)
    ssec = np.zeros((n, single_stride), dtype=abi.RESULT_DTYPE); nssec = np.zeros((n, 2), np.uint32)
    b = np.ascontiguousarray(bases).reshape(-1); q = np.ascontiguousarray(quals).reshape(-1)
    o = np.ascontiguousarray(offsets, dtype=np.uint64)
    rc = lib.pairedhost_align_secondary(C.byref(v), rix.handle, C.byref(p), C.byref(pp), C.byref(sp), C.c_int(stage), C.c_uint32(n), abi.ptr(b),
                                        abi.ptr(q), abi.ptr(o), abi.ptr(prim), abi.ptr(alt), abi.ptr(sec), C.c_uint32(stride), abi.ptr(nsec),
                                        abi.ptr(ssec), C.c_uint32(single_stride), abi.ptr(nssec))
    assert rc == 0
    return prim, alt, sec, nsec, ssec, nssec


@pytest.mark.parametrize("tag", ["150", "100"])
def test_secondary_results_match_reference_fixture(golden_pairs, ref_index, tag):
    """The paired-end control flow with -om / -omax / -mpc (recording in Phases 1-3, the final filtering, the single-end secondary
    results of the chimeric fallback) against tests/golden/paired_secondary.npz (scripts/make_golden_paired_secondary.py)."""
    from tests.pairs_util import compare_paired_secondary, load_paired_secondary_sets
    rix, gi = ref_index
    z = np.load(os.path.join(util.GOLDEN, "paired_secondary.npz"))
    b, q, o = golden_pairs["b" + tag], golden_pairs["q" + tag], golden_pairs["o" + tag]
    if tag == "150":
        o = o[:1201]; b = b[:int(o[-1])]; q = q[:int(o[-1])]
    for name, kw, pkw, om, omax, mpc in load_paired_secondary_sets(z):
        key = "%s_%s_" % (name, tag)
        ref_t = tuple(z[key + k] for k in ("primary", "alt", "secondary", "nsec", "single_secondary", "nssec"))
        p = abi.default_params(max_read_len=160, **kw)
        pp = abi.default_paired_params(**pkw)
        got = host_align_secondary(gi, rix, p, pp, abi.secondary_params(om, omax, mpc), b, q, o, 0, ref_t[2].shape[1], ref_t[4].shape[1])
        # excluded: what the reference itself answers differently from run to run (fixture), stale affine-gap cells (`reserved`),
        # and the pairs on which its answer is an accident of its buffer size (SNAPGPU_PAIR_REF_BUFFER_DEPENDENT)
        exclude = z[key + "unstable"] | (got[0]["reserved"] != 0) | ((got[0]["flags"] & 2) != 0)
        assert int(exclude.sum()) <= 2 + got[0].size // 100, name
        assert not compare_paired(ref_t[0], got[0], verbose=3, exclude=exclude).any(), name
        problems = compare_paired_secondary(ref_t, got, exclude)
        assert not problems, (name, problems)
        assert int(got[3].sum()) > 0 and int(got[5].sum()) > 0


# ---------------------------------------------------------------------------------------- without the compiled reference in the loop

@pytest.fixture
def restatement_single_aligner():
    """Plug oracle/align_oracle.c (the C restatement of BaseAligner, Hamming pass and alignAffineGap included) into the chimeric
    fallback instead of the compiled reference's BaseAligner: the whole paired-end path then runs on restatements only."""
    lib = C.CDLL(HOSTLIB)
    lib.pairedhost_use_restatement(1)
    yield
    lib.pairedhost_use_restatement(0)


@pytest.mark.parametrize("name", ["default_d8", "lvonly_d12", "spacing_d8"])
def test_chimeric_align_on_restatements_only(golden_pairs, ref_index, restatement_single_aligner, name):
    rix, gi = ref_index
    kw, pkw = OPTS[name]
    z = golden_pairs
    for tag in ("150", "100"):
        prim, alt, cnt = host_align(gi, rix, abi.default_params(max_read_len=160, **kw), abi.default_paired_params(**pkw),
                                    z["b" + tag], z["q" + tag], z["o" + tag], 0)
        key = "%s_%s_s0" % (name, tag)
        exclude = z[key + "_unstable"] | (prim["reserved"] != 0)
        assert int(exclude.sum()) <= 2 + prim.size // 100
        assert not compare_paired(z[key + "_primary"], prim, verbose=3, exclude=exclude).any()
        assert (alt["status"] == z[key + "_alt"]["status"]).all()


def test_secondary_results_on_restatements_only(golden_pairs, ref_index, restatement_single_aligner):
    from tests.pairs_util import compare_paired_secondary, load_paired_secondary_sets
    rix, gi = ref_index
    z = np.load(os.path.join(util.GOLDEN, "paired_secondary.npz"))
    b, q, o = golden_pairs["b100"], golden_pairs["q100"], golden_pairs["o100"]
    for name, kw, pkw, om, omax, mpc in load_paired_secondary_sets(z):
        key = "%s_100_" % name
        ref_t = tuple(z[key + k] for k in ("primary", "alt", "secondary", "nsec", "single_secondary", "nssec"))
        got = host_align_secondary(gi, rix, abi.default_params(max_read_len=160, **kw), abi.default_paired_params(**pkw),
                                   abi.secondary_params(om, omax, mpc), b, q, o, 0, ref_t[2].shape[1], ref_t[4].shape[1])
        exclude = z[key + "unstable"] | (got[0]["reserved"] != 0) | ((got[0]["flags"] & 2) != 0)
        assert int(exclude.sum()) <= 2 + got[0].size // 100, name
        assert not compare_paired(ref_t[0], got[0], verbose=3, exclude=exclude).any(), name
        assert not compare_paired_secondary(ref_t, got, exclude), name
