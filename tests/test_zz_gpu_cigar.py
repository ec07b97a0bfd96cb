"""SAMFormat::computeCigar (Landau-Vishkin variant) on the device, through the C ABI (snapgpu_compute_cigar_lv), against the
reference's answers (tests/golden/cigar_lv.npz, scripts/make_golden_cigar.py) and against the C restatement on fresh items.

Written in a round that had no GPU time left: the kernel was verified on the wavefront emulator (tests/test_emu_kernels.py) and
compiled for gfx950, not yet run on hardware -- hence the file name, which makes it the last module of the `-m gpu` run."""
import numpy as np
import pytest

from snap_amd import abi
from tests import util

pytestmark = pytest.mark.gpu


def check_against_fixture(aligner, z, use_m, sel=slice(None)):
    got = aligner.computeCigar(z["data"], z["off"][sel], z["length"][sel], z["loc"][sel], z["extra_before"][sel], bool(use_m), ops_stride=256)
    pre = "m%d_" % use_m
    for k in ("n_ops", "edit_distance", "add_front_clipping", "extra_clipped_after"):
        bad = np.nonzero(got[k] != z[pre + k][sel])[0]
        assert bad.size == 0, (k, bad[:5], got[k][bad[:5]], z[pre + k][sel][bad[:5]])
    exp_ops, exp_n = z[pre + "ops"][sel], z[pre + "n_ops"][sel]
    for i in range(len(exp_n)):
        assert util.cigar_text(got["ops"][i], got["n_ops"][i]) == util.cigar_text(exp_ops[i], exp_n[i]), i
    return got


def cigar_properties(got, length, extra_before):
    """What any CIGAR of a read of that length satisfies: query-consuming ops (M I = X) add up to the bases that were aligned,
    no two neighbouring ops share a code, never a leading D or a trailing I / D."""
    for i in range(len(length)):
        n = int(got["n_ops"][i])
        if n <= 0:
            continue
        ops = got["ops"][i, :n]
        codes, counts = ops & 15, ops >> 4
        assert (counts > 0).all()
        assert (codes[1:] != codes[:-1]).all(), i
        q = int(counts[np.isin(codes, (0, 1, 7, 8))].sum())
        assert q == int(length[i]) - int(extra_before[i]) - int(got["extra_clipped_after"][i]), i
        assert codes[0] != 2 and codes[-1] not in (1, 2), i
        assert int(got["edit_distance"][i]) == int(counts[np.isin(codes, (1, 2, 8))].sum()) or (codes == 0).any(), i


@pytest.fixture(scope="module")
def cig_aligner(golden_index):
    from snap_amd.aligner import BaseAligner
    a = BaseAligner(golden_index, abi.default_params(max_k=8, max_read_len=160))
    yield a
    a.close()


@pytest.fixture(scope="module")
def golden_cigar():
    import os
    return np.load(os.path.join(util.GOLDEN, "cigar_lv.npz"))


@pytest.mark.parametrize("use_m", [0, 1])
def test_compute_cigar_vs_reference_fixture(cig_aligner, golden_cigar, use_m):
    z = golden_cigar
    got = check_against_fixture(cig_aligner, z, use_m)
    cigar_properties(got, z["length"], z["extra_before"])


def test_compute_cigar_vs_restatement_on_fresh_reads(cig_aligner, golden_index):
    """20 000 reads cut from the golden genome with substitutions and indels (up to 12 edits), half of them shifted by a few
    bases: device == C restatement, and the CIGAR properties hold."""
    ix = golden_index
    rng = np.random.default_rng(77)
    pad = (ix.genome_padded.size - ix.n_bases) // 2
    G = ix.genome_padded[pad:]
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    cb = [int(x) for x in ix.contig_begin] + [int(ix.n_bases)]
    items, locs = [], []
    while len(items) < 20000:
        c = int(rng.integers(0, len(cb) - 1)); L = int(rng.choice([100, 150, 250]))
        lo, hi = cb[c], cb[c + 1] - ix.chromosome_padding - L - 20
        if hi <= lo:
            continue
        p = int(rng.integers(lo, hi))
        r = list(G[p:p + L + 16])
        for _ in range(int(rng.integers(0, 13))):
            j = int(rng.integers(0, L)); t = rng.random()
            if t < 0.6: r[j] = int(acgt[rng.integers(0, 4)])
            elif t < 0.8: del r[j]
            else: r.insert(j, int(acgt[rng.integers(0, 4)]))
        items.append(bytes(r[:L])); locs.append(p + (int(rng.integers(-3, 4)) if rng.random() < 0.5 else 0))
    data = np.frombuffer(b"".join(items), dtype=np.uint8)
    length = np.array([len(x) for x in items], dtype=np.int32)
    off = np.zeros(len(items), dtype=np.uint64); off[1:] = np.cumsum(length)[:-1]
    loc = np.array(locs, dtype=np.int64); xb = np.zeros(len(items), dtype=np.int32)
    for use_m in (False, True):
        got = cig_aligner.computeCigar(data, off, length, loc, xb, use_m, ops_stride=64)
        cigar_properties(got, length, xb)
        sub = np.arange(0, len(items), 10)                       # the scalar restatement on every tenth item
        exp = util.oracle_compute_cigar_lv(ix, data, off[sub], length[sub], loc[sub], xb[sub], use_m, ops_stride=64)
        for k in ("n_ops", "edit_distance", "add_front_clipping", "extra_clipped_after"):
            assert (got[k][sub] == exp[k]).all(), k
        for j, i in enumerate(sub):
            assert util.cigar_text(got["ops"][i], got["n_ops"][i]) == util.cigar_text(exp["ops"][j], exp["n_ops"][j]), i


def test_compute_cigar_argument_errors(cig_aligner, golden_index):
    from snap_amd.aligner import SnapGpuError
    d = np.frombuffer(b"ACGT" * 25, dtype=np.uint8)
    one = lambda **kw: cig_aligner.computeCigar(d, kw.get("off", [0]), kw.get("length", [100]), kw.get("loc", [2000]), kw.get("xb", [0]))
    with pytest.raises(SnapGpuError):
        one(off=[50])                                            # read runs past the data buffer
    with pytest.raises(SnapGpuError):
        one(loc=[int(golden_index.n_bases) + 5])                 # location outside the genome
    with pytest.raises(SnapGpuError):
        one(xb=[101])
    r = cig_aligner.computeCigar(d, [0], [100], [2000], [0], ops_stride=2)      # far too few op slots: the reference's -2
    assert int(r["edit_distance"][0]) in (-2, -1) or int(r["n_ops"][0]) <= 2


# ---------------------------------------------------------------------------------------------- affine-gap variant
AG_KEYS = ("n_ops", "edit_distance", "add_front_clipping", "extra_clipped_after", "back_clipping_missed")


def check_ag_against_fixture(aligner, z, use_m, step=1):
    """Device == a fresh reference object (the answer that is a function of the item alone); every item the reference answers
    differently depending on its object's history must be flagged by the device."""
    sel = slice(0, None, step)
    got = aligner.computeCigarAffineGap(z["data"], z["quals"], z["off"][sel], z["length"][sel], z["loc"][sel], z["extra_before"][sel],
                                        z["score"][sel], bool(use_m), ops_stride=256)
    pre = "m%d_" % use_m
    for k in AG_KEYS:
        bad = np.nonzero(got[k] != z[pre + k][sel])[0]
        assert bad.size == 0, (k, bad[:5], got[k][bad[:5]], z[pre + k][sel][bad[:5]])
    exp_ops, exp_n = z[pre + "ops"][sel], z[pre + "n_ops"][sel]
    for i in range(len(exp_n)):
        assert util.cigar_text(got["ops"][i], got["n_ops"][i]) == util.cigar_text(exp_ops[i], exp_n[i]), i
    assert not (z[pre + "unstable"][sel] & (got["stale"] == 0)).any(), "history-dependent item not flagged"
    return got


@pytest.fixture(scope="module")
def golden_cigar_ag():
    import os
    return np.load(os.path.join(util.GOLDEN, "cigar_ag.npz"))


@pytest.mark.parametrize("use_m", [0, 1])
def test_compute_cigar_ag_vs_reference_fixture(cig_aligner, golden_cigar_ag, use_m):
    z = golden_cigar_ag
    got = check_ag_against_fixture(cig_aligner, z, use_m)
    # query-consuming ops never exceed the bases that were aligned (a tail insertion is left to the caller to soft-clip)
    for i in range(len(z["off"])):
        n = int(got["n_ops"][i])
        if n <= 0:
            continue
        ops = got["ops"][i, :n]; codes, counts = ops & 15, ops >> 4
        q = int(counts[np.isin(codes, (0, 1, 7, 8))].sum())
        assert q <= int(z["length"][i]) - int(z["extra_before"][i]) - int(got["extra_clipped_after"][i]), i
        assert (counts > 0).all()


def test_compute_cigar_ag_vs_live_reference(cig_aligner, golden_index, tmp_path):
    """Fresh reads with indels against the compiled reference (a fresh AffineGapVectorizedWithCigar per item), where oracle/_ref
    is on the box: needs the index directory, which is rebuilt here from the golden genome."""
    import os
    from oracle import ref
    from snap_amd import synth
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not on this box")
    ix = golden_index
    pad = (ix.genome_padded.size - ix.n_bases) // 2
    G = ix.genome_padded[pad:]
    cb = [int(x) for x in ix.contig_begin] + [int(ix.n_bases)]
    contigs = [(c.name, G[cb[i]:cb[i + 1] - (ix.chromosome_padding if i + 1 < len(cb) - 1 or True else 0)].copy()) for i, c in enumerate(ix.contigs)]
    # strip the padding the index added after each contig: bases are ACGTN, padding is 'n'
    contigs = [(n, s[:int(np.nonzero(s != ord('n'))[0][-1]) + 1]) for n, s in contigs]
    fa = str(tmp_path / "ref.fa"); synth.write_fasta(fa, contigs)
    alt = [c.name for c in ix.contigs if c.is_alt]
    ref.build_index(fa, str(tmp_path / "idx"), ix.seed_len, threads=4, extra=sum((["-altContigName", a] for a in alt), []))
    from snap_amd.index import GenomeIndex
    ix2 = GenomeIndex.load_from_directory(str(tmp_path / "idx"))
    assert (ix2.contig_begin == ix.contig_begin).all() and (ix2.genome_padded == ix.genome_padded).all()
    ri = ref.RefIndex(str(tmp_path / "idx"))
    rng = np.random.default_rng(123)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    items, quals, locs, ks = [], [], [], []
    while len(items) < 4000:
        c = int(rng.integers(0, len(cb) - 1)); L = int(rng.choice([100, 150, 250]))
        lo, hi = cb[c], cb[c + 1] - ix.chromosome_padding - L - 20
        if hi <= lo:
            continue
        p = int(rng.integers(lo, hi))
        r = list(G[p:p + L + 16]); ne = int(rng.integers(0, 10))
        for _ in range(ne):
            j = int(rng.integers(0, L)); t = rng.random()
            if t < 0.5: r[j] = int(acgt[rng.integers(0, 4)])
            elif t < 0.75: del r[j]
            else: r.insert(j, int(acgt[rng.integers(0, 4)]))
        items.append(bytes(r[:L])); quals.append(rng.integers(35, 74, size=L).astype(np.uint8).tobytes())
        locs.append(p + (int(rng.integers(-3, 4)) if rng.random() < 0.3 else 0)); ks.append(ne + int(rng.integers(0, 3)))
    data = np.frombuffer(b"".join(items), dtype=np.uint8); q = np.frombuffer(b"".join(quals), dtype=np.uint8)
    length = np.array([len(x) for x in items], dtype=np.int32)
    off = np.zeros(len(items), dtype=np.uint64); off[1:] = np.cumsum(length)[:-1]
    loc = np.array(locs, dtype=np.int64); xb = np.zeros(len(items), dtype=np.int32); k = np.array(ks, dtype=np.int32)
    for use_m in (False, True):
        exp = ri.compute_cigar_ag(data, q, off, length, loc, xb, k, use_m, fresh_object=True, ops_stride=128)
        got = cig_aligner.computeCigarAffineGap(data, q, off, length, loc, xb, k, use_m, ops_stride=128)
        for key in AG_KEYS:
            assert (got[key] == exp[key]).all(), key
        for i in range(len(items)):
            assert util.cigar_text(got["ops"][i], got["n_ops"][i]) == util.cigar_text(exp["ops"][i], exp["n_ops"][i]), i


# ---------------------------------------------------------------------------------------------- result -> SAM record fields
SAMF_SETS = ["default", "lvonly", "eqx", "lvonly_eqx", "clipfront", "clipfront_lvonly"]


def check_sam_fields_against_reference_cli(golden_index, z, tag, step=1):
    """FLAG / RNAME / POS / MAPQ / CIGAR / NM as the unmodified reference CLI printed them (tests/golden/sam_fields.npz,
    scripts/make_golden_sam_fields.py) from the reads, Read::clip's outcome and the reference aligner's results."""
    from snap_amd.aligner import BaseAligner
    kw = dict(use_affine_gap=0) if "lvonly" in tag else {}
    a = BaseAligner(golden_index, abi.default_params(max_k=14, max_read_len=400, **kw))
    try:
        n = len(z[tag + "_front_clip"])
        sel = np.arange(0, n, step)
        offs = z["offsets"]
        lens = (offs[1:] - offs[:-1])[sel]
        o2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        take = np.concatenate([np.arange(offs[i], offs[i + 1]) for i in sel]).astype(np.int64)
        got = a.samFields(z["bases"][take], z["quals"][take], o2, z[tag + "_front_clip"][sel], z[tag + "_data_len"][sel], z[tag + "_results"][sel],
                          bool(z[tag + "_use_m"]))
    finally:
        a.close()
    for k in ("flag", "contig", "pos", "mapq", "nm", "n_ops"):
        bad = np.nonzero(got[k] != z[tag + "_" + k][sel])[0]
        assert bad.size == 0, (tag, k, sel[bad[:5]], got[k][bad[:5]], z[tag + "_" + k][sel][bad[:5]])
    for j, i in enumerate(sel):
        assert util.cigar_text(got["ops"][j], got["n_ops"][j]) == util.cigar_text(z[tag + "_ops"][i], z[tag + "_n_ops"][i]), (tag, i)
    return got


@pytest.mark.parametrize("tag", SAMF_SETS)
def test_sam_fields_vs_reference_cli_fixture(golden_index, tag):
    import os
    z = np.load(os.path.join(util.GOLDEN, "sam_fields.npz"))
    got = check_sam_fields_against_reference_cli(golden_index, z, tag)
    assert int((got["flag"] & 4 != 0).sum()) > 100 and int((got["flag"] & 16 != 0).sum()) > 1000


def check_align_sam_single_against_reference_cli(golden_index, z, tag, n=None):
    """snapgpu_align_sam_single -- one upload, the align kernel applying Read::clip's outcome itself, results handed to the SAM-field kernels
    (k_samf_dp8 + k_sam_fields) in HBM -- from the unclipped reads alone: the SingleAlignmentResults must be the reference aligner's and the
    fields what the unmodified reference CLI printed (the same fixture as above, which keeps both).  Also against the two calls it replaces."""
    from snap_amd.aligner import BaseAligner
    kw = dict(use_affine_gap=0) if "lvonly" in tag else {}
    prm = abi.default_params(max_read_len=400, **kw)                       # the CLI's defaults: -d 27 (scripts/make_golden_sam_fields.py)
    n = n or len(z[tag + "_front_clip"])
    offs = z["offsets"][:n + 1]
    bases, quals = z["bases"][:int(offs[-1])], z["quals"][:int(offs[-1])]
    fc, dl = z[tag + "_front_clip"][:n], z[tag + "_data_len"][:n]
    # SingleAligner.cpp:211-232: shorter than -mrl 50 or more Ns than maxDist: not given to the aligner
    skip = np.array([dl[i] < 50 or int((bases[int(offs[i]) + fc[i]:int(offs[i]) + fc[i] + dl[i]] == ord("N")).sum()) > int(prm.max_k) for i in range(n)], dtype=np.uint8)
    a = BaseAligner(golden_index, prm)
    try:
        res, alt, got = a.alignSam(bases, quals, offs, fc, dl, skip, bool(z[tag + "_use_m"]))
        exp = z[tag + "_results"][:n]
        keep = np.nonzero(skip == 0)[0]
        bad = util.compare_results(exp[keep], res[keep])
        assert not bad, (tag, bad[:3] if isinstance(bad, list) else bad)
        assert (res["status"][skip != 0] == 0).all() and (res["score"][skip != 0] == -1).all()          # NotFound, as the reference's writer sees them
        for k in ("flag", "contig", "pos", "mapq", "nm", "n_ops"):
            ne = np.nonzero(got[k] != z[tag + "_" + k][:n])[0]
            assert ne.size == 0, (tag, k, ne[:5], got[k][ne[:5]], z[tag + "_" + k][:n][ne[:5]])
        for i in range(n):
            assert util.cigar_text(got["ops"][i], got["n_ops"][i]) == util.cigar_text(z[tag + "_ops"][i], z[tag + "_n_ops"][i]), (tag, i)
        two = a.samFields(bases, quals, offs, fc, dl, res, bool(z[tag + "_use_m"]))
        for k in ("flag", "contig", "pos", "mapq", "nm", "n_ops", "stale"):
            assert (two[k] == got[k]).all(), (tag, k)
        assert (two["ops"] == got["ops"]).all()
        # an empty batch is a call like any other
        r0, _, g0 = a.alignSam(bases[:0], quals[:0], np.zeros(1, np.uint64), fc[:0], dl[:0], skip[:0])
        assert r0.size == 0 and g0["flag"].size == 0
    finally:
        a.close()
    return int((skip != 0).sum())


@pytest.mark.parametrize("tag", ["default", "lvonly", "eqx", "clipfront"])
def test_align_sam_single_vs_reference_cli_fixture(golden_index, tag):
    import os
    z = np.load(os.path.join(util.GOLDEN, "sam_fields.npz"))
    assert check_align_sam_single_against_reference_cli(golden_index, z, tag) > 10


# ---------------------------------------------------------------------------------------------- paired-end writer
def check_sam_fields_paired_against_reference_cli(z, tag, n_pairs=None):
    """All 9 computed fields of both records of each pair and the order of the two records, as the unmodified reference CLI printed them
    (tests/golden/sam_fields_paired.npz, scripts/make_golden_sam_fields_paired.py)."""
    from snap_amd.aligner import BaseAligner
    kw = dict(use_affine_gap=0) if tag.startswith("lvonly") else {}
    a = BaseAligner(util.load_golden_index("paired_index.npz"), abi.default_params(max_read_len=400, **kw))
    try:
        npairs = len(z[tag + "_first_written"]) if n_pairs is None else n_pairs
        n = 2 * npairs
        offs = z["offsets"][:n + 1]
        got = a.samFieldsPaired(z["bases"][:int(offs[-1])], z["quals"][:int(offs[-1])], offs, z["front_clip"][:n], z["data_len"][:n],
                                z[tag + "_results"][:npairs], bool(z[tag + "_use_m"]))
    finally:
        a.close()
    for k in ("flag", "contig", "pos", "mapq", "nm", "n_ops", "rnext", "pnext", "tlen"):
        bad = np.nonzero(got[k] != z[tag + "_" + k][:n])[0]
        assert bad.size == 0, (tag, k, bad[:5], got[k][bad[:5]], z[tag + "_" + k][:n][bad[:5]])
    for i in range(n):
        assert util.cigar_text(got["ops"][i], got["n_ops"][i]) == util.cigar_text(z[tag + "_ops"][i], z[tag + "_n_ops"][i]), (tag, i)
    assert (got["first_written"] == z[tag + "_first_written"][:npairs]).all()
    return got


@pytest.mark.parametrize("tag", ["default", "lvonly", "eqx"])
def test_sam_fields_paired_vs_reference_cli_fixture(tag):
    import os
    z = np.load(os.path.join(util.GOLDEN, "sam_fields_paired.npz"))
    got = check_sam_fields_paired_against_reference_cli(z, tag)
    assert int((got["flag"] & 2 != 0).sum()) > 1500 and int((got["first_written"] == 1).sum()) > 300


def run_sam_fields_device_form(aligner, z, tag, n, to_dev, to_host):
    """snapgpu_sam_fields_single_device with every array in device memory (to_dev / to_host move numpy arrays there and back)."""
    import ctypes as C
    offs = z["offsets"][:n + 1]
    tot = int(offs[-1])
    ins = [z["bases"][:tot], z["quals"][:tot], offs.astype(np.uint64), z[tag + "_front_clip"][:n].astype(np.int32), z[tag + "_data_len"][:n].astype(np.int32),
           np.ascontiguousarray(z[tag + "_results"][:n])]
    stride = 64
    outs = [np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int64), np.zeros(n, np.int32), np.zeros((n, stride), np.uint32),
            np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)]
    d_in = [to_dev(x) for x in ins]; d_out = [to_dev(x) for x in outs]
    ptr = lambda t: C.c_void_p(t[1])
    rc = aligner.lib.snapgpu_sam_fields_single_device(aligner.handle, C.c_uint32(n), C.c_uint32(160), *[ptr(t) for t in d_in], C.c_int(int(z[tag + "_use_m"])),
                                                      *[ptr(t) for t in d_out[:5]], C.c_uint32(stride), *[ptr(t) for t in d_out[5:]], None)
    assert rc == 0, rc
    return [to_host(t, o) for t, o in zip(d_out, outs)]


def test_sam_fields_device_pointer_form(golden_index):
    """Same answers as the host-pointer form, with everything resident in HBM.  Device buffers come from hipMalloc / hipMemcpy of the HIP
    runtime libsnapgpu.so itself is linked against (tests/util.py: HipBuffers) -- not from torch, whose bundled runtime cannot be
    initialised once another copy of libamdhip64 owns the device (GPUTEST_r01: 'No HIP GPUs are available')."""
    import os
    from snap_amd.aligner import BaseAligner
    z = np.load(os.path.join(util.GOLDEN, "sam_fields.npz"))
    n = 1200
    hip = util.HipBuffers()
    def to_dev(x):
        return (None, hip.upload(x))
    def to_host(t, like):
        return hip.download(t[1], like)
    a = BaseAligner(golden_index, abi.default_params(max_k=14, max_read_len=400))
    try:
        flag, contig, pos, mapq, ops, n_ops, nm, stale = run_sam_fields_device_form(a, z, "default", n, to_dev, to_host)
    finally:
        hip.free_all()
        a.close()
    for k, v in (("flag", flag), ("contig", contig), ("pos", pos), ("mapq", mapq), ("nm", nm), ("n_ops", n_ops)):
        assert (v == z["default_" + k][:n]).all(), k
    for i in range(n):
        assert util.cigar_text(ops[i], n_ops[i]) == util.cigar_text(z["default_ops"][i], z["default_n_ops"][i]), i
