"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI
(libsnapgpu.so); expectations come from the committed golden fixtures the compiled reference
produced (scripts/make_golden.py) and, where oracle/_ref travelled to the box, from the
reference itself on fresh seeded inputs.  Integer fields and FP64 probabilities must be
bit-identical."""
import json
import os

import numpy as np
import pytest

from snap_amd import abi, synth
from tests import util
from oracle import ref

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(util.GOLDEN, "reference_kats.json")))


@pytest.fixture(scope="module")
def aligner(golden_index):
    from snap_amd.aligner import BaseAligner
    a = BaseAligner(golden_index, abi.default_params(max_k=8, max_read_len=160))
    yield a
    a.close()


def _loaded_native(name="libsnapgpu.so"):
    return any(name in line for line in open("/proc/self/maps"))


def test_native_library_is_what_runs(aligner):
    assert _loaded_native(), "libsnapgpu.so is not mapped into this process"


def test_tables_match_restatement(aligner):
    lib = util.oracle_lib()
    t = aligner.debug_tables()
    assert (t["phred"] == np.ctypeslib.as_array(lib.oracle_phred_table(), (256,))).all()
    assert (t["indel"] == np.ctypeslib.as_array(lib.oracle_indel_table(), (1001,))).all()
    assert (t["perfect"] == np.ctypeslib.as_array(lib.oracle_perfect_table(), (1001,))).all()
    assert t["seed_prob"] == lib.oracle_seed_prob(20)
    assert [int(x) for x in t["wrapped"][:20]] == [lib.oracle_wrapped_next_seed(20, i) for i in range(20)]
    # MAPQ thresholds reproduce (int)(-10*log10(x)) of the host libm
    rng = np.random.default_rng(1)
    thr = t["mapq_threshold"]
    for _ in range(5000):
        pa = float(rng.random() * 5); pb = float(pa * rng.random() ** 3)
        x = 1 - pb / pa
        m = max(i for i in range(71) if x <= thr[i]) if pb / pa < 1 else 70
        assert m == lib.oracle_compute_mapq(pa, pb, 0, 0)


def test_lookup_seeds_vs_reference_fixture(aligner, golden_primitives):
    z = golden_primitives
    nh, hits = aligner.lookupSeed32(z["seeds"], z["seed_hits"].shape[2])
    assert (nh == z["seed_n_hits"]).all()
    assert (hits == z["seed_hits"]).all()


def test_lookup_kernels_agree_counts_only_and_odd_batch_sizes(aligner, golden_primitives):
    """The stand-alone probe kernel (lookup16.h: sixteen probes per wave pass) in its counts-only form (hit lists read, not stored) and
    with batch sizes that leave the last pass partly empty, against the fixture; SNAPGPU_LOOKUP8 selects the eight-probe kernel, which
    must give the same bytes."""
    import os
    z = golden_primitives
    seeds = np.ascontiguousarray(z["seeds"], dtype=np.uint8)
    exp_nh, exp_hits = z["seed_n_hits"], z["seed_hits"]
    for n in (1, 7, 8, 9, 33, seeds.shape[0]):
        nh, hits = aligner.lookupSeed32(seeds[:n], exp_hits.shape[2])
        assert (nh == exp_nh[:n]).all() and (hits == exp_hits[:n]).all(), n
    os.environ["SNAPGPU_LOOKUP8"] = "1"
    try:
        nh8, hits8 = aligner.lookupSeed32(seeds, exp_hits.shape[2])
    finally:
        del os.environ["SNAPGPU_LOOKUP8"]
    assert (nh8 == exp_nh).all() and (hits8 == exp_hits).all()
    # counts only, through the device-pointer entry (torch on hardware; under the emulator host memory is device memory)
    n = seeds.shape[0]
    # (device buffers through the HIP runtime libsnapgpu.so itself uses -- tests/util.HipBuffers -- never torch: its bundled runtime
    #  cannot initialise the GPU once another copy of libamdhip64 owns it, and the attempt breaks later hipMallocs of this process)
    hip = util.HipBuffers(emu=hasattr(aligner.lib, "emu_total_ops"))
    d_seeds = hip.upload(np.ascontiguousarray(seeds.reshape(-1)))
    d_nh = hip.upload(np.zeros((n, 2), dtype=np.int64))
    aligner.counters(reset=True)
    aligner.lookup_device(n, d_seeds, d_nh, 0, 300)
    got = hip.download(d_nh, np.zeros((n, 2), dtype=np.int64))
    assert (got == exp_nh).all()
    c = aligner.counters(reset=True)
    valid = exp_nh[:, 0] >= 0
    assert c["n_hash_table_lookups"] == int(valid.sum())
    assert c["n_hits_consumed"] == int(np.minimum(np.maximum(exp_nh[valid], 0), 300).sum())
    assert c["n_overflow_lists"] == int((exp_nh[valid] > 1).sum())


def test_lv_known_answers_and_fixture(aligner, golden_primitives):
    texts = [c["text"].encode() for c in KATS["lv"]]
    pats = [c["pattern"].encode() for c in KATS["lv"]]
    got = aligner.computeEditDistance(1, texts, pats, [b"2" * len(p) for p in pats], [c["k"] for c in KATS["lv"]])
    assert got["score"].tolist() == [c["expect"] for c in KATS["lv"]]
    z = golden_primitives
    for d in (1, -1):
        tt = [t if d == 1 else t[::-1] for t in z["lv_texts"]]
        got = aligner.computeEditDistance(d, tt, list(z["lv_pats"]), list(z["lv_quals"]), z["lv_k"])
        for key in ("score", "match_probability", "net_indel", "total_indels", "text_span"):
            assert (got[key] == z["lv%+d_%s" % (d, key)]).all(), (d, key)


def test_affine_gap_known_answers(golden_index):
    from snap_amd.aligner import BaseAligner
    # the reference's test fixture uses a 3' bonus of 5 (AffineGapVectorizedTest.cpp:9)
    a = BaseAligner(golden_index, abi.default_params(max_k=8, max_read_len=160, three_prime_end_bonus=5))
    texts = [c["text"].encode() for c in KATS["ag"]]
    pats = [c["pattern"].encode() for c in KATS["ag"]]
    got = a.computeScoreAffine(1, texts, pats, [b"2" * len(p) for p in pats], [c["w"] for c in KATS["ag"]],
                               [c["score_init"] for c in KATS["ag"]], [0] * len(texts), [0] * len(texts))
    a.close()
    assert got["ag_score"].tolist() == [c["expect"] for c in KATS["ag"]]


def test_affine_gap_vs_reference_fixture(aligner, golden_primitives):
    z = golden_primitives
    n_checked = 0
    for d in (1, -1):
        tt = [t if d == 1 else t[::-1] for t in z["ag_texts"]]
        got = aligner.computeScoreAffine(d, tt, list(z["ag_pats"]), list(z["ag_quals"]), z["ag_w"], z["ag_si"], z["ag_rc"],
                                         z["ag_banded"], z["ag_clip"])
        for i in range(len(tt)):
            o = util.oracle_ag(d, z["ag_banded"][i], tt[i], z["ag_pats"][i], z["ag_quals"][i], z["ag_w"][i], z["ag_si"][i],
                               z["ag_rc"][i], z["ag_clip"][i])
            if o["stale_reads"]:
                continue            # reference result undefined (depends on its object's history)
            n_checked += 1
            exp = z["ag%+d_ag_score" % d][i]
            assert got["ag_score"][i] == exp, (d, i)
            if exp != -1:
                for key in ("text_offset", "pattern_offset", "n_edits", "match_probability"):
                    assert got[key][i] == z["ag%+d_%s" % (d, key)][i], (d, i, key)
    assert n_checked > 1000


def check_affine_gap_call_sequences(aligner, z, tags=("short", "long"), step=1):
    """tests/golden/ag_sequence.npz (scripts/make_golden_ag_sequence.py): affine-gap problems as calls in order on ONE newly constructed
    reference object; snapgpu_affine_gap_sequence -- one wave, the exact form over an image of the object's array kept from call to call --
    must give the reference's answer for EVERY call, the ones whose traceback read what earlier calls left behind included."""
    n_dep = 0
    for tag in tags:
        n = len(z[tag + "_texts"]) // step
        texts, pats, quals = list(z[tag + "_texts"][:n]), list(z[tag + "_pats"][:n]), list(z[tag + "_quals"][:n])
        for d in (1, -1):
            tt = [t if d == 1 else t[::-1] for t in texts]
            got = aligner.computeScoreAffine(d, tt, pats, quals, z[tag + "_w"][:n], z[tag + "_si"][:n], z[tag + "_rc"][:n], z[tag + "_banded"][:n], sequence=True)
            pre = "%s%+d_" % (tag, d)
            assert (got["ag_score"] == z[pre + "ag_score"][:n]).all(), (tag, d)
            ok = z[pre + "ag_score"][:n] != -1
            for key in ("text_offset", "pattern_offset", "n_edits", "match_probability"):
                assert (got[key][ok] == z[pre + key][:n][ok]).all(), (tag, d, key)
            dep = z[pre + "depends_on_history"][:n]
            assert ((got["stale_steps"][dep] & 0xffff) > 0).all()           # an answer can only depend on earlier calls through such steps
            n_dep += int(dep.sum())
    return n_dep


def test_affine_gap_call_sequences_vs_reference_fixture(golden_index):
    from snap_amd.aligner import BaseAligner
    import os
    z = np.load(os.path.join(util.GOLDEN, "ag_sequence.npz"), allow_pickle=True)
    a = BaseAligner(golden_index, abi.default_params(max_k=8, max_read_len=160))
    try:
        assert check_affine_gap_call_sequences(a, z) > 40
    finally:
        a.close()


def check_affine_gap_wide_bands(aligner, n=1200, seed=1001, max_len=380):
    """Seeded fuzz over WIDE bands (w 13 .. 31: segments of 40 .. 64 positions -- ag_win.h: ag_banded_win2, every banded call of `-d 20` and the
    limit-30 calls of the paired path), patterns up to 380 bases (the 192-, 256- and 384-position instantiations), three clipping modes, both
    directions, against the C restatement (pinned to the reference by tests/test_oracle.py)."""
    rng = np.random.default_rng(seed)
    texts, pats, quals, ws, sis, rcs, bands, clips = [], [], [], [], [], [], [], []
    for _ in range(n):
        big = _ % 5 == 4                  # a fifth of the problems beyond the window forms: w 32 .. 45 (banded, segments of 72 .. 96 positions) -- more than
        w = int(rng.integers(32, 46)) if big else int(rng.integers(13, 32))      # 192 positions: the LDS form inside the 256- / 384-position instantiations (round 6)
        L = int(rng.integers(3 * (2 * w + 1), max(3 * (2 * w + 1) + 1, max_len)))
        t = bytes(rng.choice(list(b"ACGT"), size=L + 80).astype(np.uint8))
        p = bytearray(t[:L])
        for _e in range(int(rng.integers(0, 8))):
            j = int(rng.integers(0, len(p)))
            r = rng.random()
            if r < 0.5: p[j] = b"ACGT"[rng.integers(0, 4)]
            elif r < 0.75 and len(p) > 1: del p[j:j + int(rng.integers(1, max(2, w // 2)))]
            else: p[j:j] = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, max(2, w // 2)))).astype(np.uint8))
        if rng.random() < 0.3:
            k = int(rng.integers(3, max(4, L // 3)))
            p[len(p) - k:] = bytes(rng.choice(list(b"ACGT"), size=k).astype(np.uint8))
        p = bytes(p[:L])
        if len(p) < 3 * (2 * w + 1): p = p + t[len(p):3 * (2 * w + 1)]
        texts.append(t[:len(p) + w]); pats.append(p)
        quals.append(bytes(rng.integers(35, 74, size=len(p), dtype=np.uint8)))
        ws.append(w); sis.append(int(rng.integers(20, 400))); rcs.append(int(rng.integers(0, 2)))
        bands.append(0 if (big and _ % 10 == 9) else 1); clips.append(int(rng.integers(0, 3)))        # (and some of those as full, unbanded problems)
    n_cmp = 0
    order = np.argsort([len(p) for p in pats], kind="stable")
    for lo, hi in ((0, 193), (193, 257), (257, 1 << 20)):                 # one call per instantiation of the batch kernel
        def positions(i):                 # num_seg * seg_len of the problem (ag_dims)
            if not bands[i]: return -(-len(pats[i]) // 8) * 8
            sl = -(-(2 * ws[i] + 1) // 8) * 8
            return -(-len(pats[i]) // sl) * sl
        sel = [int(i) for i in order if lo <= positions(i) < hi]
        if not sel: continue
        for d in (1, -1):
            tt = [texts[i] if d == 1 else texts[i][::-1] for i in sel]
            got = aligner.computeScoreAffine(d, tt, [pats[i] for i in sel], [quals[i] for i in sel], [ws[i] for i in sel], [sis[i] for i in sel],
                                             [rcs[i] for i in sel], [bands[i] for i in sel], [clips[i] for i in sel])
            for x, i in enumerate(sel):
                o = util.oracle_ag(d, bands[i], tt[x], pats[i], quals[i], ws[i], sis[i], rcs[i], clips[i])
                if o["stale_reads"]:
                    continue
                n_cmp += 1
                assert got["ag_score"][x] == o["ag_score"], (d, i, len(pats[i]), ws[i])
                if o["ag_score"] != -1:
                    for key in ("text_offset", "pattern_offset", "n_edits", "match_probability"):
                        assert got[key][x] == o[key], (d, i, key, clips[i], len(pats[i]), ws[i])
    return n_cmp


def test_affine_gap_wide_bands_vs_restatement(golden_index):
    from snap_amd.aligner import BaseAligner
    a = BaseAligner(golden_index, abi.default_params(max_k=20, max_read_len=400))
    try:
        assert check_affine_gap_wide_bands(a) > 1500
    finally:
        a.close()


def test_affine_gap_wide_band_call_sequences_vs_reference_fixture(golden_index, step=1):
    """tests/golden/ag_sequence_wide.npz (scripts/make_golden_ag_sequence_wide.py): calls in order on one reference object with w 13 .. 31 --
    the exact (image-keeping) instantiation of ag_banded_win2."""
    from snap_amd.aligner import BaseAligner
    import os
    z = np.load(os.path.join(util.GOLDEN, "ag_sequence_wide.npz"), allow_pickle=True)
    a = BaseAligner(golden_index, abi.default_params(max_k=20, max_read_len=200))
    try:
        assert check_affine_gap_call_sequences(a, z, tags=("wide",), step=step) >= (4 if step == 1 else 0)
    finally:
        a.close()


def test_affine_gap_clipping_modes_vs_restatement(aligner):
    """Seeded fuzz over the three clipping modes (0 off, 1 useClippingOptimizations, 2 = with useAltLiftover), banded/full and
    window/register forms, both directions, against the C restatement (itself pinned to the reference by tests/test_oracle.py)."""
    rng = np.random.default_rng(78)
    texts, pats, quals, ws, sis, rcs, bands, clips = [], [], [], [], [], [], [], []
    for _ in range(1500):
        L = int(rng.integers(8, 150))
        t = bytes(rng.choice(list(b"ACGT"), size=L + 60).astype(np.uint8))
        p = bytearray(t[:L])
        for _e in range(int(rng.integers(0, 5))):
            j = int(rng.integers(0, len(p)))
            r = rng.random()
            if r < 0.5: p[j] = b"ACGT"[rng.integers(0, 4)]
            elif r < 0.75 and len(p) > 1: del p[j]
            else: p.insert(j, b"ACGT"[rng.integers(0, 4)])
        if rng.random() < 0.3:
            k = int(rng.integers(3, max(4, L // 3)))
            p[len(p) - k:] = bytes(rng.choice(list(b"ACGT"), size=k).astype(np.uint8))
        p = bytes(p[:L]) or b"A"
        w = int(rng.integers(1, 30))
        texts.append(t[:len(p) + w]); pats.append(p)
        quals.append(bytes(rng.integers(35, 74, size=len(p), dtype=np.uint8)))
        ws.append(w); sis.append(int(rng.integers(20, 200))); rcs.append(int(rng.integers(0, 2)))
        bands.append(1 if len(p) >= 3 * (2 * w + 1) else 0); clips.append(int(rng.integers(0, 3)))
    n_cmp = 0
    for d in (1, -1):
        tt = texts if d == 1 else [x[::-1] for x in texts]
        got = aligner.computeScoreAffine(d, tt, pats, quals, ws, sis, rcs, bands, clips)
        for i in range(len(tt)):
            o = util.oracle_ag(d, bands[i], tt[i], pats[i], quals[i], ws[i], sis[i], rcs[i], clips[i])
            if o["stale_reads"]:
                continue
            n_cmp += 1
            assert got["ag_score"][i] == o["ag_score"], (d, i)
            if o["ag_score"] != -1:
                for key in ("text_offset", "pattern_offset", "n_edits", "match_probability"):
                    assert got[key][i] == o[key], (d, i, key, clips[i])
    assert n_cmp > 2500


@pytest.mark.parametrize("name,kw", [("default_d8", dict(max_k=8)), ("lvonly_d8", dict(max_k=8, use_affine_gap=0)),
                                     ("default_d27", dict(max_k=27)), ("emitalt_d8", dict(max_k=8, emit_alt_alignments=1))])
def test_align_read_vs_reference_fixture(golden_index, golden_reads, name, kw):
    from snap_amd.aligner import BaseAligner
    z = golden_reads
    a = BaseAligner(golden_index, abi.default_params(max_read_len=160, **kw))
    for tag, L in (("100", 100), ("150", 150)):
        b, q = z["b" + tag], z["q" + tag]
        offs = np.arange(b.shape[0] + 1, dtype=np.uint64) * L
        prim, alt = a.AlignRead(b, q, offs)
        # EVERY read is compared, none excluded: the expectation is what a reference aligner newly constructed in zero-filled memory
        # answers for the read (tests/golden/fresh_overrides.npz over the committed shared-object run; util.with_fresh_overrides).
        # `reserved` != 0 marks the reads whose banded affine-gap traceback left the band (redone by the exact pass): every read on
        # which the shared-object reference run differs from the fresh one, or moved with its history, must be among them.
        key = "%s_%s_" % (name, tag)
        flagged = (prim["reserved"] & 0x3fffffff) != 0
        exp_prim, patched = util.with_fresh_overrides(z[key + "primary"], key + "primary")
        history_dependent = z[key + "unstable"].copy(); history_dependent[patched] = True
        assert not (history_dependent & ~flagged).any(), "reference-unstable read not flagged"
        problems = util.compare_results(exp_prim, prim)
        ea, _ = util.with_fresh_overrides(z[key + "alt"], key + "alt")
        assert (ea["status"] == alt["status"]).all()
        sel = ea["status"] != 0
        problems += util.compare_results(ea[sel], alt[sel], "firstALT")
        assert not problems, problems
        c = a.counters(reset=True)
        exp = z[key + "counters"]
        assert [c["n_hash_table_lookups"], c["n_lv_locations"], c["n_ag_locations"]] == exp.tolist()
    a.close()


def test_results_do_not_depend_on_batch_order_or_size(golden_index, golden_reads):
    from snap_amd.aligner import BaseAligner
    z = golden_reads
    a = BaseAligner(golden_index, abi.default_params(max_k=8, max_read_len=160))
    b, q = z["b100"], z["q100"]
    n = b.shape[0]
    perm = np.random.default_rng(4).permutation(n)
    offs = np.arange(n + 1, dtype=np.uint64) * 100
    p1, _ = a.AlignRead(b, q, offs)
    p2, _ = a.AlignRead(b[perm], q[perm], offs)
    assert not util.compare_results(p1[perm], p2)
    p3, _ = a.AlignRead(b[:17], q[:17], offs[:18])
    assert not util.compare_results(p1[:17], p3)
    a.close()


def test_ragged_and_degenerate_reads(golden_index):
    """Empty batch, reads shorter than a seed, mixed lengths, reads longer than max_read_len."""
    from snap_amd.aligner import BaseAligner, SnapGpuError
    a = BaseAligner(golden_index, abi.default_params(max_k=8, max_read_len=160))
    g = golden_index.genome
    c0 = golden_index.contigs[0].begin
    lens = [0, 5, 19, 20, 21, 63, 64, 65, 100, 160]
    bases = np.concatenate([g[c0 + 300:c0 + 300 + L] for L in lens]).astype(np.uint8)
    quals = np.full(bases.size, ord("I"), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    prim, _ = a.AlignRead(bases, quals, offs)
    for i, L in enumerate(lens):
        if L < 20:
            assert prim["status"][i] == abi.NOT_FOUND and prim["location"][i] == abi.INVALID_GENOME_LOCATION_32
        else:
            # an exact copy of the reference aligns where it came from with score 0
            assert prim["status"][i] != abi.NOT_FOUND and prim["score"][i] == 0, (L, prim[i])
            # (the origin may be one copy of a repeat: require that the reported place spells the read)
            loc = int(prim["location"][i])
            seq = g[loc:loc + L]
            if prim["direction"][i] == 1:
                seq = synth._COMP[seq[::-1]]
            assert (seq == g[c0 + 300:c0 + 300 + L]).all()
    e, _ = a.AlignRead(np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert e.size == 0
    with pytest.raises(SnapGpuError):
        a.AlignRead(np.full(200, ord("A"), np.uint8), np.full(200, ord("I"), np.uint8), np.array([0, 200], np.uint64))
    a.close()


def test_property_exact_copies_align_to_origin_at_scale(golden_index):
    """Size-independent property on a large batch: error-free reads map to their origin, score 0,
    and the reverse complement of a read maps to the same place on the other strand."""
    from snap_amd.aligner import BaseAligner
    ix = golden_index
    rng = np.random.default_rng(8)
    L, n = 150, 200_000
    c = ix.contigs[1]
    end = ix.contigs[2].begin - ix.chromosome_padding
    pos = rng.integers(c.begin, end - L, size=n)
    b = ix.genome[pos[:, None] + np.arange(L)[None, :]]
    ok = (b != ord("N")).all(axis=1) & (b != ord("n")).all(axis=1)
    b = np.ascontiguousarray(b[ok]); pos = pos[ok]
    q = np.full(b.shape, ord("I"), dtype=np.uint8)
    offs = np.arange(b.shape[0] + 1, dtype=np.uint64) * L
    a = BaseAligner(ix, abi.default_params(max_k=8, max_read_len=160))
    fwd, _ = a.AlignRead(b, q, offs)
    rc, _ = a.AlignRead(np.ascontiguousarray(synth._COMP[b[:, ::-1]]), q, offs)
    a.close()
    assert (fwd["status"] != abi.NOT_FOUND).all() and (fwd["score"] == 0).all()
    uniq = fwd["mapq"] >= 10
    assert uniq.mean() > 0.3
    assert (fwd["location"][uniq] == pos[uniq]).all() and (fwd["direction"][uniq] == 0).all()
    assert (rc["score"] == 0).all()
    assert (rc["location"][uniq] == pos[uniq]).all() and (rc["direction"][uniq] == 1).all()
    # (MAPQ need not be strand-symmetric: seed order differs between the two orientations)


@pytest.mark.skipif(not ref.available() or not os.path.exists(ref.CLI_PATH), reason="oracle/_ref not on this box")
def test_align_read_vs_live_reference_on_fresh_genome(tmp_path):
    """Fresh seeded genome with repeats, index built by the reference's builder on this box, 30k reads."""
    from snap_amd.aligner import BaseAligner
    from snap_amd.index import GenomeIndex
    g = synth.make_genome(77, 3_000_000, n_contigs=3, repeat_frac=0.45, max_copies=400, n_run_frac=0.002)
    fa = str(tmp_path / "ref.fa"); synth.write_fasta(fa, g)
    ref.build_index(fa, str(tmp_path / "idx"), 20, threads=max(1, os.cpu_count() or 1))
    ix = GenomeIndex.load_from_directory(str(tmp_path / "idx"))
    ri = ref.RefIndex(str(tmp_path / "idx"))
    # 150 / 250 bp: register affine-gap variants (3 and 6 chunks); 500 bp: the LDS formulation (AGC = 0)
    for L, kw, mrl, n in ((150, dict(max_k=8), 256, 15000), (250, dict(max_k=20), 256, 15000), (500, dict(max_k=27), 512, 3000)):
        p = abi.default_params(max_read_len=mrl, **kw)
        rd = synth.make_reads(78 + L, g, n, L, sub=0.015, ins=0.002, dele=0.002, n_frac=0.0005)
        with ref.fresh_objects():       # a newly constructed reference aligner per read: its answer is a function of the read alone
            pr, ar, cr, _ = ri.align_single(p, rd["bases"], rd["quals"], rd["offsets"], threads=os.cpu_count() or 1)
        ps, _, _, _ = ri.align_single(p, rd["bases"], rd["quals"], rd["offsets"], threads=2)     # ... and the usual long-lived objects
        a = BaseAligner(ix, p)
        pg, ag = a.AlignRead(rd["bases"], rd["quals"], rd["offsets"])
        c = a.counters()
        a.close()
        assert not util.compare_results(pr, pg)                      # every read, no exclusion
        flagged = (pg["reserved"] & 0x3fffffff) != 0
        shared_differs = np.zeros(len(pg), bool)
        for f in pr.dtype.names:
            if f != "reserved":
                shared_differs |= (pr[f] != ps[f]) & ((pr["status"] != 0) | (f not in util.UNDEFINED_WHEN_NOT_FOUND))
        assert not (shared_differs & ~flagged).any(), "the reference's answer depends on its object's history for a read that is not flagged"
        assert [c["n_hash_table_lookups"], c["n_lv_locations"], c["n_ag_locations"]] == [cr["lookups"], cr["lv"], cr["ag"]]
