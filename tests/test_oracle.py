"""CPU tests that pin the oracle.

1. oracle/snap_oracle.c (plain-C restatement) against the reference's own known-answer tests
   (tests/golden/reference_kats.json <- tests/LandauVishkinTest.cpp:11-32,
   tests/AffineGapVectorizedTest.cpp:39-67).
2. The restatement against fixtures produced by the compiled reference (tests/golden/primitives.npz,
   script scripts/make_golden.py) -- bit-exact, FP64 included.
3. When oracle/_ref is present (this container): the restatement against the reference itself on
   fresh seeded fuzz, and the reference against the committed AlignRead fixtures.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests import util
from oracle import ref

KATS = json.load(open(os.path.join(util.GOLDEN, "reference_kats.json")))
have_ref = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")


def test_lv_known_answers():
    for c in KATS["lv"]:
        t, p = c["text"].encode(), c["pattern"].encode()
        got = util.oracle_lv(1, t, p, b"2" * len(p), c["k"])
        assert got["score"] == c["expect"], c


def test_affine_gap_known_answers():
    for c in KATS["ag"]:
        t, p = c["text"].encode(), c["pattern"].encode()
        got = util.oracle_ag(1, 0, t, p, b"2" * len(p), c["w"], c["score_init"], 0, 0, params=tuple(c["params"]))
        assert got["ag_score"] == c["expect"], (c, got)


def test_lv_restatement_vs_reference_fixture(golden_primitives):
    z = golden_primitives
    for d in (1, -1):
        for i in range(len(z["lv_texts"])):
            t = z["lv_texts"][i] if d == 1 else z["lv_texts"][i][::-1]
            got = util.oracle_lv(d, t, z["lv_pats"][i], z["lv_quals"][i], z["lv_k"][i])
            for key in ("score", "match_probability", "net_indel", "total_indels", "text_span"):
                assert got[key] == z["lv%+d_%s" % (d, key)][i], (d, i, key)


def test_affine_gap_restatement_vs_reference_fixture(golden_primitives):
    z = golden_primitives
    n_checked = n_stale = 0
    for d in (1, -1):
        for i in range(len(z["ag_texts"])):
            t = z["ag_texts"][i] if d == 1 else z["ag_texts"][i][::-1]
            got = util.oracle_ag(d, z["ag_banded"][i], t, z["ag_pats"][i], z["ag_quals"][i], z["ag_w"][i], z["ag_si"][i],
                                 z["ag_rc"][i], z["ag_clip"][i])
            if got["stale_reads"]:
                n_stale += 1        # the reference's answer depends on what its object computed before
                continue
            n_checked += 1
            exp_score = z["ag%+d_ag_score" % d][i]
            assert got["ag_score"] == exp_score, (d, i)
            if exp_score != -1:
                for key in ("text_offset", "pattern_offset", "n_edits", "match_probability"):
                    assert got[key] == z["ag%+d_%s" % (d, key)][i], (d, i, key)
    assert n_checked > 1000 and n_stale < 20


def test_lookup_restatement_vs_reference_fixture(golden_index, golden_primitives):
    z = golden_primitives
    seeds, nh, hits = z["seeds"], z["seed_n_hits"], z["seed_hits"]
    for i in range(len(seeds)):
        r = util.oracle_lookup(golden_index, seeds[i].tobytes())
        if r is None:
            assert nh[i, 0] == -1
            continue
        for d in range(2):
            assert r[d][0] == nh[i, d], (i, d)
            n = min(int(nh[i, d]), hits.shape[2])
            assert (r[d][1][:n] == hits[i, d, :n]).all()


def test_hits_are_sorted_descending(golden_index, golden_primitives):
    # GenomeIndex.cpp:879-889: overflow lists are stored in descending genome order
    z = golden_primitives
    for i in range(0, len(z["seeds"]), 7):
        r = util.oracle_lookup(golden_index, z["seeds"][i].tobytes())
        if r:
            for d in range(2):
                h = r[d][1].astype(np.int64)
                assert (np.diff(h) < 0).all()


@have_ref
def test_tables_and_scalars_vs_reference():
    lib = util.oracle_lib()
    ph, ind, pf = ref.tables(1001, 1001)
    assert (np.ctypeslib.as_array(lib.oracle_phred_table(), (256,)) == ph).all()
    assert (np.ctypeslib.as_array(lib.oracle_indel_table(), (1001,)) == ind).all()
    assert (np.ctypeslib.as_array(lib.oracle_perfect_table(), (1001,)) == pf).all()
    for s in range(16, 33):
        assert lib.oracle_seed_prob(s) == ref.seed_prob(s)
        for w in range(s):
            assert lib.oracle_wrapped_next_seed(s, w) == ref.wrapped_next_seed(s, w)
    rng = np.random.default_rng(3)
    for _ in range(20000):
        pa = float(rng.random() * 5); pb = float(pa * rng.random() ** 4)
        sk = int(rng.integers(0, 40))
        assert lib.oracle_compute_mapq(pa, pb, 0, sk) == ref.compute_mapq(pa, pb, 0, sk)


@have_ref
def test_restatement_vs_reference_fresh_fuzz():
    rng = np.random.default_rng(2026)
    texts, pats, quals, ks = [], [], [], []
    for _ in range(1500):
        L = int(rng.integers(1, 200))
        t = bytes(rng.choice(list(b"ACGT"), size=L + 40).astype(np.uint8))
        p = bytearray(t[:L])
        for _e in range(int(rng.integers(0, 6))):
            j = int(rng.integers(0, len(p)))
            r = rng.random()
            if r < 0.5: p[j] = b"ACGT"[rng.integers(0, 4)]
            elif r < 0.75 and len(p) > 1: del p[j]
            else: p.insert(j, b"ACGT"[rng.integers(0, 4)])
        p = bytes(p[:L]) or b"A"
        texts.append(t); pats.append(p); quals.append(bytes(rng.integers(35, 74, size=len(p), dtype=np.uint8)))
        ks.append(int(rng.integers(0, 40)))
    for d in (1, -1):
        tt = texts if d == 1 else [x[::-1] for x in texts]
        r = ref.landau_vishkin(d, tt, pats, quals, ks)
        for i in range(len(tt)):
            got = util.oracle_lv(d, tt[i], pats[i], quals[i], ks[i])
            for key in r:
                assert got[key] == r[key][i], (d, i, key)


@have_ref
def test_affine_gap_restatement_vs_reference_fresh_fuzz():
    """All three clipping modes (0 off, 1 useClippingOptimizations, 2 = 1 + useAltLiftover), banded and full, both directions."""
    rng = np.random.default_rng(77)
    texts, pats, quals, ws, sis, rcs, bands, clips = [], [], [], [], [], [], [], []
    for _ in range(1200):
        L = int(rng.integers(8, 160))
        t = bytes(rng.choice(list(b"ACGT"), size=L + 60).astype(np.uint8))
        p = bytearray(t[:L])
        for _e in range(int(rng.integers(0, 5))):
            j = int(rng.integers(0, len(p)))
            r = rng.random()
            if r < 0.5: p[j] = b"ACGT"[rng.integers(0, 4)]
            elif r < 0.75 and len(p) > 1: del p[j]
            else: p.insert(j, b"ACGT"[rng.integers(0, 4)])
        if rng.random() < 0.3:                              # garbage tail: something to clip
            k = int(rng.integers(3, max(4, L // 3)))
            p[len(p) - k:] = bytes(rng.choice(list(b"ACGT"), size=k).astype(np.uint8))
        p = bytes(p[:L]) or b"A"
        w = int(rng.integers(1, 30))
        texts.append(t[:len(p) + w]); pats.append(p)
        quals.append(bytes(rng.integers(35, 74, size=len(p), dtype=np.uint8)))
        ws.append(w); sis.append(int(rng.integers(20, 200))); rcs.append(int(rng.integers(0, 2)))
        bands.append(1 if len(p) >= 3 * (2 * w + 1) else 0); clips.append(int(rng.integers(0, 3)))
    for d in (1, -1):
        tt = texts if d == 1 else [x[::-1] for x in texts]
        r = ref.affine_gap(d, tt, pats, quals, ws, sis, rcs, bands, clips)
        n_cmp = 0
        for i in range(len(tt)):
            got = util.oracle_ag(d, bands[i], tt[i], pats[i], quals[i], ws[i], sis[i], rcs[i], clips[i])
            if got["stale_reads"]:
                continue                                    # the reference's answer depends on its object's history here
            n_cmp += 1
            assert got["ag_score"] == r["ag_score"][i], (d, i)
            if got["ag_score"] != -1:
                for key in ("text_offset", "pattern_offset", "n_edits", "match_probability"):
                    assert got[key] == r[key][i], (d, i, key, clips[i])
        assert n_cmp > 1000


@have_ref
def test_reference_reproduces_committed_alignread_fixtures(golden_reads, tmp_path):
    """The fixtures really are what the reference computes (guards against stale goldens)."""
    from snap_amd import abi
    import subprocess
    # rebuild the index from the committed genome bytes is not possible without the FASTA; instead
    # re-run scripts/make_golden.py's reference calls against its cached index if present
    idx_dir = "/tmp/snap_golden/idx"
    if not os.path.exists(os.path.join(idx_dir, "GenomeIndex")):
        pytest.skip("golden working directory not present (run scripts/make_golden.py)")
    ri = ref.RefIndex(idx_dir)
    z = golden_reads
    p = abi.default_params(max_k=8, max_read_len=160)
    offs = np.arange(z["b100"].shape[0] + 1, dtype=np.uint64) * 100
    prim, alt, cnt, _ = ri.align_single(p, z["b100"], z["q100"], offs, threads=2)
    # a handful of reads are history-dependent in the reference itself (see scripts/make_golden.py)
    unstable = np.zeros(len(prim), bool)
    for k in z.files:
        if k.endswith("_100_unstable"):
            unstable |= z[k]
    assert not util.compare_results(z["default_d8_100_primary"], prim, exclude=unstable)


def test_secondary_fixture_obeys_the_reference_contract():
    """tests/golden/secondary_reads.npz (the compiled reference's answers with -om / -omax / -mpc): what
    finalizeSecondaryResults promises (BaseAligner.cpp:2423-2553) holds for every read -- scores within min(maxK, best + om),
    at most -omax results, at most -mpc per contig counting the primary, supplementary == is-ALT."""
    import ast
    z = np.load(os.path.join(util.GOLDEN, "secondary_reads.npz"))
    ix = util.load_golden_index()
    begins = np.array([c.begin for c in ix.contigs], dtype=np.int64)
    first_alt = min(c.begin for c in ix.contigs if c.is_alt)
    for name, kw, om, omax, mpc in [(str(r[0]), ast.literal_eval(str(r[1])), int(r[2]), int(r[3]), int(r[4])) for r in z["sets"]]:
        for tag in ("100", "150"):
            key = "%s_%s_" % (name, tag)
            prim, sec, nsec = z[key + "primary"], z[key + "secondary"], z[key + "nsec"]
            assert int(nsec.max()) <= min(omax, sec.shape[1]) and int(nsec.sum()) > 0
            live = np.arange(sec.shape[1])[None, :] < nsec[:, None]
            worst = np.minimum(kw["max_k"], prim["score"] + om)
            assert (sec["score"][live] <= np.broadcast_to(worst[:, None], live.shape)[live]).all()
            assert (sec["score_prior_to_clipping"][live] == sec["score"][live]).all()
            assert (sec["status"][live] == 2).all() and (sec["mapq"][live] == 0).all()
            if kw.get("alt_awareness", 1):
                assert ((sec["supplementary"][live] != 0) == (sec["location"][live] >= first_alt)).all()
            if mpc > 0:
                for i in np.nonzero(nsec > 0)[0]:
                    contigs = np.searchsorted(begins, sec["location"][i, :nsec[i]], side="right") - 1
                    counts = np.bincount(contigs, minlength=len(begins))
                    if prim["status"][i] != 0:
                        counts[np.searchsorted(begins, prim["location"][i], side="right") - 1] += 1
                        assert counts.max() <= mpc, (name, tag, int(i))


# ---------------------------------------------------------------------------------------- the restatement of AlignRead (oracle/align_oracle.c)

RESTATEMENT_SETS = dict(default_d8=dict(max_k=8), lvonly_d8=dict(max_k=8, use_affine_gap=0), default_d27=dict(max_k=27),
                        emitalt_d8=dict(max_k=8, emit_alt_alignments=1))


@pytest.mark.parametrize("name", list(RESTATEMENT_SETS))
def test_align_read_restatement_vs_reference_fixture(golden_index, golden_reads, name):
    """oracle_align_read -- the plain-C restatement of BaseAligner::AlignRead -- against what the compiled reference returned
    for the 4 000 golden reads (tests/golden/tiny_reads.npz): every field of the primary result, bit for bit."""
    from snap_amd import abi
    for tag in ("100", "150"):
        b, q = golden_reads["b" + tag], golden_reads["q" + tag]
        n, L = b.shape
        prim, alt = util.oracle_align_reads(golden_index, abi.default_params(max_read_len=160, **RESTATEMENT_SETS[name]), b, q,
                                            np.arange(n + 1, dtype=np.uint64) * L)
        key = "%s_%s_" % (name, tag)
        # every read, none excluded: the restatement keeps the traceback arrays of one aligner's two affine-gap objects across the
        # calls of a read (snap_oracle.c: oracle_ag_bind_objects), i.e. it answers like a newly constructed reference aligner --
        # which is what the fixture holds once the fresh-object overrides are patched in
        exp, patched = util.with_fresh_overrides(golden_reads[key + "primary"], key + "primary")
        assert not util.compare_results(exp, prim)
        moved = golden_reads[key + "unstable"].copy(); moved[patched] = True
        assert not (moved & (prim["reserved"] == 0)).any()
        e_alt, _ = util.with_fresh_overrides(golden_reads[key + "alt"], key + "alt")
        assert (e_alt["status"] == alt["status"]).all()
        if name == "emitalt_d8":
            assert not util.compare_results(e_alt, alt, "first ALT", exclude=alt["status"] == 0)


def test_align_read_restatement_with_secondary_results_vs_reference_fixture(golden_index, golden_reads):
    """... and with -om / -omax / -mpc: secondary results, order included (tests/golden/secondary_reads.npz)."""
    import ast
    from snap_amd import abi
    z = np.load(os.path.join(util.GOLDEN, "secondary_reads.npz"))
    for r in z["sets"]:
        name, kw, om, omax, mpc = str(r[0]), ast.literal_eval(str(r[1])), int(r[2]), int(r[3]), int(r[4])
        for tag in ("100", "150"):
            b, q = golden_reads["b" + tag], golden_reads["q" + tag]
            n, L = b.shape
            key = "%s_%s_" % (name, tag)
            prim, alt, sec, nsec = util.oracle_align_reads(golden_index, abi.default_params(max_read_len=160, **kw), b, q,
                                                           np.arange(n + 1, dtype=np.uint64) * L, secondary=abi.secondary_params(om, omax, mpc),
                                                           sec_stride=max(64, z[key + "secondary"].shape[1]))
            e_prim, _ = util.with_fresh_overrides(z[key + "primary"], "sec_" + key + "primary")
            e_sec, _ = util.with_fresh_overrides(z[key + "secondary"], "sec_" + key + "secondary")
            e_nsec, _ = util.with_fresh_overrides(z[key + "nsec"], "sec_" + key + "nsec")
            problems = util.compare_results(e_prim, prim)                      # every read, no exclusion
            problems += util.compare_secondary(e_sec, e_nsec, sec, nsec, np.zeros(n, bool))
            assert not problems, (name, tag, problems)


@have_ref
def test_align_read_restatement_vs_live_reference_on_fresh_reads(tmp_path):
    """Fresh seeded reads (ragged lengths, N-rich, reads at contig ends, unalignable) on a fresh repeat-rich genome with an ALT contig:
    the restatement against the compiled reference, run here."""
    from snap_amd import abi, synth
    from snap_amd.index import GenomeIndex
    d = str(tmp_path)
    g = synth.make_genome(4242, 300_000, n_contigs=3, repeat_frac=0.4, max_copies=80, repeat_len=(120, 1500), n_run_frac=0.003)
    rng = np.random.default_rng(17)
    altc = g[1][1][30_000:38_000].copy()
    m = rng.random(altc.size) < 0.012
    altc[m] = synth._ACGT[rng.integers(0, 4, size=int(m.sum()))]
    g.append(("chrB_alt", altc))
    synth.write_fasta(d + "/ref.fa", g)
    ref.build_index(d + "/ref.fa", d + "/idx", 20, threads=4, extra=["-altContigName", "chrB_alt"])
    gi = GenomeIndex.load_from_directory(d + "/idx")
    ri = ref.RefIndex(d + "/idx")
    rd = synth.make_reads(5, g, 1500, 130, sub=0.02, ins=0.004, dele=0.004, n_frac=0.002)
    bases, quals, offs = [], [], [0]
    cat0 = g[0][1]
    for i in range(rd["bases"].shape[0]):
        L = int(rng.integers(18, 131))                       # ragged, some shorter than the seed
        b, q = rd["bases"][i, :L].copy(), rd["quals"][i, :L].copy()
        if i % 97 == 0:
            b[rng.integers(0, L, size=min(L, 10))] = ord("N")
        if i % 131 == 0:
            b = cat0[:L].copy()                              # the very start of a contig
        if i % 137 == 0:
            b = cat0[-L:].copy()                             # ... and its end
        if i % 89 == 0:
            b = synth._ACGT[rng.integers(0, 4, size=L)]      # unalignable
        bases.append(b); quals.append(q); offs.append(offs[-1] + L)
    bases = np.concatenate(bases); quals = np.concatenate(quals); offs = np.array(offs, dtype=np.uint64)
    for kw in (dict(max_k=8), dict(max_k=14, extra_search_depth=2), dict(max_k=8, use_affine_gap=0, num_seeds=0, seed_coverage=4.0)):
        p = abi.default_params(max_read_len=160, **kw)
        with ref.fresh_objects():
            rp, ra, _, _ = ri.align_single(p, bases, quals, offs, threads=4)
        op, oa = util.oracle_align_reads(gi, p, bases, quals, offs)
        assert not util.compare_results(rp, op), kw                            # every read, no exclusion
        assert (ra["status"] == oa["status"]).all()


# ---------------------------------------------------------------------------------------- CIGAR (first piece of SURVEY.md 8(f) rank 1)

def _lv_cigar(lib, fn_ref, text_buf, t0, text_len, pat_buf, p0, pattern_len, k, use_m):
    """oracle_lv_cigar (or, fn_ref set, the reference's LandauVishkinWithCigar::computeEditDistance) on strings placed inside
    larger buffers -- the reference looks past both ends, so both sides must see the same neighbours."""
    tb = C.create_string_buffer(bytes(text_buf), len(text_buf)); pb = C.create_string_buffer(bytes(pat_buf), len(pat_buf))
    out = C.create_string_buffer(1024); tu = C.c_int(-1); ni = C.c_int(0)
    tp, pp = C.c_void_p(C.addressof(tb) + t0), C.c_void_p(C.addressof(pb) + p0)
    if fn_ref is not None:
        r = fn_ref(tp, text_len, pp, pattern_len, k, use_m, out, 1024, C.byref(tu), C.byref(ni))
    else:
        r = lib.oracle_lv_cigar(tp, text_len, -t0, len(text_buf) - t0, pp, pattern_len, -p0, len(pat_buf) - p0, k, use_m, out, 1024,
                                C.byref(tu), C.byref(ni))
    return r, out.value.decode(), (tu.value if r >= 0 else None), (ni.value if r > 0 else 0)


def test_lv_cigar_restatement_on_the_references_own_vectors():
    """tests/LandauVishkinTest.cpp:34-121: 15 string pairs x {=/X, M} CIGARs."""
    lib = util.oracle_lib()
    for c in KATS["lv_cigar"]:
        t, p = c["text"].encode(), c["pattern"].encode()
        for use_m, key in ((0, "cigar"), (1, "cigar_m")):
            got = _lv_cigar(lib, None, b"\x01" * 16 + t + b"\x00" * 16, 16, len(t), b"\x02" * 16 + p + b"\x00" * 16, 16, len(p), c["k"], use_m)
            assert got[1] == c[key], (c, use_m, got)


@have_ref
def test_lv_cigar_restatement_vs_live_reference_fuzz():
    """Seeded mutated / indel'd / truncated strings: edit distance, CIGAR, text used and net indel equal the reference's."""
    lib = util.oracle_lib()
    rlib = ref.lib()
    rng = np.random.default_rng(2026)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    n_checked = 0
    for it in range(3000):
        L = int(rng.integers(1, 160))
        g = rng.choice(acgt, size=L + 220)
        if it % 3 == 0:
            g[60:60 + L // 2] = g[60 + L // 2:60 + 2 * (L // 2)][:L // 2]       # low-complexity: tandem duplication (ambiguous indel placement)
        text0 = 40
        pat = bytearray(g[text0:text0 + L].tobytes())
        for _ in range(int(rng.integers(0, 5))):                                 # substitutions
            pat[int(rng.integers(0, len(pat)))] = int(rng.choice(acgt))
        for _ in range(int(rng.integers(0, 3))):                                 # indels
            pos = int(rng.integers(0, len(pat)))
            if rng.random() < 0.5 and len(pat) > 2:
                del pat[pos:pos + int(rng.integers(1, 4))]
            else:
                pat[pos:pos] = bytes(rng.choice(acgt, size=int(rng.integers(1, 4))).tolist())
        pat = bytes(pat) or b"A"
        text_len = len(pat) + int(rng.integers(-6, 20)) if it % 7 == 0 else len(pat) + 30      # sometimes shorter than the pattern
        text_len = max(0, min(text_len, g.size - text0 - 8))
        k = int(rng.integers(0, 14))
        pbuf = bytes(rng.choice(acgt, size=24).tolist()) + pat + bytes(rng.choice(acgt, size=24).tolist())
        for use_m in (0, 1):
            a = _lv_cigar(lib, None, g.tobytes(), text0, text_len, pbuf, 24, len(pat), k, use_m)
            b = _lv_cigar(lib, rlib.snapref_lv_cigar, g.tobytes(), text0, text_len, pbuf, 24, len(pat), k, use_m)
            assert a == b, (it, use_m, k, text_len, pat, a, b)
            n_checked += a[0] > 0
    assert n_checked > 1500


def test_compute_cigar_restatement_vs_reference_fixture(golden_index):
    """oracle/cigar_oracle.c: SAMFormat::computeCigar (LV variant) == what the compiled reference returned for the 3 460 items of
    tests/golden/cigar_lv.npz (aligned reads, shifted reads with leading D / I, reads hanging off a contig, "*" cases)."""
    z = np.load(os.path.join(util.GOLDEN, "cigar_lv.npz"))
    for use_m in (0, 1):
        sel = np.arange(len(z["off"]))[::2 if use_m == 0 else 5]            # every second item; with M instead of = / X (the op alphabet only) every fifth
        o = util.oracle_compute_cigar_lv(golden_index, z["data"], z["off"][sel], z["length"][sel], z["loc"][sel], z["extra_before"][sel], bool(use_m), ops_stride=256)
        pre = "m%d_" % use_m
        for k in ("n_ops", "edit_distance", "add_front_clipping", "extra_clipped_after"):
            assert (o[k] == z[pre + k][sel]).all(), k
        for j, i in enumerate(sel):
            assert util.cigar_text(o["ops"][j], o["n_ops"][j]) == util.cigar_text(z[pre + "ops"][i], z[pre + "n_ops"][i]), i
    assert int((z["m0_add_front_clipping"] > 0).sum()) > 50 and int((z["m0_add_front_clipping"] < 0).sum()) > 50
    assert int((z["m0_extra_clipped_after"] > 0).sum()) > 30 and int((z["m0_n_ops"] < 0).sum()) >= 1


def test_restatement_stop_on_first_hit_and_explore_popular_seeds_vs_live_reference(tmp_path):
    """-f / -x in oracle/align_oracle.c (oracle_set_aligner_flags) against the compiled reference running with the same flags: every read,
    every field, on a repeat-rich genome with a low -h (so that -x changes what is found)."""
    import os
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built here")
    from snap_amd import abi, synth
    from snap_amd.index import GenomeIndex
    g = synth.make_genome(4711, 600_000, n_contigs=2, repeat_frac=0.5, max_copies=900, repeat_len=(150, 1200), max_divergence=0.03)
    fa = str(tmp_path / "ref.fa"); synth.write_fasta(fa, g)
    ref.build_index(fa, str(tmp_path / "idx"), 20, threads=max(1, os.cpu_count() or 1))
    ix = GenomeIndex.load_from_directory(str(tmp_path / "idx"))
    ri = ref.RefIndex(str(tmp_path / "idx"))
    p = abi.default_params(max_k=8, max_read_len=160)
    p.max_hits = 40
    rd = synth.make_reads(4712, g, 1500, 120, sub=0.02, ins=0.001, dele=0.001)
    lib = util.oracle_lib()
    with ref.fresh_objects():
        base, _, _, _ = ri.align_single(p, rd["bases"], rd["quals"], rd["offsets"], threads=os.cpu_count() or 1)
    for f, x in ((True, False), (False, True), (True, True)):
        with ref.fresh_objects(), ref.aligner_flags(stop_on_first_hit=f, explore_popular_seeds=x):
            pr, ar, _, _ = ri.align_single(p, rd["bases"], rd["quals"], rd["offsets"], threads=os.cpu_count() or 1)
        lib.oracle_set_aligner_flags(1 if f else 0, 1 if x else 0)
        try:
            po, ao = util.oracle_align_reads(ix, p, rd["bases"], rd["quals"], rd["offsets"])
        finally:
            lib.oracle_set_aligner_flags(0, 0)
        assert not util.compare_results(pr, po), (f, x)
        assert not util.compare_results(ar, ao, what="first_alt"), (f, x)
        assert any((pr[k] != base[k]).any() for k in ("status", "location", "score", "mapq")), (f, x)      # (the flags do something here)
