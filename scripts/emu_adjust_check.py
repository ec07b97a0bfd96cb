"""-ae on the emulator (or, with SNAPGPU_TEST_LIB unset and a GPU, on the device) against the live reference: BaseAligner::AlignRead with
secondary results and ignoreAlignmentAdjustmentsForOm = false on reads made to need the adjuster -- indels in the first / last bases,
reads that hang over either end of a contig, both strands.  Usage: python scripts/emu_adjust_check.py [n_reads] [om] [read_len]"""
import os, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def adjust_reads(seed, contigs, n, L):
    """reads that exercise AlignmentAdjuster: a third plain (substitutions + scattered indels), a third with an indel of 1-4 bases within
    the first or last 6 bases, a third hanging 1-25 bases over the start or the end of their contig; half of everything reverse-complemented"""
    from snap_amd import synth
    rng = np.random.default_rng(seed)
    base = synth.make_reads(seed, contigs, n, L, sub=0.01, ins=0.002, dele=0.002)
    b, q = base["bases"].copy(), base["quals"].copy()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.zeros(256, np.uint8); comp[:] = ord("N")
    for x, y in zip(b"ACGT", b"TGCA"): comp[x] = y
    lens = [len(g) for _, g in contigs]
    for i in range(n):
        kind = i % 3
        if kind == 0: continue
        ci = int(rng.integers(len(contigs))); g = contigs[ci][1]
        if len(g) < 3 * L: continue
        if kind == 1:
            pos = int(rng.integers(50, len(g) - 2 * L - 50)); d = int(rng.integers(1, 5)); at = int(rng.integers(1, 7))
            if rng.random() < 0.5: at = L - at - d
            if rng.random() < 0.5:                      # deletion of d bases from the read at `at`
                r = np.concatenate([g[pos:pos + at], g[pos + at + d:pos + L + d]])
            else:                                       # insertion of d bases
                r = np.concatenate([g[pos:pos + at], acgt[rng.integers(0, 4, d)], g[pos + at:pos + L - d]])
        else:
            over = int(rng.integers(1, 26))
            if rng.random() < 0.5: r = np.concatenate([acgt[rng.integers(0, 4, over)], g[:L - over]])
            else: r = np.concatenate([g[len(g) - (L - over):], acgt[rng.integers(0, 4, over)]])
        r = r[:L].copy()
        sub = rng.random(L) < 0.01
        r[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        if rng.random() < 0.5: r = comp[r[::-1]]
        b[i] = r
    return b, q


if __name__ == "__main__":
    import snap_amd.aligner as al
    if os.environ.get("SNAPGPU_TEST_LIB", "emu") != "gpu":
        al.LIB_PATH = os.environ.get("SNAPGPU_TEST_LIB", os.path.join(ROOT, "tests", "emu", "_build", "libsnapgpu_emu.so")); al._lib = None
    from snap_amd import synth, abi
    from snap_amd.index import GenomeIndex
    from snap_amd.aligner import BaseAligner
    from oracle import ref
    from tests import util
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    om = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    d = tempfile.mkdtemp(prefix="adjust")
    contigs = synth.make_genome(77, 400_000, n_contigs=40, repeat_frac=0.3, max_copies=30, repeat_len=(150, 600), max_divergence=0.03)
    synth.write_fasta(d + "/g.fa", contigs)
    ref.build_index(d + "/g.fa", d + "/idx", seed_len=20, threads=8)
    gi = GenomeIndex.load_from_directory(d + "/idx")
    b, q = adjust_reads(5, contigs, n, L)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    p = abi.default_params(max_k=10, max_read_len=L + 10, extra_search_depth=2)
    rix = ref.RefIndex(d + "/idx")
    with ref.fresh_objects(), ref.adjust_alignments():
        e_prim, _, e_sec, e_nsec = rix.align_single_secondary(p, om, b, q, offs, threads=8)
    with ref.fresh_objects():
        p_prim, _, p_sec, p_nsec = rix.align_single_secondary(p, om, b, q, offs, threads=8)
    a = BaseAligner(gi, p)
    a.enable_secondary(om, adjust_alignments=1)
    prim, alt, sec, nsec = a.AlignReadSecondary(b, q, offs, stride=8)
    a.close()
    problems = util.compare_results(e_prim, prim, "primary")
    problems += util.compare_secondary(e_sec, e_nsec, sec, nsec, np.zeros(n, bool))
    moved = (e_prim["location"] != p_prim["location"]) | (e_prim["status"] != p_prim["status"])
    print("reads", n, "aligned", int((e_prim["status"] != 0).sum()), "secondary results", int(e_nsec.sum()), "(without -ae:", int(p_nsec.sum()), ")",
          "primaries the adjuster moved / dropped:", int(moved.sum()), "clipped:", int((e_prim["clipping_for_read_adjustment"] != 0).sum()),
          "score changed:", int((e_prim["score"] != p_prim["score"]).sum()))
    print("problems:", problems[:10] if problems else "none")
    shutil.rmtree(d, ignore_errors=True)
    sys.exit(1 if problems else 0)
