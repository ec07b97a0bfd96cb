"""AlignmentAdjuster::AdjustAlignment, item by item: snapgpu_adjust_alignments (emulator, or the GPU with SNAPGPU_TEST_LIB=gpu) against
snapref_adjust_alignments on results whose location is deliberately off by -6 .. +6 (leading deletions / insertions, repeated moves),
on reads at both ends of their contigs (overhang, moves that leave the contig -> NotFound), both strands.  Usage: [n_items]"""
import os, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


if __name__ == "__main__":
    import snap_amd.aligner as al
    if os.environ.get("SNAPGPU_TEST_LIB", "emu") != "gpu":
        al.LIB_PATH = os.environ.get("SNAPGPU_TEST_LIB", os.path.join(ROOT, "tests", "emu", "_build", "libsnapgpu_emu.so")); al._lib = None
    from snap_amd import synth, abi
    from snap_amd.index import GenomeIndex
    from snap_amd.aligner import BaseAligner
    from oracle import ref
    from tests.adjust_util import adjust_cases as make_cases
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    L = 100
    d = tempfile.mkdtemp(prefix="adjunit")
    contigs = synth.make_genome(78, 300_000, n_contigs=30, repeat_frac=0.1, max_copies=10, repeat_len=(150, 400), max_divergence=0.03)
    synth.write_fasta(d + "/g.fa", contigs)
    ref.build_index(d + "/g.fa", d + "/idx", seed_len=20, threads=8)
    gi = GenomeIndex.load_from_directory(d + "/idx")
    cstart = [int(c.begin) for c in gi.contigs]
    b, res = make_cases(int(sys.argv[2]) if len(sys.argv) > 2 else 9, contigs, cstart, n, L)
    off = np.arange(n, dtype=np.uint64) * L; length = np.full(n, L, np.int32)
    exp = ref.RefIndex(d + "/idx").adjust_alignments(b, off, length, res)
    a = BaseAligner(gi, abi.default_params(max_k=8, max_read_len=L + 10))
    got = a.AdjustAlignments(b, off, length, res)
    a.close()
    bad = [(i, f, int(exp[f][i]), int(got[f][i])) for f in ("status", "location", "score", "clipping_for_read_adjustment") for i in np.nonzero(exp[f] != got[f])[0][:5]]
    print("items", n, "moved", int((exp["location"] != res["location"]).sum()), "clipped", int((exp["clipping_for_read_adjustment"] != 0).sum()),
          "dropped", int(((exp["status"] == 0) & (res["status"] != 0)).sum()), "mismatches:", bad if bad else "none")
    shutil.rmtree(d, ignore_errors=True)
    sys.exit(1 if bad else 0)
