"""-ae on the emulator (or, with SNAPGPU_TEST_LIB unset and a GPU, on the device) against the live reference: BaseAligner::AlignRead with
secondary results and ignoreAlignmentAdjustmentsForOm = false on reads made to need the adjuster -- indels in the first / last bases,
reads that hang over either end of a contig, both strands.  Usage: python scripts/emu_adjust_check.py [n_reads] [om] [read_len]"""
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
    from tests import util
    from tests.adjust_util import adjust_reads
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
