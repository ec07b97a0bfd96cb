"""GPU bring-up check for the paired-end path: device vs the reference (oracle/_ref) on seeded hard pairs."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from snap_amd import synth, abi
from snap_amd.index import GenomeIndex
from snap_amd.aligner import ChimericPairedEndAligner
from oracle import ref
from tests.pairs_util import hard_pairs, compare_paired

def main():
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    maxk = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    d = tempfile.mkdtemp(prefix="pe_", dir="/tmp")
    contigs = synth.make_genome(21, 3_000_000, n_contigs=3, repeat_frac=0.35, max_copies=400, n_run_frac=0.002)
    synth.write_fasta(d + "/g.fa", contigs)
    ref.build_index(d + "/g.fa", d + "/idx", seed_len=20, threads=16)
    rix = ref.RefIndex(d + "/idx"); gi = GenomeIndex.load_from_directory(d + "/idx")
    pr = hard_pairs(5, contigs, npairs, L, insert_mean=400 if L < 200 else 600)
    p = abi.default_params(max_k=maxk, max_read_len=max(160, L + 10)); pp = abi.default_paired_params()
    t0 = time.time()
    rp, ra, rc_, secs = rix.align_paired(p, pp, pr["bases"], pr["quals"], pr["offsets"], threads=32, stage=0)
    print("reference: %.2fs" % secs, rc_, flush=True)
    al = ChimericPairedEndAligner(gi, p, pp)
    t0 = time.time()
    gp, ga = al.align(pr["bases"], pr["quals"], pr["offsets"])
    print("gpu call %.2fs kernel %s" % (time.time() - t0, al.kernel_time()), al.counters(), flush=True)
    bad = compare_paired(rp, gp, verbose=5)
    print("stale flagged:", int((gp["reserved"] != 0).sum()), "overflow:", int((gp["flags"] != 0).sum()))
    nb = int(bad.sum())
    print("RESULT mismatching pairs: %d of %d" % (nb, npairs))
    return 0 if nb == 0 else 1

if __name__ == "__main__":
    sys.exit(main())
