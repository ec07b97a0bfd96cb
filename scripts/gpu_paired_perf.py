"""Throughput check of the paired-end path: GPU vs the reference on all host threads (C3-shaped: 2x150 bp, -d 8)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from snap_amd import synth, abi
from snap_amd.index import GenomeIndex
from snap_amd.aligner import ChimericPairedEndAligner
from oracle import ref
from tests.pairs_util import compare_paired

def main():
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    mb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    maxk = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    d = tempfile.mkdtemp(prefix="pep_", dir="/tmp")
    contigs = synth.make_genome(31, mb * 1_000_000, n_contigs=4, repeat_frac=0.3, max_copies=2000)
    synth.write_fasta(d + "/g.fa", contigs)
    t0 = time.time(); ref.build_index(d + "/g.fa", d + "/idx", seed_len=20, threads=64); print("index %.1fs" % (time.time() - t0), flush=True)
    rix = ref.RefIndex(d + "/idx"); gi = GenomeIndex.load_from_directory(d + "/idx")
    pr = synth.make_pairs(9, contigs, npairs, 150)
    p = abi.default_params(max_k=maxk, max_read_len=160); pp = abi.default_paired_params()
    nthreads = os.cpu_count()
    rp, ra, rc_, secs = rix.align_paired(p, pp, pr["bases"], pr["quals"], pr["offsets"], threads=nthreads, stage=0)
    print("reference: %.3fs on %d threads = %.0f pairs/s" % (secs, nthreads, npairs / secs), rc_, flush=True)
    al = ChimericPairedEndAligner(gi, p, pp)
    al.align(pr["bases"][:300 * 200], pr["quals"][:300 * 200], pr["offsets"][:201])      # warm-up
    al.kernel_time(reset=True); al.counters(reset=True)
    gp, ga = al.align(pr["bases"], pr["quals"], pr["offsets"])
    ms, nl = al.kernel_time()
    print("gpu kernel %.1f ms = %.0f pairs/s  (%.0f reads/s)" % (ms, npairs / ms * 1e3, 2 * npairs / ms * 1e3), al.counters(), flush=True)
    bad = compare_paired(rp, gp, verbose=3, exclude=gp["reserved"] != 0)
    print("RESULT mismatching pairs: %d of %d, flagged %d, as pair %.3f" % (int(bad.sum()), npairs, int((gp["reserved"] != 0).sum()), gp["aligned_as_pair"].mean()))

if __name__ == "__main__":
    main()
