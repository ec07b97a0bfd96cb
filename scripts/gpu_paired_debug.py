"""Find paired-end mismatches on a repeat-dense genome and dump the offending pairs (gpurun_out/pe_bad.npz)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from snap_amd import synth, abi
from snap_amd.index import GenomeIndex
from snap_amd.aligner import ChimericPairedEndAligner
from oracle import ref
from tests.pairs_util import compare_paired

def main():
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    mb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    d = tempfile.mkdtemp(prefix="ped_", dir="/tmp")
    contigs = synth.make_genome(20260925, mb * 1_000_000, n_contigs=max(1, min(24, mb // 8)), repeat_frac=0.30, max_copies=5000,
                                repeat_len=(200, 3000), max_divergence=0.05)
    synth.write_fasta(d + "/g.fa", contigs)
    ref.build_index(d + "/g.fa", d + "/idx", seed_len=20, threads=64)
    rix = ref.RefIndex(d + "/idx"); gi = GenomeIndex.load_from_directory(d + "/idx")
    pr = synth.make_pairs(20260925 + 1000, contigs, npairs, 150)
    p = abi.default_params(max_k=8, max_read_len=160); pp = abi.default_paired_params()
    rp, ra, rc_, secs = rix.align_paired(p, pp, pr["bases"], pr["quals"], pr["offsets"], threads=os.cpu_count(), stage=0)
    print("reference %.2fs" % secs, rc_, flush=True)
    al = ChimericPairedEndAligner(gi, p, pp)
    import ctypes as C
    gp = np.zeros(npairs, dtype=abi.PAIRED_RESULT_DTYPE); ga = np.zeros(npairs, dtype=abi.PAIRED_RESULT_DTYPE)
    b = np.ascontiguousarray(pr["bases"]).reshape(-1); q = np.ascontiguousarray(pr["quals"]).reshape(-1); o = pr["offsets"]
    rc = al.lib.snapgpu_align_paired(al.handle, C.c_uint32(npairs), abi.ptr(b), abi.ptr(q), abi.ptr(o), abi.ptr(gp), abi.ptr(ga))
    print("rc", rc, al.lib.snapgpu_last_error(al.handle)[:100], "kernel", al.kernel_time(), flush=True)
    bad = compare_paired(rp, gp, verbose=4, exclude=gp["reserved"] != 0)
    idx = np.nonzero(bad)[0]
    print("mismatching", idx.size, "of", npairs, "flags set among them", int((gp["flags"][idx] != 0).sum()), "flags total", int((gp["flags"] != 0).sum()))
    os.makedirs("gpurun_out", exist_ok=True)
    keep = idx[:200]
    L = 150
    np.savez_compressed("gpurun_out/pe_bad.npz", idx=keep, bases=pr["bases"].reshape(-1, L)[np.concatenate([[2 * i, 2 * i + 1] for i in keep]).astype(int)] if keep.size else np.zeros((0, L), np.uint8),
                        quals=pr["quals"].reshape(-1, L)[np.concatenate([[2 * i, 2 * i + 1] for i in keep]).astype(int)] if keep.size else np.zeros((0, L), np.uint8),
                        ref=rp[keep], gpu=gp[keep])

if __name__ == "__main__":
    main()
