"""MI355X: first timing of the SAM side, for `rocprofv3 --kernel-trace --stats -- python scripts/gpu_sam_perf.py`.
Aligns N reads (150 bp, 1.5 % substitutions, 0.2 % indels each way) of a seeded genome, then times snapgpu_sam_fields_single on
them (host pointers: copy-inclusive wall time; the kernels k_sam_fields / k_cigar_* show up in the rocprof kernel table) and, for the
same reads, the two cigar batch primitives.  Prints one JSON line.  Algorithmic bytes per read: 3 * len + MAX_K in, 4 * n_ops + 32 out."""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import abi, synth
from snap_amd.aligner import BaseAligner
if os.environ.get("SNAPGPU_AB_LIB"):            # an A/B build of the library (scripts/ab_bench.py build <name>): measurement only
    import snap_amd.aligner as _al
    _al.LIB_PATH = os.path.abspath(os.environ["SNAPGPU_AB_LIB"]); _al._lib = None
from snap_amd.index import GenomeIndex
from oracle import ref

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
d = tempfile.mkdtemp(prefix="samperf")
g = synth.make_genome(77, 8_000_000, n_contigs=3, repeat_frac=0.1)
synth.write_fasta(d + "/g.fa", g)
ref.build_index(d + "/g.fa", d + "/idx", 20, threads=max(1, os.cpu_count() or 1))
ix = GenomeIndex.load_from_directory(d + "/idx")
rd = synth.make_reads(3, g, n, 150, sub=0.015, ins=0.002, dele=0.002)
a = BaseAligner(ix, abi.default_params(max_k=14, max_read_len=160))
t0 = time.time(); prim, _ = a.AlignRead(rd["bases"], rd["quals"], rd["offsets"]); t_align = time.time() - t0
fc = np.zeros(n, np.int32); dl = np.full(n, 150, np.int32)
out = {"reads": n, "align_s_incl_copies": t_align}
for use_m in (True, False):
    a.samFields(rd["bases"], rd["quals"], rd["offsets"], fc, dl, prim, use_m)          # warm-up
    a.kernel_time(reset=True)
    t0 = time.time(); r = a.samFields(rd["bases"], rd["quals"], rd["offsets"], fc, dl, prim, use_m); dt = time.time() - t0
    kms, kn = a.kernel_time(reset=True)
    out["sam_fields_%s" % ("M" if use_m else "eqx")] = {"s_incl_copies": dt, "reads_per_s": n / dt, "kernel_ms": kms, "kernel_reads_per_s": n / (kms / 1e3) if kms else None,
                                                          "algorithmic_GBps": (n * (3 * 150 + 127 + 4 * float(r["n_ops"][r["n_ops"] > 0].mean()) + 32)) / (kms / 1e3) / 1e9 if kms else None, "mapped": int((r["flag"] & 4 == 0).sum()),
                                                          "mean_ops": float(r["n_ops"][r["n_ops"] > 0].mean())}
a.close()
print(json.dumps(out))
