"""Effect of the experimental heavy-first ordering on a repeat-rich workload (bench.py's genome recipe at 24 Mb, 60 000 pairs)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from snap_amd import abi, synth
from snap_amd.index import GenomeIndex
from snap_amd.aligner import ChimericPairedEndAligner
from oracle import ref
t0 = time.time()
d = tempfile.mkdtemp()
g = synth.make_genome(20260925, 24_000_000, n_contigs=3, repeat_frac=0.30, max_copies=5000, repeat_len=(200, 3000), max_divergence=0.05)
synth.write_fasta(d + "/ref.fa", g)
ref.build_index(d + "/ref.fa", d + "/idx", 20, threads=os.cpu_count() or 8)
gi = GenomeIndex.load_from_directory(d + "/idx")
pr = synth.make_pairs(7, g, 60_000, 150)
print("setup %.1fs" % (time.time() - t0), flush=True)
res = {}
for mode in ("0", "1", "0", "1"):
    os.environ["SNAPGPU_PAIRED_HEAVY_FIRST"] = mode
    a = ChimericPairedEndAligner(gi, abi.default_params(max_k=8, max_read_len=160), abi.default_paired_params())
    a.kernel_time(reset=True)
    p, _ = a.align(pr["bases"], pr["quals"], pr["offsets"])
    ms, n = a.kernel_time()
    print("heavy_first", mode, "kernel ms %.1f" % ms, flush=True)
    res.setdefault(mode, p)
    a.close()
print("identical results:", all((res["0"][f] == res["1"][f]).all() for f in res["0"].dtype.names))
