"""Correctness check of the experimental heavy-first ordering (SNAPGPU_PAIRED_HEAVY_FIRST=1): same results as the default order and as the
reference fixture, on the golden pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from snap_amd import abi
from snap_amd.aligner import ChimericPairedEndAligner
from tests import util
from tests.pairs_util import compare_paired

gi = util.load_golden_index("paired_index.npz")
z = np.load(os.path.join(util.GOLDEN, "paired_reads.npz"))
b, q, o = z["b150"], z["q150"], z["o150"]
out = {}
for mode in ("0", "1"):
    os.environ["SNAPGPU_PAIRED_HEAVY_FIRST"] = mode
    a = ChimericPairedEndAligner(gi, abi.default_params(max_k=8, max_read_len=160), abi.default_paired_params())
    t = time.time()
    out[mode], _ = a.align(b, q, o)
    print("mode", mode, "%.3fs" % (time.time() - t), "reads counted", a.counters()["n_reads"])
    a.close()
same = all((out["0"][f] == out["1"][f]).all() for f in out["0"].dtype.names)
bad = compare_paired(z["default_d8_150_s0_primary"], out["1"], verbose=2, exclude=z["default_d8_150_s0_unstable"] | (out["1"]["reserved"] != 0))
print("heavy-first == default order:", same, "| vs reference fixture: mismatching", int(bad.sum()))
