"""MI355X: the Phase-4 help slots under stress (SNAPGPU_PAIRED_HELP_MIN=2: nearly every pair publishes its candidate list): the golden pairs
must still equal the reference, and the watchdog counters say whether any wait was given up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from tests.pairs_util import compare_paired
from snap_amd import abi
from snap_amd.aligner import ChimericPairedEndAligner
zp = np.load(os.path.join(util.GOLDEN, "paired_reads.npz"))
ix = util.load_golden_index("paired_index.npz")
key = "default_d8_150_s0"
np_ = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
o = zp["o150"][:2 * np_ + 1]
a = ChimericPairedEndAligner(ix, abi.default_params(max_read_len=160, max_k=8), abi.default_paired_params())
for rep in range(2):
    a.counters(reset=True)
    t0 = time.time(); prim, alt = a.align(zp["b150"][:int(o[-1])], zp["q150"][:int(o[-1])], o); dt = time.time() - t0
    c = a.counters()
    ref, _ = util.with_fresh_overrides(zp[key + "_primary"], "pe_" + key + "_primary")
    bad = compare_paired(ref[:np_], prim, verbose=0)
    print("rep", rep, "pairs", prim.size, "bad", int(bad.sum()), "%.2fs" % dt, "watchdog", c["help_watchdog_events"], hex(c["help_watchdog_last"]),
          "lists published", c["help_lists_published"], "answers used", c["help_answers_used"], "of", c["n_ag_locations"], "affine-gap locations", flush=True)
a.close()
