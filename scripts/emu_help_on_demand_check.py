"""TEST INFRASTRUCTURE (host, wavefront emulator): the on-demand Phase-4 help (paired.h / paired_dev.h) on pairs that HAVE long candidate lists --
a small genome made mostly of high-copy repeats -- against the reference with fresh aligner objects, every pair compared.
  SNAPGPU_PAIRED_HELP_MIN=32 SNAPGPU_EMU_HELP_SPIN=1 SNAPGPU_EMU_CUS=2 python scripts/emu_help_on_demand_check.py [n_pairs]
prints how many lists were published and how many speculative answers the ordered walks used."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import snap_amd.aligner as al
al.LIB_PATH = os.path.join(ROOT, "tests", "emu", "_build", "libsnapgpu_emu.so"); al._lib = None
from snap_amd import abi, synth
from snap_amd.aligner import ChimericPairedEndAligner
from snap_amd.index import GenomeIndex
from oracle import ref
from tests.pairs_util import compare_paired

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 160
d = tempfile.mkdtemp(prefix="helpod")
g = synth.make_genome(11, 3_000_000, n_contigs=2, repeat_frac=0.8, max_copies=2500, repeat_len=(400, 1500), max_divergence=0.012)
synth.write_fasta(d + "/g.fa", g)
ref.build_index(d + "/g.fa", d + "/idx", 20, threads=8)
ix = GenomeIndex.load_from_directory(d + "/idx")
pairs = synth.make_pairs(5, g, n_pairs, 150)
params, pparams = abi.default_params(max_k=8, max_read_len=160), abi.default_paired_params()
ri = ref.RefIndex(d + "/idx")
with ref.fresh_objects():
    exp = ri.align_paired(params, pparams, pairs["bases"], pairs["quals"], pairs["offsets"], threads=8, stage=0)[0]
a = ChimericPairedEndAligner(ix, params, pparams)
a.counters(reset=True)
t0 = time.time(); got, _ = a.align(pairs["bases"], pairs["quals"], pairs["offsets"]); dt = time.time() - t0
c = a.counters()
bad = compare_paired(exp, got, verbose=2)
print("pairs", n_pairs, "differ", int(bad.sum()), "%.1fs" % dt, "ag locations", c["n_ag_locations"], "lists published", c["help_lists_published"],
      "answers used", c["help_answers_used"], "watchdog", c["help_watchdog_events"], "replayed", int(((got["flags"] & 4) != 0).sum()))
a.close()
sys.exit(1 if bad.any() else 0)
