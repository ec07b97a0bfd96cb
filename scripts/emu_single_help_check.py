"""TEST INFRASTRUCTURE (host, wavefront emulator): the help for heavy single-end reads (snap_amd/csrc/se_help.h) on reads that HAVE long forced
walks -- a small genome made mostly of diverged high-copy repeats -- against the reference with fresh aligner objects, every read compared.
  SNAPGPU_SINGLE_HELP_EAGER=1 python scripts/emu_single_help_check.py [n_reads]                       # lists published, the owner alone
  SNAPGPU_EMU_HELP_SPIN=1 SNAPGPU_EMU_CUS=2 python scripts/emu_single_help_check.py [n_reads]         # idle waves evaluate candidates
prints how many lists were published and how many stored evaluations the ordered walks used."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import snap_amd.aligner as al
al.LIB_PATH = os.environ.get("SNAPGPU_TEST_LIB", os.path.join(ROOT, "tests", "emu", "_build", "libsnapgpu_emu.so")); al._lib = None
from snap_amd import abi, synth
from snap_amd.aligner import BaseAligner
from snap_amd.index import GenomeIndex
from oracle import ref
from tests import util

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 300
d = tempfile.mkdtemp(prefix="sehelp")
g = synth.make_genome(13, 2_000_000, n_contigs=2, repeat_frac=0.85, max_copies=280, repeat_len=(400, 1500), max_divergence=0.03)
synth.write_fasta(d + "/g.fa", g)
ref.build_index(d + "/g.fa", d + "/idx", 20, threads=8)
ix = GenomeIndex.load_from_directory(d + "/idx")
reads = synth.make_reads(7, g, n_reads, 150)
params = abi.default_params(max_k=8, max_read_len=160)
ri = ref.RefIndex(d + "/idx")
with ref.fresh_objects():
    exp, _, rc, _ = ri.align_single(params, reads["bases"], reads["quals"], reads["offsets"], threads=8)
a = BaseAligner(ix, params)
a.counters(reset=True)
t0 = time.time(); got, _ = a.AlignRead(reads["bases"], reads["quals"], reads["offsets"]); dt = time.time() - t0
c = a.counters()
problems = util.compare_results(exp, got)
print("reads", n_reads, "problems", problems, "%.1fs" % dt, "lv", c["n_lv_locations"], "ag", c["n_ag_locations"], "ref lv/ag", rc,
      "lists published", c.get("help_lists_published"), "answers used", c.get("help_answers_used"), "watchdog", c.get("help_watchdog_events"),
      "replayed", int(((got["reserved"] & 0x80000000) != 0).sum()))
a.close()
sys.exit(1 if problems else 0)
