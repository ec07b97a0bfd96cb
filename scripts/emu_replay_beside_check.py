"""Emulator check of the exact replay beside the paired main pass: hard pairs on a repeat-rich genome, the library run with
SNAPGPU_PAIRED_REPLAY_BESIDE=1 and =0 and the reference with fresh aligner objects -- all three equal, byte for byte between the two
library runs (flags included).  Usage: python scripts/emu_replay_beside_check.py [n_pairs] [max_k] [read_len]"""
import os, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snap_amd.aligner as al
al.LIB_PATH = os.environ.get("SNAPGPU_TEST_LIB", os.path.join(ROOT, "tests", "emu", "_build", "libsnapgpu_emu.so")); al._lib = None
import numpy as np
from snap_amd import synth, abi
from snap_amd.index import GenomeIndex
from snap_amd.aligner import ChimericPairedEndAligner
from oracle import ref
from tests.pairs_util import hard_pairs, compare_paired

n = int(sys.argv[1]) if len(sys.argv) > 1 else 800
maxk = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = int(sys.argv[3]) if len(sys.argv) > 3 else 150
d = tempfile.mkdtemp(prefix="beside")
if os.environ.get("BESIDE_CHECK_GENOME") == "repeats":     # long candidate lists: what makes a traceback's stale step land in a LATER call of its object
    contigs = synth.make_genome(11, 1_200_000, n_contigs=2, repeat_frac=0.8, max_copies=1200, repeat_len=(400, 1200), max_divergence=0.012)
else:
    contigs = synth.make_genome(21 + L, 3_000_000, n_contigs=3, repeat_frac=0.35, max_copies=400, n_run_frac=0.002)
synth.write_fasta(d + "/g.fa", contigs)
ref.build_index(d + "/g.fa", d + "/idx", seed_len=20, threads=16)
gi = GenomeIndex.load_from_directory(d + "/idx")
pr = hard_pairs(5 + maxk, contigs, n, L, insert_mean=400, insert_max=1000)
p = abi.default_params(max_k=maxk, max_read_len=L + 10); pp = abi.default_paired_params(max_spacing=1000)
with ref.fresh_objects():
    rp = ref.RefIndex(d + "/idx").align_paired(p, pp, pr["bases"], pr["quals"], pr["offsets"], threads=16, stage=0)[0]
out = {}
for mode in ("1", "0"):
    os.environ["SNAPGPU_PAIRED_REPLAY_BESIDE"] = mode
    a = ChimericPairedEndAligner(gi, p, pp)
    got, _ = a.align(pr["bases"], pr["quals"], pr["offsets"]); a.close()
    bad = compare_paired(rp, got, verbose=3)
    print("beside =", mode, "flags&4:", int(((got["flags"] & 4) != 0).sum()), "other flag bits:", hex(int(np.bitwise_or.reduce(got["flags"])) & ~7),
          "mismatching pairs vs reference:", int(bad.sum()))
    out[mode] = got
print("identical bytes:", out["1"].tobytes() == out["0"].tobytes())
shutil.rmtree(d, ignore_errors=True)
