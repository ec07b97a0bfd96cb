"""SNAPGPU_SINGLE_RESOLVE=1 with the exact replay switched OFF over every single-end fixture set (4 option sets x 100 / 150 bp x 4 000
reads): the fast form that answers calls leaving their band in place (ag_resolve.h) against the fresh-object reference answers of
tests/golden/tiny_reads.npz.  Emulator by default; SNAPGPU_TEST_LIB=gpu on an MI355X."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import snap_amd.aligner as al
if os.environ.get("SNAPGPU_TEST_LIB", "emu") != "gpu":
    al.LIB_PATH = os.environ.get("SNAPGPU_TEST_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "emu", "_build", "libsnapgpu_emu.so")); al._lib = None
import numpy as np
from snap_amd import abi
from snap_amd.aligner import BaseAligner
from tests import util
gi = util.load_golden_index(); z = np.load(os.path.join(util.GOLDEN, "tiny_reads.npz"))
sets = [k[:-len("_100_primary")] for k in z.files if k.endswith("_100_primary")]
print(sets)
for name in sets:
    kw = {"default_d8": dict(max_k=8), "default_d27": dict(max_k=27), "lvonly_d8": dict(max_k=8, use_affine_gap=0), "emitalt_d8": dict(max_k=8, emit_alt_alignments=1)}.get(name)
    if kw is None: continue
    for tag in ("100", "150"):
        key = "%s_%s_" % (name, tag)
        b, q = z["b" + tag], z["q" + tag]; n, L = b.shape; offs = np.arange(n + 1, dtype=np.uint64) * L
        exp, _ = util.with_fresh_overrides(z[key + "primary"], key + "primary")
        os.environ["SNAPGPU_SINGLE_RESOLVE"] = "1"; os.environ["SNAPGPU_NO_EXACT_REPLAY"] = "1"
        a = BaseAligner(gi, abi.default_params(max_read_len=160, **kw)); prim, alt = a.AlignRead(b, q, offs); a.close()
        bad = util.compare_results(exp, prim)
        print(key, "flagged:", int(((prim["reserved"] >> 30) & 1).sum()), "problems:", bad[:2] if bad else "none", flush=True)
