"""Generate tests/golden/ag_sequence.npz with the compiled reference (oracle/_ref): affine-gap problems as CALLS IN ORDER on one newly
constructed AffineGapVectorized<dir> in zero-filled memory (ref_driver.cpp: snapref_affine_gap under snapref_set_fresh_objects(1)) -- 1 200
calls with patterns up to 150 (the 192-position register form on the device) and 500 with patterns up to 420 (the LDS form), both
directions.  Calls whose banded traceback leaves the band read what EARLIER calls of the sequence left in the array: the fixture pins the
exact (image-keeping) forms of the kernels, which the replay passes run."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref
from tests import adjust_util

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
out = {}
for tag, seed, n, max_len in (("short", 20260929, 1200, 150), ("long", 20260930, 500, 420)):
    texts, pats, quals, ws, sis, rcs, bands = adjust_util.ag_call_sequence(seed, n, max_len)
    out[tag + "_texts"] = np.array(texts, dtype=object); out[tag + "_pats"] = np.array(pats, dtype=object); out[tag + "_quals"] = np.array(quals, dtype=object)
    out[tag + "_w"] = np.array(ws, np.int32); out[tag + "_si"] = np.array(sis, np.int32); out[tag + "_rc"] = np.array(rcs, np.uint8); out[tag + "_banded"] = np.array(bands, np.uint8)
    for d in (1, -1):
        tt = [t if d == 1 else t[::-1] for t in texts]
        with ref.fresh_objects():
            seq = ref.affine_gap(d, tt, pats, quals, ws, sis, rcs, bands)
        # every call alone on an object of its own: which answers depend on the sequence at all
        alone = {k: np.zeros_like(v) for k, v in seq.items()}
        for i in range(n):
            with ref.fresh_objects():
                r = ref.affine_gap(d, [tt[i]], [pats[i]], [quals[i]], [ws[i]], [sis[i]], [rcs[i]], [bands[i]])
            for k in alone: alone[k][i] = r[k][0]
        dep = np.zeros(n, bool)
        for k in seq: dep |= (seq[k] != alone[k]) & (seq["ag_score"] != -1)
        for k, v in seq.items(): out["%s%+d_%s" % (tag, d, k)] = v
        out["%s%+d_depends_on_history" % (tag, d)] = dep
        print(tag, d, "calls", n, "answers that depend on earlier calls:", int(dep.sum()))
np.savez_compressed(OUT + '/ag_sequence.npz', **out)
print('wrote', OUT + '/ag_sequence.npz', os.path.getsize(OUT + '/ag_sequence.npz'), 'bytes')
