"""Generate tests/golden/ag_sequence_wide.npz with the compiled reference (oracle/_ref): the twin of make_golden_ag_sequence.py for WIDE bands
(w 13 .. 31: segments of 40 .. 64 positions, the window form ag_banded_win2 of ag_win.h) -- 500 calls in order on one newly constructed
AffineGapVectorized<dir>, patterns of 81 .. 191 bases (the 192-position register form on the device, which is the one the exact sequence
kernel has), both directions."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref
from tests import adjust_util

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
out = {}
for tag, seed, n, max_len in (("wide", 20261001, 500, 186),):
    texts, pats, quals, ws, sis, rcs, bands = adjust_util.ag_call_sequence(seed, n, max_len, w_range=(13, 32), min_len=81)
    out[tag + "_texts"] = np.array(texts, dtype=object); out[tag + "_pats"] = np.array(pats, dtype=object); out[tag + "_quals"] = np.array(quals, dtype=object)
    out[tag + "_w"] = np.array(ws, np.int32); out[tag + "_si"] = np.array(sis, np.int32); out[tag + "_rc"] = np.array(rcs, np.uint8); out[tag + "_banded"] = np.array(bands, np.uint8)
    for d in (1, -1):
        tt = [t if d == 1 else t[::-1] for t in texts]
        with ref.fresh_objects():
            seq = ref.affine_gap(d, tt, pats, quals, ws, sis, rcs, bands)
        alone = {k: np.zeros_like(v) for k, v in seq.items()}
        for i in range(n):
            with ref.fresh_objects():
                r = ref.affine_gap(d, [tt[i]], [pats[i]], [quals[i]], [ws[i]], [sis[i]], [rcs[i]], [bands[i]])
            for k in alone: alone[k][i] = r[k][0]
        dep = np.zeros(n, bool)
        for k in seq: dep |= (seq[k] != alone[k]) & (seq["ag_score"] != -1)
        for k, v in seq.items(): out["%s%+d_%s" % (tag, d, k)] = v
        out["%s%+d_depends_on_history" % (tag, d)] = dep
        print(tag, d, "calls", n, "banded", int(np.sum(bands)), "answers that depend on earlier calls:", int(dep.sum()))
np.savez_compressed(OUT + '/ag_sequence_wide.npz', **out)
print('wrote', OUT + '/ag_sequence_wide.npz', os.path.getsize(OUT + '/ag_sequence_wide.npz'), 'bytes')
