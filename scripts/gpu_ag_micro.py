"""Micro-benchmark of the affine-gap batch kernel (for rocprofv3 --kernel-trace)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import abi
from snap_amd.aligner import BaseAligner
from tests.util import load_golden_index
ix = load_golden_index()
al = BaseAligner(ix, abi.default_params(max_k=8, max_read_len=160))
rng = np.random.default_rng(1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
texts, pats, quals = [], [], []
for i in range(N):
    g = bytes(rng.choice(list(b'ACGT'), size=257).astype(np.uint8))
    p = bytearray(g[:130])
    for _ in range(4): p[int(rng.integers(0, 130))] = b'ACGT'[rng.integers(0, 4)]
    texts.append(g); pats.append(bytes(p)); quals.append(b'I' * 130)
for banded in (0, 1):
    for rep in range(2):
        t0 = time.time()
        r = al.computeScoreAffine(1, texts, pats, quals, [9] * N, [150] * N, [0] * N, [banded] * N)
        print('banded', banded, 'rep', rep, 'wall', time.time() - t0, 'found', int((r['ag_score'] > 0).sum()))
# LV for comparison
t0 = time.time(); r = al.computeEditDistance(1, texts, pats, quals, [9] * N); print('lv wall', time.time() - t0)
