"""Diagnostic for tests/test_gpu_paired.py::test_results_do_not_depend_on_batch_composition: which records / fields differ between the
full golden batch and a permuted half of it, over several repetitions and settings (prints, asserts nothing)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from snap_amd import abi
from tests import util
from snap_amd.aligner import ChimericPairedEndAligner

gi = util.load_golden_index("paired_index.npz")
z = np.load(os.path.join(util.GOLDEN, "paired_reads.npz"))
o = z["o150"].astype(np.int64)


def run(env, reps=3):
    for k, v in env.items():
        os.environ[k] = v
    a = ChimericPairedEndAligner(gi, abi.default_params(max_k=8, max_read_len=160), abi.default_paired_params())
    for rep in range(reps):
        a.counters(reset=True)
        prim, _ = a.align(z["b150"], z["q150"], z["o150"])
        c = a.counters()
        n = prim.size
        order = np.random.default_rng(3 + rep).permutation(n)[: n // 2]
        bb = np.concatenate([z["b150"][o[2 * i]:o[2 * i + 2]] for i in order])
        qq = np.concatenate([z["q150"][o[2 * i]:o[2 * i + 2]] for i in order])
        lens = np.concatenate([[o[2 * i + 1] - o[2 * i], o[2 * i + 2] - o[2 * i + 1]] for i in order])
        oo = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        sub, _ = a.align(bb, qq, oo)
        full = prim[order]
        bad = [(int(order[i]), [f for f in sub.dtype.names if not np.array_equal(sub[i][f], full[i][f])]) for i in range(sub.size) if sub[i].tobytes() != full[i].tobytes()]
        print(env, "rep", rep, "published", c.get("help_lists_published"), "used", c.get("help_answers_used"),
              "flags&4 full", int(((prim["flags"] & 4) != 0).sum()), "sub", int(((sub["flags"] & 4) != 0).sum()),
              "reserved>0 full", np.nonzero(prim["reserved"])[0].tolist(), "diff", bad, flush=True)
        for i, fields in bad[:4]:
            j = int(np.nonzero(order == i)[0][0])
            for f in fields:
                print("   pair", i, f, "full", prim[i][f], "sub", sub[j][f])
    a.close()
    for k in env:
        del os.environ[k]


run({})
run({"SNAPGPU_PAIRED_HELP_MIN": "0"})
run({"SNAPGPU_NO_EXACT_REPLAY": "1"})
