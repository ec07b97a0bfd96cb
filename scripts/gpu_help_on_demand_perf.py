"""MI355X: does the on-demand Phase-4 help shorten a launch whose pairs are nearly all heavy?  A 3 Mb genome made of high-copy repeats
(the recipe of scripts/emu_help_on_demand_check.py), N pairs, the same batch aligned with help off and with SNAPGPU_PAIRED_HELP_MIN=64
(contexts created one after the other: the switch is read by snapgpu_enable_paired); results must be the same bytes, and a sample is
compared with the reference (fresh objects).  Prints one JSON line."""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from snap_amd import abi, synth
from snap_amd.aligner import ChimericPairedEndAligner
from snap_amd.index import GenomeIndex
from oracle import ref
from tests.pairs_util import compare_paired

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
d = tempfile.mkdtemp(prefix="helpod")
if len(sys.argv) > 2:            # "bench:<Mb>": the bench genome's recipe (30 % planted repeats, copy numbers up to 5 000) at another size
    mb = int(sys.argv[2].split(":")[1])
    g = synth.make_genome(20260925, mb * 1_000_000, n_contigs=max(1, min(24, mb // 8)), repeat_frac=0.30, max_copies=5000, repeat_len=(200, 3000), max_divergence=0.05)
else:
    g = synth.make_genome(11, 3_000_000, n_contigs=2, repeat_frac=0.8, max_copies=2500, repeat_len=(400, 1500), max_divergence=0.012)
out_genome = sys.argv[2] if len(sys.argv) > 2 else "3 Mb, 80 % repeats"
synth.write_fasta(d + "/g.fa", g)
ref.build_index(d + "/g.fa", d + "/idx", 20, threads=os.cpu_count() or 8)
ix = GenomeIndex.load_from_directory(d + "/idx")
pairs = synth.make_pairs(5, g, n_pairs, 150)
params, pparams = abi.default_params(max_k=8, max_read_len=160), abi.default_paired_params()
out = {"pairs": n_pairs, "genome": out_genome}
res = {}
for tag, env in (("help_off", None), ("help_on_demand_64", "64")):
    if env is None:
        os.environ.pop("SNAPGPU_PAIRED_HELP_MIN", None)
    else:
        os.environ["SNAPGPU_PAIRED_HELP_MIN"] = env
    a = ChimericPairedEndAligner(ix, params, pparams)
    a.align(pairs["bases"], pairs["quals"], pairs["offsets"])                 # warm-up
    a.counters(reset=True); a.kernel_time(reset=True)
    t0 = time.time(); got, _ = a.align(pairs["bases"], pairs["quals"], pairs["offsets"]); dt = time.time() - t0
    c = a.counters(); kms, _ = a.kernel_time()
    res[tag] = got
    out[tag] = {"wall_s": dt, "kernel_ms": kms, "ag_locations": c["n_ag_locations"], "lists_published": c["help_lists_published"],
                "answers_used": c["help_answers_used"], "watchdog": c["help_watchdog_events"], "replayed": int(((got["flags"] & 4) != 0).sum())}
    a.close()
out["same_bytes"] = res["help_off"].tobytes() == res["help_on_demand_64"].tobytes()
k = min(n_pairs, 1500 if len(sys.argv) <= 2 else 20000)
ri = ref.RefIndex(d + "/idx")
with ref.fresh_objects():
    exp = ri.align_paired(params, pparams, pairs["bases"][:2 * k], pairs["quals"][:2 * k], pairs["offsets"][:2 * k + 1], threads=os.cpu_count() or 8, stage=0)[0]
out["reference_sample"] = {"pairs": k, "differ": int(compare_paired(exp, res["help_on_demand_64"][:k], verbose=0).sum())}
print(json.dumps(out))
