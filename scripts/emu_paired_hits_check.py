"""Build container (emulator; SNAPGPU_TEST_LIB=gpu on an MI355X): the paired-end set intersection with the hits staged in LDS
(paired_dev.h: hs_stage / hs_get) on pairs whose hit lists are LONG -- a genome built of high-copy repeats, so that every lookup's window is
reloaded many times and the binary-search jumps of getNextHitLessThanOrEqualTo start inside and outside staged windows -- under both
window sizes (num_seeds 8: 16 hits per lookup; seed coverage instead of a seed count: 30 lookups per set, 8 hits each) and a narrow
spacing (many jumps).  Every field of every pair against the compiled reference with fresh aligner objects; work counters too.

    python scripts/emu_paired_hits_check.py [n_pairs=60]
"""
import os, shutil, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.environ.get("SNAPGPU_TEST_LIB", os.path.join(ROOT, "tests", "emu", "_build", "libsnapgpu_emu.so"))
import snap_amd.aligner as al
if lib != "gpu":
    al.LIB_PATH = lib; al._lib = None
from snap_amd import abi, synth
from snap_amd.aligner import ChimericPairedEndAligner
from snap_amd.index import GenomeIndex
from oracle import ref
from tests.pairs_util import compare_paired

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
only = sys.argv[2] if len(sys.argv) > 2 else None
d = tempfile.mkdtemp(prefix="emuhits")
g = synth.make_genome(23, 1_500_000, n_contigs=2, repeat_frac=0.75, max_copies=900, repeat_len=(300, 1500), max_divergence=0.02)
synth.write_fasta(d + "/g.fa", g)
ref.build_index(d + "/g.fa", d + "/idx", 20, threads=8)
gi = GenomeIndex.load_from_directory(d + "/idx")
rix = ref.RefIndex(d + "/idx")
bad_total = 0
for tag, pkw in (("n8", {}), ("coverage", dict(num_seeds=0, seed_coverage=4.0)), ("narrow", dict(min_spacing=50, max_spacing=350, num_seeds=12)),
                 ("indels", dict(num_seeds=16))):
    if only and tag != only:
        continue
    # "indels": deletions of 2 .. 30 bases in most reads, so that seeds either side of one hit 2 .. 30 apart: the seed-hinted indel limits of Phase 2a
    pairs = synth.make_pairs(7 + len(tag), g, n, 150, long_indel_frac=0.8 if tag == "indels" else 0.0, long_indel_max=30)
    params, pparams = abi.default_params(max_k=12 if tag == "indels" else 8, max_read_len=160), abi.default_paired_params(**pkw)
    with ref.fresh_objects():
        exp = rix.align_paired(params, pparams, pairs["bases"], pairs["quals"], pairs["offsets"], threads=4, stage=0)[0]
    a = ChimericPairedEndAligner(gi, params, pparams)
    try:
        a.counters(reset=True)
        got, _ = a.align(pairs["bases"], pairs["quals"], pairs["offsets"])
        c = a.counters()
    finally:
        a.close()
    bad = compare_paired(exp, got, verbose=3)
    print("%s: %d pairs, %d differ; per read: %.0f hits, %.1f LV, %.1f AG" % (tag, n, int(bad.sum()), c["n_hits_consumed"] / (2 * n), c["n_lv_locations"] / (2 * n), c["n_ag_locations"] / (2 * n)))
    bad_total += int(bad.sum())
shutil.rmtree(d, ignore_errors=True)
sys.exit(1 if bad_total else 0)
