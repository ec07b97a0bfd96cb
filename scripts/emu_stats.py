#!/usr/bin/env python
"""TEST INFRASTRUCTURE / analysis: what a read costs in events -- Landau-Vishkin calls and levels, affine-gap calls, rows, lazy-F rounds,
traceback steps -- counted by the wavefront emulator's build of the device code (EMU_STAT in snap_amd/csrc/dev_common.h; nothing of it
exists in a device build).  Runs the 1 000 committed 150-bp golden reads (tests/golden/tiny_reads.npz) through k_align_single and prints
the counters per read.  Builds its own copy of the emulator library under /tmp/snapgpu_emu_stats (-DSNAPGPU_AG_WIN_STATS).

    python scripts/emu_stats.py [n_reads] [--bench-like] [--sam]
--bench-like: reads drawn from the fixture genome the way bench.py draws them (synth.make_reads with its defaults: 1 % substitutions, 0.05 %
insertions and deletions), checked against the C restatement (oracle/) instead of the committed reference results.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SNAPGPU_EMU_BDIR"] = "/tmp/snapgpu_emu_stats"
os.environ.setdefault("SNAPGPU_EMU_CUS", "8")

import numpy as np                                     # noqa: E402
import tests.emu.build as eb                           # noqa: E402

NAMES = {0: "LV calls", 1: "LV calls ending in the perfect-match prefix", 2: "LV levels (e >= 1)", 3: "LV calls reaching the planes level", 4: "LV calls above the limit",
         5: "LV calls with an answer at e >= 1", 6: "sum of e over those", 8: "affine-gap window calls", 9: "affine-gap window rows", 10: "window slides",
         11: "rows whose X changed after the first segment's rounds", 15: "second segments in rounds because a later stripe end beats stripe 0's flow", 16: "... because some cell's offer is below T_fp",
         32: "SAM: affine-gap CIGAR items", 33: "SAM: banded calls", 34: "SAM: full (unbanded) calls", 35: "SAM: banded rows", 36: "SAM: full rows",
         37: "SAM: banded first-pass vectors", 38: "SAM: banded lazy-F vector steps", 39: "SAM: full first-pass vectors", 40: "SAM: full lazy-F vector steps", 41: "SAM: traceback gathers", 42: "SAM: banded calls whose row loop k_samf_dp8 had run", 12: "rows with two segments", 13: "traceback gathers (64 cells each)", 14: "rows on which (nk0, nk1) changed"}


def main():
    bench_like = "--bench-like" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if args else 1000
    os.makedirs(eb.BDIR, exist_ok=True)
    stats_src = os.path.join(eb.BDIR, "stats.cpp")
    with open(stats_src, "w") as f:
        f.write("unsigned long long g_emu_stats[64]; unsigned long long g_agwin_stats[64]; unsigned long long g_agform_stats[64];\n")
    eb.FLAGS.append("-DSNAPGPU_AG_WIN_STATS")
    base_units = eb.units
    eb.units = lambda: base_units() + [("stats.o", stats_src, [])]
    lib_path = eb.build(verbose=True)
    import snap_amd.aligner as al
    al._lib, al.LIB_PATH = None, lib_path
    from snap_amd import abi
    from snap_amd.aligner import BaseAligner
    from tests import util
    z = np.load(os.path.join(ROOT, "tests", "golden", "tiny_reads.npz"))
    b, q = z["b150"][:n], z["q150"][:n]
    ix = util.load_golden_index()
    prm = abi.default_params(max_k=8, max_read_len=160)
    if bench_like:
        from snap_amd import synth
        pad = (ix.genome_padded.size - ix.n_bases) // 2
        ends = [c.begin for c in ix.contigs[1:]] + [ix.n_bases]
        contigs = [(c.name, ix.genome_padded[pad + c.begin: pad + e - ix.chromosome_padding]) for c, e in zip(ix.contigs, ends)]
        rd = synth.make_reads(20260925, contigs, n, 150)
        b, q = rd["bases"], rd["quals"]
    a = BaseAligner(ix, prm)
    offs = np.arange(b.shape[0] + 1, dtype=np.uint64) * 150
    prim, _ = a.AlignRead(b, q, offs)
    if bench_like:
        ref, _ = util.oracle_align_reads(ix, prm, b, q, offs)
        bad = util.compare_results(ref, prim)
    else:
        bad = util.compare_results(z["default_d8_150_primary"][:n], prim, exclude=z["default_d8_150_unstable"][:n])
    assert not bad, bad
    if "--sam" in sys.argv:                  # the SAM side over the same reads: SAMFormat::computeCigar's affine-gap variant (cigar_ag.h)
        nrd = len(prim)
        a.samFields(b, q, offs, np.zeros(nrd, np.int32), np.full(nrd, 150, np.int32), prim)
    a.close()
    h = C.CDLL(lib_path)
    st = (C.c_ulonglong * 64).in_dll(h, "g_emu_stats")
    ag = (C.c_ulonglong * 64).in_dll(h, "g_agwin_stats")
    nr = float(len(prim))
    print("%d reads (results identical to the reference's)" % len(prim))
    for i in range(64):
        if st[i]:
            print("  %-60s %10d  %8.2f per read" % (NAMES.get(i, "stat %d" % i), st[i], st[i] / nr))
    print("  lazy-F rounds run, first segment : " + " ".join("%d" % ag[16 + r] for r in range(7)) + "   closed form for the second: %d" % ag[4])
    print("  lazy-F rounds run, second segment: " + " ".join("%d" % ag[24 + r] for r in range(7)))
    print("  ag_compute_reg rows (unbanded AGC 1..3+ | banded): " + " ".join("%d" % ag[32 + i] for i in range(8)) + "   positions x rows: %d | %d" % (ag[40], ag[41]))
    print("  ag_compute_reg lazy-F rounds run, unbanded: " + " ".join("%d" % ag[48 + r] for r in range(8)) + "   banded: " + " ".join("%d" % ag[56 + r] for r in range(8)))


if __name__ == "__main__":
    main()
