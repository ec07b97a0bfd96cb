#!/usr/bin/env python
"""TEST INFRASTRUCTURE / analysis: which affine-gap FORM the paired-end kernel's calls take (ag_win.h: ag_dispatch_inl) on a bench-like
workload, counted by the wavefront emulator's build (-DSNAPGPU_AG_WIN_STATS; nothing of it exists in a device build).

    python scripts/emu_paired_stats.py [n_pairs] [--c5]          # --c5: 2 x 250 bp, -d 20, insert N(600, 80^2) (BASELINE configs[4]); default 2 x 150 bp, -d 8... (configs[2])
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


def main():
    c5 = "--c5" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if args else 300
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
    from snap_amd import abi, synth
    from snap_amd.aligner import ChimericPairedEndAligner
    from tests import util
    ix = util.load_golden_index()
    pad = (ix.genome_padded.size - ix.n_bases) // 2
    ends = [c.begin for c in ix.contigs[1:]] + [ix.n_bases]
    contigs = [(c.name, ix.genome_padded[pad + c.begin: pad + e - ix.chromosome_padding]) for c, e in zip(ix.contigs, ends)]
    if c5:
        L, prm = 250, abi.default_params(max_k=20, max_read_len=256)
        pr = synth.make_pairs(20260925, contigs, n, L, insert_mean=600, insert_sd=80, long_indel_frac=0.002)
    else:
        L, prm = 150, abi.default_params(max_k=8, max_read_len=160)
        pr = synth.make_pairs(20260925, contigs, n, L)
    a = ChimericPairedEndAligner(ix, prm, abi.default_paired_params())
    a.align(pr["bases"].reshape(-1), pr["quals"].reshape(-1), pr["offsets"])
    print("counters:", a.counters())
    a.close()
    h = C.CDLL(lib_path)
    fs = (C.c_ulonglong * 64).in_dll(h, "g_agform_stats")
    names = {1: "banded, window (2 segments in one register)", 2: "banded, wide window (win2)", 3: "banded, chunked register form",
             4: "unbanded <= 64 positions (window FULL)", 6: "unbanded > 64 positions (chunked register form)"}
    print("%d pairs of 2 x %d" % (n, L))
    for f, nm in names.items():
        if fs[f]:
            print("  %-52s calls %7d (%.2f / pair)  rows %9d (%.0f / call)  positions / call %.0f" % (nm, fs[f], fs[f] / n, fs[8 + f], fs[8 + f] / fs[f], fs[16 + f] / fs[f]))
    print("  unbanded > 64 by 64-position chunks: " + "  ".join("%d chunks: %d calls, %d rows" % (c, fs[24 + c], fs[32 + c]) for c in range(8) if fs[24 + c]))
    aw = (C.c_ulonglong * 64).in_dll(h, "g_agwin_stats")
    if aw[5]:
        print("  wide window (win2): %d rows; lazy-F rounds per row: first segment %.2f, second segment %.2f; second segments answered in closed form: %d (%.0f %% of rows)" % (aw[5], aw[6] / aw[5], aw[8] / aw[5], aw[7], 100.0 * aw[7] / aw[5]))
    print("  calls by (w >= 32, positions > 192, > 256): " + " ".join("%s:%d" % (("w<32" if not (k & 1) else "w>=32") + ("/<=192" if k < 2 else ("/<=256" if k < 4 else "/>256")), fs[56 + k]) for k in range(6)))
    print("  calls by band half-width w (15 = 15 and above): " + " ".join("%d:%d" % (w, fs[40 + w]) for w in range(16) if fs[40 + w]))


if __name__ == "__main__":
    main()
