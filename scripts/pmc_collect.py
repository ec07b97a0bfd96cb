#!/usr/bin/env python
"""PMC passes of the bench's dominant kernel in the configuration bench.py TIMES (three feeders unless told otherwise), one rocprofv3
--pmc pass per counter group (counters only -- no trace domains beside --pmc; MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and
WRITE_SIZE cannot share a pass), summed over the dispatches of the kernel and divided by their number.

    python scripts/pmc_collect.py <out-dir> [--genome-mb 3100] [--workload single|paired] [--feeders 0] [--steps 3] [-- extra bench args]

Writes <out-dir>/pmc_entry.json: one entry in the format bench.py's attach_pmc() replays (profiles/pmc_latest.json: {"entries": [...]}),
carrying `kernel_source_hash` = the hash of the device sources the passes ran on, so that bench.py only ever replays counters of the
build it is timing.  Raw per-pass CSVs stay in <out-dir>/pmc_<i>/.  Run on the GPU box (gpurun), from the repository root.
"""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GROUPS = ["FETCH_SIZE", "WRITE_SIZE",
          "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES",
          "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_ANY"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--genome-mb", type=int, default=3100)
    ap.add_argument("--workload", default="single")
    ap.add_argument("--feeders", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--groups", type=int, default=len(GROUPS), help="only the first N counter groups")
    ap.add_argument("--timeout", type=int, default=240)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per launch (bench.py --reads)")
    ap.add_argument("--tag", default="", help="the entry's workload label when it is not --workload (bench.py's c5 leg: --tag c5 --read-len 250 --max-k 20 ...)")
    a, a.rest = ap.parse_known_args()               # anything else goes to bench.py (--read-len 250 --max-k 20 ...)
    a.rest = [x for x in a.rest if x != "--"]
    os.makedirs(a.out, exist_ok=True)
    from bench import kernel_source_hash
    env = dict(os.environ, TMPDIR="/tmp")
    bench = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", "0", "--batches", "2", "--skip-cpu", "--skip-refwalk",
             "--skip-breakdown", "--no-extra-legs", "--genome-mb", str(a.genome_mb), "--workload", a.workload, "--reads", str(a.reads)] + (["--feeders", str(a.feeders)] if a.feeders else []) + a.rest
    tot = collections.defaultdict(float)
    rows = collections.defaultdict(int)
    meta = {}
    for i, grp in enumerate(GROUPS[:a.groups], 1):
        d = os.path.join(a.out, "pmc_%d" % i)
        cmd = ["rocprofv3", "--pmc"] + grp.split() + ["--output-format", "csv", "-d", os.path.abspath(d), "-o", "bench", "--"] + bench + ([] if i == 1 else ["--skip-probe"])
        with open(os.path.join(a.out, "pmc_%d.json" % i), "w") as fo, open(os.path.join(a.out, "pmc_%d.err" % i), "w") as fe:
            try:
                rc = subprocess.run(cmd, stdout=fo, stderr=fe, stdin=subprocess.DEVNULL, env=env, cwd="/tmp", timeout=a.timeout).returncode
            except subprocess.TimeoutExpired:
                rc = -9
        print("pass %d (%s): rc=%s" % (i, grp.split()[0], rc), flush=True)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                kn = r["Kernel_Name"]
                k = "align" if ("k_align_single" in kn or "k_align_paired" in kn) else "probe" if "k_lookup_seeds" in kn else None
                if k is None:
                    continue
                # the dominant instantiation only: the replay / large-buffer passes share the template name
                key = (k, kn.split("(")[0].strip())
                tot[(key, r["Counter_Name"])] += float(r["Counter_Value"]); rows[(key, r["Counter_Name"])] += 1
                meta[key] = {"scratch_bytes_per_lane": int(r.get("Scratch_Size", 0) or 0), "vgprs": int(r.get("VGPR_Count", 0) or 0),
                             "accum_vgprs": int(r.get("Accum_VGPR_Count", 0) or 0), "sgprs": int(r.get("SGPR_Count", 0) or 0), "lds_block_bytes": int(r.get("LDS_Block_Size", 0) or 0)}
    # dominant align kernel = the instantiation with the most wave cycles (or the most fetched bytes when that pass is missing)
    aligns = {key for (key, c) in tot if key[0] == "align"}

    def weight(key):
        return tot.get((key, "SQ_WAVE_CYCLES"), 0.0) or tot.get((key, "FETCH_SIZE"), 0.0)
    if not aligns:
        raise SystemExit("no dispatch of the align kernel in the counter files under " + a.out)
    dom = max(aligns, key=weight)

    def per_launch(key, c):
        return tot[(key, c)] / rows[(key, c)] if rows.get((key, c)) else None
    n = a.reads
    e = {"workload": a.tag or a.workload, "genome_mb": a.genome_mb, "reads_per_launch": n, "feeders": a.feeders or 3, "kernel": dom[1], "kernel_resources": meta.get(dom),
         "kernel_source_hash": kernel_source_hash(), "dispatches_per_pass": rows.get((dom, "FETCH_SIZE")),
         "source": "%s: scripts/pmc_collect.py, %d separate rocprofv3 --pmc passes (counters only) of `%s`; per dispatch of the dominant instantiation"
                   % (a.out, min(a.groups, len(GROUPS)), " ".join(os.path.basename(x) if x.endswith("bench.py") else x for x in bench[1:])),
         "fetch_size_kb": per_launch(dom, "FETCH_SIZE"), "write_size_kb": per_launch(dom, "WRITE_SIZE"),
         "note": "(FETCH_SIZE + WRITE_SIZE) KB x 1024, each from its own pass.  No gfx950 x2 correction: MI355X_MICROARCH.md calibrates it for wide coalesced "
                 "streaming reads only; this kernel reads narrow / random and writes bytes.",
         "valu_insts_per_launch": per_launch(dom, "SQ_INSTS_VALU"), "salu_insts_per_launch": per_launch(dom, "SQ_INSTS_SALU"),
         "vmem_rd_insts_per_launch": per_launch(dom, "SQ_INSTS_VMEM_RD"), "vmem_wr_insts_per_launch": per_launch(dom, "SQ_INSTS_VMEM_WR"),
         "lds_insts_per_launch": per_launch(dom, "SQ_INSTS_LDS"), "wave_cycles": per_launch(dom, "SQ_WAVE_CYCLES"),
         "wait_inst_any": per_launch(dom, "SQ_WAIT_INST_ANY"), "wait_any": per_launch(dom, "SQ_WAIT_ANY"), "active_inst_any": per_launch(dom, "SQ_ACTIVE_INST_ANY"),
         "active_inst_valu": per_launch(dom, "SQ_ACTIVE_INST_VALU"), "thread_cycles_valu": per_launch(dom, "SQ_THREAD_CYCLES_VALU"),
         # SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = active lanes x cycles per instruction; r03z: 5.08e12 / 1.08e11 = 47 with one count per lane
         "thread_cycles_per_lane_inst": 1.0}
    if e["fetch_size_kb"] is not None and e["write_size_kb"] is not None:
        e["hbm_bytes_per_launch"] = (e["fetch_size_kb"] + e["write_size_kb"]) * 1024.0
    probes = [key for key in {k for (k, c) in tot} if key[0] == "probe"]
    if probes:
        e["probe_fetch_size_kb"] = per_launch(probes[0], "FETCH_SIZE")
    others = {"%s:%s" % (key[1], c): {"sum": v, "dispatch_rows": rows[(key, c)]} for (key, c), v in tot.items() if key != dom}
    json.dump(e, open(os.path.join(a.out, "pmc_entry.json"), "w"), indent=1)
    json.dump(others, open(os.path.join(a.out, "pmc_other_kernels.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(e)[:1500])


if __name__ == "__main__":
    main()
