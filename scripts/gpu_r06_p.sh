#!/bin/bash
# r06p: paired-end knobs, environment only (256 Mb): Phase-4 help threshold, feeders, wave slots per CU
O=gpurun_out/${1:-r06p}; mkdir -p $O
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1 --workload paired --steps 6"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $COMMON $EXTRA > $O/paired_$tag.json 2> $O/paired_$tag.err; }
EXTRA=""
run base_1 X=1; run help32 SNAPGPU_PAIRED_HELP_MIN=32; run help128 SNAPGPU_PAIRED_HELP_MIN=128; run help16 SNAPGPU_PAIRED_HELP_MIN=16
run wpc16 SNAPGPU_PAIRED_WAVES_PER_CU=16; run wpc8 SNAPGPU_PAIRED_WAVES_PER_CU=8
EXTRA="--feeders 2"; run f2 X=1
EXTRA="--feeders 4"; run f4 X=1
EXTRA="--feeders 6"; run f6 X=1
EXTRA=""; run base_2 X=1
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline())
        print("%-24s %9.0f reads/s  ms/step %7.1f  feeders %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("feeders_per_gpu")))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
