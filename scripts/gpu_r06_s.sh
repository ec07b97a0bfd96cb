#!/bin/bash
# r06s: grids sized to a share of the chip, single end (environment only): SNAPGPU_WAVES_PER_CU x feeders; and more paired points
O=gpurun_out/${1:-r06s}; mkdir -p $O
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
run() { tag=$1; wpc=$2; f=$3; SNAPGPU_WAVES_PER_CU=$wpc timeout 600 python bench.py --workload single --steps 12 --feeders $f $COMMON > $O/single_$tag.json 2> $O/single_$tag.err; }
run wpc24_f3_a 24 3; run wpc8_f3 8 3; run wpc12_f3 12 3; run wpc6_f4 6 4; run wpc8_f4 8 4; run wpc12_f2 12 2; run wpc4_f6 4 6; run wpc16_f3 16 3; run wpc24_f3_b 24 3
prun() { tag=$1; wpc=$2; f=$3; SNAPGPU_PAIRED_WAVES_PER_CU=$wpc timeout 600 python bench.py --workload paired --steps 8 --feeders $f $COMMON > $O/paired_$tag.json 2> $O/paired_$tag.err; }
prun wpc4_f5 4 5; prun wpc4_f6 4 6; prun wpc8_f3 8 3; prun wpc4_f4 4 4
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline())
        print("%-28s %9.0f reads/s  ms/step %7.1f  feeders %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("feeders_per_gpu")))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
