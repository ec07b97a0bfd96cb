#!/bin/bash
# r06u: paired_grid_share with the recent-peak ramp guard; how much of the chip three feeders' grids should ask for together (SNAPGPU_PAIRED_GRID_OVER: 1, 1.5, 2 shares each)
O=gpurun_out/${1:-r06u}; mkdir -p $O
C5="--workload paired --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002"
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
for over in 1 1.5 2; do
  SNAPGPU_PAIRED_GRID_OVER=$over timeout 900 python bench.py --workload paired --steps 9 --feeders 3 $COMMON > $O/paired_over$over.json 2> $O/paired_over$over.err
  SNAPGPU_PAIRED_GRID_OVER=$over timeout 900 python bench.py $C5 --steps 9 --feeders 3 $COMMON > $O/c5_over$over.json 2> $O/c5_over$over.err
done
SNAPGPU_PAIRED_GRID_OVER=1.33 timeout 900 python bench.py --workload paired --steps 8 --feeders 4 $COMMON > $O/paired_f4_over1.33.json 2> $O/paired_f4_over1.33.err
timeout 900 python bench.py --workload paired --steps 8 --feeders 4 $COMMON > $O/paired_f4_over1.json 2> $O/paired_f4_over1.err
timeout 900 python bench.py --workload paired --steps 6 --feeders 2 $COMMON > $O/paired_f2.json 2> $O/paired_f2.err
timeout 900 python bench.py --workload paired --steps 4 --feeders 1 $COMMON > $O/paired_f1.json 2> $O/paired_f1.err
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline())
        print("%-28s %9.0f reads/s  ms/step %7.1f  feeders %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("feeders_per_gpu")))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
