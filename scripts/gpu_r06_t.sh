#!/bin/bash
# r06t: the library with paired_grid_share (a paired-end launch asks for its share of the chip: calls in flight, at most 3) -- three / four feeders, the control
# (SNAPGPU_PAIRED_GRID_SHARE=1 = whole-chip grids), configs[4], both genome sizes; the paired GPU tests
O=gpurun_out/${1:-r06t}; mkdir -p $O
C5="--workload paired --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002"
for mb in 256 3100; do
  COMMON="--genome-mb $mb --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
  for f in 3 4; do
    timeout 900 python bench.py --workload paired --steps 8 --feeders $f $COMMON > $O/paired_${mb}_f$f.json 2> $O/paired_${mb}_f$f.err
    timeout 900 python bench.py $C5 --steps 8 --feeders $f $COMMON > $O/c5_${mb}_f$f.json 2> $O/c5_${mb}_f$f.err
  done
  SNAPGPU_PAIRED_GRID_SHARE=1 timeout 900 python bench.py --workload paired --steps 8 --feeders 3 $COMMON > $O/paired_${mb}_f3_share1.json 2> $O/paired_${mb}_f3_share1.err
  SNAPGPU_PAIRED_GRID_SHARE=4 timeout 900 python bench.py --workload paired --steps 8 --feeders 4 $COMMON > $O/paired_${mb}_f4_share4.json 2> $O/paired_${mb}_f4_share4.err
done
timeout 1500 python -m pytest tests -m gpu -x -q -k "paired or repeat" > $O/pytest_paired.txt 2>&1; tail -3 $O/pytest_paired.txt
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline())
        print("%-28s %9.0f reads/s  ms/step %7.1f  feeders %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("feeders_per_gpu")))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
