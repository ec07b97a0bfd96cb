#!/bin/bash
# r06w: with one stream per context and eight hardware queues, how much should a paired-end launch ask for?  SNAPGPU_PAIRED_GRID_OVER 1 / 1.25 / 1.5 / 2 / whole chip,
# three feeders (four at 256 Mb), both genome sizes, paired and configs[4]
O=gpurun_out/${1:-r06w}; mkdir -p $O
C5="--workload paired --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002"
for mb in 256 3100; do
  COMMON="--genome-mb $mb --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
  for over in 1 1.5 1.25 2 3; do
    SNAPGPU_PAIRED_GRID_OVER=$over timeout 900 python bench.py --workload paired --steps 9 --feeders 3 $COMMON > $O/paired_${mb}_over$over.json 2> $O/paired_${mb}_over$over.err
  done
  for over in 1 1.5; do
    SNAPGPU_PAIRED_GRID_OVER=$over timeout 900 python bench.py $C5 --steps 9 --feeders 3 $COMMON > $O/c5_${mb}_over$over.json 2> $O/c5_${mb}_over$over.err
  done
done
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
timeout 900 python bench.py --workload paired --steps 8 --feeders 4 $COMMON > $O/paired_256_f4.json 2> $O/paired_256_f4.err
SNAPGPU_PAIRED_WAVES_PER_CU=4 timeout 900 python bench.py --workload paired --steps 8 --feeders 4 $COMMON > $O/paired_256_f4_wpc4.json 2> $O/paired_256_f4_wpc4.err
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline())
        print("%-28s %9.0f reads/s  ms/step %7.1f  feeders %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("feeders_per_gpu")))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
