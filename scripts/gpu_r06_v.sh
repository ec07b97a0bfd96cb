#!/bin/bash
# r06v: the runtime's hardware queues (GPU_MAX_HW_QUEUES, default 4; three feeders = six streams share them) against the paired-end grids' shares; the new GPU test
O=gpurun_out/${1:-r06v}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multi_ctx.py -m gpu -x -q > $O/pytest_multi_ctx.txt 2>&1; tail -2 $O/pytest_multi_ctx.txt
C5="--workload paired --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002"
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
p() { tag=$1; shift; timeout 900 env "$@" python bench.py --workload paired --steps 9 --feeders 3 $COMMON > $O/paired_$tag.json 2> $O/paired_$tag.err; }
p base A=1
p hwq8 GPU_MAX_HW_QUEUES=8
p hwq8_share1 GPU_MAX_HW_QUEUES=8 SNAPGPU_PAIRED_GRID_SHARE=1
p hwq8_over1.5 GPU_MAX_HW_QUEUES=8 SNAPGPU_PAIRED_GRID_OVER=1.5
p hwq2 GPU_MAX_HW_QUEUES=2
timeout 900 env SNAPGPU_PAIRED_WAVES_PER_CU=4 python bench.py --workload paired --steps 8 --feeders 4 $COMMON > $O/paired_f4_wpc4.json 2> $O/paired_f4_wpc4.err
timeout 900 env SNAPGPU_PAIRED_WAVES_PER_CU=4 GPU_MAX_HW_QUEUES=8 python bench.py --workload paired --steps 8 --feeders 4 $COMMON > $O/paired_f4_wpc4_hwq8.json 2> $O/paired_f4_wpc4_hwq8.err
timeout 900 env SNAPGPU_PAIRED_WAVES_PER_CU=4 python bench.py --workload paired --steps 12 --feeders 4 $COMMON > $O/paired_f4_wpc4_s12.json 2> $O/paired_f4_wpc4_s12.err
timeout 900 env GPU_MAX_HW_QUEUES=8 python bench.py $C5 --steps 9 --feeders 3 $COMMON > $O/c5_hwq8.json 2> $O/c5_hwq8.err
timeout 900 env GPU_MAX_HW_QUEUES=8 python bench.py --workload single --steps 12 --feeders 3 $COMMON > $O/single_hwq8.json 2> $O/single_hwq8.err
timeout 900 python bench.py --workload single --steps 12 --feeders 3 $COMMON > $O/single_base.json 2> $O/single_base.err
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline())
        print("%-28s %9.0f reads/s  ms/step %7.1f  feeders %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("feeders_per_gpu")))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
