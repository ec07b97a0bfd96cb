#!/bin/bash
# r06l: Landau-Vishkin inlined in the single-end kernels (-DSNAPGPU_LV_INLINE; same spills since the kernel has one score() site): A/B + parity of the variant
O=gpurun_out/${1:-r06l}; mkdir -p $O
ls -la --time-style=full-iso snap_amd/libsnapgpu.so snap_amd/ab/*.so > $O/libs.txt; cat $O/libs.txt
SNAPGPU_TEST_LIB=snap_amd/ab/libsnapgpu_lvinl.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_secondary.py tests/test_gpu_flags.py -m gpu -q --timeout 600 > $O/pytest_lvinl.txt 2>&1; tail -3 $O/pytest_lvinl.txt
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
for rep in 1 2 3; do
  timeout 600 python bench.py --workload single --steps 12 $COMMON > $O/single_new_$rep.json 2> $O/single_new_$rep.err
  timeout 600 python scripts/ab_bench.py run lvinl --workload single --steps 12 $COMMON > $O/single_lvinl_$rep.json 2> $O/single_lvinl_$rep.err
done
timeout 600 python scripts/ab_bench.py run lvinl --workload single --steps 6 --read-len 250 --max-k 20 $COMMON > $O/single250_lvinl.json 2> $O/single250_lvinl.err
timeout 600 python bench.py --workload single --steps 6 --read-len 250 --max-k 20 $COMMON > $O/single250_new.json 2> $O/single250_new.err
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline())
        print("%-28s %9.0f reads/s  ms/step %7.1f" % (os.path.basename(f), d["value"], d["ms_per_step"]))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
