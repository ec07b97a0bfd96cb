#!/bin/bash
# r06j: ag_banded_win2's closed form for the second segment (A/B on c5 against r06i); single-end heavy-first with three feeders (env); the tool's pass timeline
O=gpurun_out/${1:-r06j}; mkdir -p $O
ls -la --time-style=full-iso snap_amd/libsnapgpu.so snap_amd/ab/*.so snap_amd/snapgpu-sam > $O/libs.txt; python -c "import bench; print('kernel_source_hash', bench.kernel_source_hash())" >> $O/libs.txt 2>&1; cat $O/libs.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py -m gpu -q --timeout 600 > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
C5="--workload paired --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002"
for rep in 1 2 3; do
  timeout 600 python scripts/ab_bench.py run r06i $C5 --steps 4 $COMMON > $O/c5_r06i_$rep.json 2> $O/c5_r06i_$rep.err
  timeout 600 python bench.py $C5 --steps 4 $COMMON > $O/c5_new_$rep.json 2> $O/c5_new_$rep.err
done
for rep in 1 2; do
  timeout 600 python bench.py --workload single --steps 12 $COMMON > $O/single_default_$rep.json 2> $O/single_default_$rep.err
  SNAPGPU_SINGLE_HEAVY_FIRST=1 timeout 600 python bench.py --workload single --steps 12 $COMMON > $O/single_heavyfirst_$rep.json 2> $O/single_heavyfirst_$rep.err
done
timeout 900 python bench.py $C5 --steps 4 --genome-mb 256 --skip-probe --no-extra-legs --warmup 1 > $O/c5_full.json 2> $O/c5_full.err
timeout 900 python bench.py --genome-mb 256 --workload single --steps 4 --warmup 1 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-c5-leg --paired-leg-steps 1 > $O/e2e_256.json 2> $O/e2e_256.err
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline()); r=d["roofline"]; c=d["config"]
        print("%-28s %9.0f reads/s  ms/step %7.1f  parity %s/%s" % (os.path.basename(f), d["value"], d["ms_per_step"], c.get("parity_units"), c.get("parity_mismatching")), {k: c[k] for k in c if k.startswith(("e2e_"))})
        if "e2e" in d: print("   e2e tail:", *d["e2e"].get("tool_tail", []), sep="\n      ")
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
