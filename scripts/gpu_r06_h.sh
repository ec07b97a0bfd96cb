#!/bin/bash
# r06h: the tool's reaper thread (written batches freed off the writer's thread); single-end experiments: 7 waves per SIMD, 2 / 4 feeders
O=gpurun_out/${1:-r06h}; mkdir -p $O
ls -la --time-style=full-iso snap_amd/libsnapgpu.so snap_amd/ab/*.so snap_amd/snapgpu-sam > $O/libs.txt; python -c "import bench; print('kernel_source_hash', bench.kernel_source_hash())" >> $O/libs.txt 2>&1; cat $O/libs.txt
timeout 1200 python -m pytest tests/test_zz_gpu_native_sam.py tests/test_zz_gpu_datatest.py -m gpu -q --timeout 600 > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
for rep in 1 2; do
  timeout 600 python bench.py --workload single --steps 12 $COMMON > $O/single_new_$rep.json 2> $O/single_new_$rep.err
  SNAPGPU_WAVES_PER_CU=28 timeout 600 python scripts/ab_bench.py run w7 --workload single --steps 12 $COMMON > $O/single_w7_$rep.json 2> $O/single_w7_$rep.err
  timeout 600 python bench.py --workload single --steps 12 --feeders 2 $COMMON > $O/single_f2_$rep.json 2> $O/single_f2_$rep.err
  timeout 600 python bench.py --workload single --steps 12 --feeders 4 $COMMON > $O/single_f4_$rep.json 2> $O/single_f4_$rep.err
done
timeout 900 python bench.py --genome-mb 256 --workload single --steps 4 --warmup 1 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-c5-leg --paired-leg-steps 1 > $O/e2e_256.json 2> $O/e2e_256.err
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline()); r=d["roofline"]; c=d["config"]
        print("%-24s %9.0f reads/s  ms/step %7.1f  feeders %s" % (os.path.basename(f), d["value"], d["ms_per_step"], c.get("feeders_per_gpu")), {k: c[k] for k in c if k.startswith(("e2e_"))})
        if "e2e" in d: print("   e2e tail:", *d["e2e"].get("tool_tail", []), sep="\n      ")
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
