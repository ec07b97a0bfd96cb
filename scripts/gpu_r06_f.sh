#!/bin/bash
# r06f: the candidate table's directory in a vector register (REGDIR), the wide-band window form's scalar diet, the streamed index loader's sizing,
# snapgpu-sam -passes.  Parity suites; A/B at 256 Mb against the commit before (r06e); then the driver's own command at GRCh38 scale.
O=gpurun_out/${1:-r06f}; mkdir -p $O
ls -la --time-style=full-iso snap_amd/libsnapgpu.so snap_amd/ab/*.so snap_amd/snapgpu-sam > $O/libs.txt; python -c "import bench; print('kernel_source_hash', bench.kernel_source_hash())" >> $O/libs.txt 2>&1; cat $O/libs.txt
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_secondary.py tests/test_gpu_repeats.py tests/test_gpu_flags.py tests/test_zy_gpu_index_shapes.py tests/test_zz_gpu_native_sam.py -m gpu -q --timeout 500 > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
C5="--workload paired --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002"
for rep in 1 2; do
for v in r06e new; do
  if [ $v = new ]; then CMD="python bench.py"; else CMD="python scripts/ab_bench.py run $v"; fi
  timeout 600 $CMD --workload single --steps 12 $COMMON > $O/single_${v}_$rep.json 2> $O/single_${v}_$rep.err
  timeout 600 $CMD --workload paired --steps 6 $COMMON > $O/paired_${v}_$rep.json 2> $O/paired_${v}_$rep.err
  timeout 600 $CMD $C5 --steps 4 $COMMON > $O/c5_${v}_$rep.json 2> $O/c5_${v}_$rep.err
done; done
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err; echo "driver cmd rc=$?" | tee $O/driver_cmd.rc
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline()); r=d["roofline"]; c=d["config"]
        print("%-24s %9.0f reads/s  ms/step %7.1f  parity %s/%s" % (os.path.basename(f), d["value"], d["ms_per_step"], c.get("parity_units"), c.get("parity_mismatching")), {k: c[k] for k in c if k.startswith(("paired_","c5_","e2e_"))})
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
