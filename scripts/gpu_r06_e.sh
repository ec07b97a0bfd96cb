#!/bin/bash
# r06e: the affine-gap row's scalar diet (per-position best-local words, masks / constants kept across rows, incremental band scalars, 64 text codes per
# LDS read) + the streamed index loader.  Parity suites, then A/B at 256 Mb: nowin2 = round 5's single-end path, r06d = the commit before, new = this one.
O=gpurun_out/${1:-r06e}; mkdir -p $O
ls -la --time-style=full-iso snap_amd/libsnapgpu.so snap_amd/ab/*.so > $O/libs.txt; python -c "import bench; print('kernel_source_hash', bench.kernel_source_hash())" >> $O/libs.txt 2>&1; cat $O/libs.txt
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_secondary.py tests/test_gpu_repeats.py tests/test_gpu_flags.py tests/test_zy_gpu_index_shapes.py -m gpu -q --timeout 500 > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
C5="--workload paired --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002"
for rep in 1 2; do
for v in r06d new; do
  if [ $v = new ]; then CMD="python bench.py"; else CMD="python scripts/ab_bench.py run $v"; fi
  timeout 600 $CMD --workload single --steps 12 $COMMON > $O/single_${v}_$rep.json 2> $O/single_${v}_$rep.err
  timeout 600 $CMD --workload paired --steps 6 $COMMON > $O/paired_${v}_$rep.json 2> $O/paired_${v}_$rep.err
  timeout 600 $CMD $C5 --steps 4 $COMMON > $O/c5_${v}_$rep.json 2> $O/c5_${v}_$rep.err
done; done
# the single-end leg with the reference beside it (parity of every read of the batches compared) and the timed breakdown
timeout 900 python bench.py --workload single --steps 12 --genome-mb 256 --skip-probe --no-extra-legs --warmup 1 > $O/single_full.json 2> $O/single_full.err
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline()); r=d["roofline"]
        print("%-24s %9.0f reads/s  ms/step %7.1f  parity %s/%s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("parity_units"), d["config"].get("parity_mismatching")), {k: round(v,3) for k,v in (r.get("wave_cycle_breakdown") or {}).items()})
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
