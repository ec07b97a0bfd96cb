#!/bin/bash
# r06b: the paired-end kernel's HBM pools / LDS state behind typed pointers (global_* / ds_* instead of flat_*): parity, A/B against round 5's
# library (paired, c5, single), the phase timers of the new build
O=gpurun_out/${1:-r06b}; mkdir -p $O
ls -la --time-style=full-iso snap_amd/libsnapgpu.so snap_amd/ab/*.so > $O/libs.txt; python -c "import bench; print('kernel_source_hash', bench.kernel_source_hash())" >> $O/libs.txt 2>&1; cat $O/libs.txt
timeout 1200 python -m pytest tests/test_gpu_paired.py tests/test_gpu_parity.py tests/test_gpu_secondary.py tests/test_gpu_repeats.py -m gpu -q -x --timeout 400 > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
for rep in 1 2; do
for v in r05 new; do
  if [ $v = new ]; then CMD="python bench.py"; else CMD="python scripts/ab_bench.py run $v"; fi
  timeout 600 $CMD --workload paired --steps 6 $COMMON > $O/paired_${v}_$rep.json 2> $O/paired_${v}_$rep.err
  timeout 600 $CMD --workload paired --steps 3 --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002 $COMMON > $O/c5_${v}_$rep.json 2> $O/c5_${v}_$rep.err
  timeout 600 $CMD --workload single --steps 12 $COMMON > $O/single_${v}_$rep.json 2> $O/single_${v}_$rep.err
done; done
timeout 600 python scripts/ab_bench.py run pt --workload paired --steps 3 $COMMON > $O/paired_pt.json 2> $O/paired_pt.err
timeout 600 python scripts/ab_bench.py run pt --workload paired --steps 3 --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002 $COMMON > $O/c5_pt.json 2> $O/c5_pt.err
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline()); r=d["roofline"]
        print(os.path.basename(f), "%.0f reads/s" % d["value"], "ms/step %.0f" % d["ms_per_step"], "breakdown", {k: round(v,3) for k,v in (r.get("wave_cycle_breakdown") or {}).items()}, "cyc/read", r.get("wave_cycles_per_read"))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
