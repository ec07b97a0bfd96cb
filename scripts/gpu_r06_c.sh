#!/bin/bash
# r06c: ag_banded_win2 (wide bands, w 13 .. 31, one segment per register set): parity (wide-band fuzz + call sequences + the paired / single suites),
# A/B against the same sources compiled -DSNAPGPU_NO_AG_WIN2 (c5, paired, single-end d20/250), phase timers of the new build
O=gpurun_out/${1:-r06c}; mkdir -p $O
ls -la --time-style=full-iso snap_amd/libsnapgpu.so snap_amd/ab/*.so > $O/libs.txt; python -c "import bench; print('kernel_source_hash', bench.kernel_source_hash())" >> $O/libs.txt 2>&1; cat $O/libs.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_secondary.py tests/test_gpu_repeats.py -m gpu -q -x --timeout 500 > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
C5="--workload paired --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002"
for rep in 1 2; do
for v in nowin2 new; do
  if [ $v = new ]; then CMD="python bench.py"; else CMD="python scripts/ab_bench.py run $v"; fi
  timeout 600 $CMD $C5 --steps 6 $COMMON > $O/c5_${v}_$rep.json 2> $O/c5_${v}_$rep.err
  timeout 600 $CMD --workload paired --steps 6 $COMMON > $O/paired_${v}_$rep.json 2> $O/paired_${v}_$rep.err
done; done
timeout 600 python scripts/ab_bench.py run pt $C5 --steps 3 $COMMON > $O/c5_pt.json 2> $O/c5_pt.err
timeout 600 python scripts/ab_bench.py run pt --workload paired --steps 3 $COMMON > $O/paired_pt.json 2> $O/paired_pt.err
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline()); r=d["roofline"]
        print("%-24s %9.0f reads/s  ms/step %7.1f  parity %s/%s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("parity_units"), d["config"].get("parity_mismatching")), "breakdown", {k: round(v,3) for k,v in (r.get("wave_cycle_breakdown") or {}).items()}, "cyc/read", r.get("wave_cycles_per_read"))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
