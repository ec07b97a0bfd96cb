#!/bin/bash
# r06d: ONE call site for score() in AlignRead (kernel 12 400 -> 7 033 instructions) + the lazy-F verdict as a vector AND against per-lane group masks
# instead of the scalar fold (ag_win.h: kgrp).  Parity (the suites that cover single / paired / secondary / repeats / wide bands), then A/B at 256 Mb:
#   nowin2 = the commit before (its single-end and -d 8 paths are round 5's), fold = this commit with the scalar fold, new = this commit
O=gpurun_out/${1:-r06d}; mkdir -p $O
ls -la --time-style=full-iso snap_amd/libsnapgpu.so snap_amd/ab/*.so > $O/libs.txt; python -c "import bench; print('kernel_source_hash', bench.kernel_source_hash())" >> $O/libs.txt 2>&1; cat $O/libs.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_secondary.py tests/test_gpu_repeats.py tests/test_gpu_flags.py -m gpu -q --timeout 500 > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
C5="--workload paired --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002"
for rep in 1 2; do
for v in nowin2 fold new; do
  if [ $v = new ]; then CMD="python bench.py"; else CMD="python scripts/ab_bench.py run $v"; fi
  timeout 600 $CMD --workload single --steps 12 $COMMON > $O/single_${v}_$rep.json 2> $O/single_${v}_$rep.err
  timeout 600 $CMD --workload paired --steps 6 $COMMON > $O/paired_${v}_$rep.json 2> $O/paired_${v}_$rep.err
  timeout 600 $CMD $C5 --steps 4 $COMMON > $O/c5_${v}_$rep.json 2> $O/c5_${v}_$rep.err
done; done
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline()); r=d["roofline"]
        print("%-24s %9.0f reads/s  ms/step %7.1f  parity %s/%s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("parity_units"), d["config"].get("parity_mismatching")))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
