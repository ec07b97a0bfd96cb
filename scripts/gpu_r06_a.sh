#!/bin/bash
# r06a: the paired-end kernel's frame (Aligner / DevPL / PairedCore) in LDS instead of per-lane scratch: parity, then A/B against round 5's library
O=gpurun_out/${1:-r06a}; mkdir -p $O
ls -la --time-style=full-iso snap_amd/libsnapgpu.so snap_amd/ab/*.so > $O/libs.txt; python -c "import bench; print('kernel_source_hash', bench.kernel_source_hash())" >> $O/libs.txt 2>&1; cat $O/libs.txt
timeout 900 python -m pytest tests/test_gpu_paired.py -m gpu -q -x --timeout 300 > $O/pytest_paired.txt 2>&1; tail -3 $O/pytest_paired.txt
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
for rep in 1 2; do
for v in r05 new; do
  if [ $v = new ]; then CMD="python bench.py"; else CMD="python scripts/ab_bench.py run $v"; fi
  timeout 600 $CMD --workload paired --steps 6 $COMMON > $O/paired_${v}_$rep.json 2> $O/paired_${v}_$rep.err
  timeout 600 $CMD --workload paired --steps 3 --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002 $COMMON > $O/c5_${v}_$rep.json 2> $O/c5_${v}_$rep.err
done; done
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline()); pc=d.get("parity_check",{})
        print(os.path.basename(f), "%.0f reads/s" % d["value"], "ms/step %.0f" % d["ms_per_step"], "parity", {k:pc.get(k) for k in pc if k.startswith(("pairs","reads","mism"))})
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
