#!/bin/bash
# r06k: exact twins of the AGC 4 / 6 kernels (the replay of a long-read batch's flagged pairs no longer goes through the LDS form): parity, then c5 at 3 100 Mb and at 256 Mb
O=gpurun_out/${1:-r06k}; mkdir -p $O
ls -la --time-style=full-iso snap_amd/libsnapgpu.so snap_amd/ab/*.so > $O/libs.txt; python -c "import bench; print('kernel_source_hash', bench.kernel_source_hash())" >> $O/libs.txt 2>&1; cat $O/libs.txt
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_secondary.py tests/test_zy_gpu_index_shapes.py -m gpu -q --timeout 600 > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
C5="--workload paired --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002"
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
for rep in 1 2; do
  timeout 600 python scripts/ab_bench.py run r06i $C5 --steps 4 $COMMON > $O/c5_r06i_$rep.json 2> $O/c5_r06i_$rep.err
  timeout 600 python bench.py $C5 --steps 4 $COMMON > $O/c5_new_$rep.json 2> $O/c5_new_$rep.err
done
timeout 1500 python bench.py $C5 --steps 6 --genome-mb 3100 --skip-probe --no-extra-legs --warmup 1 > $O/c5_3100_full.json 2> $O/c5_3100_full.err
timeout 900 python scripts/ab_bench.py run r06i $C5 --steps 6 --genome-mb 3100 --skip-cpu --skip-probe --no-extra-legs --warmup 1 > $O/c5_3100_r06i.json 2> $O/c5_3100_r06i.err
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline()); r=d["roofline"]; c=d["config"]
        print("%-28s %9.0f reads/s  ms/step %7.1f  genome %s  parity %s/%s cpu %s" % (os.path.basename(f), d["value"], d["ms_per_step"], c.get("genome_mb"), c.get("parity_units"), c.get("parity_mismatching"), (d.get("cpu_baseline") or {}).get("value")))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
