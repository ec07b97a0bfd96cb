#!/bin/bash
# r06m: Landau-Vishkin micro-cuts (one masked store per bitmap, the rank selects behind one test, a plane-less entry): A/B against r06z's library, parity
O=gpurun_out/${1:-r06m}; mkdir -p $O
ls -la --time-style=full-iso snap_amd/libsnapgpu.so snap_amd/ab/*.so > $O/libs.txt; python -c "import bench; print('kernel_source_hash', bench.kernel_source_hash())" >> $O/libs.txt 2>&1; cat $O/libs.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_secondary.py tests/test_gpu_planes.py tests/test_zz_gpu_cigar.py -m gpu -q --timeout 600 > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
for rep in 1 2 3; do
  timeout 600 python scripts/ab_bench.py run r06z --workload single --steps 12 $COMMON > $O/single_r06z_$rep.json 2> $O/single_r06z_$rep.err
  timeout 600 python bench.py --workload single --steps 12 $COMMON > $O/single_new_$rep.json 2> $O/single_new_$rep.err
done
for rep in 1 2; do
  timeout 600 python scripts/ab_bench.py run r06z --workload paired --steps 6 $COMMON > $O/paired_r06z_$rep.json 2> $O/paired_r06z_$rep.err
  timeout 600 python bench.py --workload paired --steps 6 $COMMON > $O/paired_new_$rep.json 2> $O/paired_new_$rep.err
done
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline())
        print("%-28s %9.0f reads/s  ms/step %7.1f" % (os.path.basename(f), d["value"], d["ms_per_step"]))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
