#!/bin/bash
# r06n: two Landau-Vishkin problems to a wavefront (lv.h: lv_compute_pair_inl) as an A/B library: parity of the variant, then single-end A/B at 256 Mb
O=gpurun_out/${1:-r06n}; mkdir -p $O
ls -la --time-style=full-iso snap_amd/libsnapgpu.so snap_amd/ab/*.so > $O/libs.txt; cat $O/libs.txt
SNAPGPU_TEST_LIB=snap_amd/ab/libsnapgpu_lvpair.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_secondary.py tests/test_gpu_flags.py tests/test_gpu_repeats.py tests/test_gpu_adjust.py tests/test_zy_gpu_index_shapes.py -m gpu -q --timeout 600 > $O/pytest_lvpair.txt 2>&1; tail -4 $O/pytest_lvpair.txt
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
for rep in 1 2 3; do
  timeout 600 python bench.py --workload single --steps 12 $COMMON > $O/single_base_$rep.json 2> $O/single_base_$rep.err
  timeout 600 python scripts/ab_bench.py run lvpair --workload single --steps 12 $COMMON > $O/single_lvpair_$rep.json 2> $O/single_lvpair_$rep.err
done
timeout 900 python scripts/ab_bench.py run lvpair --workload single --steps 12 --genome-mb 256 --skip-probe --no-extra-legs --warmup 1 > $O/single_lvpair_full.json 2> $O/single_lvpair_full.err
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline()); c=d["config"]
        print("%-28s %9.0f reads/s  ms/step %7.1f  parity %s/%s" % (os.path.basename(f), d["value"], d["ms_per_step"], c.get("parity_units"), c.get("parity_mismatching")), {k: round(v,3) for k,v in (d["roofline"].get("wave_cycle_breakdown") or {}).items()})
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
