#!/bin/bash
# r06g: AGC 4 / 6 kernels at AGC 3's register budget (three chunks + LDS form beyond 192 positions: 3 waves per SIMD for the c5 leg), REGDIR off again,
# the tool's wait accounting.  Parity suites (incl. the 250 bp / -d 20 paired fixtures and the wide-band fuzz beyond the window forms); A/B at 256 Mb against
# r06f; the e2e leg alone at 256 Mb with SNAPGPU_SAM_VERBOSE.
O=gpurun_out/${1:-r06g}; mkdir -p $O
ls -la --time-style=full-iso snap_amd/libsnapgpu.so snap_amd/ab/*.so snap_amd/snapgpu-sam > $O/libs.txt; python -c "import bench; print('kernel_source_hash', bench.kernel_source_hash())" >> $O/libs.txt 2>&1; cat $O/libs.txt
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_secondary.py tests/test_gpu_repeats.py tests/test_gpu_flags.py tests/test_gpu_adjust.py tests/test_zy_gpu_index_shapes.py tests/test_zz_gpu_native_sam.py -m gpu -q --timeout 600 > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
C5="--workload paired --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002"
for rep in 1 2; do
for v in r06f new; do
  if [ $v = new ]; then CMD="python bench.py"; else CMD="python scripts/ab_bench.py run $v"; fi
  timeout 600 $CMD $C5 --steps 4 $COMMON > $O/c5_${v}_$rep.json 2> $O/c5_${v}_$rep.err
  timeout 600 $CMD --workload single --steps 12 $COMMON > $O/single_${v}_$rep.json 2> $O/single_${v}_$rep.err
  timeout 600 $CMD --workload single --steps 6 --read-len 250 --max-k 20 $COMMON > $O/single250_${v}_$rep.json 2> $O/single250_${v}_$rep.err
done; done
# c5 with the reference beside it (parity of every pair of the sample)
timeout 900 python bench.py $C5 --steps 4 --genome-mb 256 --skip-probe --no-extra-legs --warmup 1 > $O/c5_full.json 2> $O/c5_full.err
timeout 900 python bench.py --genome-mb 256 --workload single --steps 4 --warmup 1 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-c5-leg --paired-leg-steps 1 > $O/e2e_256.json 2> $O/e2e_256.err
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline()); r=d["roofline"]; c=d["config"]
        print("%-24s %9.0f reads/s  ms/step %7.1f  parity %s/%s" % (os.path.basename(f), d["value"], d["ms_per_step"], c.get("parity_units"), c.get("parity_mismatching")), {k: c[k] for k in c if k.startswith(("paired_","c5_","e2e_"))})
        if "e2e" in d: print("   e2e tail:", *d["e2e"].get("tool_tail", []), sep="\n      ")
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
