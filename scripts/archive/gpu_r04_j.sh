#!/bin/bash
# r04j: -f / -x on hardware; the probe kernel with the grid sized to what is resident (and 64-VGPR variant pB); FASTQ -> SAM end to end at 20 M reads
O=gpurun_out/${1:-r04j}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_flags.py -m gpu -q > $O/pytest_flags.txt 2>&1; tail -2 $O/pytest_flags.txt
probe() { tag=$1; shift; timeout 200 "$@" --genome-mb 256 --no-extra-legs --steps 3 --warmup 1 --skip-breakdown --skip-cpu > $O/probe_$tag.out 2> $O/probe_$tag.err
  python - "$O/probe_$tag.out" $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); p=d["roofline"]["probe"]
    print("== %s: probe %s" % (sys.argv[2], {k:p.get(k) for k in ("lookups_per_s","frac","frac_bucket_lines","avg_launch_ms")}))
except Exception as e: print("== %s FAILED %s" % (sys.argv[2], e))
PY
}
probe prod python bench.py
SNAPGPU_LOOKUP_BLOCKS_PER_CU=5 probe prod_b5 python bench.py
SNAPGPU_LOOKUP_BLOCKS_PER_CU=6 probe prod_b6 python bench.py
SNAPGPU_LOOKUP_BLOCKS_PER_CU=8 probe prod_b8 python bench.py
probe pB python scripts/ab_bench.py run pB
SNAPGPU_LOOKUP_BLOCKS_PER_CU=7 probe pB_b7 python scripts/ab_bench.py run pB
timeout 900 python scripts/gpu_e2e_sam.py ${E2E_N:-20000000} > $O/e2e_sam.json 2> $O/e2e_sam.err; tail -c 1500 $O/e2e_sam.json; tail -3 $O/e2e_sam.err
