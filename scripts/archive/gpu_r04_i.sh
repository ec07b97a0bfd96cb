#!/bin/bash
# r04i: the probe kernel's variants (pA: dynamic pass distribution + 4-deep list loads; pB: + 8 blocks per CU (64 VGPRs); pC: 8-deep list loads),
# the -f / -x test against the live reference, and the lookup tests on the product build
O=gpurun_out/${1:-r04i}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_flags.py tests/test_gpu_parity.py -m gpu -q -k "flags or stop_on_first or lookup or native_library" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for v in pA pB pC; do
  timeout 200 python scripts/ab_bench.py run $v --genome-mb 256 --no-extra-legs --steps 3 --warmup 1 --skip-breakdown --skip-cpu > $O/probe_$v.out 2> $O/probe_$v.err
  python - "$O/probe_$v.out" $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); p=d["roofline"]["probe"]
print("== %s: align %.0f reads/s; probe %s" % (sys.argv[2], d["value"], {k:p.get(k) for k in ("lookups_per_s","frac","frac_bucket_lines","avg_launch_ms")}))
PY
done
