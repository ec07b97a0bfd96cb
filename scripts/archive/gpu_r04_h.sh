#!/bin/bash
# r04h: the product build after the row-maximum shortcut and the probe kernel's four-deep list loads: lookup tests, single-end line at 256 Mb with
# the probe roofline, paired with 6 steps
O=gpurun_out/${1:-r04h}; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "lookup or native_library" > $O/pytest_lookup.txt 2>&1; tail -2 $O/pytest_lookup.txt
run() { tag=$1; shift; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ); python - "$O/$tag.out" "$tag" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); r=d["roofline"]; pc=d.get("parity_check",{}); p=r.get("probe",{})
    print("== %s: %.0f reads/s, %.1f ms/step, parity %s, probe %s" % (sys.argv[2], d["value"], d["ms_per_step"],
          {k:pc.get(k) for k in ("reads","pairs","mismatching_fields","mismatching_pairs")}, {k:p.get(k) for k in ("lookups_per_s","frac","frac_bucket_lines","avg_launch_ms")}))
except Exception as e:
    print("== %s: FAILED %s" % (sys.argv[2], e)); print(open(sys.argv[1].replace(".out",".err")).read()[-800:])
PY
}
run s256 python bench.py --genome-mb 256 --no-extra-legs --steps 12 --warmup 3 --skip-breakdown --cpu-seconds 3
run p256 python bench.py --genome-mb 256 --workload paired --no-extra-legs --steps 9 --warmup 3 --cpu-seconds 3
