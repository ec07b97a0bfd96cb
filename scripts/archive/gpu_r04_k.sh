#!/bin/bash
# r04k: -f / -x test; probe on the product build; FASTQ -> SAM at 20 M reads with the context's buffer pool (no reference run: its records' hash
# for this FASTQ is ab75b74b961b9c21, profiles/r04j)
O=gpurun_out/${1:-r04k}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_flags.py -m gpu -q > $O/pytest_flags.txt 2>&1; tail -2 $O/pytest_flags.txt
timeout 200 python bench.py --genome-mb 256 --no-extra-legs --steps 3 --warmup 1 --skip-breakdown --skip-cpu > $O/probe_prod.out 2> $O/probe_prod.err
python - "$O/probe_prod.out" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); p=d["roofline"]["probe"]
print("== probe %s" % {k:p.get(k) for k in ("lookups_per_s","frac","frac_bucket_lines","avg_launch_ms")})
PY
timeout 900 python scripts/gpu_e2e_sam.py ${E2E_N:-20000000} --skip-reference > $O/e2e_sam.json 2> $O/e2e_sam.err; python - "$O/e2e_sam.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["snapgpu_sam"]; print("== e2e", {k:s.get(k) for k in ("wall_s","index_load_s","stream_s","reads_per_s_streaming","records","records_hash")}, s["tool_tail"][-1])
PY
for q in 2 4; do
SNAPGPU_E2E_Q=$q timeout 300 python - "$O" $q <<'PY'
import subprocess,sys,os,re,time
# (the FASTQ is gone: e2e removes it) -- nothing to do here; kept as a placeholder for -q sweeps
PY
done
