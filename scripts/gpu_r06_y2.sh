#!/bin/bash
# r06y2: snapgpu-sam's short first calls (2, then 4 batches per call at the start of a pass) against the closing build's tool, e2e leg at 3 100 Mb
O=gpurun_out/${1:-r06y2}; mkdir -p $O
ARGS="--gpus 1 --steps 3 --warmup 1 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-c5-leg --paired-leg-steps 1"
timeout 400 python bench.py $ARGS > $O/e2e_new.json 2> $O/e2e_new.err
cp snap_amd/snapgpu-sam /tmp/snapgpu-sam.new; cp snap_amd/ab/snapgpu-sam.before snap_amd/snapgpu-sam
timeout 400 python bench.py $ARGS > $O/e2e_before.json 2> $O/e2e_before.err
cp /tmp/snapgpu-sam.new snap_amd/snapgpu-sam
python - $O <<'PY' | tee $O/summary.txt
import json,sys
for t in ("new","before"):
    try:
        d=json.loads(open(sys.argv[1]+"/e2e_%s.json"%t).readline()); e=d["e2e"]
        print(t, e.get("value"), e.get("pass_values"), "identical:", e.get("identical_records"), [l for l in e.get("tool_tail",[]) if "timeline" in l][-1:])
    except Exception as ex: print(t, "ERR", ex)
PY
