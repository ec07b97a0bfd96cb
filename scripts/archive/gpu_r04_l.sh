#!/bin/bash
# r04l: FASTQ -> SAM at 20 M reads with the host threads' time split printed
O=gpurun_out/${1:-r04l}; mkdir -p $O
timeout 900 python scripts/gpu_e2e_sam.py ${E2E_N:-20000000} --skip-reference > $O/e2e_sam.json 2> $O/e2e_sam.err; python - "$O/e2e_sam.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,s in d.items():
    if k.startswith("snapgpu_sam"):
        print("== e2e", k, {q:s.get(q) for q in ("wall_s","index_load_s","stream_s","reads_per_s_streaming","records","records_hash")}); print("\n".join(s["tool_tail"][-2:]))
PY
