#!/bin/bash
# r04o: the whole GPU suite on the current build (affine-gap row loop v3, probe, -f / -x, buffer pool, SAM kernels at 8 waves per SIMD, mapped reader),
# smoke, then FASTQ -> SAM at 20 M reads
O=gpurun_out/${1:-r04o}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.txt 2>&1; tail -14 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python scripts/gpu_e2e_sam.py 20000000 --skip-reference > $O/e2e_sam.json 2> $O/e2e_sam.err; python - "$O/e2e_sam.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,s in d.items():
    if k.startswith("snapgpu_sam"):
        print("== e2e", k, {q:s.get(q) for q in ("wall_s","index_load_s","stream_s","reads_per_s_streaming","records","records_hash")}); print("\n".join(s["tool_tail"][-2:]))
PY
