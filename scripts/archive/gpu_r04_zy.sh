#!/bin/bash
# r04zy: FASTQ -> SAM at 20 M reads on the closing build, with more feeders / other batch sizes beside the defaults (a launch lasts as long as its
# slowest read, ~150 ms for a batch that holds one of the heavy reads: small batches need more launches in flight to hide that).
# The reference CLI's side of the comparison is in profiles/r04z: same FASTQ generator, same seed, records hash ab75b74b961b9c21.
O=gpurun_out/${1:-r04zy}; mkdir -p $O
E2E_SWEEP="-q 6;-q 8;-b 262144 -q 6" timeout 500 python scripts/gpu_e2e_sam.py 20000000 --skip-reference > $O/e2e_sam.json 2> $O/e2e_sam.err
python - $O/e2e_sam.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["snapgpu_sam"]; r=d.get("snap_aligner_reference",{})
print("== e2e 20 M: snapgpu-sam %.0f reads/s streaming (wall %.1f s), records hash %s" % (s.get("reads_per_s_streaming",0), s["wall_s"], s.get("records_hash")))
for k,v in d.items():
    if k.startswith("snapgpu_sam ") : print("   %s: %.0f reads/s streaming, hash %s | %s" % (k, v.get("reads_per_s_streaming",0), v.get("records_hash"), v["tool_tail"][-1][:200]))
PY
