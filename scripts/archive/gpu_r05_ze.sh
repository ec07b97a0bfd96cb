#!/bin/bash
# r05ze: snapgpu-sam with its group buffers page-locked (hipHostRegister): the SAM-identity tests of the tool on the hardware, then FASTQ -> SAM
# at 20 M reads (256 Mb genome) with the registration on / off / on / off over one FASTQ (first run hashed, the others timed only).
O=gpurun_out/${1:-r05ze}; mkdir -p $O
timeout 300 python -m pytest tests/test_zz_gpu_native_sam.py -m gpu -q -k "page_locked or (identical_to_reference_cli and single) or fastq_to_sam_identical" --timeout 150 > $O/pytest_native_sam.txt 2>&1; tail -4 $O/pytest_native_sam.txt
E2E_SWEEP="SNAPGPU_SAM_PIN=0;SNAPGPU_SAM_PIN=1;SNAPGPU_SAM_PIN=0" timeout 400 python scripts/gpu_e2e_sam.py 20000000 --skip-reference --no-hash > $O/e2e_pin.json 2> $O/e2e_pin.err
python - $O/e2e_pin.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline())
for k,v in d.items():
    if isinstance(v,dict) and "stream_s" in v: print(k, v.get("reads_per_s_streaming"), v.get("records_hash"), v["tool_tail"][-1][:200])
PY
