#!/bin/bash
# r04n: the SAM-field kernel at other occupancies (launch bounds 4 / 6 / 8 waves per SIMD x blocks per CU of its persistent grid)
O=gpurun_out/${1:-r04n}; mkdir -p $O
t() { tag=$1; lib=$2; b=$3; SNAPGPU_AB_LIB=$lib SNAPGPU_SAMF_BLOCKS_PER_CU=$b timeout 200 python scripts/gpu_sam_perf.py 300000 > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); print("== %s: M %.0f reads/s (kernel %.0f), =X %.0f (kernel %.0f)" % (sys.argv[2], d["sam_fields_M"]["reads_per_s"], d["sam_fields_M"]["kernel_reads_per_s"], d["sam_fields_eqx"]["reads_per_s"], d["sam_fields_eqx"]["kernel_reads_per_s"]))
except Exception as e: print("== %s FAILED %s" % (sys.argv[2], e))
PY
}
t w4_b4 "" 4
t w4_b3 "" 3
t w6_b6 snap_amd/ab/libsnapgpu_s6.so 6
t w8_b8 snap_amd/ab/libsnapgpu_s8.so 8
t w8_b6 snap_amd/ab/libsnapgpu_s8.so 6
