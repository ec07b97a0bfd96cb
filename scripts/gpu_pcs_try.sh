#!/bin/bash
for cfg in "stochastic cycles 65536" "stochastic cycles 4096" "host_trap time 100" "host_trap time 1"; do
  set -- $cfg
  echo "== trying $cfg"
  METHOD=$1 UNIT=$2 INTERVAL=$3 T=240 bash scripts/gpu_pcsample.sh r04g_$1_$3 --feeders 1 2>&1 | tail -12 | cut -c1-400
  if ls gpurun_out/r04g_$1_$3/*_hist.csv > /dev/null 2>&1; then echo "== worked: $cfg"; break; fi
done
