#!/bin/bash
# r03n: diagnostic for the batch-composition test + the plane Landau-Vishkin hybrid A/B (see gpu_r03_m.sh)
tag=${1:-r03n}; out=gpurun_out/$tag; mkdir -p $out
timeout 300 python scripts/diag_batch_composition.py > $out/diag.txt 2>&1
tail -40 $out/diag.txt
timeout 200 python -m pytest tests/test_gpu_paired.py -x -q -m gpu -k "batch_composition or fixture" 2>&1 | tail -5
bash scripts/gpu_r03_m.sh $tag
