#!/bin/bash
# r03u: the wide-location index shapes (5 .. 7-byte locations, narrowed on load) on hardware: probe, AlignRead, paired end vs the live reference
O=gpurun_out/${1:-r03u}; mkdir -p $O
timeout 140 python -m pytest tests/test_zy_gpu_index_shapes.py -m gpu -q -k "wide" > $O/pytest_wide.txt 2>&1; tail -3 $O/pytest_wide.txt
