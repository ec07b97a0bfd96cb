#!/bin/bash
# r03x: the last seconds of the round's GPU budget: single-end parity + adjuster fixture tests on the final build, and the single-end
# fixture tests through the plane instantiations (SNAPGPU_LV_PLANES=1)
O=gpurun_out/${1:-r03x}; mkdir -p $O
SNAPGPU_LV_PLANES=1 timeout 15 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fixture" > $O/pytest_planes.txt 2>&1; tail -1 $O/pytest_planes.txt
timeout 35 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adjust.py tests/test_gpu_secondary.py -m gpu -q -x -k "not live" > $O/pytest_single.txt 2>&1; tail -1 $O/pytest_single.txt
