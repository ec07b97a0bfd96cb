#!/bin/bash
# r03w: the default kernels without the plane Landau-Vishkin code (its own instantiations now): the driver's bench command and the single-end fixture tests
O=gpurun_out/${1:-r03w}; mkdir -p $O
timeout 60 python bench.py --gpus 1 --steps 20 --warmup 5 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 250 $O/bench_driver_cmd.json
timeout 30 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fixture" > $O/pytest_parity.txt 2>&1; tail -2 $O/pytest_parity.txt
