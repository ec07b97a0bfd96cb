#!/bin/bash
# r03r: after "heavy-first only where a context has the GPU to itself": the driver's bench command again, one context, and the tests
# added after r03q (snapgpu-sam -ae) plus the single-end parity and multi-context files
O=gpurun_out/${1:-r03r}; mkdir -p $O
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 500 $O/bench_driver_cmd.json
timeout 200 python bench.py --feeders 1 --steps 4 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown > $O/single_f1.json 2> $O/single_f1.err; tail -c 300 $O/single_f1.json
timeout 400 python -m pytest tests/test_zz_gpu_native_sam.py tests/test_gpu_parity.py tests/test_gpu_multi_ctx.py -m gpu -q -k "ae or parity or multi_ctx or feeders or replica" > $O/pytest_subset.txt 2>&1; tail -3 $O/pytest_subset.txt
