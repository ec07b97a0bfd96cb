#!/bin/bash
# r03q: the round's last hardware run: the whole GPU suite (with the tests added after r03z: -ae, the replay beside the main pass, the
# shim's -ae cases), smoke, the driver's bench command, and the same with the heavy-first dequeue off (three feeders: never measured)
O=gpurun_out/${1:-r03q}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 700 $O/bench_driver_cmd.json
SNAPGPU_SINGLE_HEAVY_FIRST=0 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown > $O/bench_heavy_first_off.json 2> $O/bench_heavy_first_off.err; tail -c 400 $O/bench_heavy_first_off.json
