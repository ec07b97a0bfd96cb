#!/bin/bash
# r03p: the exact replay beside the paired main pass (launch_paired, PairedArgs::rq) on hardware: the paired tests (with the new
# concurrency test), then one context with and without it, then three feeders with it forced on
O=gpurun_out/${1:-r03p}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-240} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-260}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160)"; }
timeout 400 python -m pytest tests/test_gpu_paired.py tests/test_gpu_multi_ctx.py -m gpu -q -x > $O/pytest_paired.txt 2>&1; tail -3 $O/pytest_paired.txt
run p_f1_beside python bench.py --workload paired --feeders 1 --steps 2 --skip-cpu
SNAPGPU_PAIRED_REPLAY_BESIDE=0 run p_f1_after python bench.py --workload paired --feeders 1 --steps 2 --skip-cpu
SNAPGPU_PAIRED_REPLAY_BESIDE=1 run p_f3_beside python bench.py --workload paired --skip-cpu
