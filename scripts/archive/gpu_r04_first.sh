#!/bin/bash
# What to run first next round (nothing here has been measured: the round-3 GPU minutes were spent when these were written).
#  1. the RESOLVE instantiation's first hardware run (tests/test_zzz_gpu_resolve.py), then the whole suite
#  2. single end with SNAPGPU_SINGLE_RESOLVE=1 (fast form + calls that leave their band answered in place, no images, no replay)
#     against the default (exact form as the one pass), three feeders and one context, parity on
#  3. PMC passes of the RESOLVE kernel: the point of it is the traffic (84 GB per launch now, mostly traceback images)
#  4. paired end, one context: heavy-first dequeue + the exact replay beside the main pass together (the flagged pairs are the heaviest:
#     started first, their replay overlaps the rest of the batch)
O=gpurun_out/${1:-r04a}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-260}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160)"; }
SNAPGPU_TEST_UNMEASURED=1 timeout 120 python -m pytest tests/test_zzz_gpu_resolve.py -m gpu -q > $O/pytest_resolve.txt 2>&1; tail -2 $O/pytest_resolve.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
run f3_default python bench.py --gpus 1 --steps 20 --warmup 5 --skip-probe --skip-refwalk --skip-breakdown
SNAPGPU_SINGLE_RESOLVE=1 run f3_resolve python bench.py --gpus 1 --steps 20 --warmup 5 --skip-probe --skip-refwalk --skip-breakdown
run f1_default python bench.py --feeders 1 --steps 6 --skip-probe --skip-refwalk --skip-breakdown
SNAPGPU_SINGLE_RESOLVE=1 run f1_resolve python bench.py --feeders 1 --steps 6 --skip-probe --skip-refwalk --skip-breakdown
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  SNAPGPU_SINGLE_RESOLVE=1 timeout 150 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_resolve_$i -o bench -- python bench.py --steps 1 --warmup 0 --feeders 1 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown > $O/pmc_resolve_$i.json 2> $O/pmc_resolve_$i.err < /dev/null
done
run p_f1_default python bench.py --workload paired --feeders 1 --steps 2 --skip-cpu
SNAPGPU_PAIRED_HEAVY_FIRST=1 SNAPGPU_PAIRED_REPLAY_BESIDE=1 run p_f1_heavy_first_beside python bench.py --workload paired --feeders 1 --steps 2 --skip-cpu
