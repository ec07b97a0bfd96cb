#!/bin/bash
# r04a: the round's first hardware call (trimmed scripts/gpu_r04_first.sh + the GRCh38-scale run with bench.py's default options, timed phase by phase)
O=gpurun_out/${1:-r04a}; mkdir -p $O
run() { tag=$1; shift; t0=$SECONDS; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ); echo "wall=$((SECONDS-t0)) s" >> $O/$tag.err ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-420}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160) $(tail -n 1 $O/$tag.err)"; }
SNAPGPU_TEST_UNMEASURED=1 timeout 180 python -m pytest tests/test_zzz_gpu_resolve.py -m gpu -q > $O/pytest_resolve.txt 2>&1; tail -2 $O/pytest_resolve.txt
run f3_default python bench.py --gpus 1 --steps 20 --warmup 5 --skip-probe --skip-refwalk --skip-breakdown --skip-cpu
SNAPGPU_SINGLE_RESOLVE=1 run f3_resolve python bench.py --gpus 1 --steps 20 --warmup 5 --skip-probe --skip-refwalk --skip-breakdown
run f1_default python bench.py --feeders 1 --steps 6 --skip-probe --skip-refwalk --skip-breakdown --skip-cpu
SNAPGPU_SINGLE_RESOLVE=1 run f1_resolve python bench.py --feeders 1 --steps 6 --skip-probe --skip-refwalk
T=400 run g3100_default python bench.py --genome-mb 3100
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  SNAPGPU_SINGLE_RESOLVE=1 timeout 150 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_resolve_$i -o bench -- python bench.py --steps 1 --warmup 0 --feeders 1 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown > $O/pmc_resolve_$i.json 2> $O/pmc_resolve_$i.err < /dev/null
done
run p_f1_default python bench.py --workload paired --feeders 1 --steps 2 --skip-cpu
SNAPGPU_PAIRED_HEAVY_FIRST=1 SNAPGPU_PAIRED_REPLAY_BESIDE=1 run p_f1_heavy_first_beside python bench.py --workload paired --feeders 1 --steps 2 --skip-cpu
run p_f3_default python bench.py --workload paired --skip-cpu
