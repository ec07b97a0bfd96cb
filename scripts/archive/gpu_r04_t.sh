#!/bin/bash
# r04t: three feeders with the FAST form as the main pass + exact replay of the flagged reads (SNAPGPU_NO_ALWAYS_EXACT=1) against the default
# (exact form as the one pass): speed, parity, and FETCH_SIZE / WRITE_SIZE of each
O=gpurun_out/${1:-r04t}; mkdir -p $O
t() { tag=$1; shift; timeout 200 "$@" > $O/$tag.out 2> $O/$tag.err
  python -c "import json;d=json.loads(open('$O/$tag.out').readline());print('== $tag: %.0f reads/s, %.1f ms/step, launch %.1f ms, parity %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('parity_check')))" 2>&1 | tail -1; }
t exact_main python bench.py --genome-mb 256 --no-extra-legs --steps 12 --warmup 3 --skip-probe --skip-refwalk --skip-breakdown --cpu-seconds 3
SNAPGPU_NO_ALWAYS_EXACT=1 t fast_replay python bench.py --genome-mb 256 --no-extra-legs --steps 12 --warmup 3 --skip-probe --skip-refwalk --skip-breakdown --cpu-seconds 3
t exact_main_b python bench.py --genome-mb 256 --no-extra-legs --steps 12 --warmup 3 --skip-probe --skip-refwalk --skip-breakdown --skip-cpu
SNAPGPU_NO_ALWAYS_EXACT=1 t fast_replay_b python bench.py --genome-mb 256 --no-extra-legs --steps 12 --warmup 3 --skip-probe --skip-refwalk --skip-breakdown --skip-cpu
SNAPGPU_NO_ALWAYS_EXACT=1 timeout 400 python scripts/pmc_collect.py $O/pmc_fast --genome-mb 256 --groups 2 > $O/pmc_fast.txt 2>&1; tail -c 700 $O/pmc_fast.txt
