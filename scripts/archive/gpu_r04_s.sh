#!/bin/bash
# r04s: the single-end kernel built for 5 / 4 waves per SIMD (96 / 128 VGPRs) against 6 (80): fewer spills against fewer waves
O=gpurun_out/${1:-r04s}; mkdir -p $O
t() { tag=$1; lib=$2; wpc=$3; SNAPGPU_WAVES_PER_CU=$wpc timeout 200 python scripts/ab_bench.py run $lib --genome-mb 256 --no-extra-legs --steps 12 --warmup 3 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown > $O/$tag.out 2> $O/$tag.err
  python -c "import json;d=json.loads(open('$O/$tag.out').readline());print('== $tag: %.0f reads/s, %.1f ms/step, launch %.1f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))" 2>&1 | tail -1; }
t base_w6 base 24
t w5 w5 20
t w4 w4 16
t base_w6b base 24
