#!/bin/bash
# r04v: the affine-gap row loop with fewer SCALAR instructions (masks that follow (nk0, nk1) cached, lazy-F rounds as nested instantiations,
# X by one readlane per round, open - ext in a vector register, ...) against the committed build; the same with 32 dummy scalar / vector
# instructions per row (which unit the row loop's time follows); and the kernel built for 7 / 8 waves per SIMD.
O=gpurun_out/${1:-r04v}; mkdir -p $O
t() { tag=$1; lib=$2; wpc=$3; SNAPGPU_WAVES_PER_CU=$wpc timeout 200 python scripts/ab_bench.py run $lib --genome-mb 256 --no-extra-legs --steps 12 --warmup 3 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown > $O/$tag.out 2> $O/$tag.err
  python -c "import json;d=json.loads(open('$O/$tag.out').readline());print('== $tag: %.0f reads/s, %.1f ms/step, launch %.1f ms, parity %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['parity_check'].get('mismatching_fields')))" 2>&1 | tail -1; }
t base base 24
t s1 s1 24
t s1ds s1ds 24
t s1dv s1dv 24
t w7 w7 28
t w8 w8 32
t s1b s1 24
