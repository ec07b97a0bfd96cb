#!/bin/bash
# r04w: s2 = s1 (profiles/r04v) + the unbanded one-segment affine-gap calls through the window form's row (ag_banded_win<.., FULL>) + reads /
# qualities / window handed to the Landau-Vishkin and affine-gap functions as LDS pointers (LdsSeq: ds_read_u8 instead of FLAT loads);
# s2 once more with the reference beside it (parity of every sampled read), and its paired-end leg.
O=gpurun_out/${1:-r04w}; mkdir -p $O
t() { tag=$1; lib=$2; shift 2; timeout 300 python scripts/ab_bench.py run $lib --genome-mb 256 --no-extra-legs --steps 12 --warmup 3 --skip-probe --skip-refwalk --skip-breakdown "$@" > $O/$tag.out 2> $O/$tag.err
  python -c "import json;d=json.loads(open('$O/$tag.out').readline());print('== $tag: %.0f reads/s, %.1f ms/step, launch %.1f ms, parity %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], json.dumps(d.get('parity_check'))[:200]))" 2>&1 | tail -1; }
t s1 s1 --skip-cpu
t s2 s2 --skip-cpu
t s2_parity s2 --cpu-seconds 4
t s2_paired s2 --workload paired --steps 6 --warmup 2 --cpu-seconds 4
t s1b s1 --skip-cpu
t s2b s2 --skip-cpu
