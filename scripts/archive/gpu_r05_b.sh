#!/bin/bash
# round 5, session b: the paired kernel with the windowed hit sets -- parity on hardware (long hit lists + the GPU suite's paired file), then
# A/B against the previous sources on one box (256 Mb), its phase timers, and the new bench legs at test size
O=gpurun_out/r05b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SNAPGPU_TEST_LIB=gpu timeout 300 python scripts/emu_paired_hits_check.py 400 > $O/hits_check.txt 2>&1; echo "hits check rc=$?"; grep -v "^pairs" $O/hits_check.txt | tail -5
timeout 600 python -m pytest tests/test_gpu_paired.py -x -q > $O/pytest_paired.txt 2>&1; tail -2 $O/pytest_paired.txt
t() { tag=$1; lib=$2; shift 2; timeout 400 python scripts/ab_bench.py run $lib --workload paired --genome-mb 256 --steps 6 --warmup 1 --no-extra-legs "$@" > $O/$tag.out 2> $O/$tag.err
  python - <<P
import json
try:
    d=json.loads(open('$O/$tag.out').readline()); r=d['roofline']
    print('== $tag: %.0f reads/s, %.1f ms/step, launch avg %.1f ms [%s .. %s .. %s]; breakdown %s; parity %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r.get('launch_ms_min'), r.get('launch_ms_median'), r.get('launch_ms_max'), r.get('wave_cycle_breakdown'), d.get('parity_check')))
except Exception as e: print('$tag failed', e)
P
}
t base base --skip-cpu
t new new --cpu-seconds 8
t pt2 pt2 --skip-cpu --steps 3
timeout 900 python -m pytest tests/test_zzzz_gpu_bench.py -x -q -k extra_legs > $O/pytest_bench.txt 2>&1; tail -3 $O/pytest_bench.txt
