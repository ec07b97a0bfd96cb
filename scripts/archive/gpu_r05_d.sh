#!/bin/bash
# round 5, session d: the paired kernel with its LDS state addressed as LDS (llvm.assume(is.shared)) -- parity, A/B on one box, phase timers;
# the native tool's paired -om run that ran out of memory in session c
O=gpurun_out/r05d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SNAPGPU_TEST_LIB=gpu timeout 300 python scripts/emu_paired_hits_check.py 300 > $O/hits_check.txt 2>&1; echo "hits check rc=$?"; grep -v "^pairs" $O/hits_check.txt | tail -5
timeout 600 python -m pytest tests/test_gpu_paired.py tests/test_zz_gpu_native_sam.py -q > $O/pytest_paired.txt 2>&1; tail -3 $O/pytest_paired.txt
t() { tag=$1; lib=$2; shift 2; timeout 400 python scripts/ab_bench.py run $lib --workload paired --genome-mb 256 --steps 6 --warmup 1 --no-extra-legs "$@" > $O/$tag.out 2> $O/$tag.err
  python - <<P
import json
try:
    d=json.loads(open('$O/$tag.out').readline()); r=d['roofline']
    print('== $tag: %.0f reads/s, %.1f ms/step, launch [%.0f .. %.0f .. %.0f]; cycles/read %s breakdown %s; parity %s' % (d['value'], d['ms_per_step'], r.get('launch_ms_min',0), r.get('launch_ms_median',0), r.get('launch_ms_max',0), r.get('wave_cycles_per_read'), r.get('wave_cycle_breakdown'), d.get('parity_check')))
except Exception as e: print('$tag failed', e)
P
}
t base base --skip-cpu
t new new --cpu-seconds 8
t pt2 pt2 --skip-cpu --steps 3
t base2 base --skip-cpu
