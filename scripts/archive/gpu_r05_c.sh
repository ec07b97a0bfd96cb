#!/bin/bash
# round 5, session c: where Phase 2's cycles go (sub-timers), and FASTQ -> SAM with the one-call path / read-length classes / 1 M-read batches
O=gpurun_out/r05c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 400 python scripts/ab_bench.py run pt3 --workload paired --genome-mb 256 --steps 3 --warmup 1 --skip-cpu --no-extra-legs > $O/paired_pt3.out 2> $O/paired_pt3.err
python - <<P
import json
try:
    d=json.loads(open('$O/paired_pt3.out').readline()); r=d['roofline']
    print('paired pt3: %.0f reads/s; [lookup=queries, lv=best_possible, ag=records, single_fallback=phase2a, hits=all of phase 2] %s; cycles/read %.0f' % (d['value'], r.get('wave_cycle_breakdown'), r.get('wave_cycles_per_read',0)))
except Exception as e: print('paired pt3 failed', e)
P
timeout 900 python -m pytest tests/test_zz_gpu_native_sam.py -x -q > $O/pytest_native_sam.txt 2>&1; tail -3 $O/pytest_native_sam.txt
E2E_SWEEP="-q 3;-q 6;-b 524288" timeout 600 python scripts/gpu_e2e_sam.py 20000000 --skip-reference --keep > $O/e2e.json 2> $O/e2e.err
python - <<P
import json
try:
    d=json.loads(open('$O/e2e.json').readline())
    for k,v in d.items():
        if isinstance(v,dict) and 'tool_tail' in v: print(k, v.get('reads_per_s_streaming'), v['records_hash'], v['tool_tail'][-1][:220])
except Exception as e: print('e2e failed', e)
P
D=${SNAP_BENCH_DIR:-/tmp/snap_bench}
IDX=$(ls -d $D/*256*/idx 2>/dev/null | head -1); FQ=$(ls $D/*/e2e.fq | head -1)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/e2e_stats -o sam -- snap_amd/snapgpu-sam single $IDX $FQ -d 8 -o /tmp/e2e_prof.sam > $O/e2e_prof.txt 2>&1
head -6 $O/e2e_stats/*kernel_stats.csv 2>/dev/null | cut -c1-200
