#!/bin/bash
# round 5, session g: register-resident hit-set cursors (paired), the seed-probability fix on hardware, the SAM side with two-vector row
# loops ahead of the records and grouped GPU calls in snapgpu-sam
O=gpurun_out/r05g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SNAPGPU_TEST_LIB=gpu timeout 300 python scripts/emu_paired_hits_check.py 200 > $O/hits_check.txt 2>&1; echo "hits check rc=$?"; grep -v "^pairs" $O/hits_check.txt | tail -5
timeout 900 python -m pytest tests/test_gpu_paired.py tests/test_zz_gpu_native_sam.py tests/test_zz_gpu_cigar.py -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
t() { tag=$1; lib=$2; shift 2; timeout 400 python scripts/ab_bench.py run $lib --workload paired --genome-mb 256 --no-extra-legs --skip-cpu "$@" > $O/$tag.out 2> $O/$tag.err
  python - <<P
import json
try:
    d=json.loads(open('$O/$tag.out').readline()); r=d['roofline']
    print('== $tag: %.0f reads/s, hipEvent launch avg %.1f ms, calls [%.0f .. %.0f .. %.0f] cycles/read %s %s' % (d['value'], r['avg_launch_ms'], r.get('launch_ms_min',0), r.get('launch_ms_median',0), r.get('launch_ms_max',0), r.get('wave_cycles_per_read'), r.get('wave_cycle_breakdown')))
except Exception as e: print('$tag failed', e)
P
}
t pt2_f3 pt2 --steps 3 --warmup 1
t new_f3 new --steps 6 --warmup 1
t base_f3 base --steps 6 --warmup 1
timeout 200 python scripts/gpu_sam_perf.py 400000 > $O/sam_perf.json 2> $O/sam_perf.err; python -c "
import json; d=json.loads(open('$O/sam_perf.json').readline()); print('sam_fields: kernel %.2f M reads/s (M), %.2f M (=/X)' % (d['sam_fields_M']['kernel_reads_per_s']/1e6, d['sam_fields_eqx']['kernel_reads_per_s']/1e6))"
E2E_SWEEP="-g 4;-q 3;-q 6" timeout 600 python scripts/gpu_e2e_sam.py 20000000 --skip-reference --keep > $O/e2e.json 2> $O/e2e.err
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
