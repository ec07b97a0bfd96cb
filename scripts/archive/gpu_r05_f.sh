#!/bin/bash
# round 5, session f: sessions d and e ran a library whose link had failed (a device-only builtin in the host pass): everything since the
# hit-set windows measured HERE for the first time -- LDS state addressed as LDS, the eight-reads-per-wave row loops of the SAM side
O=gpurun_out/r05f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SNAPGPU_TEST_LIB=gpu timeout 300 python scripts/emu_paired_hits_check.py 200 > $O/hits_check.txt 2>&1; echo "hits check rc=$?"; grep -v "^pairs" $O/hits_check.txt | tail -4
timeout 900 python -m pytest tests/test_gpu_paired.py tests/test_zz_gpu_native_sam.py tests/test_zz_gpu_cigar.py tests/test_gpu_planes.py tests/test_zz_gpu_datatest.py -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
t() { tag=$1; lib=$2; shift 2; timeout 400 python scripts/ab_bench.py run $lib --workload paired --genome-mb 256 --no-extra-legs --skip-cpu "$@" > $O/$tag.out 2> $O/$tag.err
  python - <<P
import json
try:
    d=json.loads(open('$O/$tag.out').readline()); r=d['roofline']
    print('== $tag: %.0f reads/s, hipEvent launch avg %.1f ms, calls [%.0f .. %.0f .. %.0f] cycles/read %s %s' % (d['value'], r['avg_launch_ms'], r.get('launch_ms_min',0), r.get('launch_ms_median',0), r.get('launch_ms_max',0), r.get('wave_cycles_per_read'), r.get('wave_cycle_breakdown')))
except Exception as e: print('$tag failed', e)
P
}
export SNAPGPU_PAIRED_HELP_MIN=0
t base_d1 base --feeders 1 --steps 3 --warmup 1 --batches 3
t new_d1 new --feeders 1 --steps 3 --warmup 1 --batches 3
unset SNAPGPU_PAIRED_HELP_MIN
t base_f3 base --steps 6 --warmup 1
t new_f3 new --steps 6 --warmup 1
t pt2_f3 pt2 --steps 3 --warmup 1
for v in 1 0; do SNAPGPU_SAMF_DP8=$v timeout 200 python scripts/gpu_sam_perf.py 400000 > $O/sam_perf_dp8_$v.json 2> $O/sam_perf_$v.err; python -c "
import json; d=json.loads(open('$O/sam_perf_dp8_$v.json').readline()); print('sam_fields DP8=$v: kernel %.2f M reads/s (M), %.2f M (=/X)' % (d['sam_fields_M']['kernel_reads_per_s']/1e6, d['sam_fields_eqx']['kernel_reads_per_s']/1e6))"; done
timeout 600 python scripts/gpu_e2e_sam.py 20000000 --skip-reference --keep > $O/e2e.json 2> $O/e2e.err
python - <<P
import json
try:
    d=json.loads(open('$O/e2e.json').readline())
    for k,v in d.items():
        if isinstance(v,dict) and 'tool_tail' in v: print(k, v.get('reads_per_s_streaming'), v['records_hash'], v['tool_tail'][-1][:220])
except Exception as e: print('e2e failed', e)
P
