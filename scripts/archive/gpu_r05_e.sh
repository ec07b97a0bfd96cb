#!/bin/bash
# round 5, session e: (1) the paired kernel, A/B made deterministic (one feeder, Phase-4 help off: the launch time is then the kernel's own),
# base / new alternating; (2) the SAM side with and without the eight-reads-per-wave row loops; (3) the GPU suite's SAM files; (4) FASTQ -> SAM
O=gpurun_out/r05e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
t() { tag=$1; lib=$2; shift 2; SNAPGPU_PAIRED_HELP_MIN=0 timeout 400 python scripts/ab_bench.py run $lib --workload paired --genome-mb 256 --feeders 1 --steps 3 --warmup 1 --batches 3 --no-extra-legs --skip-cpu "$@" > $O/$tag.out 2> $O/$tag.err
  python - <<P
import json
try:
    d=json.loads(open('$O/$tag.out').readline()); r=d['roofline']
    print('== $tag: %.0f reads/s, hipEvent launch avg %.1f ms, calls [%.0f .. %.0f .. %.0f]' % (d['value'], r['avg_launch_ms'], r.get('launch_ms_min',0), r.get('launch_ms_median',0), r.get('launch_ms_max',0)))
except Exception as e: print('$tag failed', e)
P
}
t base_1 base
t new_1 new
t base_2 base
t new_2 new
for v in 1 0; do SNAPGPU_SAMF_DP8=$v timeout 200 python scripts/gpu_sam_perf.py 400000 > $O/sam_perf_dp8_$v.json 2> $O/sam_perf_$v.err; python -c "
import json; d=json.loads(open('$O/sam_perf_dp8_$v.json').readline()); print('sam_fields DP8=$v: kernel %.2f M reads/s (M), %.2f M (=/X)' % (d['sam_fields_M']['kernel_reads_per_s']/1e6, d['sam_fields_eqx']['kernel_reads_per_s']/1e6))"; done
timeout 900 python -m pytest tests/test_zz_gpu_native_sam.py tests/test_zz_gpu_cigar.py -q > $O/pytest_sam.txt 2>&1; tail -4 $O/pytest_sam.txt
timeout 600 python scripts/gpu_e2e_sam.py 20000000 --skip-reference --keep > $O/e2e.json 2> $O/e2e.err
python - <<P
import json
try:
    d=json.loads(open('$O/e2e.json').readline())
    for k,v in d.items():
        if isinstance(v,dict) and 'tool_tail' in v: print(k, v.get('reads_per_s_streaming'), v['records_hash'], v['tool_tail'][-1][:220])
except Exception as e: print('e2e failed', e)
P
