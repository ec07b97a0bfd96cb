#!/bin/bash
# round 5, session a: where the paired kernel's wave cycles go (phase-timer build `pt`), and how FASTQ -> SAM's GPU time splits (kernel trace)
O=gpurun_out/r05a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 400 python scripts/ab_bench.py run pt --workload paired --genome-mb 256 --steps 3 --warmup 1 --skip-cpu --no-extra-legs > $O/paired_pt.out 2> $O/paired_pt.err
python - <<P
import json
try:
    d=json.loads(open('$O/paired_pt.out').readline()); r=d['roofline']
    print('paired pt: %.0f reads/s; breakdown %s; cycles/read %.0f' % (d['value'], r.get('wave_cycle_breakdown'), r.get('wave_cycles_per_read',0)))
except Exception as e: print('paired pt failed', e)
P
timeout 500 python scripts/gpu_e2e_sam.py 20000000 --skip-reference --keep > $O/e2e_first.json 2> $O/e2e_first.err
D=${SNAP_BENCH_DIR:-/tmp/snap_bench}
IDX=$(ls -d $D/*256*/idx 2>/dev/null | head -1); FQ=$(ls $D/*/e2e.fq | head -1)
echo "idx=$IDX fq=$FQ"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/e2e_stats -o sam -- snap_amd/snapgpu-sam single $IDX $FQ -d 8 -o /tmp/e2e_prof.sam > $O/e2e_prof.txt 2>&1
head -8 $O/e2e_stats/*kernel_stats.csv 2>/dev/null | cut -c1-200
tail -3 $O/e2e_prof.txt
cat $O/e2e_first.json | cut -c1-1500
timeout 200 python scripts/gpu_sam_perf.py 400000 > $O/sam_perf.json 2> $O/sam_perf.err; cat $O/sam_perf.json
