#!/bin/bash
# r06x: the kernel-trace timeline of three paired-end feeders on the closing build (1.5 shares per launch, one stream per context, eight hardware queues), 256 Mb
O=gpurun_out/${1:-r06x}; mkdir -p $O
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_x -o kt -- python bench.py --workload paired --steps 9 $COMMON > $O/paired_traced.json 2> $O/paired_traced.err
f=$(find /tmp/kt_x -name "*kernel_trace.csv" | head -1); python - "$f" $O/paired_kernel_trace.csv <<'PY' | tee $O/summary.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
keep=[r for r in rows if 'k_align_paired' in r.get('Kernel_Name','')]
w=csv.writer(open(sys.argv[2],'w')); w.writerow(['kernel','queue','start_ns','end_ns'])
t0=min(int(r['Start_Timestamp']) for r in keep)
ev=[]
for r in keep:
    s,e=int(r['Start_Timestamp'])-t0, int(r['End_Timestamp'])-t0
    w.writerow([r['Kernel_Name'][:40], r.get('Queue_Id',''), s, e])
    if e-s > 1_000_000: ev.append((s,1,'true>' in r['Kernel_Name'][:40])); ev.append((e,-1,'true>' in r['Kernel_Name'][:40]))
ev.sort()
# time with k kernels running (launches longer than 1 ms), over the span from the 4th start to the last end but 3
hist={}; cur=0; last=None
for t,d,_ in ev:
    if last is not None: hist[cur]=hist.get(cur,0)+(t-last)
    cur+=d; last=t
tot=sum(hist.values())
print("queues:", sorted({r.get('Queue_Id','') for r in keep}))
print("time with k paired kernels running:", {k: round(v/tot,3) for k,v in sorted(hist.items())})
PY
python -c "
import json; d=json.loads(open('$O/paired_traced.json').readline()); print('traced run: %.0f reads/s' % d['value'])" | tee -a $O/summary.txt
