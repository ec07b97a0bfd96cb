#!/bin/bash
# r06q: kernel-trace TIMELINE of the paired leg (start / end of every launch): do the three feeders' kernels overlap?
O=gpurun_out/${1:-r06q}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_paired -o kt -- python bench.py --workload paired --steps 6 $COMMON > $O/paired.json 2> $O/paired.err
f=$(find /tmp/kt_paired -name "*kernel_trace.csv" | head -1); python - "$f" $O/paired_kernel_trace.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
keep=[r for r in rows if 'k_align_paired' in r.get('Kernel_Name','')]
w=csv.writer(open(sys.argv[2],'w'))
w.writerow(['kernel','queue','start_ns','end_ns'])
t0=min(int(r['Start_Timestamp']) for r in keep)
for r in keep: w.writerow([r['Kernel_Name'][:40], r.get('Queue_Id',''), int(r['Start_Timestamp'])-t0, int(r['End_Timestamp'])-t0])
print(len(keep),'paired launches')
PY
C5="--workload paired --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_single -o kt -- python bench.py --workload single --steps 12 $COMMON > $O/single.json 2> $O/single.err
f=$(find /tmp/kt_single -name "*kernel_trace.csv" | head -1); python - "$f" $O/single_kernel_trace.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
keep=[r for r in rows if 'k_align_single' in r.get('Kernel_Name','')]
w=csv.writer(open(sys.argv[2],'w'))
w.writerow(['kernel','queue','start_ns','end_ns'])
t0=min(int(r['Start_Timestamp']) for r in keep)
for r in keep: w.writerow([r['Kernel_Name'][:40], r.get('Queue_Id',''), int(r['Start_Timestamp'])-t0, int(r['End_Timestamp'])-t0])
print(len(keep),'single launches')
PY
head -3 $O/paired_kernel_trace.csv
