#!/bin/bash
# r06r: each feeder's paired-end grid sized to a THIRD of the chip (SNAPGPU_PAIRED_WAVES_PER_CU=4: three feeders' main kernels are then all resident at once and the small
# second-pass / exact-replay launches find free slots in the tails instead of waiting behind the other feeders' pending blocks -- the timeline of profiles/r06q)
O=gpurun_out/${1:-r06r}; mkdir -p $O
COMMON="--genome-mb 256 --skip-cpu --skip-refwalk --skip-breakdown --skip-probe --no-extra-legs --warmup 1"
C5="--workload paired --reads 200000 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002"
for rep in 1 2; do
  timeout 600 python bench.py --workload paired --steps 6 $COMMON > $O/paired_wpc12_$rep.json 2> $O/paired_wpc12_$rep.err
  SNAPGPU_PAIRED_WAVES_PER_CU=4 timeout 600 python bench.py --workload paired --steps 6 $COMMON > $O/paired_wpc4_$rep.json 2> $O/paired_wpc4_$rep.err
  SNAPGPU_PAIRED_WAVES_PER_CU=4 timeout 600 python bench.py --workload paired --steps 6 --feeders 4 $COMMON > $O/paired_wpc4_f4_$rep.json 2> $O/paired_wpc4_f4_$rep.err
  timeout 600 python bench.py $C5 --steps 6 $COMMON > $O/c5_wpc12_$rep.json 2> $O/c5_wpc12_$rep.err
  SNAPGPU_PAIRED_WAVES_PER_CU=4 timeout 600 python bench.py $C5 --steps 6 $COMMON > $O/c5_wpc4_$rep.json 2> $O/c5_wpc4_$rep.err
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SNAPGPU_PAIRED_WAVES_PER_CU=4 timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_paired4 -o kt -- python bench.py --workload paired --steps 6 $COMMON > $O/paired_wpc4_traced.json 2> $O/paired_wpc4_traced.err
f=$(find /tmp/kt_paired4 -name "*kernel_trace.csv" | head -1); python - "$f" $O/paired_wpc4_kernel_trace.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
keep=[r for r in rows if 'k_align_paired' in r.get('Kernel_Name','')]
w=csv.writer(open(sys.argv[2],'w')); w.writerow(['kernel','queue','start_ns','end_ns'])
t0=min(int(r['Start_Timestamp']) for r in keep)
for r in keep: w.writerow([r['Kernel_Name'][:40], r.get('Queue_Id',''), int(r['Start_Timestamp'])-t0, int(r['End_Timestamp'])-t0])
PY
python - $O <<'PY' | tee $O/summary.txt
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        d=json.loads(open(f).readline())
        print("%-28s %9.0f reads/s  ms/step %7.1f  feeders %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("feeders_per_gpu")))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
