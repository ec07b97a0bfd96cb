#!/bin/bash
# r06z: the closing run of round 6 on the committed build: the whole GPU suite, smoke, PMC passes of the single-end / paired / c5 legs at 3 100 Mb (-> profiles/pmc_latest.json),
# the driver's bench command, and the same command under rocprofv3 --kernel-trace --stats.   Usage: bash scripts/gpu_r06_z.sh <out-dir> [nosuite]
O=gpurun_out/${1:-r06z}; mkdir -p $O
ls -la --time-style=full-iso snap_amd/libsnapgpu.so snap_amd/snapgpu-sam > $O/libs.txt; python -c "import bench; print('kernel_source_hash', bench.kernel_source_hash())" >> $O/libs.txt 2>&1; cat $O/libs.txt
if [ "$2" != nosuite ]; then
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python scripts/pmc_collect.py $O/pmc_3100 --genome-mb 3100 > $O/pmc_3100.txt 2>&1; tail -c 300 $O/pmc_3100.txt; echo
timeout 900 python scripts/pmc_collect.py $O/pmc_paired_3100 --genome-mb 3100 --workload paired --steps 3 > $O/pmc_paired_3100.txt 2>&1; tail -c 300 $O/pmc_paired_3100.txt; echo
timeout 700 python scripts/pmc_collect.py $O/pmc_c5_3100 --genome-mb 3100 --workload paired --steps 3 --reads 200000 --tag c5 --timeout 200 -- --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002 > $O/pmc_c5_3100.txt 2>&1; tail -c 300 $O/pmc_c5_3100.txt; echo
python - $O <<'PY'
import json,sys,os
O=sys.argv[1]; es=[]
for d in ("pmc_3100","pmc_paired_3100","pmc_c5_3100"):
    f=os.path.join(O,d,"pmc_entry.json")
    if os.path.exists(f): es.append(json.load(open(f)))
if es:
    json.dump({"entries":es}, open("profiles/pmc_latest.json","w"), indent=1)
    json.dump({"entries":es}, open(os.path.join(O,"pmc_latest.json"),"w"), indent=1)
    print("== pmc_latest.json: %d entries %s, hash %s" % (len(es), [e.get("workload") for e in es], sorted({e.get("kernel_source_hash") for e in es})))
PY
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "driver cmd rc=$?" | tee $O/bench_driver_cmd.rc; grep "bench +" $O/bench_driver_cmd.err > $O/bench_driver_cmd.log
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-e2e-leg > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv 2>/dev/null; rm -rf $O/stats
python - $O <<'PY' | tee $O/summary.txt
import json,sys,os
O=sys.argv[1]
for f in ("bench_driver_cmd.json","bench_under_rocprof.json"):
    try:
        d=json.loads(open(os.path.join(O,f)).readline()); r=d["roofline"]; c=d["config"]
        print("== %s: %.0f reads/s (%s Mb), %.1f ms/step, cpu %.0f; bound %s %s; traffic %s; achieved %s frac %s" % (f, d["value"], c["genome_mb"], d["ms_per_step"], d["cpu_baseline"]["value"], r.get("bound"), r.get("bound_fractions"), r.get("traffic"), r.get("achieved"), r.get("frac")))
        print("   ", {k: c[k] for k in c if k.startswith(("parity","paired_","c5_","e2e_"))})
        for k in ("valu_per_read","salu_per_read","wave_cycle_breakdown","per_read","avg_launch_ms","launch_event_ms_median"): print("   ", k, r.get(k))
        if "e2e" in d: print("   e2e tail:", *d["e2e"].get("tool_tail", []), sep="\n      ")
    except Exception as e: print(f, "ERR", e)
try:
    import csv
    rows=list(csv.DictReader(open(os.path.join(O,"bench_kernel_stats.csv"))))
    for r in rows[:8]: print("   stats:", r.get("Name","")[:70], r.get("Calls"), r.get("AverageNs"), r.get("Percentage"))
except Exception as e: print("stats ERR", e)
PY
