# Round 3, closing run: the suite, smoke, the default benches under rocprofv3 --kernel-trace --stats, one-context runs, PMC passes of the
# dominant kernel and of the probe kernel (separate passes, counters only), C5 as SURVEY 8(d) defines it, and the GRCh38-scale run.
O=gpurun_out/${1:-r03z}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-200}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160)"; }
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null
tail -c 600 $O/bench_default.json; head -4 $O/stats/bench_kernel_stats.csv
run single_f1 python bench.py --feeders 1 --steps 4 --skip-cpu --skip-probe
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pstats -o bench -- python bench.py --workload paired > $O/bench_paired.json 2> $O/bench_paired.err < /dev/null
tail -c 500 $O/bench_paired.json; head -4 $O/pstats/bench_kernel_stats.csv
run paired_f1 python bench.py --workload paired --feeders 1 --steps 2 --skip-cpu
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  SNAPGPU_SINGLE_HELP=0 timeout 150 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$i -o bench -- python bench.py --steps 1 --warmup 0 --feeders 1 --skip-cpu --skip-refwalk --skip-breakdown > $O/pmc_$i.json 2> $O/pmc_$i.err < /dev/null
done
python - <<PY
import csv, glob, json, collections
tot = collections.defaultdict(float); calls = collections.defaultdict(int)
for f in glob.glob("$O/pmc_*/bench_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        k = "align" if "k_align_single" in kn else "probe" if "k_lookup_seeds" in kn else None
        if k:
            tot[k + ":" + r["Counter_Name"]] += float(r["Counter_Value"]); calls[k + ":" + r["Counter_Name"]] += 1
out = {k: {"sum": v, "dispatch_rows": calls[k]} for k, v in tot.items()}
json.dump(out, open("$O/pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, sort_keys=True)[:600])
PY
W=300 T=500 run c5 python bench.py --workload paired --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002 --feeders 2 --steps 2 --warmup 1 --batches 2 --cpu-sample 20000
W=300 T=600 run bench_3g python bench.py --genome-mb 3100 --cpu-sample 100000 --batches 3
