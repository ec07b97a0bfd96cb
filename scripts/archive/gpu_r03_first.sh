# First hardware call of the next round (~7 GPU-minutes): the measurements round 2 ran out of minutes for.  Each step under its own timeout;
# outputs under gpurun_out/<name>/ (copy what should be judged into profiles/<name>/).
#   1. the on-demand Phase-4 help on the real paired bench batch (256 Mb / 500 k pairs), off vs on, one context and three feeders
#   2. the paired exact kernel as the main pass on the same batch, function-form build (faulted in the inlined build, profiles/r02g)
#   3. PMC passes of the paired kernel (never collected)
O=gpurun_out/${1:-r03a}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-200} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-220}) $(grep -m1 -i fault $O/$tag.err | cut -c1-100)"; }
T=260 run p_f1_help_off python bench.py --workload paired --steps 2 --warmup 1 --feeders 1 --skip-cpu
SNAPGPU_PAIRED_HELP_MIN=64 run p_f1_help_on python bench.py --workload paired --steps 2 --warmup 1 --feeders 1 --skip-cpu
SNAPGPU_PAIRED_HELP_MIN=64 T=260 W=900 run p_f3_help_on python bench.py --workload paired --steps 6 --warmup 1 --feeders 3
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d $O/ppmc_$i -o bench -- python bench.py --workload paired --steps 1 --warmup 0 --feeders 1 --skip-cpu > $O/ppmc_$i.json 2> $O/ppmc_$i.err < /dev/null
done
python - <<PY
import csv, glob, json, collections
tot = collections.defaultdict(float)
for f in glob.glob("$O/ppmc_*/bench_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_align_paired" in r["Kernel_Name"]:
            tot[("exact:" if "true>" in r["Kernel_Name"] else "fast:") + r["Counter_Name"]] += float(r["Counter_Value"])
json.dump(dict(tot), open("$O/paired_pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(dict(tot), sort_keys=True))
PY
# last: the run that faulted in round 2 (a fault aborts the process; nothing after it depends on the GPU)
SNAPGPU_PAIRED_ALWAYS_EXACT=1 AMD_LOG_LEVEL=1 T=150 run p_f1_always_exact python bench.py --workload paired --steps 2 --warmup 1 --feeders 1 --skip-cpu
