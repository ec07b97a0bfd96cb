# PMC passes over one launch of the single-end bench (counters only, no trace domains; each group in its own rocprofv3 pass as the slots
# allow: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Usage: bash scripts/gpu_pmc_all.sh <out-dir-under-gpurun_out> [extra bench args]
O=gpurun_out/${1:-pmc}; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$i -o bench -- python bench.py --steps 1 --warmup 0 --skip-cpu --skip-probe "$@" > $O/pmc_$i.json 2> $O/pmc_$i.err < /dev/null
done
python - <<PY
import csv, glob, json, collections
tot = collections.defaultdict(float)
for f in glob.glob("$O/pmc_*/bench_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = "main" if "true>" not in r["Kernel_Name"] else "exact"
        if "k_align_" in r["Kernel_Name"]:
            tot[k + ":" + r["Counter_Name"]] += float(r["Counter_Value"])
json.dump(dict(tot), open("$O/pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(dict(tot), sort_keys=True))
PY
