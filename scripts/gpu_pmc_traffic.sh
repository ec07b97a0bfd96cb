# HBM traffic of one k_align_single launch: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (MI355X_MICROARCH.md),
# counters only (no trace domains).  Usage: bash scripts/gpu_pmc_traffic.sh <out-dir-under-gpurun_out>
O=gpurun_out/${1:-pmc}; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 100 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o bench -- python bench.py --steps 1 --warmup 0 --skip-cpu > $O/pmc_$c.json 2> $O/pmc_$c.err < /dev/null
  python - <<PY
import csv
tot = 0.0; n = 0
for r in csv.DictReader(open("$O/pmc_$c/bench_counter_collection.csv")):
    if "k_align_single" in r["Kernel_Name"] and r["Counter_Name"] == "$c":
        tot += float(r["Counter_Value"]); n += 1
print("$c", tot, "launches", n)
PY
done
