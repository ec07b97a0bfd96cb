# Round 3, third hardware call: the kernels touched since r03b on hardware (probe kernel, paired-end LV split / 4 waves per SIMD), the
# single-end launch profile (where the tail is), PMC passes, and the GRCh38-scale run.
O=gpurun_out/${1:-r03c}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-300}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160)"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_secondary.py tests/test_zy_gpu_index_shapes.py -m gpu -q > $O/pytest_subset.txt 2>&1; tail -3 $O/pytest_subset.txt
W=200 run bench_default python bench.py
W=200 T=400 run paired_f1_w16 python bench.py --workload paired --feeders 1 --steps 2 --skip-cpu
SNAPGPU_PAIRED_WAVES_PER_CU=8 W=200 T=400 run paired_f1_w8 python bench.py --workload paired --feeders 1 --steps 2 --skip-cpu
W=200 T=400 run paired_f3_w16 python bench.py --workload paired
W=200 T=400 run paired_f2_w16 python bench.py --workload paired --feeders 2 --steps 4 --skip-cpu
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$i -o bench -- python bench.py --steps 1 --warmup 0 --feeders 1 --skip-cpu --skip-refwalk --skip-breakdown > $O/pmc_$i.json 2> $O/pmc_$i.err < /dev/null
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
print(json.dumps(out, sort_keys=True))
PY
W=400 T=900 run bench_3g python bench.py --genome-mb 3100 --cpu-sample 100000 --batches 2
grep "index build\|GPU index\|generated\|resident" $O/bench_3g.err
W=300 T=600 run paired_3g python bench.py --workload paired --genome-mb 3100 --feeders 2 --steps 4 --batches 2 --cpu-sample 20000
