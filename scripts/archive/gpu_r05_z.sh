#!/bin/bash
# r05z: the closing run of round 5 on the committed build: the GPU suite, smoke, PMC passes of the two timed configurations at 3 100 Mb
# (single end and paired end: four counter-only rocprofv3 passes each), the driver's bench command (with its paired / c5 / e2e legs),
# kernel-trace stats of the single-end and of the paired-end run.
O=gpurun_out/${1:-r05z}; mkdir -p $O
timeout 1300 python -m pytest tests -m gpu -q --durations=5 --timeout 150 > $O/pytest_gpu.txt 2>&1; tail -9 $O/pytest_gpu.txt
if tail -1 $O/pytest_gpu.txt | grep -Eq "failed|error|Timeout"; then echo "== GPU suite not green"; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.txt | head -20; [ "$2" = "go-on" ] || exit 1; fi
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python scripts/pmc_collect.py $O/pmc_3100 --genome-mb 3100 > $O/pmc_3100.txt 2>&1; tail -c 300 $O/pmc_3100.txt; echo
timeout 900 python scripts/pmc_collect.py $O/pmc_paired_3100 --genome-mb 3100 --workload paired --steps 3 > $O/pmc_paired_3100.txt 2>&1; tail -c 300 $O/pmc_paired_3100.txt; echo
python - $O <<'PY'
import json,sys,os
O=sys.argv[1]; es=[]
for d in ("pmc_3100","pmc_paired_3100"):
    f=os.path.join(O,d,"pmc_entry.json")
    if os.path.exists(f): es.append(json.load(open(f)))
if es:
    json.dump({"entries":es}, open("profiles/pmc_latest.json","w"), indent=1)
    json.dump({"entries":es}, open(os.path.join(O,"pmc_latest.json"),"w"), indent=1)
    print("== pmc_latest.json: %d entries, hash %s" % (len(es), es[0].get("kernel_source_hash")))
PY
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; grep "bench +" $O/bench_driver_cmd.err > $O/bench_driver_cmd.log
python - $O/bench_driver_cmd.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d["roofline"]
print("== bench: %.0f reads/s (%s Mb), %.1f ms/step, parity %s, cpu %.0f; bound %s %s; probe frac %s; residency %s; launch ms [%s %s %s]; wall %.0f s, rss %.1f GB" % (d["value"], d["config"]["genome_mb"], d["ms_per_step"],
      {k:d["parity_check"][k] for k in ("reads","mismatching_fields")}, d["cpu_baseline"]["value"], r.get("bound"), r.get("bound_fractions"), r.get("probe_frac"), r.get("mean_wave_residency"),
      r.get("launch_ms_min"), r.get("launch_ms_median"), r.get("launch_ms_max"), d["bench_wall_s"], d.get("host_peak_rss_gb",0)))
for leg in ("paired","c5"):
    if leg in d: print("   %s: %s reads/s, %s ms/step, parity %s, cpu %s, traffic %s, bound %s" % (leg, d[leg].get("value"), d[leg].get("ms_per_step"), d[leg].get("parity_check"), d[leg].get("cpu_baseline",{}).get("value"), d[leg].get("roofline",{}).get("traffic"), d[leg].get("roofline",{}).get("bound")) if "error" not in d[leg] else "   %s: %s" % (leg, d[leg]))
if "e2e" in d: print("   e2e:", {k:d["e2e"].get(k) for k in ("value","index_load_s","stream_s","identical_records","records_compared","speedup_vs_reference_cli_own_figure","error")}, d["e2e"].get("reference_cli"))
PY
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/stats -o bench -- python bench.py --no-extra-legs --skip-cpu > $O/bench_stats.json 2> $O/bench_stats.err < /dev/null
head -4 $O/stats/bench_kernel_stats.csv | cut -c1-160
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/stats_paired -o bench -- python bench.py --workload paired --steps 6 --warmup 1 --no-extra-legs --skip-cpu > $O/bench_paired_stats.json 2> $O/bench_paired_stats.err < /dev/null
head -4 $O/stats_paired/bench_kernel_stats.csv | cut -c1-160
