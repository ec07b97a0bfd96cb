#!/bin/bash
# r04zz: the closing run again, on the build with the round's last kernel changes (profiles/r04v - r04x): the GPU suite, smoke, the driver's bench
# command, PMC passes of the timed configuration (3 100 Mb), kernel-trace stats of the default run.  (FASTQ -> SAM: scripts/gpu_r04_zy.sh)
O=gpurun_out/${1:-r04zz}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --durations=5 --timeout 120 > $O/pytest_gpu.txt 2>&1; tail -9 $O/pytest_gpu.txt
if tail -1 $O/pytest_gpu.txt | grep -Eq "failed|error|Timeout"; then echo "== GPU suite not green: stopping here"; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.txt | head -20; exit 1; fi
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python scripts/pmc_collect.py $O/pmc_3100 --genome-mb 3100 > $O/pmc_3100.txt 2>&1; tail -c 300 $O/pmc_3100.txt
python - $O <<'PY'
import json,sys,os
O=sys.argv[1]; es=[]
for d in ("pmc_3100","pmc_256"):
    f=os.path.join(O,d,"pmc_entry.json")
    if os.path.exists(f): es.append(json.load(open(f)))
if es:
    json.dump({"entries":es}, open("profiles/pmc_latest.json","w"), indent=1)
    json.dump({"entries":es}, open(os.path.join(O,"pmc_latest.json"),"w"), indent=1)
    print("== pmc_latest.json: %d entries, hash %s" % (len(es), es[0].get("kernel_source_hash")))
PY
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; grep "bench +" $O/bench_driver_cmd.err > $O/bench_driver_cmd.log
python - $O/bench_driver_cmd.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d["roofline"]
print("== bench: %.0f reads/s (%s Mb), %.1f ms/step, parity %s, cpu %.0f; probe %.2f G/s frac %.3f; paired %s; g256 %s; wall %.0f s, rss %.1f GB" % (d["value"], d["config"]["genome_mb"], d["ms_per_step"],
      {k:d["parity_check"][k] for k in ("reads","mismatching_fields")}, d["cpu_baseline"]["value"], r["probe"]["lookups_per_s"]/1e9, r["probe"]["frac"],
      {k:d["paired"].get(k) for k in ("value","ms_per_step")} if "paired" in d else None, d.get("genome_256mb",{}).get("value"), d["bench_wall_s"], d.get("host_peak_rss_gb",0)))
if "paired" in d: print("   paired parity", d["paired"].get("parity_check"), "cpu", d["paired"].get("cpu_baseline",{}).get("value"))
PY
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/stats -o bench -- python bench.py --no-extra-legs --skip-cpu > $O/bench_stats.json 2> $O/bench_stats.err < /dev/null
head -4 $O/stats/bench_kernel_stats.csv | cut -c1-160
