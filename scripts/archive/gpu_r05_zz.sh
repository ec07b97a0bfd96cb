#!/bin/bash
# r05zz: after the closing run (r05z) the index builder learned the other key sizes and include/snapgpu.h's text about it changed -- the
# header is part of kernel_source_hash, so the counter passes are taken again on the final sources (same kernels, byte for byte), then
# the driver's bench command once more.  First the tests that are new since r05z, the bench test and smoke; last a counter pass of the c5 leg.
O=gpurun_out/${1:-r05zz}; mkdir -p $O
timeout 600 python -m pytest tests/test_zx_gpu_index_build.py tests/test_zzzz_gpu_bench.py tests/test_gpu_parity.py -m gpu -q --durations=5 --timeout 150 > $O/pytest_gpu_part.txt 2>&1; tail -8 $O/pytest_gpu_part.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python scripts/pmc_collect.py $O/pmc_3100 --genome-mb 3100 > $O/pmc_3100.txt 2>&1; tail -c 300 $O/pmc_3100.txt; echo
timeout 900 python scripts/pmc_collect.py $O/pmc_paired_3100 --genome-mb 3100 --workload paired --steps 3 > $O/pmc_paired_3100.txt 2>&1; tail -c 300 $O/pmc_paired_3100.txt; echo
merge() {
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
}
merge
timeout 600 python scripts/pmc_collect.py $O/pmc_c5_3100 --genome-mb 3100 --workload paired --steps 3 --reads 200000 --tag c5 --timeout 200 -- --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002 > $O/pmc_c5_3100.txt 2>&1; tail -c 300 $O/pmc_c5_3100.txt; echo
merge
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; grep "bench +" $O/bench_driver_cmd.err > $O/bench_driver_cmd.log
python - $O/bench_driver_cmd.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d["roofline"]
print("== bench: %.0f reads/s (%s Mb), %.1f ms/step, parity %s, cpu %.0f; bound %s %s; traffic %s; wall %.0f s" % (d["value"], d["config"]["genome_mb"], d["ms_per_step"],
      {k:d["parity_check"][k] for k in ("reads","mismatching_fields")}, d["cpu_baseline"]["value"], r.get("bound"), r.get("bound_fractions"), r.get("traffic"), d["bench_wall_s"]))
for leg in ("paired","c5"):
    if leg in d: print("   %s: %s reads/s, parity %s, cpu %s, traffic %s, bound %s" % (leg, d[leg].get("value"), d[leg].get("parity_check"), d[leg].get("cpu_baseline",{}).get("value"), d[leg].get("roofline",{}).get("traffic"), d[leg].get("roofline",{}).get("bound")) if "error" not in d[leg] else "   %s: %s" % (leg, d[leg]))
if "e2e" in d: print("   e2e:", {k:d["e2e"].get(k) for k in ("value","index_load_s","stream_s","identical_records","records_compared","speedup_vs_reference_cli_own_figure","error")})
PY
