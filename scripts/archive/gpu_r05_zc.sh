#!/bin/bash
# r05zc: the counter pass of the c5 leg (configs[4] on one GPU) that r05zz's command line lost to argparse; merged into profiles/pmc_latest.json
# beside r05zz's two entries (same sources, same hash).
O=gpurun_out/${1:-r05zc}; mkdir -p $O
timeout 700 python scripts/pmc_collect.py $O/pmc_c5_3100 --genome-mb 3100 --workload paired --steps 3 --reads 200000 --tag c5 --timeout 170 --read-len 250 --max-k 20 --insert-mean 600 --insert-sd 80 --long-indel-frac 0.002 > $O/pmc_c5_3100.txt 2>&1; tail -c 600 $O/pmc_c5_3100.txt; echo
python - $O <<'PY'
import json,sys,os
O=sys.argv[1]
t=json.load(open("profiles/pmc_latest.json")); es=[e for e in t["entries"] if e.get("workload")!="c5"]
f=os.path.join(O,"pmc_c5_3100","pmc_entry.json")
if os.path.exists(f): es.append(json.load(open(f)))
json.dump({"entries":es}, open(os.path.join(O,"pmc_latest.json"),"w"), indent=1)
print("== %d entries %s, hashes %s" % (len(es), [e.get("workload") for e in es], sorted({e.get("kernel_source_hash") for e in es})))
PY
