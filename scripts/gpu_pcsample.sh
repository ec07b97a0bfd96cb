#!/bin/bash
# PC sampling of the single-end bench (rocprofv3 --pc-sampling-beta-enabled, stochastic): a histogram of sampled program counters, compacted on
# the box (the raw CSV can be hundreds of MB), analysed against the library's own disassembly by scripts/pcsample_report.py
O=gpurun_out/${1:-pcs}; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout ${T:-240} rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method ${METHOD:-stochastic} --pc-sampling-unit ${UNIT:-cycles} --pc-sampling-interval ${INTERVAL:-1048576} \
  --output-format csv -d /tmp/pcs_out -o pcs -- python bench.py --genome-mb 256 --no-extra-legs --steps 3 --warmup 1 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown "$@" > $O/bench.out 2> $O/bench.err < /dev/null
echo "rc=$?"; tail -c 300 $O/bench.out; ls -la /tmp/pcs_out/* | head; 
python - "$O" <<'PY'
import csv, glob, sys, collections, json, os
out = sys.argv[1]
files = glob.glob("/tmp/pcs_out/**/*pc_sampling*.csv", recursive=True)
print("files", files)
for f in files:
    rd = csv.reader(open(f, newline=""))
    hdr = next(rd)
    print(os.path.basename(f), hdr)
    idx = {h: i for i, h in enumerate(hdr)}
    hist = collections.Counter(); n = 0; first = []
    for row in rd:
        n += 1
        if len(first) < 5: first.append(row)
        key = tuple(row[idx[h]] for h in hdr if h in ("Instruction", "Instruction_Comment", "Code_Object_Id", "Code_Object_Offset", "Wave_Issued_Instruction", "Instruction_Type", "Stall_Reason") and h in idx)
        hist[key] += 1
    print("rows", n, "distinct", len(hist)); print(first[:3])
    keys = [h for h in hdr if h in ("Instruction", "Instruction_Comment", "Code_Object_Id", "Code_Object_Offset", "Wave_Issued_Instruction", "Instruction_Type", "Stall_Reason")]
    with open(os.path.join(out, os.path.basename(f).replace(".csv", "_hist.csv")), "w", newline="") as fo:
        w = csv.writer(fo); w.writerow(keys + ["samples"])
        for k, v in hist.most_common(): w.writerow(list(k) + [v])
PY
ls -la $O
