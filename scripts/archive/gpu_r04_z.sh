#!/bin/bash
# r04z: the round's closing run on the committed build: the GPU suite, smoke, the driver's bench command, PMC passes of the timed configuration at both
# genome sizes, kernel-trace stats of the default run, the N > 1 code path on one rank at GRCh38 scale, FASTQ -> SAM at 20 M reads with the reference CLI beside it
O=gpurun_out/${1:-r04z}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.txt 2>&1; tail -9 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; grep "bench +" $O/bench_driver_cmd.err > $O/bench_driver_cmd.log
python - $O/bench_driver_cmd.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d["roofline"]
print("== bench: %.0f reads/s (%s Mb), %.1f ms/step, parity %s, cpu %.0f; probe %.2f G/s frac %.3f; paired %s; g256 %s; wall %.0f s, rss %.1f GB" % (d["value"], d["config"]["genome_mb"], d["ms_per_step"],
      {k:d["parity_check"][k] for k in ("reads","mismatching_fields")}, d["cpu_baseline"]["value"], r["probe"]["lookups_per_s"]/1e9, r["probe"]["frac"],
      {k:d["paired"].get(k) for k in ("value","ms_per_step")} if "paired" in d else None, d.get("genome_256mb",{}).get("value"), d["bench_wall_s"], d.get("host_peak_rss_gb",0)))
if "paired" in d: print("   paired parity", d["paired"].get("parity_check"), "cpu", d["paired"].get("cpu_baseline",{}).get("value"))
PY
timeout 900 python scripts/pmc_collect.py $O/pmc_3100 --genome-mb 3100 > $O/pmc_3100.txt 2>&1; tail -c 300 $O/pmc_3100.txt
timeout 600 python scripts/pmc_collect.py $O/pmc_256 --genome-mb 256 > $O/pmc_256.txt 2>&1; tail -c 300 $O/pmc_256.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/stats -o bench -- python bench.py --no-extra-legs --skip-cpu > $O/bench_stats.json 2> $O/bench_stats.err < /dev/null
head -4 $O/stats/bench_kernel_stats.csv | cut -c1-160
SNAP_BENCH_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 6 --warmup 3 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown > $O/bench_forced_rccl_3100.json 2> $O/bench_forced_rccl_3100.err
python -c "import json;d=json.loads(open('$O/bench_forced_rccl_3100.json').readline());print('== forced RCCL path, one rank, %s Mb: %.0f reads/s, wall %.0f s, rank-0 host peak %.1f GB' % (d['config']['genome_mb'], d['value'], d['bench_wall_s'], d.get('host_peak_rss_gb',0)))" 2>&1 | tail -1
timeout 900 python scripts/gpu_e2e_sam.py 20000000 > $O/e2e_sam.json 2> $O/e2e_sam.err
python - $O/e2e_sam.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["snapgpu_sam"]; r=d.get("snap_aligner_reference",{})
print("== e2e 20 M: snapgpu-sam %.0f reads/s streaming (wall %.1f s), reference %s reads/s own figure (wall %.1f s), identical %s" % (s.get("reads_per_s_streaming",0), s["wall_s"], r.get("reads_per_s_own_figure"), r.get("wall_s",0), d.get("identical_records")))
PY
