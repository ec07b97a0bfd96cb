# round 2, first hardware call: the whole -m gpu suite WITHOUT -x (a failure must not hide the rest), smoke, first SAM-side timing
O=gpurun_out/${1:-r02a}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/samstats -o sam -- python scripts/gpu_sam_perf.py > $O/sam_perf.json 2> $O/sam_perf.err < /dev/null
tail -c 900 $O/sam_perf.json; head -8 $O/samstats/sam_kernel_stats.csv
