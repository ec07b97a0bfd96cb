O=gpurun_out/${1:-r02a}; mkdir -p $O      # copy what should be judged into profiles/<same name>/
timeout 1200 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --steps 3 --warmup 1 > $O/bench_stats.json 2> $O/bench_stats.err < /dev/null
tail -c 900 $O/bench_stats.json; head -3 $O/stats/bench_kernel_stats.csv
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pstats -o bench -- python bench.py --workload paired --steps 2 --warmup 1 > $O/bench_paired.json 2> $O/bench_paired.err < /dev/null
tail -c 700 $O/bench_paired.json; head -3 $O/pstats/bench_kernel_stats.csv
# the SAM side: written when no GPU time was left; run this first thing in the next round
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/samstats -o sam -- python scripts/gpu_sam_perf.py > $O/sam_perf.json 2> $O/sam_perf.err < /dev/null
tail -c 600 $O/sam_perf.json; head -4 $O/samstats/sam_kernel_stats.csv
