# round 2, second hardware call: whole -m gpu suite (no -x), then the single-end bench under rocprofv3 (kernel stats), then the paired bench
O=gpurun_out/${1:-r02b}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --steps 3 --warmup 1 > $O/bench_stats.json 2> $O/bench_stats.err < /dev/null
tail -c 2500 $O/bench_stats.json; head -5 $O/stats/bench_kernel_stats.csv
timeout 600 python bench.py --workload paired --steps 2 --warmup 1 > $O/bench_paired.json 2> $O/bench_paired.err < /dev/null
tail -c 1500 $O/bench_paired.json
