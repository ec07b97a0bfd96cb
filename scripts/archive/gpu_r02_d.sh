# round 2: benches (single under rocprofv3 kernel stats, paired under rocprofv3 kernel stats), tight timeouts
O=gpurun_out/${1:-r02d}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 330 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --steps 3 --warmup 1 > $O/bench_stats.json 2> $O/bench_stats.err < /dev/null
tail -c 400 $O/bench_stats.json; head -4 $O/stats/bench_kernel_stats.csv
timeout 330 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pstats -o bench -- python bench.py --workload paired --steps 2 --warmup 1 > $O/bench_paired.json 2> $O/bench_paired.err < /dev/null
tail -c 400 $O/bench_paired.json; head -6 $O/pstats/bench_kernel_stats.csv
timeout 60 python -m pytest tests/test_gpu_multi_ctx.py -m gpu -x -q 2>&1 | tail -2
