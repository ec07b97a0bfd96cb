# round 2, third hardware call: benches only (single under rocprofv3, paired), after the SGPR-base traceback store and the register-form exact replay
O=gpurun_out/${1:-r02c}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --steps 3 --warmup 1 > $O/bench_stats.json 2> $O/bench_stats.err < /dev/null
tail -c 600 $O/bench_stats.json; head -4 $O/stats/bench_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pstats -o bench -- python bench.py --workload paired --steps 2 --warmup 1 > $O/bench_paired.json 2> $O/bench_paired.err < /dev/null
tail -c 600 $O/bench_paired.json; head -6 $O/pstats/bench_kernel_stats.csv
