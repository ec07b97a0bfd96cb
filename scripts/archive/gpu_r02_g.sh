O=gpurun_out/${1:-r02g}; mkdir -p $O
run() { tag=$1; shift; ( timeout 100 "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
run t_fix python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py -m gpu -x -q -k "fixture"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --steps 3 --warmup 1 > $O/bench_stats.json 2> $O/bench_stats.err < /dev/null
tail -c 300 $O/bench_stats.json; head -4 $O/stats/bench_kernel_stats.csv
SNAPGPU_PAIRED_HELP_MIN=0 timeout 300 python bench.py --workload paired --steps 2 --warmup 1 > $O/bench_paired.json 2> $O/bench_paired.err < /dev/null
tail -c 500 $O/bench_paired.json
