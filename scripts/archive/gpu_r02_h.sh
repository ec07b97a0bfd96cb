O=gpurun_out/${1:-r02h}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-150} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-260}) $(grep -m1 -i fault $O/$tag.err | cut -c1-100)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
run t_fix python -m pytest tests/test_gpu_parity.py tests/test_gpu_paired.py tests/test_gpu_multi_ctx.py -m gpu -x -q -k "fixture or feeders or replica"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
# paired end: A = one context (the function-form kernel alone), B = two feeders
T=400 W=900 run paired_f1 python bench.py --workload paired --steps 2 --warmup 1 --feeders 1
T=400 W=500 run paired_f2 python bench.py --workload paired --steps 4 --warmup 1 --feeders 2 --skip-cpu
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --steps 3 --warmup 1 > $O/bench_stats.json 2> $O/bench_stats.err < /dev/null
tail -c 400 $O/bench_stats.json; head -4 $O/stats/bench_kernel_stats.csv
timeout 600 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
