# Round-2 closing run: suite, smoke, both benches under rocprofv3, feeder sweep, PMC passes, C5 line.  Copy what should be judged from
# gpurun_out/<name>/ into profiles/<name>/.
O=gpurun_out/${1:-r02i}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-150} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-260}) $(grep -m1 -i fault $O/$tag.err | cut -c1-100)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
timeout 600 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --steps 3 --warmup 1 > $O/bench_stats.json 2> $O/bench_stats.err < /dev/null
tail -c 300 $O/bench_stats.json; echo; head -3 $O/stats/bench_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pstats -o bench -- python bench.py --workload paired --steps 4 --warmup 1 --feeders 2 > $O/bench_paired.json 2> $O/bench_paired.err < /dev/null
tail -c 600 $O/bench_paired.json; echo; head -3 $O/pstats/bench_kernel_stats.csv
T=200 W=160 run paired_f3 python bench.py --workload paired --steps 6 --warmup 1 --feeders 3 --skip-cpu
T=200 W=160 run paired_f4 python bench.py --workload paired --steps 8 --warmup 1 --feeders 4 --skip-cpu
T=120 W=160 run single_f2 python bench.py --steps 4 --warmup 1 --feeders 2 --skip-cpu --skip-probe
bash scripts/gpu_pmc_all.sh ${1:-r02i}/pmc > $O/pmc_all.out 2>&1; tail -c 1200 $O/pmc_all.out; echo
T=300 W=700 run c5 python bench.py --workload paired --read-len 250 --max-k 20 --steps 2 --warmup 1 --feeders 2 --cpu-sample 20000
