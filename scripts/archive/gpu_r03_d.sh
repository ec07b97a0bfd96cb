# Round 3, fourth hardware call: the suite on the current build, and what the wave priority of heavy units does to the launch times.
O=gpurun_out/${1:-r03d}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-260}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160)"; }
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
run single_f1 python bench.py --feeders 1 --steps 4 --skip-cpu --skip-probe --skip-refwalk
run single_f2 python bench.py --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
run single_f3 python bench.py --feeders 3 --steps 6 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
T=400 run paired_f1 python bench.py --workload paired --feeders 1 --steps 2 --skip-cpu
T=400 run paired_f3 python bench.py --workload paired --skip-cpu
