# Round 3, fifth hardware call: the suite with the bit-plane Landau-Vishkin on hardware, and what it does to the launch times.
O=gpurun_out/${1:-r03e}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-260}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160)"; }
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
run single_f1 python bench.py --feeders 1 --steps 4 --skip-cpu --skip-probe --skip-refwalk
SNAPGPU_NO_PLANES=1 run single_f1_noplanes python bench.py --feeders 1 --steps 4 --skip-cpu --skip-probe --skip-refwalk
run single_f3 python bench.py --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
T=400 run paired_f3 python bench.py --workload paired --skip-cpu
