# Round 3, second hardware call: the GPU suite incl. the index builder on hardware, the default benches on the new set-up (GPU-built
# index, rotating batches, heavy-first dequeue, help on), the heavy-first A/B with one context, and a first 1 Gb genome.
O=gpurun_out/${1:-r03b}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-400}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160)"; }
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
W=1500 run bench_default python bench.py
grep "index build\|GPU index" $O/bench_default.err
run single_f1_heavy_on python bench.py --feeders 1 --steps 4 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
SNAPGPU_SINGLE_HEAVY_FIRST=0 run single_f1_heavy_off python bench.py --feeders 1 --steps 4 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
W=1200 T=400 run paired_default python bench.py --workload paired
SNAPGPU_PAIRED_HEAVY_FIRST=1 run paired_f1_heavy_first python bench.py --workload paired --feeders 1 --steps 2 --skip-cpu
W=1500 T=600 run bench_1g python bench.py --genome-mb 1000 --cpu-sample 100000
grep "index build\|GPU index\|generated" $O/bench_1g.err
