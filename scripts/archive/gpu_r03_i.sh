# Round 3, ninth hardware call: defaults after the feeder policy; what the helpers' evaluations are worth (diagnostic counters).
O=gpurun_out/${1:-r03i}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-200}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160)"; }
run f1 python bench.py --feeders 1 --steps 4 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
SNAPGPU_SINGLE_HELP_KEEP=1 run f1_keep1 python bench.py --feeders 1 --steps 4 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
run f3 python bench.py --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
