O=gpurun_out/${1:-r03j}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-200}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160)"; }
SNAPGPU_PHASE_TIMERS=1 run f1_timed_help python bench.py --feeders 1 --steps 2 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
SNAPGPU_PHASE_TIMERS=1 SNAPGPU_SINGLE_HELP=0 run f1_timed_nohelp python bench.py --feeders 1 --steps 2 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
