# Round 3, seventh hardware call: the help for heavy single-end reads after the polling fix.
O=gpurun_out/${1:-r03g}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-260}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160)"; }
run single_f1_help python bench.py --feeders 1 --steps 4 --skip-probe --skip-refwalk --cpu-sample 200000
run single_f3_help python bench.py --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
run single_f2_help python bench.py --feeders 2 --steps 4 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
SNAPGPU_SINGLE_HELP=0 run single_f3_nohelp python bench.py --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
