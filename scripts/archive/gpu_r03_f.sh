# Round 3, sixth hardware call: the help for heavy single-end reads on hardware (suite, parity on the bench batch, launch times).
O=gpurun_out/${1:-r03f}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-260}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160)"; }
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
SNAPGPU_LV_PLANES=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_zy_gpu_index_shapes.py "tests/test_gpu_paired.py::test_align_paired_vs_reference_live" -m gpu -q > $O/pytest_planes.txt 2>&1; tail -2 $O/pytest_planes.txt
run single_f1_help python bench.py --feeders 1 --steps 4 --skip-probe --skip-refwalk
SNAPGPU_SINGLE_HELP=0 run single_f1_nohelp python bench.py --feeders 1 --steps 4 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
SNAPGPU_SINGLE_HELP=0 SNAPGPU_NO_ALWAYS_EXACT=1 run single_f1_nohelp_fast python bench.py --feeders 1 --steps 4 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
run single_f3_help python bench.py --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
run single_f2_help python bench.py --feeders 2 --steps 4 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
