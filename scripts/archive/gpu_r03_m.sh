# the hybrid plane Landau-Vishkin (bytes for levels 0-2, planes from level 3) against the byte form: parity subset, then three feeders and the phase breakdown
O=gpurun_out/${1:-r03m}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-200}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160)"; }
SNAPGPU_LV_PLANES=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_zy_gpu_index_shapes.py tests/test_gpu_paired.py tests/test_gpu_secondary.py -m gpu -q > $O/pytest_planes.txt 2>&1; tail -2 $O/pytest_planes.txt
run f3_bytes python bench.py --skip-cpu --skip-probe --skip-refwalk
SNAPGPU_LV_PLANES=1 run f3_planes python bench.py --skip-cpu --skip-probe --skip-refwalk
SNAPGPU_LV_PLANES=1 run f1_planes python bench.py --feeders 1 --steps 4 --skip-probe --skip-refwalk --skip-breakdown --cpu-sample 200000
