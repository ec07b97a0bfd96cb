# experiment: the single-end exact kernel built for 8 waves per SIMD (64 VGPRs) against the 6-wave build, three feeders
O=gpurun_out/${1:-r03k}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-200}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160)"; }
run f3_w6 python bench.py --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
cp snap_amd/libsnapgpu.so /tmp/libsnapgpu_w6.so; cp snap_amd/libsnapgpu_w8.so snap_amd/libsnapgpu.so
SNAPGPU_WAVES_PER_CU=32 run f3_w8 python bench.py --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
run f3_w8_grid24 python bench.py --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
SNAPGPU_WAVES_PER_CU=32 run f4_w8 python bench.py --feeders 4 --steps 8 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
cp /tmp/libsnapgpu_w6.so snap_amd/libsnapgpu.so
run f4_w6 python bench.py --feeders 4 --steps 8 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown
