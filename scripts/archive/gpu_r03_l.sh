# experiment: the paired-end kernel built for 3 waves per SIMD (168 VGPRs, 12 waves per CU) against the 2-wave build
O=gpurun_out/${1:-r03l}; mkdir -p $O
run() { tag=$1; shift; ( timeout ${T:-400} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ) ; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-200}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160)"; }
run p_f1_w2 python bench.py --workload paired --feeders 1 --steps 2 --skip-cpu
cp snap_amd/libsnapgpu.so /tmp/libsnapgpu_keep.so; cp snap_amd/libsnapgpu_pw3.so snap_amd/libsnapgpu.so
run p_f1_w3 python bench.py --workload paired --feeders 1 --steps 2 --skip-cpu
run p_f3_w3 python bench.py --workload paired --skip-cpu
run p_f2_w3 python bench.py --workload paired --feeders 2 --steps 4 --skip-cpu
cp /tmp/libsnapgpu_keep.so snap_amd/libsnapgpu.so
