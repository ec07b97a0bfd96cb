O=gpurun_out/${1:-r02j}; mkdir -p $O
timeout 200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?"; tail -c 700 $O/bench_default.json; echo
timeout 110 python scripts/gpu_e2e_sam.py 500000 > $O/e2e_sam.json 2> $O/e2e_sam.err; echo "rc=$?"; tail -c 1500 $O/e2e_sam.json; tail -3 $O/e2e_sam.err
