O=gpurun_out/${1:-r02e}; mkdir -p $O
timeout 200 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err < /dev/null; tail -c 300 $O/bench.json
bash scripts/gpu_pmc_all.sh ${1:-r02e}/pmc
