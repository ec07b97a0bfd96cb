#!/bin/bash
# r04b: the new bench line (GRCh38-scale default + paired leg + stand-in leg) with the driver's command, bench.py's own GPU tests, PMC passes of
# HEAD's default configuration (three feeders) at 3100 Mb and 256 Mb, and the kernel-trace stats of the default run
O=gpurun_out/${1:-r04b}; mkdir -p $O
run() { tag=$1; shift; t0=$SECONDS; ( timeout ${T:-300} "$@" > $O/$tag.out 2> $O/$tag.err; echo "rc=$?" >> $O/$tag.out ); echo "wall=$((SECONDS-t0)) s" >> $O/$tag.err; echo "== $tag: $(tail -n 2 $O/$tag.out | tr '\n' ' ' | cut -c1-${W:-300}) $(grep -m1 -i 'fault\|error' $O/$tag.err | cut -c1-160) $(tail -n 1 $O/$tag.err)"; }
timeout 600 python -m pytest tests/test_zzzz_gpu_bench.py -m gpu -q -x > $O/pytest_bench.txt 2>&1; tail -5 $O/pytest_bench.txt
T=600 run bench_driver_cmd python bench.py --gpus 1 --steps 20 --warmup 5
timeout 900 python scripts/pmc_collect.py $O/pmc_3100 --genome-mb 3100 > $O/pmc_3100.txt 2>&1; tail -c 600 $O/pmc_3100.txt
timeout 600 python scripts/pmc_collect.py $O/pmc_256 --genome-mb 256 > $O/pmc_256.txt 2>&1; tail -c 600 $O/pmc_256.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/stats -o bench -- python bench.py --no-extra-legs --skip-cpu > $O/bench_stats.json 2> $O/bench_stats.err < /dev/null
head -5 $O/stats/bench_kernel_stats.csv 2>/dev/null || find $O/stats -name '*kernel_stats.csv' | head
