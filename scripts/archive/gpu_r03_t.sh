#!/bin/bash
# r03t: kernel stats of the default bench on the build the round ends on (heavy-first limited to one-feeder contexts)
O=gpurun_out/${1:-r03t}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --skip-cpu > $O/bench_default.json 2> $O/bench_default.err < /dev/null
tail -c 300 $O/bench_default.json; head -5 $O/stats/bench_kernel_stats.csv
