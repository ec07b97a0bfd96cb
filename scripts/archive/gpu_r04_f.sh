#!/bin/bash
# r04f: kernel-trace stats of the paired bench over two builds (which kernel got slower)
O=gpurun_out/${1:-r04f}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in ${VARIANTS:-dup0 v3}; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/st_$v -o bench -- python scripts/ab_bench.py run $v --genome-mb 256 --workload paired --no-extra-legs --steps 3 --warmup 1 --skip-cpu > $O/p_$v.out 2> $O/p_$v.err < /dev/null
  echo "== $v: $(python -c "import json;d=json.loads(open('$O/p_$v.out').readline());print(d['value'], d['roofline']['avg_launch_ms'])")"
  head -6 $O/st_$v/bench_kernel_stats.csv | cut -c1-200
done
