#!/bin/bash
# r04x: s3 = s2 (profiles/r04w) without lds_seq()'s aperture test (it stopped the paired-end kernel: an empty backward half's pattern pointer is one
# byte below LDS address 0) + the next lazy-F round's gather issued before this round's verdict + the rare row paths marked unlikely.
# s4 = s3 + tagged traceback images (dev_common.h: bt_cell): the exact form's images are cleared once per fifteen reads instead of once per read.
# s5 = s4 + the traceback's diagonal runs taken at once (a ballot, population counts; only mismatches are visited one by one).
# First the paired-end path on a small genome (under a timeout: r04w's paired run never came back), then s2 against s3 against s4, then 4 feeders.
O=gpurun_out/${1:-r04x}; mkdir -p $O
SNAP_BENCH_DIR=/tmp/snap_bench_small timeout 150 python scripts/ab_bench.py run s5 --genome-mb 24 --reads 20000 --steps 2 --warmup 1 --batches 2 --workload paired --no-extra-legs --skip-probe --skip-refwalk --skip-breakdown --cpu-seconds 2 > $O/paired_small.out 2> $O/paired_small.err
PRC=$?; echo "== paired small: rc=$PRC"; python -c "import json;d=json.loads(open('$O/paired_small.out').readline());print('   %.0f reads/s, parity %s' % (d['value'], json.dumps(d.get('parity_check'))[:200]))" 2>&1 | tail -1
t() { tag=$1; lib=$2; shift 2; timeout 200 python scripts/ab_bench.py run $lib --genome-mb 256 --no-extra-legs --steps 12 --warmup 3 --skip-probe --skip-refwalk --skip-breakdown "$@" > $O/$tag.out 2> $O/$tag.err
  python -c "import json;d=json.loads(open('$O/$tag.out').readline());print('== $tag: %.0f reads/s, %.1f ms/step, launch %.1f ms, parity %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], json.dumps(d.get('parity_check'))[:160]))" 2>&1 | tail -1; }
t s2 s2 --skip-cpu
t s3 s3 --skip-cpu
t s4 s4 --skip-cpu
t s5 s5 --cpu-seconds 3
t s5_f4 s5 --skip-cpu --feeders 4
t s3b s3 --skip-cpu
t s4b s4 --skip-cpu
t s5b s5 --skip-cpu
[ $PRC -eq 0 ] && t s5_paired s5 --workload paired --steps 6 --warmup 2 --cpu-seconds 3
