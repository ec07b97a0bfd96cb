#!/bin/bash
# What to run first next round (nothing here has been measured on the closing build of round 4, device sources ea4d2638ea359a6f):
#   build the variants first, in the build container:
#     python scripts/ab_bench.py build base
#     python scripts/ab_bench.py build w5 '-DSNAPGPU_WAVES_PER_SIMD(AGC)=5'      # 96 VGPRs: the main kernel's 162 spilled VGPRs against 5 waves per SIMD
#     python scripts/ab_bench.py build lvd -DSNAPGPU_LV_DUMMY=1                  # (to be written: 32 dummy scalar instructions per Landau-Vishkin level, as SNAPGPU_AG_DUMMY did for the row)
# 1. feeders with the faster kernel at GRCh38 scale (four were slower than three at 256 Mb: profiles/r04x)
# 2. 5 waves per SIMD again (r04s: -1.5 % with the old row loop; the kernel is more latency-bound now, so probably still worse -- but the spills were 229 then)
# 3. FASTQ -> SAM under the kernel tracer: how the 8 s of GPU time split between k_align_single, k_sam_fields and idle (Next 2)
O=gpurun_out/${1:-r05a}; mkdir -p $O
t() { tag=$1; lib=$2; shift 2; timeout 300 python scripts/ab_bench.py run $lib --no-extra-legs --steps 12 --warmup 3 --skip-cpu --skip-probe --skip-refwalk --skip-breakdown "$@" > $O/$tag.out 2> $O/$tag.err
  python -c "import json;d=json.loads(open('$O/$tag.out').readline());print('== $tag: %.0f reads/s (%s Mb), %.1f ms/step, launch %.1f ms' % (d['value'], d['config']['genome_mb'], d['ms_per_step'], d['roofline']['avg_launch_ms']))" 2>&1 | tail -1; }
t f3 base
t f2 base --feeders 2
t f4 base --feeders 4
SNAPGPU_WAVES_PER_CU=20 t w5 w5
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python scripts/gpu_e2e_sam.py 20000000 --skip-reference --keep > $O/e2e_first.json 2> $O/e2e_first.err      # leaves the FASTQ and the index under $SNAP_BENCH_DIR
D=${SNAP_BENCH_DIR:-/tmp/snap_bench}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$O/e2e_stats -o sam -- snap_amd/snapgpu-sam single $(ls -d $D/*256*/index* 2>/dev/null | head -1) $(ls $D/*/e2e.fq | head -1) -d 8 -o /tmp/e2e_prof.sam > $O/e2e_prof.txt 2>&1
head -8 $O/e2e_stats/*kernel_stats.csv 2>/dev/null | cut -c1-180
