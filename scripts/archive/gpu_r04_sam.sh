#!/bin/bash
# r04sam: the SAM side after cigar_ag.h's row bytes moved to LDS and its traceback fetches 64 cells per load -- no Python on the box:
# snap_amd/snapgpu-sam (this build) and snap_amd/ab/old/snapgpu-sam (the build before) over an index and FASTQ files made in the build
# container by scripts/make_sam_check.py (gpurun_in/sam_check: a 4 Mb genome indexed by the reference's indexer, 100 000 reads, 10 000 pairs), records compared with the
# reference CLI's (-t 1: its order is the input's; md5 of every line but @PG, computed in the build container), then both builds timed on
# the reads SAM_CAT times over (default 20: 2 M reads; 200: 20 M).
T=${SAM_TOOL:-snap_amd/snapgpu-sam}; OLD=${SAM_TOOL_OLD:-snap_amd/ab/old/snapgpu-sam}
I=gpurun_in/sam_check; O=gpurun_out/${1:-r04sam}; mkdir -p $O; W=${TMPDIR:-/tmp}/sam_check_work; mkdir -p $W
ok=1
if [ -z "$SAM_SKIP_CHECK" ]; then
timeout 100 $T single $I/index $I/single.fq -d 8 -o $W/s.sam > $O/single.txt 2>&1 || ok=0
got=$(grep -v '^@PG' $W/s.sam | md5sum | cut -c1-32); want=$(cat $I/single.md5)
echo "== single: $(grep -vc '^@' $W/s.sam) records, md5 $got, reference $want: $([ "$got" = "$want" ] && echo IDENTICAL || echo DIFFERENT)"; [ "$got" = "$want" ] || ok=0
timeout 100 $T paired $I/index $I/r1.fq $I/r2.fq -d 8 -o $W/p.sam > $O/paired.txt 2>&1 || ok=0
got=$(grep -v '^@PG' $W/p.sam | md5sum | cut -c1-32); want=$(cat $I/paired.md5)
echo "== paired: $(grep -vc '^@' $W/p.sam) records, md5 $got, reference $want: $([ "$got" = "$want" ] && echo IDENTICAL || echo DIFFERENT)"; [ "$got" = "$want" ] || ok=0
echo "== correctness: ok=$ok"
fi
[ -n "$SAM_CHECK_ONLY" ] && exit 0
N=${SAM_CAT:-20}; k=0; while [ $k -lt $N ]; do cat $I/single.fq; k=$((k+1)); done > $W/big.fq
for rep in ${SAM_REPS:-1 2}; do
  for which in ${SAM_WHICH:-new old}; do
    tool=$T; [ $which = old ] && tool=$OLD; [ $which = mid ] && tool=snap_amd/ab/mid/snapgpu-sam
    SNAPGPU_SAM_VERBOSE=1 timeout 100 $tool single $I/index $W/big.fq -d 8 -o $W/big.sam > $O/time_${which}_$rep.txt 2>&1
    echo "== $which ($rep): $(grep -o 'FASTQ -> SAM in .*reads/s' $O/time_${which}_$rep.txt) | $(grep -o 'feeders: .*' $O/time_${which}_$rep.txt | cut -c1-110)"
  done
done
