#!/bin/bash
# r03s: paired end with three feeders, Phase-4 help on (default) and off -- the single-end help and the heavy-first dequeue both turned out
# to cost with several feeders; the paired help was only ever compared with one context
O=gpurun_out/${1:-r03s}; mkdir -p $O
timeout 300 python bench.py --workload paired --skip-cpu > $O/p_f3_help_on.json 2> $O/p_f3_help_on.err; tail -c 300 $O/p_f3_help_on.json
SNAPGPU_PAIRED_HELP_MIN=0 timeout 300 python bench.py --workload paired --skip-cpu > $O/p_f3_help_off.json 2> $O/p_f3_help_off.err; tail -c 300 $O/p_f3_help_off.json
