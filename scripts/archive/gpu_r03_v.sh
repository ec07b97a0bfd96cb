#!/bin/bash
# r03v: the GRCh38-scale single-end run on the last build (heavy-first limited to one-feeder contexts), parity on a 50 000-read sample
O=gpurun_out/${1:-r03v}; mkdir -p $O
timeout 135 python bench.py --genome-mb 3100 --batches 2 --cpu-sample 50000 --skip-refwalk --skip-breakdown --skip-probe > $O/bench_3g.json 2> $O/bench_3g.err; tail -c 400 $O/bench_3g.json; tail -3 $O/bench_3g.err
