#!/bin/bash
# r04zx: PMC passes of the closing build at the 256 Mb stand-in (the bench line's genome_256mb leg) and of its paired-end leg
O=gpurun_out/${1:-r04zx}; mkdir -p $O
timeout 150 python scripts/pmc_collect.py $O/pmc_256 --genome-mb 256 --timeout 60 > $O/pmc_256.txt 2>&1; tail -c 400 $O/pmc_256.txt; echo
timeout 200 python scripts/pmc_collect.py $O/pmc_paired_256 --genome-mb 256 --workload paired --steps 2 --timeout 90 > $O/pmc_paired_256.txt 2>&1; tail -c 600 $O/pmc_paired_256.txt
