#!/bin/bash
# r05zd: the tests written after r05zz (snapgpu_align_sam_single against the reference CLI's fixture, through the ABI directly), on the hardware.
O=gpurun_out/${1:-r05zd}; mkdir -p $O
timeout 400 python -m pytest tests/test_zz_gpu_cigar.py -m gpu -q --durations=5 --timeout 150 > $O/pytest_cigar.txt 2>&1; tail -9 $O/pytest_cigar.txt
