#!/bin/bash
# r05zf: the whole GPU suite on the final commit of round 5.
O=gpurun_out/${1:-r05zf}; mkdir -p $O
timeout 450 python -m pytest tests -m gpu -q --durations=5 --timeout 150 > $O/pytest_gpu.txt 2>&1; tail -9 $O/pytest_gpu.txt
grep -E "^(FAILED|ERROR)" $O/pytest_gpu.txt | head -20
