#!/bin/bash
# r04d: where a window-form affine-gap call spends its time: the same build with its prologue (dup1), its row loop (dup2) or its traceback (dup3)
# run twice -- results unchanged, the difference in time per batch is what that part costs
VARIANTS="dup0 dup1 dup2 dup3" PVARIANTS="" bash scripts/gpu_r04_c.sh ${1:-r04d}
