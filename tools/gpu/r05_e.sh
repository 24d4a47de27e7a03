#!/bin/bash
# round 5, call E: weight lo planes of inexact checkpoints (fp16 / fp32 values) on the device: kernel checks, tiny fixtures, true dims;
# the split / strict regression tests around them.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "lo_plane or split or gemv_wg" 2>&1 | tail -5 | tee gpurun_out/r05_e_pytest_kernels.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -s -k "inexact or split_mode or strict_mode or true_dims_split or true_dims_strict" 2>&1 | grep -v "^$" | tail -25 | cut -c1-400 | tee gpurun_out/r05_e_pytest_e2e.txt
