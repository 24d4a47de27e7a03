#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -s -k "fp16_operand" 2>&1 | grep -B2 -A12 "Error\|assert" | head -60 | cut -c1-300 | tee gpurun_out/r06_h_pytest_kernels_fail.txt
