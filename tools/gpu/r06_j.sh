#!/bin/bash
# round 6, call J: the fp8 format at full 13b depth against the oracle's fp8 mode on the effective weights (measurement first)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python tools/experiments/fp8_full_depth_oracle.py 2>&1 | grep -v amdgpu | tee gpurun_out/r06_j_fp8_full_depth_vs_oracle.txt
