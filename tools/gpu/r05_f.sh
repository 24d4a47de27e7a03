#!/bin/bash
# round 5, call F: the fp8 format on the massive-activation profile (per-layer, 13b geometry), fp8 + e4m3 KV vs fp8 + bf16 KV at
# full depth (the number ADVICE r4 asked for), the lo-plane kernel test after its fix.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "lo_plane" 2>&1 | tail -3 | tee gpurun_out/r05_f_pytest_kernels.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -s -k "massive or fp8_formats_per_layer or inexact_checkpoint" 2>&1 | grep -v "^$" | tail -14 | cut -c1-600 | tee gpurun_out/r05_f_pytest_fp8_outliers.txt
timeout 900 python -m pytest tests/test_gpu_fulldepth.py -q -x -m gpu -s -k "fp8_formats_vs_bf16" 2>&1 | grep -v "^$" | tail -12 | cut -c1-600 | tee gpurun_out/r05_f_pytest_fp8_full_depth.txt
