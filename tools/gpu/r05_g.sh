#!/bin/bash
# round 5, call G: fp8 outlier per-layer numbers (both formats), fp8 fixtures (cached-step attentions over e4m3 keys), full depth.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -s -k "massive or test_fp8_weight_format" 2>&1 | grep -v "^$" | tail -14 | cut -c1-700 | tee gpurun_out/r05_g_pytest_fp8.txt
timeout 900 python -m pytest tests/test_gpu_fulldepth.py -q -x -m gpu -s -k "fp8_formats_vs_bf16" 2>&1 | grep -v "^$" | tail -8 | cut -c1-600 | tee gpurun_out/r05_g_pytest_fp8_full_depth.txt
