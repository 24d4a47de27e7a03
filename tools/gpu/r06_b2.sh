#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python tools/experiments/dbg_inexact_decode.py ds_img_depth_seg llava_img vc_img_seg 2>&1 | grep "^==\|free-running" | cut -c1-300 | tee gpurun_out/r06_b_free_running.txt
( cd .bisect/2398e5b && timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -s -k "inexact or split_mode or strict_mode or true_dims_split or true_dims_strict" 2>&1 | grep "bf16_path\|passed\|failed" | cut -c1-330 ) | tee gpurun_out/r06_b_2398e5b_full_order.txt
