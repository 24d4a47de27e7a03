#!/bin/bash
# round 6, call A: root-cause of the device-only decode-step deviation of the bf16 path on inexact checkpoints
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
if [ "$1" = "dbg" ]; then
timeout 600 python tools/experiments/dbg_inexact_decode.py ds_img_depth_seg llava_img vc_img_seg 2>&1 | grep -v "^$" | cut -c1-300 | tee gpurun_out/r06_a_dbg_inexact_decode.txt
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -s -k "inexact" 2>&1 | grep -v "^$" | tail -12 | cut -c1-400 | tee gpurun_out/r06_a_pytest_inexact.txt
fi
# the selection of round 5's call E (the run that printed 0.502): the same tests in the same order in one process
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -s -k "inexact or split_mode or strict_mode or true_dims_split or true_dims_strict" 2>&1 | grep -v "^$" | tail -25 | cut -c1-400 | tee gpurun_out/r06_a_pytest_order_of_r05_e.txt
