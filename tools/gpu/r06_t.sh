#!/bin/bash
# (RECORD: the two-wave experiment and kbench gemv_wg2 existed only in the working tree of this call — DESIGN.md 4.2c)
# round 6, call T: the failing reference_loaded case with its traceback; kbench of the two-wave x two-slice experiment for o_proj / down (split)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -m gpu -k "inexact_checkpoint_true_dims" 2>&1 | tail -60 | cut -c1-220 | tee gpurun_out/r06_t_pytest.txt
timeout 300 python tools/kbench.py gemv_wg2 2>&1 | grep gemv_wg2 | tee gpurun_out/r06_t_kbench_gemv_wg2.txt
