#!/bin/bash
# round 6, call S: parity of the split paths on the six-wave GEMV geometry (kernel level, fixtures, true dims, full depth 7b / 13b, pool)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_fulldepth.py -q -m gpu --durations=6 -k "gemv_wg or gemv_split or split or inexact or lo_plane or full_size or full_depth_7b" 2>&1 | tail -14 | cut -c1-200 | tee gpurun_out/r06_s_pytest.txt
