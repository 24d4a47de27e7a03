#!/bin/bash
# round 4, GPU call R: the three-tile geometry as the default — its own test and the pool-vs-session row-agreement test
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 36 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "three_tiles or rows_agree_between_pool" 2>&1 | tail -6 > gpurun_out/r04_r_pytest_nt3.txt; cat gpurun_out/r04_r_pytest_nt3.txt
