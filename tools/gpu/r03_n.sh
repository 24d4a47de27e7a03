#!/bin/bash
# round 3, GPU call N: decode attention with DPP row sums (no ds_bpermute in the score loop)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_n; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/kbench.py dattn_rows > $O/kbench_dattn_rows.txt 2>&1
timeout 300 python tools/kbench.py dattn_rows >> $O/kbench_dattn_rows.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "decode or dattn" 2>&1 | tail -2 > $O/pytest_subset.log
cat $O/kbench_dattn_rows.txt $O/pytest_subset.log
