#!/bin/bash
# round 3, GPU call S: flash kernel skipping the wave-tiles with no live (query, key) pair
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_s; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do timeout 200 python tools/kbench.py attn 2>&1 | grep attention >> $O/kbench_attn.txt; done
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -m gpu -k "attention or qkv or split or fixture or mask or random" 2>&1 | tail -2 > $O/pytest_subset.log
cat $O/kbench_attn.txt $O/pytest_subset.log
