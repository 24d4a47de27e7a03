#!/bin/bash
# round 3, GPU call J: flash kernel with half-tile phases (S_A S_B | softmax_A | PV_A | softmax_B | PV_B)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_j; mkdir -p $O
export TMPDIR=/tmp
for sc in 0 2; do VC_ATTN_SCHED=$sc timeout 200 python tools/kbench.py attn 2>&1 | grep attention | sed "s/^/sched=$sc /" >> $O/kbench_attn.txt; done
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention or qkv or split" 2>&1 | tail -2 > $O/pytest_subset.log
cat $O/kbench_attn.txt $O/pytest_subset.log
