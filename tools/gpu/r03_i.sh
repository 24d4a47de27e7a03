#!/bin/bash
# round 3, GPU call I: workgroup order of the flash kernel (VC_ATTN_SCHED 0..3)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_i; mkdir -p $O
export TMPDIR=/tmp
for sc in 0 1 2 3; do VC_ATTN_SCHED=$sc timeout 200 python tools/kbench.py attn 2>&1 | grep attention | sed "s/^/sched=$sc /" >> $O/kbench_attn_sched.txt; done
VC_ATTN_SCHED=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention or qkv or split" 2>&1 | tail -2 > $O/pytest_subset.log
cat $O/kbench_attn_sched.txt $O/pytest_subset.log
