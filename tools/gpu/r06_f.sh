#!/bin/bash
# round 6, call F: where the fused QKV epilogue's time goes: kbench (gemm + split vs fused, with / without split-K) and a kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python tools/kbench.py gemm_qkv 2>&1 | grep gemm_qkv | tee gpurun_out/r06_f_kbench_gemm_qkv.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o f -- python $GRAFT_REPO_ROOT/tools/kbench.py gemm_qkv > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_f -name "*.db" | head -1); echo "db: $DB"
python tools/rocpd_summary.py "$DB" gpurun_out/r06_f_kernel_stats.md; head -14 gpurun_out/r06_f_kernel_stats.md
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "qkv_fused" 2>&1 | tail -4 | tee gpurun_out/r06_f_pytest_kernels.txt
