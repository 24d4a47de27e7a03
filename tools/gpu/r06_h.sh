#!/bin/bash
# round 6, call H: the fp16-operand library (libvcoder_hip_f16.so) on hardware: kernels at true shapes, live-reference fixtures,
# true dims vs the fp32 oracle, full 7b depth vs the split path; then throughput of the two libraries side by side
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -s -k "fp16_operand" 2>&1 | tail -4 | tee gpurun_out/r06_h_pytest_kernels.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -s -k "fp16_operand or inexact_checkpoint_true_dims" 2>&1 | grep -v "^$" | tail -12 | cut -c1-400 | tee gpurun_out/r06_h_pytest_e2e.txt
timeout 1200 python -m pytest tests/test_gpu_fulldepth.py -q -x -m gpu -s -k "fp16_operand" 2>&1 | grep -v "^$" | tail -6 | cut -c1-500 | tee gpurun_out/r06_h_pytest_fulldepth.txt
