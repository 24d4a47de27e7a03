#!/bin/bash
# round 6, call D: go / no-go of the 8-phase GEMM on v_mfma_f32_32x32x16_bf16 (keep if >= 1.40 PF/s on random operands or clearly
# ahead of the 16x16x32 form on the decoder shapes): parity first, then the A/B per shape
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "test_gemm_mfma_32x32x16" 2>&1 | tail -3 | tee gpurun_out/r06_d_pytest_gemm32.txt
timeout 900 python tools/kbench.py gemm32 2>&1 | grep gemm32 | tee gpurun_out/r06_d_kbench_gemm32.txt
