#!/bin/bash
# round 4, GPU call L: is the SwiGLU epilogue what the gate/up GEMM loses against the vendor library (VERDICT r03 item 5 ii)?
# the same 9728 x 22016 x 4096 GEMM with the SwiGLU, the plain bf16 and the fp32 epilogue
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 120 python tools/kbench.py gemm 2>&1 | grep -E "gate-up|llm qkv|llm down" > gpurun_out/r04_l_kbench_gemm_gateup.txt; cat gpurun_out/r04_l_kbench_gemm_gateup.txt
