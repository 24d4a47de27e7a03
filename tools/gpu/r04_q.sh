#!/bin/bash
# round 4, GPU call Q: 17..32-row bf16 GEMV, two vs three tiles per workgroup (VC_GEMV2_NT3), timing + bit equality
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 70 python tools/kbench.py gemv_nt3 2>&1 | grep gemv_nt3 > gpurun_out/r04_q_kbench_gemv_nt3.txt; cat gpurun_out/r04_q_kbench_gemv_nt3.txt
