#!/bin/bash
# round 4, GPU call S (the round's last seconds): 17..32-row GEMV over the 256-tile matrices, 8-wave workgroup vs virtual waves
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 14 python tools/kbench.py gemv_kvirt 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_s_kbench_gemv_kvirt.txt; cat gpurun_out/r04_s_kbench_gemv_kvirt.txt
