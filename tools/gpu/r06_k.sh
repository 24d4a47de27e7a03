#!/bin/bash
# round 6, call K: the whole -m gpu suite with durations (budget: the driver's step limit is 1200 s) + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=30 2>&1 | tail -45 | cut -c1-200 | tee gpurun_out/r06_k_gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300 | tee gpurun_out/r06_k_smoke.txt
