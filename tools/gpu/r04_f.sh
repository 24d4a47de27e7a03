#!/bin/bash
# round 4, GPU call F: the whole -m gpu suite as the driver runs it, verbose with durations (per-test timeout 700 s)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -v --durations=45 --timeout=700 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -4
grep -E "FAILED|ERROR" $O/pytest_gpu.log | head
grep -E "^[0-9.]+s (call|setup)" $O/pytest_gpu.log | head -45
