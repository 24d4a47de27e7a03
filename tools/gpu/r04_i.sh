#!/bin/bash
# round 4, GPU call I: the whole -m gpu suite on the final tree, with durations
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_i; mkdir -p $O
timeout 960 python -m pytest tests -m gpu -q --durations=40 --timeout=600 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -4
grep -E "FAILED|ERROR" $O/pytest_gpu.log | head
grep -E "^[0-9.]+s (call|setup)" $O/pytest_gpu.log | head -40
