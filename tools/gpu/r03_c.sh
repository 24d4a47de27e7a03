#!/bin/bash
# round 3, GPU call C: the whole -m gpu suite (timed), as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_c; mkdir -p $O
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/ -q -m gpu -s --durations=25 ) > $O/pytest_gpu_all.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_all.log
grep -E "passed|failed|error|rc=|real" $O/pytest_gpu_all.log | tail -8
grep -E "split mode vs|bf16 path vs|per-layer relative|padded batch|projector .*rel err|strict path vs" $O/pytest_gpu_all.log | tail -30
grep -A30 "slowest" $O/pytest_gpu_all.log | head -40
