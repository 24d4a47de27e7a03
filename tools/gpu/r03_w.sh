#!/bin/bash
# round 3, GPU call W: pool stress / context limit / empty inputs on the device
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_w; mkdir -p $O
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu_e2e.py -q -m gpu -k "random_schedule_and_limits" ) > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
