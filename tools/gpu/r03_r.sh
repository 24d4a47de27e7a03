#!/bin/bash
# round 3, GPU call R: the randomised differential check on the device
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_r; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_e2e.py -q -m gpu -s -k "random_prompt_structures" ) > $O/pytest_fuzz.log 2>&1; echo "rc=$?" >> $O/pytest_fuzz.log
tail -25 $O/pytest_fuzz.log
