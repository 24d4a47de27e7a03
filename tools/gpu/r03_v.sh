#!/bin/bash
# round 3, GPU call V: the decode-step hidden-state / attention outputs on the device (all three precision modes) + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_v; mkdir -p $O
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu_e2e.py -q -m gpu -s -k "hidden_states" ) > $O/pytest_hidden.log 2>&1; echo "rc=$?" >> $O/pytest_hidden.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
tail -8 $O/pytest_hidden.log; tail -2 $O/smoke.log
