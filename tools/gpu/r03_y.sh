#!/bin/bash
# round 3, GPU call Y: smoke() of the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_y; mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
tail -2 $O/smoke.log
