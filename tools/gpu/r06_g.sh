#!/bin/bash
# round 6, call G: a 64-row decode pool (two 32-row weight passes per step, 8 calls in flight) against the 32-row pool (4 calls in
# flight), same box; kernel traces of both for the step-time-by-kernel table
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for r in 32 64 32 64; do timeout 600 python tools/experiments/pool64.py $r 3 2>&1 | tail -1; done | tee gpurun_out/r06_g_pool64.txt
cd /tmp && export TMPDIR=/tmp
for r in 32 64; do
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_g$r -o g -- python $GRAFT_REPO_ROOT/tools/experiments/pool64.py $r 2 > /dev/null 2>&1
  DB=$(find /tmp/prof_g$r -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$DB" $GRAFT_REPO_ROOT/gpurun_out/r06_g_kernel_stats_pool$r.md > /dev/null
  head -12 $GRAFT_REPO_ROOT/gpurun_out/r06_g_kernel_stats_pool$r.md
done
