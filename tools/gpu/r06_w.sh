#!/bin/bash
# round 6, call W: the trimmed full-size 7b case (one oracle row with the split ids + the same row with the bf16 path's), and
# FETCH_SIZE of the 8-phase GEMM at the engine's shapes (review item 2: last measured in round 3)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
true
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/r06_w_pmc -o pmc -- python $GRAFT_REPO_ROOT/tools/kbench.py gemm > $GRAFT_REPO_ROOT/gpurun_out/r06_w_pmc_kbench_gemm.txt 2> $GRAFT_REPO_ROOT/gpurun_out/r06_w_pmc.err
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/r06_w_pmc -name "*.db" | head -1)
python tools/pmc_gemm_traffic.py "$DB" gpurun_out/r06_w_gemm_fetch_size.md
rm -rf gpurun_out/r06_w_pmc
