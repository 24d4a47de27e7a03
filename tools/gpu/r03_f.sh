#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/profile_round.sh r03_f 2>&1 | tail -50
ls gpurun_out | grep r03_f
