#!/bin/bash
# round 4, GPU call N: rows in flight per lane of the e4m3 decode attention (VC_DATTN8_UK = 6 default / 4)
cd "$GRAFT_REPO_ROOT" || exit 1
for u in 6 4; do echo "VC_DATTN8_UK=$u"; VC_DATTN8_UK=$u timeout 60 python tools/kbench.py dattn_kv8 2>&1 | grep e4m3; done > gpurun_out/r04_n_dattn8_uk.txt; cat gpurun_out/r04_n_dattn8_uk.txt
