#!/bin/bash
# round 6, call B: which commit of round 5 printed bf16_path_decode_logits_err = 0.502 (profiles/r05_e)?  Historical trees built under
# .bisect/<commit> (git worktrees, not tracked), each running ITS OWN tests and library.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for c in "$@"; do
  ( cd .bisect/$c && echo "== $c" && timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -s -k "inexact and not true_dims" 2>&1 | grep "bf16_path\|passed\|failed\|rror" | cut -c1-330 ) | tee -a gpurun_out/r06_b_bisect.txt
done
