#!/bin/bash
# round 4, GPU call G: the multi-process bench tests after the cpu_baseline change (call F: test_bench_two_ranks_self_launched
# failed / ate ~900 s — rank 0 timed the oracle inside the 2-rank job), with durations
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_g; mkdir -p $O
timeout 700 python -m pytest -v --durations=5 --timeout=650 -m gpu "tests/test_gpu_e2e.py::test_bench_two_ranks_self_launched" \
  "tests/test_gpu_e2e.py::test_bench_force_dist_one_gpu" > $O/pytest.log 2>&1; echo "rc=$?"
grep -E "PASSED|FAILED|passed|failed|s call" $O/pytest.log | tail -12
grep -E "Error|assert " $O/pytest.log | head -10
