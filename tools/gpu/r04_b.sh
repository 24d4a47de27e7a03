#!/bin/bash
# round 4, GPU call B: (1) the workgroup-shared GEMV vs the per-wave-ring GEMV (kbench sweep), (2) durations of the tests this
# round added or changed (call A's whole-suite run hit its own 1500-s timeout without durations)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_b; mkdir -p $O
timeout 600 python tools/kbench.py gemv_wg > $O/kbench_gemv_wg.txt 2>&1; echo "kbench rc=$?"
grep -E "all GEMVs" $O/kbench_gemv_wg.txt
timeout 1200 python -m pytest -v --durations=0 --timeout=500 -m gpu \
  "tests/test_gpu_e2e.py::test_cost_answers_equal_reference_loaders" \
  "tests/test_gpu_e2e.py::test_token_comm_rccl_forced_world1" \
  "tests/test_gpu_e2e.py::test_bench_force_dist_one_gpu" \
  "tests/test_gpu_e2e.py::test_cost_harness_batched_equals_per_sample" \
  "tests/test_gpu_kernels.py" \
  "tests/test_gpu_fulldepth.py::test_full_size_13b_c3" > $O/pytest_new.log 2>&1; echo "pytest rc=$?"
grep -E "PASSED|FAILED|ERROR|passed|failed" $O/pytest_new.log | tail -12
grep -E "^[0-9.]+s (call|setup)" $O/pytest_new.log | head -25
