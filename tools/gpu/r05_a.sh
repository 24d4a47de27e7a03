#!/bin/bash
# round 5, FIRST GPU call (prepared at the end of round 4, when the GPU budget was spent): the "wide" GEMV geometry experiment
# (VC_GEMV_WIDE: ceil(tiles / 256) tiles per workgroup, one deep ring per CU — emulator-checked bit-identical, never timed) on every
# > 512-tile matrix of both models at 8 / 16 / 24 / 32 rows, then its device tests.  Decision rule: enable a class (bit NT of the
# default in decode.hip gemv_wide_now) where it wins at the row counts the bench uses (8 and 32 for 7b, 16 for 13b) AND prints
# "same bits True"; run the full -m gpu suite afterwards.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 120 python tools/kbench.py gemv_wide 2>&1 | grep gemv_wide > gpurun_out/r05_a_kbench_gemv_wide.txt
cat gpurun_out/r05_a_kbench_gemv_wide.txt
timeout 90 python tools/kbench.py gemm_chunk 2>&1 | grep gemm_chunk > gpurun_out/r05_a_kbench_gemm_chunk.txt
VC_GEMM_VARIANT=5 timeout 90 python tools/kbench.py gemm_chunk 2>&1 | grep gemm_chunk | sed 's/^/8phase /' >> gpurun_out/r05_a_kbench_gemm_chunk.txt
cat gpurun_out/r05_a_kbench_gemm_chunk.txt
VC_TEST_EXPERIMENTS=1 timeout 200 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "wide_geometry or three_tiles or virtual_waves" 2>&1 | tail -5 | tee gpurun_out/r05_a_pytest.txt
