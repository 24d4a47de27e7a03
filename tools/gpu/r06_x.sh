#!/bin/bash
# round 6, call X: the 8-phase GEMM's epilogues request what they read ahead of their bounds guards (RESID: residual values in batches
# of 8 instead of 32 dependent round trips; bias / weight scales / cos-sin rows per column group).  Parity, kbench A/B, bench A/B.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gemm or qkv" 2>&1 | tail -3 | tee gpurun_out/r06_x_pytest_kernels.txt
for w in new old; do
  L=""; [ $w = old ] && L="$GRAFT_REPO_ROOT/vcoder_amd/lib/ab/libvcoder_hip_old.so"
  VCODER_HIP_LIB=$L timeout 300 python tools/kbench.py gemm gemm_qkv 2>&1 | grep "^gemm" | sed "s/^/$w /"
done | tee gpurun_out/r06_x_kbench_gemm_ab.txt
for w in new old new old; do
  L=""; [ $w = old ] && L="$GRAFT_REPO_ROOT/vcoder_amd/lib/ab/libvcoder_hip_old.so"
  VCODER_HIP_LIB=$L timeout 400 python bench.py --gpus 1 --steps 12 --warmup 2 --no-extra-legs --no-cpu-baseline > gpurun_out/r06_x_bench_$w.json 2> gpurun_out/r06_x_bench_$w.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r06_x_bench_$w.json").read().strip().splitlines()[-1])
print("$w", "value", round(d["value"], 3), "one_batch", round(d["one_batch_at_a_time"]["value"], 3), d.get("phase_ms_one_session"), "ids_checked", d.get("ids_checked"))
PY
done 2>&1 | tee gpurun_out/r06_x_bench_ab.txt
