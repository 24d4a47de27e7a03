#!/bin/bash
# round 6, call AE: the one-workgroup-per-CU geometry of the ring GEMV for W8A16 weights at 17..32 rows.  Parity (rows agree between the
# pool's and the sessions' passes, fp8 GEMV tests), kbench pairs vs wide, then the 13b fp8 leg A/B on one box.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -x -m gpu -k "gemv_fp8 or rows_agree or fp8_weights or fp8_formats_per_layer or decode_pool" 2>&1 | tail -3 | tee gpurun_out/r06_ae_pytest.txt
timeout 300 python tools/kbench.py gemv_rows8 2>&1 | grep gemv_rows8 | tee gpurun_out/r06_ae_kbench_gemv_rows8.txt
for w in new old new old; do
  L=""; [ $w = old ] && L="$GRAFT_REPO_ROOT/vcoder_amd/lib/ab/libvcoder_hip_old.so"
  VCODER_HIP_LIB=$L timeout 400 python bench.py --model 13b --batch 16 --inflight 2 --weights fp8 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r06_ae_bench_$w.json 2> gpurun_out/r06_ae_bench_$w.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r06_ae_bench_$w.json").read().strip().splitlines()[-1])
print("$w 13b fp8 B=16 x 2 in flight:", round(d["value"], 3), "one batch", round(d["one_batch_at_a_time"]["value"], 3), {k: round(v, 1) for k, v in d["phase_ms_one_session"].items()}, "ids", d.get("ids_checked"))
PY
done 2>&1 | tee gpurun_out/r06_ae_bench_ab.txt
