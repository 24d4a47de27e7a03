#!/bin/bash
# round 6, call E: RoPE + head split + KV write fused into the QKV GEMM's epilogue (EPI_QKV): parity (kernel level bit for bit, fixtures,
# true dims), then the bench A/B on one box: fused (default) vs vc_model_set_qkv_fused(0) (bench.py --no-qkv-fused)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "qkv_fused" 2>&1 | tail -4 | tee gpurun_out/r06_e_pytest_kernels.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "test_fixture or true_dims_against_oracle or test_true_dims_13b or fp8_weight_format or decode_pool_true" 2>&1 | tail -4 | tee gpurun_out/r06_e_pytest_e2e.txt
for on in 1 0 1 0; do
  F=""; [ $on = 0 ] && F="--no-qkv-fused"; timeout 400 python bench.py --gpus 1 --steps 12 --warmup 2 --no-extra-legs --no-cpu-baseline $F > gpurun_out/r06_e_bench_fused$on.json 2> gpurun_out/r06_e_bench_fused$on.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r06_e_bench_fused$on.json").read().strip().splitlines()[-1])
print("qkv_fused=$on", "value", round(d["value"], 3), "one_batch", round(d["one_batch_at_a_time"]["value"], 3), d.get("phase_ms_one_session"), "ids_checked", d.get("ids_checked"))
PY
done 2>&1 | tee gpurun_out/r06_e_bench_ab.txt
