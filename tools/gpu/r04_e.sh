#!/bin/bash
# round 4, GPU call E: the e4m3 KV cache of the fp8 weight format — kernels, the fp8 e2e tests, decode attention rates, and the
# 13b fp8 bench leg with the cache in e4m3 / bf16 (VC_FP8_KV=1 / 0)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_e; mkdir -p $O
timeout 200 python tools/kbench.py dattn_kv8 > $O/kbench_dattn_kv8.txt 2>&1; echo "kbench rc=$?"; cat $O/kbench_dattn_kv8.txt | tail -8
timeout 900 python -m pytest -q --durations=6 --timeout=500 -m gpu "tests/test_gpu_kernels.py::test_split_small_and_attention_kernels" \
  "tests/test_gpu_e2e.py::test_fp8_weight_format" "tests/test_gpu_e2e.py::test_fp8_weights_true_dims_against_oracle" \
  "tests/test_gpu_e2e.py::test_fp8_formats_per_layer_teacher_forced" > $O/pytest_fp8.log 2>&1; echo "pytest rc=$?"
tail -12 $O/pytest_fp8.log
for kv in 1 0; do
  VC_FP8_KV=$kv timeout 500 python bench.py --model 13b --weights fp8 --batch 16 --inflight 2 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/bench_13b_fp8_kv$kv.json 2> $O/bench_13b_fp8_kv$kv.err; echo "kv$kv rc=$?"
  python - <<PY
import json
try:
    r=json.loads([l for l in open("$O/bench_13b_fp8_kv$kv.json") if l.startswith("{")][-1]); print("VC_FP8_KV=$kv", r["value"], r["ms_per_step"], r["phase_ms_one_session"], r["one_batch_at_a_time"]["value"], r["ids_checked"])
except Exception as e: print("kv$kv failed", e)
PY
done
