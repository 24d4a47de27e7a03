#!/bin/bash
# round 6, call P: decode attention — the first K batch requested behind phase 0's operand loads instead of behind the barrier.
# Parity (decode attention in its four cache formats, fixtures), kbench dattn_rows and the bench A/B on one box: new library vs the
# previous commit's (vcoder_amd/lib/ab/libvcoder_hip_old.so); then the WHOLE -m gpu suite with durations (the driver's 1200-s step).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "fused_decode or kv_cache or split_small or e4m3" 2>&1 | tail -4 | tee gpurun_out/r06_p_pytest_kernels.txt
for w in new old; do
  L=""; [ $w = old ] && L="$GRAFT_REPO_ROOT/vcoder_amd/lib/ab/libvcoder_hip_old.so"
  VCODER_HIP_LIB=$L timeout 300 python tools/kbench.py dattn_rows 2>&1 | grep -i "dattn" | sed "s/^/$w /"
done | tee gpurun_out/r06_p_kbench_dattn_rows_ab.txt
for w in new old new old; do
  L=""; [ $w = old ] && L="$GRAFT_REPO_ROOT/vcoder_amd/lib/ab/libvcoder_hip_old.so"
  VCODER_HIP_LIB=$L timeout 400 python bench.py --gpus 1 --steps 12 --warmup 2 --no-extra-legs --no-cpu-baseline > gpurun_out/r06_p_bench_$w.json 2> gpurun_out/r06_p_bench_$w.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r06_p_bench_$w.json").read().strip().splitlines()[-1])
k = d["decode_step_kernels"]
print("$w", "value", round(d["value"], 3), "one_batch", round(d["one_batch_at_a_time"]["value"], 3), d.get("phase_ms_one_session"), "ids_checked", d.get("ids_checked"),
      "attention us", round(k["attention_decode_fused_kernel"]["avg_launch_us"], 2), "frac", round(k["attention_decode_fused_kernel"]["frac"], 4), "latency p50", round(d["inter_token_latency_ms"]["p50"], 3))
PY
done 2>&1 | tee gpurun_out/r06_p_bench_ab.txt
timeout 1500 python -m pytest tests -q -m gpu --durations=25 2>&1 | tail -40 | cut -c1-200 | tee gpurun_out/r06_p_gpu_suite.txt
