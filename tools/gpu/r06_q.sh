#!/bin/bash
# round 6, call Q: the driver-shaped bench line and the round's profile pass on the tree after the GEMV prologue / epilogue and decode
# attention changes (kernel traces pooled / one batch / split / 13b fp8, FETCH_SIZE, SQ MFMA-busy)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r06_q_smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_q_bench_driver_shaped.json 2> gpurun_out/r06_q_bench.err
tail -c 300 gpurun_out/r06_q_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_q_bench_driver_shaped.json").read().strip().splitlines()[-1])
print("value", d["value"], "one batch", d["one_batch_at_a_time"]["value"], d["phase_ms_one_session"])
print("latency", {k: d["inter_token_latency_ms"][k] for k in ("p50", "p99", "max")})
print("fp16", d["parity_mode"]["fp16"].get("value"), "split", d["parity_mode"]["split"].get("value"), "roofline", d["roofline"]["frac"], d["decode_step_kernels"]["gemv_dma_kernel"]["frac"])
print("cpu_baseline", d.get("cpu_baseline"))
PY
bash tools/profile_round.sh r06_q > gpurun_out/r06_q_profile_round.log 2>&1
tail -12 gpurun_out/r06_q_profile_round.log | cut -c1-200
