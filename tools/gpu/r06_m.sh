#!/bin/bash
# round 6, call M: after the engine.hip file split and the test-time trims: the long tests with durations, the driver-shaped bench line
# (inter-token latency leg, fp16 leg), then the round's profile pass (kernel traces, FETCH_SIZE, SQ MFMA-busy)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fulldepth.py tests/test_gpu_e2e.py -q -m gpu --durations=12 -k "full or per_layer or fp8_weights_true or qkv or fixture" 2>&1 | tail -22 | cut -c1-200 | tee gpurun_out/r06_m_pytest_long.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_m_bench_driver_shaped.json 2> gpurun_out/r06_m_bench.err
tail -c 300 gpurun_out/r06_m_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_m_bench_driver_shaped.json").read().strip().splitlines()[-1])
print("value", d["value"], "one batch", d["one_batch_at_a_time"]["value"], d["phase_ms_one_session"])
print("latency", d["inter_token_latency_ms"])
print("fp16", d["parity_mode"]["fp16"].get("value"), "split", d["parity_mode"]["split"].get("value"), "roofline", d["roofline"]["frac"], d["decode_step_kernels"]["gemv_dma_kernel"]["frac"])
PY
bash tools/profile_round.sh r06_m > gpurun_out/r06_m_profile_round.log 2>&1
tail -20 gpurun_out/r06_m_profile_round.log | cut -c1-200
