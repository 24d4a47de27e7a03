#!/bin/bash
# round 6, call AH: the final tree — the WHOLE -m gpu suite with durations, smoke, the driver-shaped bench line, the profile pass
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -26 | cut -c1-200 | tee gpurun_out/r06_ah_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r06_ah_smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_ah_bench_driver_shaped.json 2> gpurun_out/r06_ah_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_ah_bench_driver_shaped.json").read().strip().splitlines()[-1])
print("value", d["value"], "one batch", d["one_batch_at_a_time"]["value"], d["phase_ms_one_session"])
print("fp16", d["parity_mode"]["fp16"].get("value"), "split", d["parity_mode"]["split"].get("value"), d["parity_mode"]["split"].get("frac_of_fast_path"), "roofline", d["roofline"]["frac"], d["decode_step_kernels"]["gemv_dma_kernel"]["frac"])
print({k: v for k, v in d.items() if k.startswith("c3") or k.startswith("c5")})
PY
bash tools/profile_round.sh r06_ah > gpurun_out/r06_ah_profile_round.log 2>&1
tail -6 gpurun_out/r06_ah_profile_round.log | cut -c1-200
