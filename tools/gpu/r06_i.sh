#!/bin/bash
# round 6, call I: fp16 kernel test after the shared rope_pair; the driver-shaped bench line with the fp16 leg and the config keys
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "fp16_operand or qkv_fused" 2>&1 | tail -3 | tee gpurun_out/r06_i_pytest_kernels.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_i_bench_driver_shaped.json 2> gpurun_out/r06_i_bench.err
tail -c 400 gpurun_out/r06_i_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_i_bench_driver_shaped.json").read().strip().splitlines()[-1])
print("value", d["value"], "one batch", d["one_batch_at_a_time"]["value"], d["phase_ms_one_session"])
print("config", json.dumps(d["config"])[-420:])
pm = d["parity_mode"]
for k in ("strict", "split", "fp16"):
    print(k, {a: pm[k].get(a) for a in ("value", "frac_of_fast_path", "ids_equal_fast_path", "ids_equal_strict", "ids_equal_strict_fraction", "ids_equal_strict_fraction_of_the_bf16_library", "ids_checked", "error")})
print("roofline", d["roofline"]["frac"], d["roofline"]["kernel"][:60]); print("c3", d["c3_13b_bf16_b16"]["value"], "c5", d["c5_slice_13b_fp8_b16"]["value"], "cpu", d["cpu_baseline"]["value"])
PY
