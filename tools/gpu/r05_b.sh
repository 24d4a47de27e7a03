#!/bin/bash
# round 5, call B: after the GEMV prune (register-staged kernel, virtual waves, bf16 wg form, NT3 knob removed; wide classes default):
# the GEMV / pool device tests, kbench gemv_rows (all GEMVs of a step at 8..32 rows), the default bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gemv" 2>&1 | tail -5 | tee gpurun_out/r05_b_pytest_gemv.txt
timeout 120 python tools/kbench.py gemv_rows 2>&1 | grep gemv_rows | tee gpurun_out/r05_b_kbench_gemv_rows.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r05_b_bench_default.json 2> gpurun_out/r05_b_bench_default.err; echo "bench rc=$?"
python - <<PY
import json
try:
    r=json.loads([l for l in open("gpurun_out/r05_b_bench_default.json") if l.startswith("{")][-1])
    print("value", r["value"], "pcie", r["pcie_inclusive"]["value"], "one", r["one_batch_at_a_time"]["value"], r["phase_ms_one_session"], "ids", r["ids_checked"])
    print("roofline", r["roofline"]["frac"], r["roofline"]["avg_launch_us"], r["roofline"]["rows_per_launch"])
    pm=r.get("parity_mode",{})
    print("split", pm.get("split",{}).get("value"), pm.get("split",{}).get("frac_of_fast_path"))
    for k in ("c3_13b_bf16_b16","c5_slice_13b_fp8_b16"):
        print(k, r[k]["value"], r[k].get("parity_mode",{}).get("split",{}).get("value"))
except Exception as e: print("bench failed", e)
PY
tail -3 gpurun_out/r05_b_bench_default.err
