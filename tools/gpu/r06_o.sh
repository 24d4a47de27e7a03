#!/bin/bash
# (RECORD of a call on commit 6135b14: gemv_xr_kernel, kbench gemv_xr and bench.py --gemv-xr were removed again afterwards — DESIGN.md 9.13)
# round 6, call O: gemv_xr_kernel — the ring kernel of o_proj / down with the activation fragments in VGPRs (count-waited asm loads) and
# weights alone in the LDS ring.  Parity first (bit-equal to the LDS-operand form), kbench by ring depth, then the bench A/B on one box.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "register_operand or test_gemv" 2>&1 | tail -4 | tee gpurun_out/r06_o_pytest_kernels.txt
timeout 300 python tools/kbench.py gemv_xr 2>&1 | grep gemv_xr | tee gpurun_out/r06_o_kbench_gemv_xr.txt
for v in 8 0 6 0 8 4; do
  timeout 400 python bench.py --gpus 1 --steps 12 --warmup 2 --no-extra-legs --no-cpu-baseline --gemv-xr $v > gpurun_out/r06_o_bench_xr$v.json 2> gpurun_out/r06_o_bench_xr$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r06_o_bench_xr$v.json").read().strip().splitlines()[-1])
k = d["decode_step_kernels"]
print("xr=$v", "value", round(d["value"], 3), "one_batch", round(d["one_batch_at_a_time"]["value"], 3), d.get("phase_ms_one_session"), "ids_checked", d.get("ids_checked"),
      "gemv frac", round(k["gemv_dma_kernel"]["frac"], 4), "latency p50", round(d["inter_token_latency_ms"]["p50"], 3))
print("   by matrix (32 rows):", {n: round(v, 2) for n, v in k["gemv_dma_kernel"]["by_rows"].get("32", {}).get("by_kind_avg_us", {}).items()}, "one batch alone:", k["gemv_dma_kernel"].get("one_batch_alone"))
PY
done 2>&1 | tee gpurun_out/r06_o_bench_ab.txt
