#!/bin/bash
# round 6, call AJ: the default bench line three times on ONE box (the run-to-run spread of `value` on the last tree)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2 3; do
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r06_aj_bench_$i.json 2> gpurun_out/r06_aj_bench_$i.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r06_aj_bench_$i.json").read().strip().splitlines()[-1])
print("run $i: value", round(d["value"], 3), "one_batch", round(d["one_batch_at_a_time"]["value"], 3), {k: round(v, 2) for k, v in d["phase_ms_one_session"].items()},
      "roofline", round(d["roofline"]["frac"], 4), "gemv", round(d["decode_step_kernels"]["gemv_dma_kernel"]["frac"], 4), "latency p50", round(d["inter_token_latency_ms"]["p50"], 3), "ids", d["ids_checked"])
PY
done 2>&1 | tee gpurun_out/r06_aj_bench_repeat.txt
