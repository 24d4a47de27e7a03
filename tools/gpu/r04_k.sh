#!/bin/bash
# round 4, GPU call K: 7b with fp8-e4m3 weights (W8A8 prefill, e4m3 KV cache), B = 8, 4 calls in flight
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_k; mkdir -p $O
timeout 150 python bench.py --weights fp8 --steps 8 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/bench_7b_fp8.json 2> $O/bench_7b_fp8.err; echo "rc=$?"
python - <<PY
import json
r=json.loads([l for l in open("$O/bench_7b_fp8.json") if l.startswith("{")][-1]); print("7b fp8", r["value"], r["ms_per_step"], r["one_batch_at_a_time"]["value"], r["phase_ms_one_session"], r["ids_checked"])
PY
