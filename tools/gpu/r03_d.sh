#!/bin/bash
# round 3, GPU call D: the tests added after call C + the default bench line (split leg with 4 calls in flight)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -s -k "hidden or lora or beam or per_layer or padded or auto_model or projector" > $O/pytest_new.log 2>&1; echo "rc=$?" >> $O/pytest_new.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
grep -E "passed|failed|rc=|per-layer|beam search:|hidden" $O/pytest_new.log | tail -12
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r03_d/bench_default.json").read().strip().splitlines()[-1])
    print("value", r["value"], "ids_checked", r["ids_checked"], r["phase_ms_one_session"])
    pm = r["parity_mode"]
    print("split", {k: pm["split"].get(k) for k in ("value", "ms_per_step", "in_flight_batches", "frac_of_fast_path", "ids_checked", "ids_equal_strict")}, pm["split"]["one_batch_at_a_time"])
    print("c3", r["c3_13b_bf16_b16"]["value"], "c5", r["c5_slice_13b_fp8_b16"]["value"])
    print("roofline", r["roofline"]["frac"], r["roofline"]["kernel"][:60])
except Exception as e:
    print("bench ERR", e)
PY
tail -3 $O/bench_default.err
