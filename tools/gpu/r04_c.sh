#!/bin/bash
# round 4, GPU call C: the split mode's new pieces on hardware — wg GEMV with the measured geometry, fp24 KV decode attention,
# folded prefill RMSNorm — kernels, tiny fixtures, the full-size 7b case, and the bench with its parity_mode legs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_c; mkdir -p $O
timeout 300 python tools/kbench.py gemv_wg dattn_split > $O/kbench.txt 2>&1; echo "kbench rc=$?"
grep -E "all GEMVs|decode attention split" $O/kbench.txt
timeout 600 python -m pytest -q --durations=8 --timeout=400 -m gpu tests/test_gpu_kernels.py \
  "tests/test_gpu_e2e.py::test_fixture_split_mode" "tests/test_gpu_e2e.py::test_fixture" \
  "tests/test_gpu_e2e.py::test_cost_answers_equal_reference_loaders" > $O/pytest_small.log 2>&1; echo "pytest small rc=$?"
tail -12 $O/pytest_small.log
for f in 1 0; do
  VC_PREFILL_FOLD=$f timeout 300 python bench.py --steps 4 --warmup 1 --inflight 1 --no-extra-legs --no-cpu-baseline > $O/bench_fold$f.json 2> $O/bench_fold$f.err; echo "fold$f rc=$?"
  python - <<PY
import json
try:
    r=json.loads([l for l in open("$O/bench_fold$f.json") if l.startswith("{")][-1]); print("fold$f", r["value"], r["phase_ms_one_session"], r["ids_checked"])
except Exception as e: print("fold$f failed", e)
PY
done
timeout 500 python bench.py --steps 12 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
try:
    r=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
    print("value", r["value"], "one", r["one_batch_at_a_time"]["value"], r["phase_ms_one_session"])
    pm=r.get("parity_mode",{})
    for k in ("strict","split"):
        if k in pm: print(k, {x: pm[k].get(x) for x in ("value","frac_of_fast_path","ids_checked","ids_equal_strict","in_flight_batches")}, pm[k].get("one_batch_at_a_time"))
    for k in ("c3_13b_bf16_b16","c5_slice_13b_fp8_b16"):
        if k in r: print(k, r[k]["value"], r[k].get("parity_mode",{}).get("split",{}).get("value"), r[k].get("parity_mode",{}).get("split",{}).get("frac_of_fast_path"))
except Exception as e: print("bench failed", e)
PY
timeout 700 python -m pytest -q -s --timeout=650 -m gpu "tests/test_gpu_fulldepth.py::test_full_size_7b_c2" > $O/pytest_full7b.log 2>&1; echo "full7b rc=$?"
grep -E "split mode|bf16 path|strict path|passed|failed|Error|assert" $O/pytest_full7b.log | head -20
