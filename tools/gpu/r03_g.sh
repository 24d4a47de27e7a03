#!/bin/bash
# round 3, GPU call G: final validation — the whole -m gpu suite as the driver runs it, smoke(), plain kbench of the decode-step
# kernels by row count, the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_g; mkdir -p $O
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu_all.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
timeout 600 python tools/kbench.py gemv_rows dattn_rows > $O/kbench_rows.txt 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
grep -E "passed|failed|rc=|real" $O/pytest_gpu_all.log | tail -5; tail -2 $O/smoke.log; grep "all GEMVs\|decode attention" $O/kbench_rows.txt
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03_g/bench_default.json").read().strip().splitlines()[-1])
print("value", r["value"], "ids_checked", r["ids_checked"], r["phase_ms_one_session"], "one", r["one_batch_at_a_time"]["value"])
print("split", r["parity_mode"]["split"]["value"], r["parity_mode"]["split"]["frac_of_fast_path"], "strict", r["parity_mode"]["strict"]["value"])
print("c3", r["c3_13b_bf16_b16"]["value"], "c5", r["c5_slice_13b_fp8_b16"]["value"], "roofline", r["roofline"]["frac"], r["roofline"]["traffic"])
PY
