#!/bin/bash
# round 3, GPU call O: validation after the flash / decode attention kernel work — the whole -m gpu suite as the driver runs it, smoke(),
# the default bench line, and the round's rocprof artefacts (tools/profile_round.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_o; mkdir -p $O
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu_all.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
bash tools/profile_round.sh r03_o > $O/profile_round.log 2>&1
grep -E "passed|failed|rc=|real" $O/pytest_gpu_all.log | tail -5; tail -2 $O/smoke.log
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03_o/bench_default.json").read().strip().splitlines()[-1])
print("value", r["value"], "ids_checked", r["ids_checked"], r["phase_ms_one_session"], "one", r["one_batch_at_a_time"]["value"])
print("split", r["parity_mode"]["split"]["value"], r["parity_mode"]["split"]["frac_of_fast_path"], "strict", r["parity_mode"]["strict"]["value"])
print("c3", r["c3_13b_bf16_b16"]["value"], "c5", r["c5_slice_13b_fp8_b16"]["value"], "roofline", r["roofline"]["frac"], r["roofline"]["traffic"])
print("composite", r["composite_roofline"]["frac_one_batch"], r["composite_roofline"]["frac_value"], r["composite_roofline"]["measured_legs_ms"])
PY
head -12 gpurun_out/r03_o_kernel_stats_one_batch.md
