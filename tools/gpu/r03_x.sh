#!/bin/bash
# round 3, GPU call X: the default bench line (all legs) of the tree as committed last
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_x; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03_x/bench_default.json").read().strip().splitlines()[-1])
print("value", r["value"], "ids_checked", r["ids_checked"], r["phase_ms_one_session"], "one", r["one_batch_at_a_time"]["value"])
print("split", r["parity_mode"]["split"]["value"], r["parity_mode"]["split"]["frac_of_fast_path"], "strict", r["parity_mode"]["strict"]["value"])
print("c3", r["c3_13b_bf16_b16"]["value"], "c5", r["c5_slice_13b_fp8_b16"]["value"], "roofline", r["roofline"]["frac"], "composite", r["composite_roofline"]["frac_one_batch"], r["composite_roofline"]["frac_value"])
PY
tail -2 $O/bench_default.err
