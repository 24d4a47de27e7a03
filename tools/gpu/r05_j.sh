#!/bin/bash
# round 5, call J: the whole -m gpu suite on the current tree + smoke + the default bench line (with the extra legs).
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out; T=${TAG:-r05_j}
mkdir -p $O
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -6 | tee $O/${T}_gpu_suite.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/${T}_smoke.log
timeout 700 python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err; echo "bench rc=$?"
python - <<PY
import json
try:
    r=json.loads([l for l in open("$O/${T}_bench_default.json") if l.startswith("{")][-1])
    print("value", r["value"], "pcie", r["pcie_inclusive"]["value"], "one", r["one_batch_at_a_time"]["value"], r["phase_ms_one_session"], "ids", r["ids_checked"])
    print("roofline", r["roofline"]["kernel"][:50], r["roofline"]["frac"], r["roofline"]["traffic"], "cpu", r.get("cpu_baseline",{}).get("value"), r.get("cpu_baseline",{}).get("cores"))
    pm=r.get("parity_mode",{})
    print("split", pm.get("split",{}).get("value"), pm.get("split",{}).get("frac_of_fast_path"), pm.get("split",{}).get("ids_equal_strict"))
    for k in ("c3_13b_bf16_b16","c5_slice_13b_fp8_b16"):
        print(k, r[k]["value"], r[k].get("parity_mode",{}).get("split",{}).get("value"), r[k].get("parity_mode",{}).get("split",{}).get("frac_of_fast_path"))
    print("composite", r["composite_roofline"]["frac_one_batch"], r["composite_roofline"]["frac_value"])
except Exception as e: print("bench failed", e)
PY
tail -3 $O/${T}_bench_default.err
