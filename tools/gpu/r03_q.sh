#!/bin/bash
# round 3, GPU call Q: queue priorities — pool stream (decode steps) vs session streams (encode + prefill)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_q; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --no-cpu-baseline --no-extra-legs > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "rc=$?" >> $O/bench_$tag.err; }
run control A=1
run pool_high VC_POOL_CU_RANGE_PRIO=high
run pool_low VC_POOL_CU_RANGE_PRIO=low
run sess_high VC_SESSION_CU_RANGE_PRIO=high
run pool_high_sess_low VC_POOL_CU_RANGE_PRIO=high VC_SESSION_CU_RANGE_PRIO=low
run control2 A=1
python - <<'PY'
import json
for t in ("control", "pool_high", "pool_low", "sess_high", "pool_high_sess_low", "control2"):
    try:
        r = json.loads(open(f"gpurun_out/r03_q/bench_{t}.json").read().strip().splitlines()[-1])
        print(t, "value", round(r["value"], 3), "ids_checked", r["ids_checked"], "one", round(r["one_batch_at_a_time"]["value"], 2))
    except Exception as e:
        print(t, "failed", e)
PY
