#!/bin/bash
# round 3, GPU call T: last check of the final tree — smoke(), the pool / true-dims tests, the default bench line without its side legs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_t; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
( time timeout 400 python -m pytest tests/test_gpu_e2e.py -q -m gpu -k "pool or true_dims" ) > $O/pytest_pool.log 2>&1; echo "rc=$?" >> $O/pytest_pool.log
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
tail -2 $O/smoke.log; grep -E "passed|failed|rc=|real" $O/pytest_pool.log | tail -3
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03_t/bench.json").read().strip().splitlines()[-1])
print("value", r["value"], "ids_checked", r["ids_checked"], r["phase_ms_one_session"], "one", r["one_batch_at_a_time"]["value"])
PY
