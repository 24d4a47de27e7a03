#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_u; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do timeout 200 python bench.py --inflight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/bench$i.json 2> $O/bench$i.err; done
python - <<'PY'
import json
for i in (1, 2):
    r = json.loads(open(f"gpurun_out/r03_u/bench{i}.json").read().strip().splitlines()[-1])
    print("value", r["value"], r["phase_ms_one_session"])
PY
