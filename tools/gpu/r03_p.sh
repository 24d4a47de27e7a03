#!/bin/bash
# round 3, GPU call P: calls in flight beyond the pool's 32 rows (requests queue for rows; the pool stays full while a new call prefills)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_p; mkdir -p $O
export TMPDIR=/tmp
for nf in 4 5 6; do
  timeout 400 python bench.py --inflight $nf --no-cpu-baseline --no-extra-legs > $O/bench_inflight$nf.json 2> $O/bench_inflight$nf.err; echo "rc=$?" >> $O/bench_inflight$nf.err
done
python - <<'PY'
import json
for nf in (4, 5, 6):
    try:
        r = json.loads(open(f"gpurun_out/r03_p/bench_inflight{nf}.json").read().strip().splitlines()[-1])
        print(nf, "value", round(r["value"], 3), "ids_checked", r["ids_checked"], "rows mix", r["roofline"]["rows_per_launch"], "pcie", round(r["pcie_inclusive"]["value"], 2))
    except Exception as e:
        print(nf, "failed", e)
PY
