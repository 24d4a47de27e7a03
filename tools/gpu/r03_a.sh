#!/bin/bash
# round 3, GPU call A: sanity of the changed pool / sampler code, the new default bench line, CU-mask overlap experiment
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03_a
O=gpurun_out/r03_a
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x -k "pool or sampling or stop" > $O/pytest_pool.log 2>&1; echo "pytest rc=$?" >> $O/pytest_pool.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
for cfg in "0:64 64:192" "0:96 96:160" "0:128 128:128" "0:64 -" "0:96 -"; do
  set -- $cfg
  tag=$(echo "pool_$1_sess_$2" | tr ':' 'x')
  if [ "$2" = "-" ]; then
    VC_POOL_CU_RANGE=$1 timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err
  else
    VC_POOL_CU_RANGE=$1 VC_SESSION_CU_RANGE=$2 timeout 300 python bench.py --no-extra-legs --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err
  fi
  echo "rc=$?" >> $O/bench_$tag.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03_a/bench_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value", round(r["value"], 2), "ids_checked", r.get("ids_checked"), "one", round(r["one_batch_at_a_time"]["value"], 2),
              "pcie", round(r.get("pcie_inclusive", {}).get("value", 0), 2))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/pytest_pool.log
