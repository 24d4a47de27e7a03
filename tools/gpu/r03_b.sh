#!/bin/bash
# round 3, GPU call B: split-precision mode on hardware (kernels, fixtures, true dims, full-size C2), then the default bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "split" > $O/pytest_split_kernels.log 2>&1; echo "rc=$?" >> $O/pytest_split_kernels.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -s -k "split" > $O/pytest_split_e2e.log 2>&1; echo "rc=$?" >> $O/pytest_split_e2e.log
timeout 1500 python -m pytest tests/test_gpu_fulldepth.py -q -x -s -k "7b" > $O/pytest_full_7b.log 2>&1; echo "rc=$?" >> $O/pytest_full_7b.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
tail -4 $O/pytest_split_kernels.log; tail -6 $O/pytest_split_e2e.log; tail -15 $O/pytest_full_7b.log
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r03_b/bench_default.json").read().strip().splitlines()[-1])
    print("value", r["value"], "ids_checked", r["ids_checked"], "prefill/decode", r["phase_ms_one_session"])
    print("parity_mode", json.dumps(r.get("parity_mode"))[:1500])
except Exception as e:
    print("bench ERR", e)
PY
tail -3 $O/bench_default.err
