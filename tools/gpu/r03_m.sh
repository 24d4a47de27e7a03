#!/bin/bash
# round 3, GPU call M: flash kernel as adopted (V^T key order, permlane maxima, v_max3, deferred maximum on the bf16 path, workgroup
# order by shape): kbench, GPU kernel + e2e suites, one-batch bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_m; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/kbench.py attn > $O/kbench_attn.txt 2>&1
( time timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -m gpu -x ) > $O/pytest_kernels_e2e.log 2>&1; echo "rc=$?" >> $O/pytest_kernels_e2e.log
timeout 600 python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-extra-legs > $O/bench_one_batch.json 2> $O/bench_one_batch.err; echo "rc=$?" >> $O/bench_one_batch.err
cat $O/kbench_attn.txt; grep -E "passed|failed|rc=" $O/pytest_kernels_e2e.log | tail -3
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03_m/bench_one_batch.json").read().strip().splitlines()[-1])
print("value", r["value"], "ids_checked", r["ids_checked"], r["phase_ms_one_session"])
PY
