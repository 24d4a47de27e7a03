#!/bin/bash
# round 3, GPU call E: GEMM yardstick (vendor library vs the 8-phase kernel), the fixed loader test, a one-batch timing check
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_e; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/experiments/gemm_yardstick.py > $O/gemm_yardstick.txt 2>&1; echo "rc=$?" >> $O/gemm_yardstick.txt
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -k "lora" > $O/pytest_lora.log 2>&1; echo "rc=$?" >> $O/pytest_lora.log
timeout 600 python bench.py --no-extra-legs --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.err
cat $O/gemm_yardstick.txt | grep -v amdgpu.ids; tail -3 $O/pytest_lora.log
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03_e/bench_short.json").read().strip().splitlines()[-1])
print("value", r["value"], r["phase_ms_one_session"], r["one_batch_at_a_time"]["value"])
PY
