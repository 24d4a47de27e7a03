#!/bin/bash
# round 6, call AG: the W8A16 wide geometry from 9 rows on — parity (rows agree, fp8 e2e incl. full depth) and the 13b fp8 leg
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_fulldepth.py -q -x -m gpu -k "gemv_fp8 or rows_agree or fp8 or decode_pool" 2>&1 | tail -3 | tee gpurun_out/r06_ag_pytest.txt
for i in 1 2; do
timeout 400 python bench.py --model 13b --batch 16 --inflight 2 --weights fp8 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r06_ag_bench_$i.json 2> gpurun_out/r06_ag_bench_$i.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r06_ag_bench_$i.json").read().strip().splitlines()[-1])
print("13b fp8 B=16 x 2 in flight:", round(d["value"], 3), "one batch", round(d["one_batch_at_a_time"]["value"], 3), {k: round(v, 1) for k, v in d["phase_ms_one_session"].items()}, "ids", d.get("ids_checked"))
PY
done 2>&1 | tee gpurun_out/r06_ag_bench.txt
