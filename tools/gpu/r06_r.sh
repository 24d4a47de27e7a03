#!/bin/bash
# round 6, call R: the split step's GEMV (gemv_wg_kernel) with six tiles per workgroup where four leave CUs with twice the tiles of
# others (7b qkv 384 -> 256 workgroups, gate / up 344 -> 230).  Parity, kbench per matrix, then the split leg of the bench A/B.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -x -m gpu -k "gemv_wg or gemv_split or split_mode or true_dims_split or inexact or decode_pool_split or lo_plane" 2>&1 | tail -4 | tee gpurun_out/r06_r_pytest.txt
timeout 300 python tools/kbench.py gemv_wg 2>&1 | grep gemv_wg | tee gpurun_out/r06_r_kbench_gemv_wg.txt
for v in chosen 2 chosen 2; do
  F=""; [ $v = 2 ] && F="--gemv-variant 2"
  timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline $F > gpurun_out/r06_r_bench_$v.json 2> gpurun_out/r06_r_bench_$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r06_r_bench_$v.json").read().strip().splitlines()[-1])
sp = d["parity_mode"]["split"]
print("$v", "value", round(d["value"], 3), "| split", round(sp["value"], 3), "frac", round(sp["frac_of_fast_path"], 4), "ids", sp.get("ids_checked"), sp.get("ids_equal_strict"),
      "| split one batch", round(sp["one_batch_at_a_time"]["value"], 3), {k: round(v, 1) for k, v in sp["one_batch_at_a_time"].items() if k.endswith("_ms")})
PY
done 2>&1 | tee gpurun_out/r06_r_bench_ab.txt
