#!/bin/bash
# round 6, call AA: row-norm kernels request a whole row at once; quant_act_rows keeps its row in registers; select_embed spreads the embedding
# row over the block.  Parity, then the bench A/B on one box: new vs the library before the three changes (7b headline, 13b bf16, 13b fp8).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -x -m gpu -k "norm or fixture or vision_tower or fp8 or true_dims or q8 or select or sampling or embed or pool or f8" 2>&1 | tail -3 | tee gpurun_out/r06_aa_pytest.txt
for w in new old new old; do
  L=""; [ $w = old ] && L="$GRAFT_REPO_ROOT/vcoder_amd/lib/ab/libvcoder_hip_old.so"
  VCODER_HIP_LIB=$L timeout 600 python bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r06_aa_bench_$w.json 2> gpurun_out/r06_aa_bench_$w.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r06_aa_bench_$w.json").read().strip().splitlines()[-1])
c3, c5 = d["c3_13b_bf16_b16"], d["c5_slice_13b_fp8_b16"]
print("$w", "value", round(d["value"], 3), "one_batch", round(d["one_batch_at_a_time"]["value"], 3), {k: round(v, 2) for k, v in d["phase_ms_one_session"].items()}, "ids", d.get("ids_checked"),
      "| 13b bf16", round(c3["value"], 3), "prefill", round(c3["one_batch_at_a_time"]["prefill_ms"], 1), "| 13b fp8", round(c5["value"], 3), "prefill", round(c5["one_batch_at_a_time"]["prefill_ms"], 1),
      "| split", round(d["parity_mode"]["split"]["value"], 3))
PY
done 2>&1 | tee gpurun_out/r06_aa_bench_ab.txt
