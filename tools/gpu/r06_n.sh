#!/bin/bash
# round 6, call N: the decode GEMV's sum-of-squares partials loaded AFTER the ring is primed, back to back (rounds 1-5: guarded loads,
# each ended by its own s_waitcnt vmcnt(0), BEFORE the first weight DMA).  Parity (bit-equal GEMV tests), kbench gemv_rows and the
# bench A/B on one box: new library vs the same tree with round 5's decode.hip (vcoder_amd/lib/ab/libvcoder_hip_old.so)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gemv or fused_decode" 2>&1 | tail -4 | tee gpurun_out/r06_n_pytest_kernels.txt
for w in new old; do
  L=""; [ $w = old ] && L="$GRAFT_REPO_ROOT/vcoder_amd/lib/ab/libvcoder_hip_old.so"
  VCODER_HIP_LIB=$L timeout 300 python tools/kbench.py gemv_rows 2>&1 | grep "gemv_rows" | sed "s/^/$w /"
done | tee gpurun_out/r06_n_kbench_gemv_rows_ab.txt
for w in new old new old; do
  L=""; [ $w = old ] && L="$GRAFT_REPO_ROOT/vcoder_amd/lib/ab/libvcoder_hip_old.so"
  VCODER_HIP_LIB=$L timeout 400 python bench.py --gpus 1 --steps 12 --warmup 2 --no-extra-legs --no-cpu-baseline > gpurun_out/r06_n_bench_$w.json 2> gpurun_out/r06_n_bench_$w.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r06_n_bench_$w.json").read().strip().splitlines()[-1])
k = d["decode_step_kernels"]
print("$w", "value", round(d["value"], 3), "one_batch", round(d["one_batch_at_a_time"]["value"], 3), d.get("phase_ms_one_session"), "ids_checked", d.get("ids_checked"),
      "gemv frac", round(k["gemv_dma_kernel"]["frac"], 4), "latency p50", round(d["inter_token_latency_ms"]["p50"], 3))
print("   by matrix (32 rows):", {n: round(v, 2) for n, v in k["gemv_dma_kernel"]["by_rows"].get("32", {}).get("by_kind_avg_us", {}).items()})
PY
done 2>&1 | tee gpurun_out/r06_n_bench_ab.txt
