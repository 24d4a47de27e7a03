#!/bin/bash
# round 4, GPU call D: folded prefill RMSNorm after the epilogue rework (A/B), and the co-residency experiment of VERDICT r03
# weak 8 with the knobs that exist: the 128x128 GEMM tile (64 KiB LDS, 2 workgroups per CU) everywhere, 4 calls in flight
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_d; mkdir -p $O
for f in 1 0; do
  VC_PREFILL_FOLD=$f timeout 300 python bench.py --steps 6 --warmup 1 --inflight 1 --no-extra-legs --no-cpu-baseline > $O/bench_fold$f.json 2> $O/bench_fold$f.err; echo "fold$f rc=$?"
  python - <<PY
import json
try:
    r=json.loads([l for l in open("$O/bench_fold$f.json") if l.startswith("{")][-1]); print("fold$f", r["value"], r["phase_ms_one_session"], r["ids_checked"])
except Exception as e: print("fold$f failed", e)
PY
done
for v in 1 3; do
  VC_PREFILL_FOLD=0 VC_GEMM_VARIANT=$v timeout 300 python bench.py --steps 12 --warmup 1 --no-extra-legs --no-cpu-baseline > $O/bench_gv$v.json 2> $O/bench_gv$v.err; echo "gemm variant $v rc=$?"
  python - <<PY
import json
try:
    r=json.loads([l for l in open("$O/bench_gv$v.json") if l.startswith("{")][-1]); print("gemm variant $v", r["value"], r["ms_per_step"], r["phase_ms_one_session"], r["ids_checked"])
except Exception as e: print("gv$v failed", e)
PY
done
