#!/bin/bash
# round 4, GPU call A: the whole -m gpu suite with per-test durations (the suite has to come under 600 s), then the
# multi-GPU code path on one GPU: bench.py plain vs --force-dist (torch / cabi gather), same steps
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=60 -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
for v in plain force_torch force_cabi; do
  case $v in
    plain) X="";;
    force_torch) X="--force-dist";;
    force_cabi) X="--force-dist --gather cabi";;
  esac
  timeout 400 python bench.py --steps 12 --warmup 1 --no-extra-legs --no-cpu-baseline $X > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$?"
  python - <<PY
import json
try:
    r=json.loads([l for l in open("$O/bench_$v.json") if l.startswith("{")][-1])
    print("$v", r["value"], r["ms_per_step"], r["config"]["token_gather"], r["ids_checked"])
except Exception as e:
    print("$v failed", e)
PY
done
