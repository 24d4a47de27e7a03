#!/bin/bash
# round 5, call I: the pool's hold policy (no decode step while a call that holds rows is still prefilling) A/B on the default bench
# and on the split-mode pooled run; pool device tests.
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out; T=r05_i
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "pool" 2>&1 | tail -3 | tee $O/${T}_pytest_pool.txt
for v in hold nohold; do
  extra=""; [ $v = nohold ] && extra="--no-pool-hold"
  timeout 400 python bench.py --steps 20 --no-cpu-baseline --no-extra-legs $extra > $O/${T}_bench_$v.json 2> $O/${T}_bench_$v.err; echo "bench $v rc=$?"
done
python - <<PY
import json
for f in ("hold","nohold"):
    try:
        r=json.loads([l for l in open("$O/${T}_bench_%s.json"%f) if l.startswith("{")][-1])
        print(f, "value", round(r["value"],3), "pcie", round(r["pcie_inclusive"]["value"],3), "one", round(r["one_batch_at_a_time"]["value"],3), "ids", r["ids_checked"], r["roofline"]["rows_per_launch"])
        rf=r["roofline"]; print("  roofline", rf["kernel"][:30], round(rf["frac"],4), round(rf["avg_launch_us"],2), rf.get("isolated_replay"))
        for k,v in r["decode_step_kernels"].items():
            print("  ", k, "frac", round(v["frac"],4), {a:(b["launches"], round(b["avg_launch_us"],2), round(b.get("frac",0),3), round(b.get("us_per_layer",0),1)) for a,b in v["by_rows"].items()})
    except Exception as e: print(f, "failed", e)
PY
timeout 300 python tools/experiments/split_mode_one_batch.py 2 4 2>&1 | tail -4 | tee $O/${T}_split_pooled.txt
