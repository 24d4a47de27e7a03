#!/bin/bash
# round 5, call C: the in-situ roofline (vc_pool_profile stamps) — bench with and without the stamps (their cost), the pool device
# tests, and a rocprofv3 kernel trace of the timed configuration with the alone / co-running split of the decode-step kernels.
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out; T=r05_c
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "pool" 2>&1 | tail -4 | tee $O/${T}_pytest_pool.txt
timeout 400 python bench.py --no-cpu-baseline --no-extra-legs > $O/${T}_bench_insitu.json 2> $O/${T}_bench_insitu.err; echo "bench rc=$?"
timeout 400 python bench.py --no-cpu-baseline --no-extra-legs --no-insitu > $O/${T}_bench_no_insitu.json 2> $O/${T}_bench_no_insitu.err; echo "bench rc=$?"
python - <<PY
import json
for f in ("insitu","no_insitu"):
    try:
        r=json.loads([l for l in open("$O/${T}_bench_%s.json"%f) if l.startswith("{")][-1])
        print(f, "value", round(r["value"],3), "one", round(r["one_batch_at_a_time"]["value"],3), "ids", r["ids_checked"])
        rf=r["roofline"]; print("  roofline", rf["kernel"][:40], round(rf["frac"],4), round(rf["avg_launch_us"],2), rf.get("isolated_replay"), rf["measured"][:20])
        for k,v in r["decode_step_kernels"].items():
            print("  ", k, round(v["frac"],4), round(v["us_per_step"],1), {a:(round(b["avg_launch_us"],2), round(b.get("frac",0),3), round(b.get("us_per_layer",0),1)) for a,b in v["by_rows"].items()})
            if "isolated_replay" in v: print("     replay", round(v["isolated_replay"]["frac"],4), round(v["isolated_replay"]["us_per_step"],1))
    except Exception as e: print(f, "failed", e)
PY
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${T}_tr -o ks -- python $ROOT/bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/${T}_traced_bench.json 2> $O/${T}_traced.err
DB=$(find $O/${T}_tr -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py "$DB" $O/${T}_kernel_stats_pooled.md > /dev/null 2>> $O/${T}_traced.err
python $ROOT/tools/rocpd_overlap.py "$DB" $O/${T}_decode_kernels_alone_vs_corun.md 2>> $O/${T}_traced.err
sqlite3 "$DB" ".schema kernels" 2>/dev/null | head -5
rm -rf $O/${T}_tr
head -16 $O/${T}_kernel_stats_pooled.md
tail -5 $O/${T}_traced.err
