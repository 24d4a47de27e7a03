#!/bin/bash
# round 5, call D: the in-situ roofline with launch PERIODS (previous launch's last workgroup end -> this launch's last workgroup end)
# next to a rocprofv3 kernel trace of the SAME run: stamps vs the trace, per kernel kind at 32 rows.
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out; T=r05_d
mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/${T}_tr -o ks -- python $ROOT/bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/${T}_traced_bench.json 2> $O/${T}_traced.err
DB=$(find $O/${T}_tr -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py "$DB" $O/${T}_kernel_stats_pooled.md > /dev/null 2>> $O/${T}_traced.err
python $ROOT/tools/rocpd_overlap.py "$DB" $O/${T}_decode_kernels_alone_vs_corun.md > /dev/null 2>> $O/${T}_traced.err
rm -rf $O/${T}_tr
cd $ROOT
python - <<PY
import json
r=json.loads([l for l in open("$O/${T}_traced_bench.json") if l.startswith("{")][-1])
print("traced value", round(r["value"],3), "ids", r["ids_checked"], r["roofline"]["rows_per_launch"])
rf=r["roofline"]; print("roofline", rf["kernel"][:30], "frac", round(rf["frac"],4), "avg_launch_us", round(rf["avg_launch_us"],2), rf.get("isolated_replay"))
for k,v in r["decode_step_kernels"].items():
    print(k, "period frac", round(v["frac"],4), "exec frac", round(v["frac_exec_only"],4), "replay", round(v["isolated_replay"]["frac"],4))
    for rows,b in v["by_rows"].items():
        print("   rows",rows, b["launches"], "period", round(b["avg_launch_us"],2), "exec", round(b["avg_exec_us"],2), {a:round(c,2) for a,c in b["by_kind_avg_us"].items()}, round(b.get("us_per_layer",0),1))
PY
grep "gemv_dma_kernel\|attention_decode" $O/${T}_kernel_stats_pooled.md | head -14
cat $O/${T}_decode_kernels_alone_vs_corun.md | head -12
