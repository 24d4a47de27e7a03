#!/bin/bash
# round 5, call M: the driver-shaped line on the final tree (--steps 20 --warmup 5), the lone batch with / without the in-situ stamps,
# a one-batch trace (cost of the stamp fold kernel after its last pass was parallelised).
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out; T=r05_m
mkdir -p $O
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_driver_shaped.json 2> $O/${T}_bench_driver_shaped.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-extra-legs > $O/${T}_one_insitu.json 2>/dev/null
timeout 300 python bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-extra-legs --no-insitu > $O/${T}_one_noinsitu.json 2>/dev/null
python - <<PY
import json
r=json.loads([l for l in open("$O/${T}_bench_driver_shaped.json") if l.startswith("{")][-1])
print("value", round(r["value"],3), "pcie", round(r["pcie_inclusive"]["value"],3), "one", round(r["one_batch_at_a_time"]["value"],3), r["phase_ms_one_session"], "ids", r["ids_checked"], r["roofline"]["rows_per_launch"])
print("roofline", r["roofline"]["kernel"][:40], round(r["roofline"]["frac"],4), round(r["roofline"]["avg_launch_us"],2), r["roofline"]["traffic"])
for k,v in r["decode_step_kernels"].items(): print("  ", k, round(v["frac"],4), round(v["isolated_replay"]["frac"],4))
pm=r.get("parity_mode",{}); print("split", pm["split"]["value"], pm["split"]["frac_of_fast_path"], "c3", r["c3_13b_bf16_b16"]["value"], r["c3_13b_bf16_b16"]["parity_mode"]["split"]["frac_of_fast_path"], "c5", r["c5_slice_13b_fp8_b16"]["value"], "cpu", r["cpu_baseline"]["value"])
print("composite", r["composite_roofline"]["frac_one_batch"], r["composite_roofline"]["frac_value"])
for f in ("one_insitu","one_noinsitu"):
    q=json.loads([l for l in open("$O/${T}_%s.json"%f) if l.startswith("{")][-1])
    print(f, round(q["value"],3), q["phase_ms_one_session"], round(q["roofline"]["frac"],4), q["roofline"]["measured"][:12])
PY
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${T}_tr -o ks -- python $ROOT/bench.py --steps 1 --warmup 1 --inflight 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $O/${T}_tr.err
DB=$(find $O/${T}_tr -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py "$DB" $O/${T}_kernel_stats_one_batch.md > /dev/null 2>> $O/${T}_tr.err
rm -rf $O/${T}_tr
grep "stamp_\|select_embed" $O/${T}_kernel_stats_one_batch.md
