#!/bin/bash
# round 5, call W (last): on the FINAL tree — the bench command under rocprofv3 (pooled = the timed configuration, one batch) with the
# bench line of the same run beside the table, and the driver-shaped line (--steps 20 --warmup 5).
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out; T=r05_w
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
trace() {  # name, command...
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/${T}_tr_$name -o ks -- "$@" > $O/${T}_$name.out 2> $O/${T}_$name.err
  local DB=$(find $O/${T}_tr_$name -name "*.db" | head -1)
  python $ROOT/tools/rocpd_summary.py "$DB" $O/${T}_kernel_stats_$name.md > /dev/null 2>> $O/${T}_$name.err
  python $ROOT/tools/rocpd_overlap.py "$DB" $O/${T}_alone_vs_corun_$name.md > /dev/null 2>> $O/${T}_$name.err
  rm -rf $O/${T}_tr_$name
  echo "trace $name done: $(head -c 300 $O/${T}_$name.out | tr '\n' ' ' | cut -c1-160)"
}
trace pooled python $ROOT/bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-extra-legs
trace one_batch python $ROOT/bench.py --steps 1 --warmup 1 --inflight 1 --no-cpu-baseline --no-extra-legs
cp $O/${T}_pooled.out $O/${T}_bench_under_rocprof_pooled.json
cp $O/${T}_one_batch.out $O/${T}_bench_under_rocprof_one_batch.json
cd $ROOT
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_driver_shaped.json 2> $O/${T}_bench_driver_shaped.err; echo "bench rc=$?"
python - <<PY
import json
r=json.loads([l for l in open("$O/${T}_bench_under_rocprof_pooled.json") if l.startswith("{")][-1])
print("traced: value", round(r["value"],3), r["roofline"]["kernel"][:34], round(r["roofline"]["frac"],4), round(r["roofline"]["avg_launch_us"],2))
for k,v in r["decode_step_kernels"].items():
    for rows,b in v["by_rows"].items():
        print("  ", k, "rows",rows, b["launches"], round(b["avg_launch_us"],2), {a:round(c,2) for a,c in b["by_kind_avg_us"].items()}, round(b.get("us_per_layer",0),1), round(b.get("frac",0),4))
r=json.loads([l for l in open("$O/${T}_bench_driver_shaped.json") if l.startswith("{")][-1])
print("driver-shaped: value", round(r["value"],3), "pcie", round(r["pcie_inclusive"]["value"],3), "one", round(r["one_batch_at_a_time"]["value"],3), r["phase_ms_one_session"], "ids", r["ids_checked"], r["roofline"]["rows_per_launch"])
print("  roofline", r["roofline"]["kernel"][:34], round(r["roofline"]["frac"],4), "gemv", round(r["decode_step_kernels"]["gemv_dma_kernel"]["frac"],4), "split", round(r["parity_mode"]["split"]["value"],2), round(r["parity_mode"]["split"]["frac_of_fast_path"],3), "c3", round(r["c3_13b_bf16_b16"]["value"],2), "c5", round(r["c5_slice_13b_fp8_b16"]["value"],2), "cpu", round(r["cpu_baseline"]["value"],5))
PY
grep "attention_decode\|gemv_dma_kernel<.*false, 4>" $O/${T}_kernel_stats_pooled.md | head -6
tail -2 $O/${T}_alone_vs_corun_pooled.md
