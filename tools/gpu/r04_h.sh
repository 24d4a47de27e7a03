#!/bin/bash
# round 4, GPU call H: the round's profile artefacts (rocprofv3 kernel traces of the bench command — pooled, one batch, split mode
# pooled, 13b fp8 — and a SEPARATE --pmc FETCH_SIZE pass over the decode-step kernels), the final default bench line, smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out; T=r04_h
export TMPDIR=/tmp
cd /tmp
trace() {  # name, command...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/${T}_tr_$name -o ks -- "$@" > $O/${T}_$name.out 2> $O/${T}_$name.err
  local DB=$(find $O/${T}_tr_$name -name "*.db" | head -1)
  python $ROOT/tools/rocpd_summary.py "$DB" $O/${T}_kernel_stats_$name.md > /dev/null 2>> $O/${T}_$name.err
  rm -rf $O/${T}_tr_$name
  echo "trace $name done: $(head -c 300 $O/${T}_$name.out | tr '\n' ' ' | cut -c1-200)"
}
trace pooled python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs
trace one_batch python $ROOT/bench.py --steps 1 --warmup 1 --inflight 1 --no-cpu-baseline --no-extra-legs
trace split_pooled python $ROOT/tools/experiments/split_mode_one_batch.py 2 4
trace 13b_fp8 python $ROOT/bench.py --model 13b --batch 16 --inflight 2 --weights fp8 --steps 2 --warmup 1 --no-cpu-baseline
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/${T}_pmc -o pmc -- python $ROOT/tools/kbench.py gemv_rows dattn_rows > $O/${T}_pmc_kbench.txt 2> $O/${T}_pmc.err
DB2=$(find $O/${T}_pmc -name "*.db" | head -1)
python $ROOT/tools/pmc_summary.py "$DB2" > $O/${T}_pmc_summary.txt 2>> $O/${T}_pmc.err
python $ROOT/tools/pmc_traffic.py "$DB2" $O/${T}_pmc_traffic.json >> $O/${T}_pmc_summary.txt 2>> $O/${T}_pmc.err
rm -rf $O/${T}_pmc
tail -12 $O/${T}_pmc_summary.txt
cd $ROOT
cp $O/${T}_pmc_traffic.json profiles/r04_pmc_traffic.json 2>/dev/null   # bench.py reads the newest profiles/rNN_pmc_traffic.json
timeout 600 python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err; echo "bench rc=$?"
python - <<PY
import json
try:
    r=json.loads([l for l in open("$O/${T}_bench_default.json") if l.startswith("{")][-1])
    print("value", r["value"], "pcie", r["pcie_inclusive"]["value"], "one", r["one_batch_at_a_time"]["value"], r["phase_ms_one_session"], "ids", r["ids_checked"])
    print("roofline", r["roofline"]["kernel"][:60], r["roofline"]["frac"], r["roofline"]["traffic"], "cpu", r.get("cpu_baseline",{}).get("value"), r.get("cpu_baseline",{}).get("cores"))
    pm=r.get("parity_mode",{})
    print("split", pm.get("split",{}).get("value"), pm.get("split",{}).get("frac_of_fast_path"), pm.get("split",{}).get("ids_equal_strict"))
    for k in ("c3_13b_bf16_b16","c5_slice_13b_fp8_b16"):
        print(k, r[k]["value"], r[k].get("parity_mode",{}).get("split",{}).get("value"), r[k].get("parity_mode",{}).get("split",{}).get("frac_of_fast_path"))
except Exception as e: print("bench failed", e)
PY
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/${T}_smoke.log
