#!/bin/bash
# round 5, call N: lone batch with / without the in-situ stamps after the fold kernel's load restructuring; pool tests
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out; T=r05_n
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "pool" 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-extra-legs > $O/${T}_one_insitu.json 2>/dev/null
timeout 300 python bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-extra-legs --no-insitu > $O/${T}_one_noinsitu.json 2>/dev/null
python - <<PY
import json
for f in ("one_insitu","one_noinsitu"):
    q=json.loads([l for l in open("$O/${T}_%s.json"%f) if l.startswith("{")][-1])
    print(f, round(q["value"],3), q["phase_ms_one_session"], round(q["roofline"]["frac"],4), q["roofline"]["measured"][:12])
PY
done
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${T}_tr -o ks -- python $ROOT/bench.py --steps 1 --warmup 1 --inflight 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $O/${T}_tr.err
DB=$(find $O/${T}_tr -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py "$DB" $O/${T}_kernel_stats_one_batch.md > /dev/null 2>> $O/${T}_tr.err
rm -rf $O/${T}_tr
grep "stamp_\|select_embed" $O/${T}_kernel_stats_one_batch.md
