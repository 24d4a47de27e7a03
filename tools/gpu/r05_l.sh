#!/bin/bash
# round 5, call L: the pooled / one-batch kernel traces again with the decode attention's launches split by grid (rows) in the table
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out; T=r05_l
export TMPDIR=/tmp
cd /tmp
trace() {  # name, command...
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/${T}_tr_$name -o ks -- "$@" > $O/${T}_$name.out 2> $O/${T}_$name.err
  local DB=$(find $O/${T}_tr_$name -name "*.db" | head -1)
  python $ROOT/tools/rocpd_summary.py "$DB" $O/${T}_kernel_stats_$name.md > /dev/null 2>> $O/${T}_$name.err
  python $ROOT/tools/rocpd_overlap.py "$DB" $O/${T}_alone_vs_corun_$name.md > /dev/null 2>> $O/${T}_$name.err
  python -c "import sqlite3,sys; c=sqlite3.connect(sys.argv[1]); print([r[1] for r in c.execute('pragma table_info(kernels)')])" "$DB"
  rm -rf $O/${T}_tr_$name
  echo "trace $name done: $(head -c 300 $O/${T}_$name.out | tr '\n' ' ' | cut -c1-160)"
}
trace pooled python $ROOT/bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-extra-legs
trace one_batch python $ROOT/bench.py --steps 1 --warmup 1 --inflight 1 --no-cpu-baseline --no-extra-legs
cp $O/${T}_pooled.out $O/${T}_bench_under_rocprof_pooled.json
cp $O/${T}_one_batch.out $O/${T}_bench_under_rocprof_one_batch.json
head -16 $O/${T}_kernel_stats_pooled.md
head -12 $O/${T}_kernel_stats_one_batch.md
tail -4 $O/${T}_pooled.err
