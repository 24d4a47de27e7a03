#!/bin/bash
# round 5, call K: the round's profile artefacts on the final tree — rocprofv3 kernel traces of the bench command (pooled = the timed
# configuration, one batch, split mode pooled, 13b fp8) with the alone / co-run split of the decode-step kernels, a SEPARATE --pmc
# FETCH_SIZE pass over the decode-step kernels, a SEPARATE SQ pass (MFMA busy, LDS conflicts) over the GEMM / flash kernels.
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
O=$ROOT/gpurun_out; T=r05_k
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
trace split_pooled python $ROOT/tools/experiments/split_mode_one_batch.py 2 4
trace 13b_fp8 python $ROOT/bench.py --model 13b --batch 16 --inflight 2 --weights fp8 --steps 2 --warmup 1 --no-cpu-baseline
cp $O/${T}_pooled.out $O/${T}_bench_under_rocprof_pooled.json
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/${T}_pmc -o pmc -- python $ROOT/tools/kbench.py gemv_rows dattn_rows > $O/${T}_pmc_kbench.txt 2> $O/${T}_pmc.err
DB2=$(find $O/${T}_pmc -name "*.db" | head -1)
python $ROOT/tools/pmc_summary.py "$DB2" > $O/${T}_pmc_summary.txt 2>> $O/${T}_pmc.err
python $ROOT/tools/pmc_traffic.py "$DB2" $O/${T}_pmc_traffic.json >> $O/${T}_pmc_summary.txt 2>> $O/${T}_pmc.err
rm -rf $O/${T}_pmc
tail -8 $O/${T}_pmc_summary.txt
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/${T}_pmcsq -o pmcsq -- python $ROOT/tools/kbench.py gemm attn gemm_f8 > $O/${T}_pmcsq_kbench.txt 2> $O/${T}_pmcsq.err
DB3=$(find $O/${T}_pmcsq -name "*.db" | head -1)
python $ROOT/tools/pmc_summary.py "$DB3" > $O/${T}_pmc_sq_summary.txt 2>> $O/${T}_pmcsq.err
rm -rf $O/${T}_pmcsq
head -30 $O/${T}_pmc_sq_summary.txt
head -14 $O/${T}_kernel_stats_pooled.md
cat $O/${T}_alone_vs_corun_pooled.md | tail -3
