#!/bin/bash
# Reproduces the per-round profile artefacts on an MI355X box (run from the repo root; outputs under gpurun_out/<tag>_*).
#   1. rocprofv3 --kernel-trace --stats of one-batch-at-a-time bench.py -> per-kernel table (tools/rocpd_summary.py)
#   2. a SEPARATE rocprofv3 --pmc FETCH_SIZE pass over the decode GEMV shapes (tools/kbench.py gemv; counters and
#      traces are never combined) -> HBM bytes per launch (tools/pmc_traffic.py, gfx950 x2 correction)
# usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o ks -- python $ROOT/bench.py --steps 1 --warmup 1 --inflight 1 --no-cpu-baseline > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
DB=$(find $OUT/${TAG}_trace -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py "$DB" $OUT/${TAG}_kernel_stats.md > /dev/null 2>> $OUT/${TAG}_trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_pmc -o pmc -- python $ROOT/tools/kbench.py gemv > $OUT/${TAG}_pmc_kbench.txt 2> $OUT/${TAG}_pmc.err
DB2=$(find $OUT/${TAG}_pmc -name "*.db" | head -1)
python $ROOT/tools/pmc_summary.py "$DB2" gemv > $OUT/${TAG}_pmc_summary.txt 2>> $OUT/${TAG}_pmc.err
python $ROOT/tools/pmc_traffic.py "$DB2" $OUT/${TAG}_pmc_traffic.json >> $OUT/${TAG}_pmc_summary.txt 2>> $OUT/${TAG}_pmc.err
# 3. SQ counters of the MFMA-bound kernels (GEMM, prefill/ViT attention) in their own pass
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/${TAG}_pmcsq -o pmcsq -- python $ROOT/tools/kbench.py gemm attn > $OUT/${TAG}_pmcsq_kbench.txt 2> $OUT/${TAG}_pmcsq.err
DB3=$(find $OUT/${TAG}_pmcsq -name "*.db" | head -1)
python $ROOT/tools/pmc_summary.py "$DB3" > $OUT/${TAG}_pmc_sq_summary.txt 2>> $OUT/${TAG}_pmcsq.err
rm -rf $OUT/${TAG}_pmcsq
# keep the merge-back small: drop the raw databases
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc
head -12 $OUT/${TAG}_kernel_stats.md; cat $OUT/${TAG}_pmc_summary.txt | tail -40
