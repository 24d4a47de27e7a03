#!/bin/bash
# Reproduces the per-round profile artefacts on an MI355X box (run from the repo root; outputs under gpurun_out/<tag>_*).
#   1. rocprofv3 --kernel-trace --stats of bench.py in its default (timed) configuration — 4 generate() calls in flight, decode
#      steps pooled — and of a lone batch (--inflight 1): per-kernel tables (tools/rocpd_summary.py)
#   2. SEPARATE rocprofv3 --pmc FETCH_SIZE passes over the two decode-step kernels at the row counts the bench runs them
#      (tools/kbench.py gemv_rows / dattn_rows; counters and traces are never combined) -> HBM bytes per launch
#      (tools/pmc_traffic.py, gfx950 x2 correction)
#   3. SQ counters of the MFMA-bound kernels (bf16 and e4m3 GEMM, prefill / ViT attention) in their own pass
#   4. kernel trace of the fp8 weight format on the 13b geometry (BASELINE configs[4])
# usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o ks -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
DB=$(find $OUT/${TAG}_trace -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py "$DB" $OUT/${TAG}_kernel_stats_pooled.md > /dev/null 2>> $OUT/${TAG}_trace.err
python $ROOT/tools/rocpd_overlap.py "$DB" $OUT/${TAG}_alone_vs_corun_pooled.md > /dev/null 2>> $OUT/${TAG}_trace.err
rm -rf $OUT/${TAG}_trace
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace1 -o ks -- python $ROOT/bench.py --steps 1 --warmup 1 --inflight 1 --no-cpu-baseline --no-extra-legs > $OUT/${TAG}_trace1_bench.json 2> $OUT/${TAG}_trace1.err
DB=$(find $OUT/${TAG}_trace1 -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py "$DB" $OUT/${TAG}_kernel_stats_one_batch.md > /dev/null 2>> $OUT/${TAG}_trace1.err
rm -rf $OUT/${TAG}_trace1
# precision mode "split": one batch of configs[1] (the split kernels' durations)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_traceS -o ks -- python $ROOT/tools/experiments/split_mode_one_batch.py 2 > $OUT/${TAG}_traceS.txt 2> $OUT/${TAG}_traceS.err
DB=$(find $OUT/${TAG}_traceS -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py "$DB" $OUT/${TAG}_kernel_stats_split_one_batch.md > /dev/null 2>> $OUT/${TAG}_traceS.err
rm -rf $OUT/${TAG}_traceS
# BASELINE configs[4] weight format on the 13b geometry: W8A8 prefill (scaled fp8 MFMA) + W8A16 decode steps
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace8 -o ks -- python $ROOT/bench.py --model 13b --batch 16 --inflight 2 --weights fp8 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_trace8_bench.json 2> $OUT/${TAG}_trace8.err
DB=$(find $OUT/${TAG}_trace8 -name "*.db" | head -1)
python $ROOT/tools/rocpd_summary.py "$DB" $OUT/${TAG}_kernel_stats_13b_fp8.md > /dev/null 2>> $OUT/${TAG}_trace8.err
rm -rf $OUT/${TAG}_trace8
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_pmc -o pmc -- python $ROOT/tools/kbench.py gemv_rows dattn_rows > $OUT/${TAG}_pmc_kbench.txt 2> $OUT/${TAG}_pmc.err
DB2=$(find $OUT/${TAG}_pmc -name "*.db" | head -1)
python $ROOT/tools/pmc_summary.py "$DB2" > $OUT/${TAG}_pmc_summary.txt 2>> $OUT/${TAG}_pmc.err
python $ROOT/tools/pmc_traffic.py "$DB2" $OUT/${TAG}_pmc_traffic.json >> $OUT/${TAG}_pmc_summary.txt 2>> $OUT/${TAG}_pmc.err
rm -rf $OUT/${TAG}_pmc
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/${TAG}_pmcsq -o pmcsq -- python $ROOT/tools/kbench.py gemm attn gemm_f8 > $OUT/${TAG}_pmcsq_kbench.txt 2> $OUT/${TAG}_pmcsq.err
DB3=$(find $OUT/${TAG}_pmcsq -name "*.db" | head -1)
python $ROOT/tools/pmc_summary.py "$DB3" > $OUT/${TAG}_pmc_sq_summary.txt 2>> $OUT/${TAG}_pmcsq.err
python $ROOT/tools/pmc_mfma_busy.py $OUT/${TAG}_pmc_sq_summary.txt $OUT/${TAG}_mfma_busy.md > /dev/null 2>> $OUT/${TAG}_pmcsq.err
rm -rf $OUT/${TAG}_pmcsq
head -14 $OUT/${TAG}_kernel_stats_pooled.md; tail -30 $OUT/${TAG}_pmc_summary.txt
