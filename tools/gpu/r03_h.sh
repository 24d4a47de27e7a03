#!/bin/bash
# round 3, GPU call H: flash kernel after the V^T key-order change (one conflict-free ds_read_b128 per PV MFMA), permlane-swap
# row maxima and the skipped no-op rescale: kbench, parity subset, SQ counters of the new kernel, prefill phase of the bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_h; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/kbench.py attn > $O/kbench_attn.txt 2>&1
( time timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -m gpu -k "attention or qkv or fixture or split" ) > $O/pytest_subset.log 2>&1; echo "rc=$?" >> $O/pytest_subset.log
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/$O/pmcsq -o pmcsq -- python $GRAFT_REPO_ROOT/tools/kbench.py attn > $GRAFT_REPO_ROOT/$O/pmcsq_kbench.txt 2> $GRAFT_REPO_ROOT/$O/pmcsq.err )
DB3=$(find $O/pmcsq -name "*.db" | head -1)
python tools/pmc_summary.py "$DB3" > $O/pmc_sq_summary.txt 2>> $O/pmcsq.err
rm -rf $O/pmcsq
timeout 600 python bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-extra-legs > $O/bench_one_batch.json 2> $O/bench_one_batch.err; echo "rc=$?" >> $O/bench_one_batch.err
cat $O/kbench_attn.txt; tail -4 $O/pytest_subset.log; grep -A9 "attention_kernel" $O/pmc_sq_summary.txt | head -24
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03_h/bench_one_batch.json").read().strip().splitlines()[-1])
print("value", r["value"], "ids_checked", r["ids_checked"], r["phase_ms_one_session"])
PY
