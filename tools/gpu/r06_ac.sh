#!/bin/bash
# round 6, call AC: the activation-row e4m3 encode on the hardware's v_cvt_pk_fp8_f32 (rmsnorm_q8, quant_act_rows): exhaustive byte
# equality with the host's software encode, the fp8 parity tests, then the 13b fp8 leg and its kernel times
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_fulldepth.py -q -x -m gpu -k "norm or f8 or fp8 or q8 or kv8 or e4m3" 2>&1 | tail -4 | tee gpurun_out/r06_ac_pytest.txt
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06_ac_trace8 -o ks -- python $GRAFT_REPO_ROOT/bench.py --model 13b --batch 16 --inflight 2 --weights fp8 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r06_ac_trace8_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_ac_trace8.err
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/r06_ac_trace8 -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" gpurun_out/r06_ac_kernel_stats_13b_fp8.md > /dev/null 2>> gpurun_out/r06_ac_trace8.err
rm -rf gpurun_out/r06_ac_trace8
grep "rmsnorm_q8\|quant_act" gpurun_out/r06_ac_kernel_stats_13b_fp8.md
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_ac_trace8_bench.json").read().strip().splitlines()[-1])
print("13b fp8 B=16 x 2 in flight (under rocprof):", round(d["value"], 3), d.get("phase_ms_one_session"), "ids", d.get("ids_checked"))
PY
