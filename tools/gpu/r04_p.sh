#!/bin/bash
# round 4, GPU call P (host-side only): bench.py's cpu_baseline leg alone, its torch pool now sized by the cgroup CPU quota
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 100 python - > gpurun_out/r04_p_cpu_baseline.json 2> gpurun_out/r04_p_cpu_baseline.err <<'PY'
import json, time
import bench
from vcoder_amd import config as vcfg
t = time.time()
r = bench.cpu_baseline(vcfg.vicuna_7b("vcoder_ds"), 128, decode_steps=6)
r["wall_s"] = time.time() - t
print(json.dumps(r))
PY
echo rc=$?; cat gpurun_out/r04_p_cpu_baseline.json; tail -3 gpurun_out/r04_p_cpu_baseline.err
