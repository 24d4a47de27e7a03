#!/bin/bash
# round 6, call C: (1) the driver-shaped bench line of the tree as round 5 left it (this round's baseline on today's box);
# (2) the same timed leg under the CPU share ONE OF EIGHT ranks gets on these boxes (16-CPU quota / 8 ranks = 2 CPUs): taskset -c 0-1
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_c_bench_driver_shaped.json 2> gpurun_out/r06_c_bench_driver_shaped.err
tail -c 600 gpurun_out/r06_c_bench_driver_shaped.err
timeout 400 taskset -c 0-1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > gpurun_out/r06_c_bench_two_cpu_rank.json 2> gpurun_out/r06_c_bench_two_cpu_rank.err
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline > gpurun_out/r06_c_bench_all_cpus.json 2> gpurun_out/r06_c_bench_all_cpus.err
python - <<'PY'
import json
for f in ("r06_c_bench_driver_shaped", "r06_c_bench_two_cpu_rank", "r06_c_bench_all_cpus"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d.get("ms_per_step"), d.get("one_batch_at_a_time"), d.get("phase_ms_one_session"))
    except Exception as e:
        print(f, "ERR", e)
PY
