#!/bin/bash
# round 4, GPU call O (host-side measurement only): fp32 matmul rate of the box's host cores by torch thread count — the
# oracle's teacher-forced pass (tests/test_gpu_fulldepth.py) and bench.py's cpu_baseline run on them
cd "$GRAFT_REPO_ROOT" || exit 1
{
lscpu | grep -E "^CPU\(s\)|Thread|Core|Socket|Model name|NUMA node\(s\)"
cat /sys/fs/cgroup/cpu.max 2>/dev/null
timeout 100 python - <<'PY'
import time, torch, os
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
a = torch.randn(2686, 4096); w = torch.randn(11008, 4096)
for n in (0, 64, 32, 16, 8):
    if n: torch.set_num_threads(n)
    torch.nn.functional.linear(a, w)
    t = time.time()
    for _ in range(3): torch.nn.functional.linear(a, w)
    dt = (time.time() - t) / 3
    print(f"threads {torch.get_num_threads():3d}: {dt*1e3:7.1f} ms  {2*2686*4096*11008/dt/1e12:.2f} TFLOP/s")
PY
} > gpurun_out/r04_o_host_matmul.txt 2>&1
cat gpurun_out/r04_o_host_matmul.txt
