"""The N>1 path on CPU: 2 and 8 processes (gloo), contiguous batch shards, one all-gather of the generated ids, no other
collective — gathered stream == the single-process stream for the same global batch.  8 ranks is the world
the driver's scaling run uses (BASELINE configs[3] / [4])."""
import os
import subprocess
import sys

import numpy as np
import pytest

from vcoder_amd.parallel import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, ctypes, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as dist
from vcoder_amd import config as vcfg, synth
from vcoder_amd.engine import HipEngine
from vcoder_amd.parallel import shard_range, gather_token_ids
import kernel_cases as kc
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cfg = vcfg.tiny("vcoder_ds")
eng = HipEngine(cfg, lib=kc.EmuBackend().lib)          # emulator injection: CPU test only
eng.load_synthetic(42); eng.finalize()
GB = %(gb)d
lo, hi = shard_range(GB, rank, world)
ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", 5, 4, sample=s) for s in range(lo, hi)])
imgs, segs, deps = synth.synth_batch(hi - lo, cfg.vit_image_size, first=lo)
local = eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=3)
allv = gather_token_ids(local, dist)
if rank == 0:
    np.save(%(out)r, allv)
dist.barrier(); dist.destroy_process_group()
'''


def test_shard_range():
    assert [shard_range(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert shard_range(2, 3, 4) == (2, 2)


@pytest.mark.parametrize("world,gb,port", [(2, 4, 29617), (8, 16, 29631)])
def test_gather_equals_single_process(tmp_path, world, gb, port):
    out = str(tmp_path / "gathered.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": out, "gb": gb})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", VC_EMU_WORKERS="2" if world == 2 else "1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)], env=env, timeout=900)
    got = np.load(out)
    # single process over the whole global batch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import kernel_cases as kc
    from vcoder_amd import config as vcfg, synth
    from vcoder_amd.engine import HipEngine

    cfg = vcfg.tiny("vcoder_ds")
    eng = HipEngine(cfg, lib=kc.EmuBackend().lib)
    eng.load_synthetic(42)
    eng.finalize()
    ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", 5, 4, sample=s) for s in range(gb)])
    imgs, segs, deps = synth.synth_batch(gb, cfg.vit_image_size)
    ref = eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=3)
    assert got.shape == (gb, 3) and np.array_equal(got, ref)


def test_forced_gather_in_a_world_of_one(tmp_path):
    """gather_token_ids(force=True) — what `bench.py --gpus 1 --force-dist` uses to run the multi-GPU gather on one GPU — really
    runs the collective in a world of one (gloo here, nccl = RCCL under -m gpu) and returns the local rows unchanged"""
    script = tmp_path / "w1.py"
    script.write_text(r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import torch.distributed as dist
from vcoder_amd.parallel import gather_token_ids
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29623")
dist.init_process_group("gloo", rank=0, world_size=1)
calls = []
orig = dist.all_gather_into_tensor
dist.all_gather_into_tensor = lambda out, t, *a, **k: (calls.append(1), orig(out, t, *a, **k))[1]
x = np.arange(24, dtype=np.int32).reshape(3, 8)
assert np.array_equal(gather_token_ids(x, dist), x) and not calls            # default: no collective in a world of one
assert np.array_equal(gather_token_ids(x, dist, force=True), x) and calls == [1]
dist.destroy_process_group()
print("ok")
''' % ROOT)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
