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


COST_WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import torch.distributed as dist
import kernel_cases as kc, e2e_cases
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
setup = e2e_cases.cost_setup(lib=kc.EmuBackend().lib, mode="split")          # emulator injection: CPU test only
out_dir = %(out)r
if %(named)r:   # world 2: the two 2-chunk runs the reference's loaders wrote golden files for
    name = ["depth_2_0", "panoptic_2_1"][rank]
    path = e2e_cases.cost_run_chunk(setup, os.path.join(out_dir, name), name=name)
else:           # one chunk per rank (scripts/v1_5/eval/cost_depth.sh:10-25: one loader process per GPU, --num-chunks = #GPUs)
    path = e2e_cases.cost_run_chunk(setup, out_dir, task="depth", num_chunks=world, chunk_idx=rank)
with open(os.path.join(out_dir, "rank%%d.path" %% rank), "w") as f:
    f.write(path)
dist.barrier()
if rank == 0 and not %(named)r:   # cost_depth.sh:31-34: the chunk files concatenated in rank order
    with open(os.path.join(out_dir, "merged.txt"), "w") as f:
        for r in range(world):
            f.write(open(open(os.path.join(out_dir, "rank%%d.path" %% r)).read()).read())
dist.barrier(); dist.destroy_process_group()
'''


def _launch(tmp_path, script_text, world, port):
    script = tmp_path / "worker.py"
    script.write_text(script_text)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", VC_EMU_WORKERS="1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)], env=env, timeout=1500)


def test_cost_harness_chunks_across_ranks(tmp_path):
    """The reference's multi-GPU COST evaluation (scripts/v1_5/eval/cost_depth.sh:10-34, eval/model_seg_loader.py:24-32): one loader
    process per GPU with --num-chunks = #GPUs, the chunk files concatenated in rank order.  (a) world 2: each rank's chunk file
    equals, byte for byte, the file the REFERENCE'S loader wrote for that chunk (tests/golden/cost: depth 2/0, panoptic 2/1);
    (b) world 8 over the 6-image folder (ranks 6 and 7 get empty chunks): the concatenation equals what ONE process writes when it
    runs the same eight chunk calls one after the other — every rank's answers are independent of who else is running."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import e2e_cases
    import kernel_cases as kc

    out2 = str(tmp_path / "w2")
    os.makedirs(out2)
    _launch(tmp_path, COST_WORKER % {"root": ROOT, "out": out2, "named": True}, 2, 29641)
    for rank, name in enumerate(("depth_2_0", "panoptic_2_1")):
        got = open(open(os.path.join(out2, f"rank{rank}.path")).read()).read()
        assert got == open(os.path.join(e2e_cases.COST_GOLD, f"answers_{name}.txt")).read(), name
    out8 = str(tmp_path / "w8")
    os.makedirs(out8)
    _launch(tmp_path, COST_WORKER % {"root": ROOT, "out": out8, "named": False}, 8, 29653)
    merged = open(os.path.join(out8, "merged.txt")).read()
    setup = e2e_cases.cost_setup(lib=kc.EmuBackend().lib, mode="split")
    one = ""
    for r in range(8):
        one += open(e2e_cases.cost_run_chunk(setup, str(tmp_path / "single"), task="depth", num_chunks=8, chunk_idx=r)).read()
    setup[0].engine.close()
    assert merged == one and merged.count("<<ANSWER>>:") == 6
