#!/usr/bin/env python
"""Experiment: N independent engine replicas on ONE GPU, each on its own HIP stream / host thread, generating
concurrently (prefill of one batch overlaps decode of another).  usage: python tools/experiments/inflight_replicas.py [n_inflight] [steps]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from vcoder_amd import config as vcfg, synth  # noqa: E402
from vcoder_amd.engine import HipEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = vcfg.vicuna_7b("vcoder_ds")
B = 8
engs = []
for i in range(n):
    e = HipEngine(cfg)
    e.load_synthetic(42)
    e.finalize()
    engs.append(e)
ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=b) for b in range(B)])
imgs, segs, deps = (torch.from_numpy(a).cuda() for a in synth.synth_batch(B, 336))
for e in engs:
    e.generate_greedy(ids, imgs, segs, deps, max_new_tokens=128)
torch.cuda.synchronize()


def worker(e, k, delay):
    time.sleep(delay)
    for _ in range(k):
        e.generate_greedy(ids, imgs, segs, deps, max_new_tokens=128)


for stagger in (0.0, 0.35):
    ths = [threading.Thread(target=worker, args=(e, steps, i * stagger)) for i, e in enumerate(engs)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"inflight={n} stagger={stagger}: {n * steps * B / dt:.2f} images/s  ({dt / (n * steps) * 1e3:.1f} ms per batch, "
          f"last timings {engs[0].last_timings()})", flush=True)
