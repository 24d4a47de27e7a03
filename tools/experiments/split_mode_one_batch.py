#!/usr/bin/env python
"""Batches of BASELINE configs[1] in precision mode "split" (for rocprofv3 --kernel-trace: the split kernels' durations).
usage: python tools/experiments/split_mode_one_batch.py [steps] [inflight]   (inflight > 1: that many generate() calls share the
decode pool — 32-row steps, one weight pass each, at 4)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from vcoder_amd import config as vcfg, synth  # noqa: E402
from vcoder_amd.engine import HipEngine  # noqa: E402

import threading  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = vcfg.vicuna_7b("vcoder_ds")
eng = HipEngine(cfg)
eng.load_synthetic(42)
eng.finalize()
eng.set_precision("split")
B = 8
ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=b) for b in range(B)])
px = tuple(torch.from_numpy(a).cuda() for a in synth.synth_batch(B, 336))
sessions = [eng] + [eng.fork() for _ in range(inflight - 1)]
for s_ in sessions[1:]:
    s_.set_precision("split")
outs = [None] * inflight


def work(i):
    for _ in range(steps):
        outs[i] = sessions[i].generate_greedy(ids, *px, max_new_tokens=128, eos_token_id=None)


ths = [threading.Thread(target=work, args=(i,)) for i in range(inflight)]
for t in ths:
    t.start()
for t in ths:
    t.join()
torch.cuda.synchronize()
assert all(np.array_equal(o, outs[0]) for o in outs)
print(eng.last_timings(), outs[0][0, :8].tolist())
