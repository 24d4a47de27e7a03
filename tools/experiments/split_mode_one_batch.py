#!/usr/bin/env python
"""One batch of BASELINE configs[1] in precision mode "split" (for rocprofv3 --kernel-trace: the split kernels' durations).
usage: python tools/experiments/split_mode_one_batch.py [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from vcoder_amd import config as vcfg, synth  # noqa: E402
from vcoder_amd.engine import HipEngine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = vcfg.vicuna_7b("vcoder_ds")
eng = HipEngine(cfg)
eng.load_synthetic(42)
eng.finalize()
eng.set_precision("split")
B = 8
ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=b) for b in range(B)])
px = tuple(torch.from_numpy(a).cuda() for a in synth.synth_batch(B, 336))
for _ in range(steps):
    out = eng.generate_greedy(ids, *px, max_new_tokens=128, eos_token_id=None)
torch.cuda.synchronize()
print(eng.last_timings(), out[0, :8].tolist())
