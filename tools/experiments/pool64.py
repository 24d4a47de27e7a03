"""Round 6: measure a 64-row decode pool before arguing about it.  `rows` = 32: four generate() calls of batch 8 in
flight (the benchmark's configuration); 64: eight calls in flight, every pooled step = TWO 32-row weight passes + the decode
attention of 64 rows (vc_pool_set_rows).  Same model, same inputs, ids of every call checked against the lone call.
usage: python tools/experiments/pool64.py ROWS [STEPS_PER_SESSION]"""
import json
import sys
import threading
import time

import numpy as np
import torch

sys.path[:0] = ['/root/repo']
from vcoder_amd import config as vcfg, synth  # noqa: E402
from vcoder_amd.engine import HipEngine  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32
per = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B, N_new = 8, 128
cfg = vcfg.vicuna_7b("vcoder_ds")
eng = HipEngine(cfg)
eng.load_synthetic(42)
eng.finalize()
eng.pool_set_rows(rows)
n_sess = rows // B
sessions = [eng] + [eng.fork() for _ in range(n_sess - 1)]
ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=i) for i in range(B)])
px = [torch.from_numpy(a).cuda() for a in synth.synth_batch(B, 336)]
lone = eng.generate_greedy(ids, *px, max_new_tokens=N_new, eos_token_id=None)


def run(k):
    outs, errs = [None] * k, []

    def worker(si):
        try:
            for j in range(si, k, n_sess):
                outs[j] = sessions[si].generate_greedy(ids, *px, max_new_tokens=N_new, eos_token_id=None)
        except BaseException as e:
            errs.append(e)
    ths = [threading.Thread(target=worker, args=(si,)) for si in range(n_sess)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    if errs:
        raise errs[0]
    return outs


run(n_sess)
torch.cuda.synchronize()
c0 = eng.pool_step_counts()
t0 = time.perf_counter()
outs = run(per * n_sess)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
c1 = eng.pool_step_counts()
ok = all(np.array_equal(o, lone) for o in outs)
print(json.dumps({"pool_rows": rows, "calls_in_flight": n_sess, "batches": per * n_sess, "images_per_s": round(per * n_sess * B / dt, 3),
                  "ms_per_batch": round(dt / (per * n_sess) * 1e3, 2), "ids_equal_lone_call": ok,
                  "pool_steps_by_span(8,16,24,>=32 rows)": [a - b for a, b in zip(c1, c0)]}))
