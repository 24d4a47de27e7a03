"""Round 6: the fp8 weight format (BASELINE configs[4]: e4m3 weights, W8A8 prefill on the scaled fp8 MFMA,
e4m3 KV cache) at FULL 13b depth against the ORACLE in its fp8 mode on the effective (dequantised) weights — not against the
device's own bf16 path.  B = 1, the C2 prompt, 8 tokens teacher-forced with the device's ids: the oracle prefill (e4m3 activation
rows) + 7 cached steps (bf16 activations over the e4m3 cache).  Prints the deviation, the logit correlation, and the same two
figures for the oracle against ITSELF with fp32 instead of bf16-emulating arithmetic (what a rounding-level perturbation costs
in this format at this depth — the bound for any implementation).  usage: python tools/experiments/fp8_full_depth_oracle.py [layers]"""
import sys
import time

import numpy as np
import torch

sys.path[:0] = ['/root/repo', '/root/repo/oracle', '/root/repo/tests']
import cpu_ref  # noqa: E402
from device_weights import device_state_dict  # noqa: E402
from vcoder_amd import config as vcfg, synth  # noqa: E402
from vcoder_amd.engine import HipEngine  # noqa: E402

cpu_ref.fit_threads()
cfg = vcfg.vicuna_13b("vcoder_ds")
if len(sys.argv) > 1:
    cfg.num_hidden_layers = int(sys.argv[1])
n_new = 8
ids = synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=0)[None]
imgs, segs, deps = synth.synth_batch(1, 336)
t0 = time.time()
eng = HipEngine(cfg)
eng.load_synthetic(42)
eng.set_weight_format("fp8")
eng.finalize()
last, _, S = eng.prefill(ids, imgs, segs, deps)
logits, toks = [last], [np.argmax(last, -1).astype(np.int32)]
for s_ in range(1, n_new):
    lg, nxt = eng.decode_step(toks[-1])
    logits.append(lg)
    toks.append(nxt)
dev_logits, dev_ids = np.stack(logits, 1)[0], np.stack(toks, 1)[0]
sd = device_state_dict(eng, cfg, 42, effective_fp8=True)
eng.close()
print(f"device fp8 run + effective weights on the host: {time.time() - t0:.0f}s; ids {dev_ids.tolist()}", flush=True)


def oracle(emu):
    om = cpu_ref.OracleModel(cfg, sd, emu_bf16=emu, act_fp8=True)
    t = torch.from_numpy
    out = []
    with torch.no_grad():
        lg, cache = om.forward(ids.tolist(), t(imgs), t(segs), t(deps), last_only=True)
        out.append(lg[0, -1].numpy())
        for s_ in range(1, n_new):
            out.append(om.decode_step([int(dev_ids[s_ - 1])], cache)[0, -1].numpy())
    return np.stack(out, 0)


def corr(a, b):
    a, b = a - a.mean(-1, keepdims=True), b - b.mean(-1, keepdims=True)
    return (a * b).sum(-1) / np.sqrt((a * a).sum(-1) * (b * b).sum(-1))


t1 = time.time()
o16 = oracle(True)
print(f"oracle fp8 mode (bf16-emulating arithmetic): {time.time() - t1:.0f}s", flush=True)
scale = float(np.abs(o16).max())
d = np.abs(dev_logits - o16).max(-1)
c = corr(dev_logits, o16)
print(f"13b x {cfg.num_hidden_layers} layers, S={S}: device fp8 vs oracle fp8: |dlogit|max / |logit|max prefill {d[0] / scale:.3f}, cached steps {d[1:].max() / scale:.3f}; "
      f"correlation min {c.min():.4f} median {np.median(c):.4f}; argmax equal at {int((np.argmax(o16, -1) == dev_ids).sum())}/{n_new} steps; |logit|max {scale:.2f}", flush=True)
t1 = time.time()
o32 = oracle(False)
d2, c2 = np.abs(o32 - o16).max(-1), corr(o32, o16)
print(f"oracle fp8 mode, fp32 arithmetic vs bf16-emulating arithmetic (the format's own sensitivity to rounding-level perturbations): prefill "
      f"{d2[0] / scale:.3f}, cached steps {d2[1:].max() / scale:.3f}; correlation min {c2.min():.4f} median {np.median(c2):.4f}  ({time.time() - t1:.0f}s)", flush=True)
