"""TEST / ANALYSIS INFRASTRUCTURE — go / no-go of an fp16-operand fast path ("next round" item 5).

The reference computes on the GPU in fp16 (vcoder_llava/model/builder.py:39 `torch_dtype=torch.float16`, :142 the tower cast to
fp16): 11 significand bits.  The benchmarked path of this repo rounds every MFMA operand — weights AND activations — to bf16 (8
bits).  gfx950's v_mfma_f32_16x16x32_f16 runs at the bf16 rate on the same byte layout, so an fp16-operand mode would cost no
throughput; the question this tool answers on the CPU before any kernel is touched: what would it buy?

It runs oracle/cpu_ref.py at the TRUE 7b / ViT-L dimensions (23 ViT layers, adapters, the C2 prompt S = 1216, the first L decoder
layers + final norm + lm_head, one sample) on the checkpoint with the REFERENCE'S value classes (fp16-valued LLM / projectors,
fp32-valued tower: synth_state_dict(dtypes="reference")) three times:
    exact   fp32 arithmetic on the original values (the reference's CPU path — the oracle)
    bf16    every rounding site of the device's fast path rounded to bf16, matrices rounded to bf16
    fp16    the same sites rounded to fp16 (saturating at 65504 as v_cvt_pk_f16_f32 / the reference's .half() do not: overflow is
            REPORTED, not hidden), matrices rounded to fp16 (exact for the fp16-valued tensors; the fp32 tower is rounded, as the
            reference's own .to(float16) does)
and prints |dlogit|max at the last prompt position relative to |logit|max.

    python oracle/fp16_go_nogo.py --layers 8 16 32 [--out profiles/r06_fp16_go_nogo.txt]

CPU only; nothing in the product imports this."""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import cpu_ref  # noqa: E402
from vcoder_amd import config as vcfg, synth  # noqa: E402


class FmtRounder(cpu_ref.Rounder):
    """every rounding site of the fast path in one storage format; counts values an fp16 operand could not hold"""

    def __init__(self):
        super().__init__(True)
        self.fmt = "exact"
        self.overflow = 0
        self.absmax = 0.0

    def __call__(self, x):
        if self.fmt == "exact":
            return x
        if self.fmt == "bf16":
            return x.to(torch.bfloat16).to(torch.float32)
        m = float(x.abs().max())
        self.absmax = max(self.absmax, m)
        if m > 65504.0:
            self.overflow += int((x.abs() > 65504.0).sum())
        return x.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)


class RoundedWeights(dict):
    """the checkpoint with every matrix rounded to the operand format WHEN IT IS READ (no second copy of a 27-GB model); vectors
    (norm weights, biases, CLS / position embeddings) stay fp32 as on the device"""

    def __init__(self, sd, R):
        super().__init__(sd)
        self.R = R

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if self.R.fmt == "exact" or v.dim() < 2 or "position_embedding" in k:
            return v
        return v.to(torch.bfloat16 if self.R.fmt == "bf16" else torch.float16).to(torch.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, nargs="+", default=[8])
    ap.add_argument("--out", default=None)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    say("# fp16-operand fast path, go / no-go on the oracle (true 7b / ViT-L dims, S = 1216, one sample, last-position logits,")
    say("# checkpoint with the reference's value classes: fp16-valued LLM / projectors, fp32-valued CLIP tower)")
    say(f"{'decoder layers':>14s} {'|logit|max':>10s} {'bf16 operands':>24s} {'fp16 operands':>24s} {'ratio':>6s} {'fp16 |operand|max':>18s} {'overflows':>9s}")
    R = FmtRounder()
    cpu_ref.Rounder = lambda *args, **kw: R
    for L in a.layers:
        cfg = vcfg.vicuna_7b("vcoder_ds")
        cfg.num_hidden_layers = L
        t0 = time.time()
        keep = lambda k: not ("depth_mm_projector" in k or "mm2_projector" in k or "vcoder_lm_emb" in k)
        sd = {}
        for key, shape, off, hw in synth.tensor_specs(cfg):
            if keep(key):
                sd[key] = torch.from_numpy(synth.synth_tensor(key, shape, 42, off, hw, synth.reference_rounding(key))).float()
        ids = synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=0)[None]
        imgs, segs, deps = (torch.from_numpy(x) for x in synth.synth_batch(1, cfg.vit_image_size))
        om = cpu_ref.OracleModel(cfg, sd, emu_bf16=True)
        om.sd = RoundedWeights(om.sd, R)

        def run(fmt):
            R.fmt, R.overflow, R.absmax = fmt, 0, 0.0
            with torch.no_grad():
                x, _ = om.prepare_inputs(ids.tolist(), imgs, segs, deps)
                lg = cpu_ref.llama_forward(x, om.sd, cfg, cpu_ref.KVCache(cfg.num_hidden_layers), True, last_only=True)
            return lg[0, -1].numpy().astype(np.float64)

        ref = run("exact")
        scale = np.abs(ref).max()
        e16 = np.abs(run("bf16") - ref).max()
        ef = np.abs(run("fp16") - ref).max()
        say(f"{L:14d} {scale:10.3f} {e16:12.3e} = {e16 / scale:8.2e} {ef:12.3e} = {ef / scale:8.2e} {e16 / ef:6.1f} {R.absmax:18.1f} {R.overflow:9d}"
            f"   # {time.time() - t0:.0f}s")
        del om, sd
    say("# go if the fp16 column is <= 5e-3 of |logit|max at 32 layers (bf16: 3.0e-2 measured on the device at full depth)")
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
