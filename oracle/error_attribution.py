"""TEST / ANALYSIS INFRASTRUCTURE — which bf16 rounding sites of the hot path carry the logit error of the bf16 path?

VERDICT r03 item 1(a): before deciding what the bar-meeting ("split") mode has to carry as bf16 hi + lo, rank the rounding
sites by what each contributes to the deviation from the fp32 reference.  oracle/cpu_ref.py rounds activations to bf16 exactly
where the device's bf16 path does; this tool runs the oracle at the TRUE 7b / ViT-L dimensions
(23 ViT layers, adapters, the C2 prompt S = 1216, the first L decoder layers + final norm + lm_head, one sample) with the
rounding enabled for ONE site group at a time (everything else fp32) and with it enabled everywhere EXCEPT one group, and
reports |dlogit|max at the last prompt position against the all-fp32 pass.

    python oracle/error_attribution.py --layers 8 [--out profiles/r04_error_attribution.txt]

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

VIT_SITES = ["vit.ln1", "vit.q", "vit.k", "vit.v", "vit.P", "vit.attn_o", "vit.ln2", "vit.gelu"]
LLM_SITES = ["llm.in_norm", "llm.q", "llm.k", "llm.v", "llm.q_rope", "llm.k_rope", "llm.P", "llm.attn_o", "llm.post_norm",
             "llm.swiglu"]
GROUPS = {
    "vit (all 23 layers + im2col + feature cast)": ["vit.cols", "vit.feats"] + VIT_SITES,
    "adapters (gelu hidden + output)": ["proj.gelu", "proj.out"],
    "llm RMSNorm outputs (qkv / gate-up GEMM inputs)": ["llm.in_norm", "llm.post_norm"],
    "llm q, k before and after RoPE": ["llm.q", "llm.k", "llm.q_rope", "llm.k_rope"],
    "llm v": ["llm.v"],
    "llm softmax numerators P": ["llm.P"],
    "llm attention output (o_proj input)": ["llm.attn_o"],
    "llm SwiGLU hidden (down_proj input)": ["llm.swiglu"],
    "llm final norm (lm_head input)": ["llm.final_norm"],
}
ALL_SITES = sorted({s for g in GROUPS.values() for s in g})


class SiteRounder(cpu_ref.Rounder):
    """names the i-th rounding call of the current context (the order of the r(...) calls in oracle/cpu_ref.py) and rounds only
    the enabled sites"""

    def __init__(self):
        super().__init__(True)
        self.enabled = set()
        self.ctx, self.i = "llm", 0

    def begin(self, ctx):
        self.ctx, self.i = ctx, 0

    def name(self):
        i = self.i
        if self.ctx == "vit":
            if i == 0:
                return "vit.cols"
            j = i - 1
            return VIT_SITES[j % 8] if j < 8 * self.vit_layers else "vit.feats"
        if self.ctx == "proj":
            return "proj.gelu" if i % 2 == 0 else "proj.out"
        j = i
        return LLM_SITES[j % 10] if j < 10 * self.llm_layers else "llm.final_norm"

    def __call__(self, x):
        n = self.name()
        self.i += 1
        return cpu_ref._bf16(x) if n in self.enabled else x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--out", default=None)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    cfg = vcfg.vicuna_7b("vcoder_ds")
    cfg.num_hidden_layers = a.layers
    t0 = time.time()
    keep = lambda k: not ("depth_mm_projector" in k or "mm2_projector" in k or "vcoder_lm_emb" in k)
    sd = {}
    for key, shape, off, hw in synth.tensor_specs(cfg):
        if keep(key):
            sd[key] = torch.from_numpy(synth.synth_tensor(key, shape, 42, off, hw)).float() if hasattr(synth, "synth_tensor") else None
    if any(v is None for v in sd.values()):
        full = synth.synth_state_dict(cfg, 42)
        sd = {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in full.items() if keep(k)}
    print(f"weights ({a.layers} decoder layers) in {time.time() - t0:.0f}s", flush=True)
    ids = synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=0)[None]
    imgs, segs, deps = (torch.from_numpy(x) for x in synth.synth_batch(1, cfg.vit_image_size))
    R = SiteRounder()
    R.vit_layers, R.llm_layers = cfg.vit_layers_used, a.layers
    cpu_ref.Rounder = lambda *args, **kw: R    # every pass of the oracle takes the shared, site-naming rounder

    orig_vit, orig_proj, orig_llama = cpu_ref.vit_forward, cpu_ref.projector_forward, cpu_ref.llama_forward

    def vit(*args, **kw):
        R.begin("vit")
        return orig_vit(*args, **kw)

    def proj(*args, **kw):
        R.begin("proj")
        return orig_proj(*args, **kw)

    def llama(*args, **kw):
        R.begin("llm")
        return orig_llama(*args, **kw)

    cpu_ref.vit_forward, cpu_ref.projector_forward, cpu_ref.llama_forward = vit, proj, llama
    om = cpu_ref.OracleModel(cfg, sd, emu_bf16=True)

    def run(enabled):
        R.enabled = set(enabled)
        with torch.no_grad():
            x, _ = om.prepare_inputs(ids.tolist(), imgs, segs, deps)
            lg = cpu_ref.llama_forward(x, om.sd, cfg, cpu_ref.KVCache(cfg.num_hidden_layers), True, last_only=True)
        return lg[0, -1].numpy().astype(np.float64)

    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    t1 = time.time()
    ref = run([])
    say(f"# error attribution, 7b dims, {a.layers} decoder layers + 23 ViT layers, S = {ids.shape[1] - 3 + 2 * cfg.num_patches}, "
        f"last-position logits; |logit|max = {np.abs(ref).max():.3f}; one pass {time.time() - t1:.0f}s")
    allr = run(ALL_SITES)
    e_all = np.abs(allr - ref).max()
    say(f"all sites rounded to bf16 (the bf16 path)              : |dlogit|max {e_all:.3e}")
    say(f"{'site group':55s} {'ONLY this group bf16':>22s} {'all BUT this group bf16':>26s}")
    for name, sites in GROUPS.items():
        only = np.abs(run(sites) - ref).max()
        but = np.abs(run([s for s in ALL_SITES if s not in sites]) - ref).max()
        say(f"{name:55s} {only:22.3e} {but:26.3e}")
    # the two candidate reduced modes of VERDICT item 1(a)
    cand = {
        "split everything but P (softmax numerators single-plane)": ["llm.P"],
        "split everything but P and the ViT + adapters": ["llm.P", "vit.cols", "vit.feats", "proj.gelu", "proj.out"] + VIT_SITES,
        "split only the residual-feeding inputs (attn_o, swiglu) and q/k": [s for s in ALL_SITES if s not in
                                                                           ("llm.attn_o", "llm.swiglu", "llm.q", "llm.k", "llm.q_rope", "llm.k_rope")],
    }
    say("# candidate reduced modes: the listed sites stay single-plane bf16, every other site exact")
    for name, sites in cand.items():
        say(f"{name:70s} |dlogit|max {np.abs(run(sites) - ref).max():.3e}")
    say(f"# total {time.time() - t0:.0f}s on {torch.get_num_threads()} threads")
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
