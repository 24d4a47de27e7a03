"""TEST INFRASTRUCTURE (build container only) — randomised differential check of oracle/cpu_ref.py against the LIVE reference
(/root/reference through oracle/ref_shim.py) on prompt STRUCTURES: random placeholder multisets and orders, text between
placeholders, modalities present or None, batches with unequal spliced lengths, with and without an attention_mask.

For every case both sides either raise (the exception class names must agree — the oracle restates the reference's quirks,
e.g. the UnboundLocalError of vcoder_ds_llava_arch.py:297) or return inputs_embeds / prefill logits that agree to fp32
round-off.  Nothing is written to tests/golden: the fixtures there are the committed vectors; this script is the wider net
that says the restatement of the splice (vcoder_ds_llava_arch.py:120-327 and its two siblings) has no untested corner.

    python oracle/fuzz_vs_reference.py [--cases 300] [--seed 0] [--variants vcoder_ds vcoder llava]
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import cpu_ref  # noqa: E402
import gen_golden  # noqa: E402
from vcoder_amd import config as vcfg  # noqa: E402
from vcoder_amd import synth  # noqa: E402

from fuzz_cases import random_case, random_overrides  # noqa: E402


def run_reference(model, cfg, rows, imgs, segs, deps, mask):
    ids = torch.tensor(rows, dtype=torch.long)
    t = lambda a: None if a is None else ([torch.from_numpy(x) for x in a] if isinstance(a, list) else torch.from_numpy(a))
    kw = {"images": t(imgs)}
    if cfg.variant != "llava":
        kw["segs"] = t(segs)
    if cfg.variant == "vcoder_ds":
        kw["depths"] = t(deps)
    with torch.no_grad():
        out = model(input_ids=ids, attention_mask=None if mask is None else torch.from_numpy(mask), use_cache=False, **kw)
    return out.logits.float().numpy()


def cached_steps_reference(model, cfg, rows, imgs, segs, deps, mask, keep: bool, n_steps: int = 3):
    """prefill with use_cache, then n cached greedy steps in the (B) form of SURVEY.md section 8(c): forward(input_ids=[[t]],
    attention_mask=..., past_key_values=pkv) without images — the mask all ones (what generate() runs) or the caller's
    extended mask carried through (`keep`).  -> step logits [B, n, V]"""
    ids = torch.tensor(rows, dtype=torch.long)
    t = lambda a: None if a is None else ([torch.from_numpy(x) for x in a] if isinstance(a, list) else torch.from_numpy(a))
    kw = {"images": t(imgs)}
    if cfg.variant != "llava":
        kw["segs"] = t(segs)
    if cfg.variant == "vcoder_ds":
        kw["depths"] = t(deps)
    am = None if mask is None else torch.from_numpy(mask)
    with torch.no_grad():
        out = model(input_ids=ids, attention_mask=am, use_cache=True, **kw)
        L = out.logits.shape[1]
        ext = torch.ones(len(rows), L, dtype=torch.long)
        if keep and am is not None:
            ext = torch.cat([torch.ones(len(rows), L - am.shape[1], dtype=torch.long), am], 1)   # left-extended by position
        pkv, last, lgs = out.past_key_values, out.logits[:, -1].float(), []
        for step in range(n_steps):
            nxt = last.argmax(-1)
            step_mask = torch.cat([ext, torch.ones(len(rows), step + 1, dtype=torch.long)], 1)
            o = model(input_ids=nxt[:, None], attention_mask=step_mask, past_key_values=pkv, use_cache=True)
            pkv, last = o.past_key_values, o.logits[:, -1].float()
            lgs.append(last.numpy())
    return np.stack(lgs, 1)


def cached_steps_oracle(oracle, rows, imgs, segs, deps, mask, keep: bool, n_steps: int = 3):
    t = lambda a: None if a is None else ([torch.from_numpy(x) for x in a] if isinstance(a, list) else torch.from_numpy(a))
    lg, cache = oracle.forward(rows, t(imgs), t(segs), t(deps), attention_mask=mask, last_only=True)
    last, lgs = lg[:, -1], []
    for _ in range(n_steps):
        last = oracle.decode_step(last.argmax(-1).tolist(), cache, keep_mask=keep)[:, -1]
        lgs.append(last.numpy())
    return np.stack(lgs, 1)


def run_oracle(oracle, rows, imgs, segs, deps, mask):
    t = lambda a: None if a is None else ([torch.from_numpy(x) for x in a] if isinstance(a, list) else torch.from_numpy(a))
    lg, _ = oracle.forward(rows, t(imgs), t(segs), t(deps), attention_mask=mask)
    return lg.numpy()


def outcome(fn):
    try:
        return "ok", fn()
    except Exception as e:   # noqa: BLE001 — the class of the failure is the datum
        return type(e).__name__, str(e)[:100]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--variants", nargs="+", default=["vcoder_ds", "vcoder", "llava"])
    ap.add_argument("--configs", type=int, default=0, help="random config variations per variant, after the tiny config itself")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for vi, variant in enumerate(v_ for v_ in args.variants for _ in range(1 + args.configs)):
            rng = np.random.RandomState(args.seed + 1000 * vi)
            cfg = vcfg.tiny(variant)
            over = random_overrides(rng, variant) if vi % (1 + args.configs) else {}   # the tiny config itself, then variations
            for k_, v_ in over.items():
                setattr(cfg, k_, v_)
            clip_dir = os.path.join(tmp, "clip_%d" % vi)
            gen_golden.make_clip_dir(cfg, clip_dir)
            sd = synth.synth_state_dict(cfg, gen_golden.SEED)
            model = gen_golden.build_reference_model(cfg, sd, clip_dir)
            oracle = cpu_ref.OracleModel(cfg, sd)
            stats, n_cached = {}, 0
            if over:
                print(f"[{variant}] config {over}", flush=True)
            for c in range(args.cases if not over else max(8, args.cases // 6)):
                rows, imgs, segs, deps, mask = random_case(rng, cfg)
                k_ref, v_ref = outcome(lambda: run_reference(model, cfg, rows, imgs, segs, deps, mask))
                k_or, v_or = outcome(lambda: run_oracle(oracle, rows, imgs, segs, deps, mask))
                key = k_ref if k_ref == k_or else f"MISMATCH ref={k_ref} oracle={k_or}"
                if k_ref == k_or == "ok":
                    if v_ref.shape != v_or.shape:
                        key = f"MISMATCH shapes ref={v_ref.shape} oracle={v_or.shape}"
                    else:
                        err = float(np.abs(v_ref - v_or).max())
                        if not err < 2e-4:
                            key = f"MISMATCH logits |d|={err:.2e}"
                        elif imgs is not None and c % 4 == 0:    # the cached steps behind it, both mask forms
                            for keep in (False, True):
                                a = cached_steps_reference(model, cfg, rows, imgs, segs, deps, mask, keep)
                                b = cached_steps_oracle(oracle, rows, imgs, segs, deps, mask, keep)
                                e2 = float(np.abs(a - b).max())
                                if not e2 < 2e-4:
                                    key = f"MISMATCH cached steps (keep={keep}) |d|={e2:.2e}"
                            n_cached += 1
                stats[key] = stats.get(key, 0) + 1
                if key.startswith("MISMATCH"):
                    bad += 1
                    print(f"[{variant} #{c}] {key}\n   rows={rows} img={imgs is not None} seg={segs is not None} "
                          f"depth={deps is not None} list={isinstance(imgs, list) or isinstance(segs, list)} mask={None if mask is None else mask.tolist()}\n"
                          f"   ref: {v_ref if k_ref != 'ok' else 'ok'}\n   oracle: {v_or if k_or != 'ok' else 'ok'}", flush=True)
            print(f"[{variant}] {sum(stats.values())} cases:", dict(sorted(stats.items())), f"(+ cached steps on {n_cached})", flush=True)
    print("fuzz_vs_reference:", "ALL AGREE" if bad == 0 else f"{bad} MISMATCHES")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
