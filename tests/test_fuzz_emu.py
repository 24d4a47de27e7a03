"""Randomised differential test: the engine (HIP sources on the lane emulator, precision modes "strict" and "split") against the oracle on
random prompt STRUCTURES — placeholder multisets / orders, text in between, modalities present or None, zero-depth sentinel,
unequal spliced lengths, with and without an attention_mask (oracle/fuzz_cases.py).  The same generator pins the oracle to the
LIVE reference in the build container (oracle/fuzz_vs_reference.py: 1200 cases, all agree), so this closes the chain
reference -> oracle -> device code for the splice and its error behaviour, not only for the committed fixtures."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import cpu_ref  # noqa: E402
import e2e_cases  # noqa: E402
import fuzz_cases  # noqa: E402
import kernel_cases as kc  # noqa: E402
from vcoder_amd import synth  # noqa: E402


@pytest.fixture(scope="module")
def emu_lib():
    return kc.EmuBackend().lib


def _outcome(fn):
    try:
        return "ok", fn()
    except Exception as e:  # noqa: BLE001 — the class of the failure is the datum
        return type(e).__name__, str(e)


def _run_cases(eng, n_cases, rng, mode="strict"):
    cfg = eng.cfg
    om = cpu_ref.OracleModel(cfg, synth.synth_state_dict(cfg, 42))
    eng.set_precision(mode)
    t = lambda a: None if a is None else ([torch.from_numpy(x) for x in a] if isinstance(a, list) else torch.from_numpy(a))
    stats = {}
    try:
        for c in range(n_cases):
            rows, imgs, segs, deps, mask = fuzz_cases.random_case(rng, cfg)
            ids = np.asarray(rows, dtype=np.int64)
            k_or, v_or = _outcome(lambda: om.forward(rows, t(imgs), t(segs), t(deps), attention_mask=mask)[0].numpy())
            k_en, v_en = _outcome(lambda: eng.prefill(ids, imgs, segs, deps, all_logits=True, attention_mask=mask)[1])
            what = f"case {c}: rows={rows} img={imgs is not None} seg={segs is not None} depth={deps is not None} list={isinstance(imgs, list) or isinstance(segs, list)} mask={None if mask is None else mask.tolist()}"
            assert k_en == k_or, f"{what}: engine {k_en} ({v_en if k_en != 'ok' else ''}) vs oracle {k_or} ({v_or if k_or != 'ok' else ''})"
            if k_or == "ok":
                assert v_en.shape == v_or.shape, f"{what}: shapes {v_en.shape} vs {v_or.shape}"
                err = float(np.abs(v_en - v_or).max())
                assert err < 1e-3, f"{what}: logits differ by {err}"
                if c % 3 == 0:   # the cached greedy loop behind it (rows of an unequal-length batch continue from their zero
                    # padding rows, in the reference as here)
                    # generate() always has a mask — HF's GenerationMixin makes one of ones when the caller passes none — so a
                    # batch of unequal spliced lengths is the reference's UnboundLocalError there (quirk 6), padded only in a
                    # bare forward()
                    unequal = _outcome(lambda: om.prepare_inputs(rows, t(imgs), t(segs), t(deps), attention_mask_given=True))[0]
                    k_gen, got = _outcome(lambda: eng.generate(ids, imgs, segs, deps, max_new_tokens=4, attention_mask=mask))
                    if unequal != "ok":
                        assert k_gen == unequal == "UnboundLocalError", f"{what}: generate -> {k_gen}, oracle -> {unequal}"
                    else:
                        # the reference's generate(): the caller's mask hides keys in the prefill; its cached multimodal steps
                        # rebuild a mask of ones (vcoder_ds_llava_arch.py:130-133)
                        lg, cache = om.forward(rows, t(imgs), t(segs), t(deps), attention_mask=mask)
                        toks = [lg[:, -1].argmax(-1)]
                        for _ in range(3):
                            toks.append(om.decode_step(toks[-1].tolist(), cache)[:, -1].argmax(-1))
                        want = torch.stack(toks, 1).numpy()
                        assert k_gen == "ok" and np.array_equal(got, want), f"{what}: greedy ids {got} vs {want.tolist()}"
                        # a host-driven decode_step loop keeps the caller's mask (forward() without images carries it)
                        lg, cache = om.forward(rows, t(imgs), t(segs), t(deps), attention_mask=mask)
                        last, _, _ = eng.prefill(ids, imgs, segs, deps, attention_mask=mask)
                        assert np.abs(last - lg[:, -1].numpy()).max() < 1e-3
                        tok_o = lg[:, -1].argmax(-1)
                        for _ in range(3):
                            lo = om.decode_step(tok_o.tolist(), cache, keep_mask=True)[:, -1]
                            le, _ = eng.decode_step(tok_o.numpy().astype(np.int32))
                            assert np.abs(le - lo.numpy()).max() < 1e-3, f"{what}: cached step with the carried mask"
                            tok_o = lo.argmax(-1)
                        eng.clear_attention_mask()
            stats[k_or] = stats.get(k_or, 0) + 1
    finally:
        eng.set_precision("bf16")
    return stats


@pytest.mark.parametrize("variant,n_cases,seed", [("vcoder_ds", 20, 11), ("vcoder", 14, 12), ("llava", 10, 13)])
def test_random_prompt_structures(emu_lib, variant, n_cases, seed):
    stats = _run_cases(e2e_cases.engine_for(variant, emu_lib), n_cases, np.random.RandomState(seed))
    assert stats.get("ok", 0) >= n_cases // 4, stats     # the generator keeps a healthy share of valid prompts
    print(variant, stats)


@pytest.mark.parametrize("variant,seed,mode", [("vcoder_ds", 21, "strict"), ("vcoder_ds", 22, "split"), ("vcoder", 24, "split"),
                                               ("vcoder", 25, "strict"), ("llava", 26, "split")])
def test_random_configs(emu_lib, variant, seed, mode):
    """the same check on random variations of the tiny architecture: feature selection (patch / cls_patch, any layer), projector
    types per modality, head dims 64 / 128 on both sides, depths, image grids, norm epsilon (oracle/fuzz_cases.random_overrides —
    the oracle is pinned to the live reference on such variations by oracle/fuzz_vs_reference.py --configs N)"""
    from vcoder_amd.engine import HipEngine

    rng = np.random.RandomState(seed)
    over = fuzz_cases.random_overrides(rng, variant)
    cfg = e2e_cases.tiny_cfg(variant, over)
    eng = HipEngine(cfg, lib=emu_lib)
    try:
        eng.load_synthetic(42)
        eng.finalize()
        stats = _run_cases(eng, 8, rng, mode)
    finally:
        eng.close()
    print(variant, mode, over, stats)
