"""`-m gpu`: FULL-SIZE parity of the benchmarked configurations against the oracle (oracle/cpu_ref.py on the box's host
cores): BASELINE configs[1] as written (VCoder-DS 7b, 32 decoder + 23 ViT layers, the C2 prompt S = 1216, B = 8, 128 greedy
tokens) and configs[2] as written (13b, 40 layers, B = 16, 128 tokens), in every precision mode the engine has.

How greedy-id equality is checked without 128 sequential oracle steps: TEACHER FORCING.  The device's own ids are appended
to the spliced prompt and the oracle runs ONE causal pass over S + n - 1 positions; its logits at positions S-1 .. S+n-2
are exactly the logits its cached greedy loop would see after the same prefix (attention is causal), so
    argmax(oracle logits at step t) == device id at step t   for every t
proves by induction that the two greedy sequences are identical.  For the split and strict modes that equality is
asserted at EVERY step with no exception; for the bf16 path a step may differ only when the oracle's margin between its
own choice and the device's choice is below twice the measured logit deviation at that step (a numerical near-tie, printed
and counted).  The oracle pass covers two rows of the batch (first and last); every row is covered by the id comparisons
between the lone call, the pooled calls and the session loop.

Weights: the seeded synthetic checkpoint generated ON THE DEVICE (bit-identical to vcoder_amd/synth.py, test_synth) and
copied back as bf16 — the host generator would need ~15 minutes for 6.7 G parameters."""
import ctypes
import time

import numpy as np
import pytest
import torch

import cpu_ref
import e2e_cases
from vcoder_amd import config as vcfg, synth
from vcoder_amd.engine import HipEngine

pytestmark = pytest.mark.gpu


from device_weights import LazyState, device_state_dict  # noqa: E402,F401  (oracle/device_weights.py: checker infrastructure)


def oracle_teacher_forced(om, ids, imgs, segs, deps, forced, checkpoints=()):
    """One causal pass over [spliced prompt | embeddings of forced[:, :-1]] -> logits at the n = forced.shape[1] decode
    positions [B, n, V] (+ last-prompt-row logits after `checkpoints` layers: the model cut to that depth)."""
    t = torch.from_numpy
    with torch.no_grad():
        x, _ = om.prepare_inputs(ids.tolist(), t(imgs), t(segs), t(deps))
        S = x.shape[1]
        n = forced.shape[1]
        if n > 1:
            extra = torch.stack([om.embed_tokens(row[:n - 1].tolist()) for row in forced], 0)
            x = torch.cat([x, extra], dim=1)
        r = cpu_ref.Rounder(om.emu)
        cache = cpu_ref.KVCache(om.cfg.num_hidden_layers)
        cut = {}
        head = lambda h: torch.nn.functional.linear(r(cpu_ref.rms_norm(h, om.sd["model.norm.weight"], om.cfg.rms_norm_eps)),
                                                    om.sd["lm_head.weight"])
        for i in range(om.cfg.num_hidden_layers):
            x = cpu_ref.llama_layer(x, om.sd, i, om.cfg, cache, 0, r)
            if i + 1 in checkpoints:
                cut[i + 1] = head(x[:, S - 1:S])[:, 0].numpy()
        logits = head(x[:, S - 1:]).numpy()
    return S, logits, cut


def _loop(eng, ids, imgs, segs, deps, n_new, keep_rows):
    """the session's own cached loop (vc_prefill + vc_decode_step) fed with its own greedy ids: ids [B, n] and the logits of
    every step for `keep_rows` [len(keep_rows), n, V]"""
    last, _, S = eng.prefill(ids, imgs, segs, deps, reserve=n_new)
    steps, toks = [last[keep_rows]], [np.argmax(last, -1).astype(np.int32)]
    for _ in range(n_new - 1):
        lg, nxt = eng.decode_step(toks[-1])
        steps.append(lg[keep_rows])
        toks.append(nxt)
    return np.stack(steps, 1), np.stack(toks, 1), S


def _loop_forced(eng, ids, imgs, segs, deps, forced):
    """the session's cached loop TEACHER-FORCED with `forced` [B, n]: the logits of every step [B, n, V] (step t sees the prompt
    and forced[:, :t])"""
    n = forced.shape[1]
    last, _, _ = eng.prefill(ids, imgs, segs, deps, reserve=n)
    steps = [last]
    for t in range(n - 1):
        lg, _ = eng.decode_step(np.ascontiguousarray(forced[:, t], dtype=np.int32))
        steps.append(lg)
    return np.stack(steps, 1)


def _concurrent(root, n_calls, fn):
    import threading

    sessions = [root] + [root.fork() for _ in range(n_calls - 1)]
    outs, errs = [None] * n_calls, []

    def work(i):
        try:
            outs[i] = fn(sessions[i])
        except BaseException as e:
            errs.append(e)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(n_calls)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return sessions, outs, errs


def run_case(cfg, B, n_new, seed, oracle_rows, checkpoints, strict_tokens, split=True, pooled_calls=4, lib=None,
             fast_vs="oracle", dtypes="bf16"):
    """The FULL-SIZE call of a BASELINE configuration (B sequences of the C2 prompt, n_new greedy tokens, every layer) on the
    device in its precision modes, checked against ONE teacher-forced fp32 oracle pass over the rows `oracle_rows`:

      fast (bf16) path   logits of every step of row oracle_rows[0] within REL_TOL_VS_FP32 of the oracle forced with the fast
                         path's own ids; ids equal except numerical near-ties; generate() == the session loop; `pooled_calls`
                         concurrent generate() calls (the configuration bench.py's `value` is measured on) == the lone call
      split mode         logits of every step of every oracle row within 1e-3 ABSOLUTE of the oracle forced with the split
                         ids, argmax(oracle) == the split id at EVERY step, no near-tie excuse: the split mode's greedy
                         sequence IS the fp32 reference's; generate() through the pool == the session loop
      strict mode        the first `strict_tokens` steps of row oracle_rows[0]: 1e-3 absolute, ids identical

    fast_vs = "split" (needs split=True): the reference the bf16 path is measured against is the SPLIT path on the device,
    teacher-forced with the bf16 path's ids (row oracle_rows[0], batch 1) — the same test proves the split path to be within 1e-3
    ABSOLUTE of the fp32 oracle with identical ids, which is 100x below the bf16 path's tolerance; it saves the oracle a third
    row (60-70 s of host time per case).

    lib: test-only injection of the CPU emulator build (tests/test_engine_emu.py runs this logic on the tiny model)."""
    assert fast_vs in ("oracle", "split") and (split or fast_vs == "oracle")
    t0 = time.time()
    eng = HipEngine(cfg, lib=lib)
    eng.load_synthetic(seed, dtypes=dtypes)      # "reference" / "reference_loaded": fp16-valued LLM, fp32- / fp16-valued tower -> weight lo planes
    eng.finalize()
    assert (eng.inexact_tensors() > 0) == (dtypes in synth.REFERENCE_CLASSES)
    ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=b) for b in range(B)])
    imgs, segs, deps = synth.synth_batch(B, cfg.vit_image_size)
    rows = list(oracle_rows)
    r0_ = rows[0]     # the sample the single-row legs (depth chart, strict mode) run on: the first oracle row
    # ---- fast path: cached decode loop fed with its own greedy ids
    fast_logits, fast_ids, S = _loop(eng, ids, imgs, segs, deps, n_new, rows)
    lone = eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=n_new)
    assert np.array_equal(lone, fast_ids), "generate() (hipGraph loop / decode pool) differs from the decode_step loop"
    if pooled_calls > 1:   # the measured configuration: concurrent calls whose decode steps share the pool
        sessions, outs, errs = _concurrent(eng, pooled_calls,
                                           lambda s_: s_.generate_greedy(ids, imgs, segs, deps, max_new_tokens=n_new))
        assert not errs, errs
        for i, o in enumerate(outs):
            assert np.array_equal(o, fast_ids), f"pooled call {i} of {pooled_calls} differs from the lone call"
        for s_ in sessions[1:]:
            s_.close()
    # depth chart: the same model cut to L layers
    cut_dev = {}
    for L in checkpoints:
        eng.set_layer_limit(L)
        cut_dev[L] = eng.prefill(ids[r0_:r0_ + 1], imgs[r0_:r0_ + 1], segs[r0_:r0_ + 1], deps[r0_:r0_ + 1])[0][0]   # the first oracle row's sample
    eng.set_layer_limit(0)
    # ---- split mode: bf16 hi + lo MFMA operands on the fast kernels
    split_logits = split_ids = None
    if split:
        eng.set_precision("split")
        split_logits, split_ids, S2 = _loop(eng, ids, imgs, segs, deps, n_new, rows)
        assert S2 == S
        assert np.array_equal(eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=n_new), split_ids), \
            "split mode: generate() through the decode pool differs from the session loop"
        if fast_vs == "split":
            r0 = rows[0]
            split_forced_fast = _loop_forced(eng, ids[r0:r0 + 1], imgs[r0:r0 + 1], segs[r0:r0 + 1], deps[r0:r0 + 1], fast_ids[r0:r0 + 1])
    # ---- strict (fp32) path
    eng.set_precision("strict")
    s_last, _, _ = eng.prefill(ids[r0_:r0_ + 1], imgs[r0_:r0_ + 1], segs[r0_:r0_ + 1], deps[r0_:r0_ + 1], reserve=strict_tokens)
    s_steps, s_toks = [s_last], [np.argmax(s_last, -1).astype(np.int32)]
    for _ in range(strict_tokens - 1):
        lg, nxt = eng.decode_step(s_toks[-1])
        s_steps.append(lg)
        s_toks.append(nxt)
    strict_logits, strict_ids = np.stack(s_steps, 1), np.stack(s_toks, 1)
    eng.set_precision("bf16")
    t_dev = time.time() - t0
    sd = (device_state_dict(eng, cfg, seed, dtypes=dtypes) if lib is None
          else cpu_ref.as_torch_state(synth.synth_state_dict(cfg, seed, dtypes=dtypes)))
    eng.close()
    t_sd = time.time() - t0 - t_dev
    # ---- ONE fp32 oracle pass (= the reference's CPU path): the oracle rows forced with the split ids, then oracle_rows[0]
    # forced with the fast path's ids
    om32 = cpu_ref.OracleModel(cfg, sd, emu_bf16=False)
    with_fast = fast_vs == "oracle"
    sel = (rows if split else []) + ([rows[0]] if with_fast else []) + ([] if split else rows[1:])
    forced = np.stack(([split_ids[r] for r in rows] if split else []) + ([fast_ids[rows[0]]] if with_fast else []) +
                      ([] if split else [fast_ids[r] for r in rows[1:]]), 0)
    S_o, o32, cut32 = oracle_teacher_forced(om32, ids[sel], imgs[sel], segs[sel], deps[sel], forced, checkpoints)
    assert S_o == S == ids.shape[1] - 3 + 2 * cfg.num_patches
    t_o32 = time.time() - t0 - t_dev - t_sd
    scale = float(np.abs(o32).max())
    n_split = len(rows) if split else 0
    # the reference rows forced with the fast path's ids: the oracle's, or the split path's own (fast_vs == "split")
    o_fast = o32[n_split:] if with_fast else split_forced_fast
    f_rows = [0] if split else list(range(len(rows)))                # index into fast_logits (which holds oracle_rows)
    err32 = np.abs(fast_logits[f_rows] - o_fast).max(-1)              # [rows, n] max |dlogit| per step
    print(f"[{cfg.num_hidden_layers}L D{cfg.hidden_size}] S={S} B={B} n={n_new} |logits|max={scale:.3f}  bf16 path vs "
          f"{'fp32 oracle' if with_fast else 'the split path forced with its ids (itself checked against the fp32 oracle below)'} "
          f"(rows {[rows[i] for i in f_rows]}): prefill {err32[:, 0].max():.4f}, decode steps max {err32[:, 1:].max():.4f} "
          f"(rel {err32.max() / scale:.2e}); times: device {t_dev:.0f}s weights {t_sd:.0f}s oracle-fp32 {t_o32:.0f}s")
    for L in checkpoints:
        e = float(np.abs(cut_dev[L] - cut32[L][0]).max())
        print(f"    depth chart: first {L:2d} layers  |dlogit|max = {e:.4f}  (rel {e / max(np.abs(cut32[L][0]).max(), 1e-9):.2e})")
    # fast path greedy ids: teacher-forced equality, near-ties excepted
    near, margins = 0, []
    for i, fr in enumerate(f_rows):
        b = rows[fr]
        for s_ in range(n_new):
            o = o_fast[i, s_]
            top = int(np.argmax(o))
            srt = np.sort(o)
            margins.append(float(srt[-1] - srt[-2]))
            if top != int(fast_ids[b, s_]):
                gap = float(o[top] - o[int(fast_ids[b, s_])])
                assert gap < 2.0 * err32[i, s_], (
                    f"greedy id mismatch at row {b} step {s_}: device {fast_ids[b, s_]} vs oracle {top}, oracle gap {gap:.4f} "
                    f"exceeds twice the measured logit deviation {err32[i, s_]:.4f}")
                near += 1
    n_steps = len(f_rows) * n_new
    print(f"    bf16 path greedy ids: {n_steps - near}/{n_steps} steps identical to the fp32 oracle, {near} numerical near-ties; "
          f"top-2 margins min {min(margins):.4f} median {float(np.median(margins)):.4f}")
    out = dict(scale=scale, err32=err32, near=near)
    # ---- split mode: the literal bar, at full size, every step, no excuse
    if split:
        o_split = o32[:n_split]
        e_split = np.abs(split_logits - o_split).max(-1)             # [rows, n]
        same = np.argmax(o_split, -1) == split_ids[rows]
        first_div = [int(np.argmax(fast_ids[b] != split_ids[b])) if (fast_ids[b] != split_ids[b]).any() else n_new for b in range(B)]
        print(f"    split mode vs fp32 oracle (rows {rows}, {n_new} steps each): |dlogit|max prefill {e_split[:, 0].max():.2e}, decode "
              f"{e_split[:, 1:].max():.2e}; argmax(oracle) == split id at {int(same.sum())}/{same.size} steps; the bf16 path follows "
              f"the split ids for {first_div} steps of {n_new} per row")
        assert same.all(), f"split mode: greedy ids differ from the fp32 oracle at steps {np.argwhere(~same).tolist()}"
        assert e_split.max() < 1e-3, f"split mode: |dlogit|max {e_split.max():.2e} exceeds BASELINE.json's 1e-3"
        out["e_split"] = float(e_split.max())
        o_strict = o_split[:1, :strict_tokens] if np.array_equal(strict_ids[0], split_ids[rows[0], :strict_tokens]) else None
    else:
        o_strict = o_fast[:1, :strict_tokens] if np.array_equal(strict_ids[0], fast_ids[rows[0], :strict_tokens]) else None
    # ---- strict path vs the fp32 oracle
    if o_strict is None:  # the strict path took another branch at a near-tie of the path that forced the oracle
        _, o_strict, _ = oracle_teacher_forced(om32, ids[r0_:r0_ + 1], imgs[r0_:r0_ + 1], segs[r0_:r0_ + 1], deps[r0_:r0_ + 1], strict_ids)
    e_strict = float(np.abs(strict_logits - o_strict).max())
    assert np.array_equal(np.argmax(o_strict, -1), strict_ids), "strict path: greedy ids differ from the fp32 oracle"
    print(f"    strict path vs fp32 oracle: |dlogit|max = {e_strict:.2e} over {strict_tokens} steps, ids identical; total {time.time() - t0:.0f}s")
    assert e_strict < 1e-3, "strict mode must meet BASELINE.json's 1e-3"
    out["e_strict"] = e_strict
    return out


# Tolerance of the bf16 path = 2x the deviation measured on MI355X, relative to max|logits| of the case.
# Measured at full depth (7b, 32 layers, |logits|max 6.77): 2.0e-2 against the fp32 oracle (0.122 prefill / 0.136 decode
# absolute), growing like sqrt(depth): 5.6e-3 @ 2 layers, 1.1e-2 @ 8, 1.35e-2 @ 16, 2.3e-2 @ 32.  Split and strict mode: 1e-3
# ABSOLUTE (BASELINE.json), asserted inside run_case.
REL_TOL_VS_FP32 = 4.0e-2
REL_TOL_VS_FP32_INEXACT = 6.0e-2   # the same path on an fp16- / fp32-valued checkpoint (weights rounded to bf16 as well): 3.0e-2 measured


def test_full_size_7b_c2():
    """BASELINE configs[1] AS WRITTEN: VCoder-DS 7b, all 32 decoder + 23 ViT layers, the C2 prompt, B = 8, 128 greedy tokens —
    lone call, 4 concurrent calls through the decode pool (what bench.py's `value` measures), split mode, strict mode; fp32
    oracle teacher-forced on row 7 (the last of the batch) with the split ids AND with the bf16 path's own ids: the benchmarked
    kernels are compared with the oracle DIRECTLY at full size (round 4 compared them with the device's split path to save an oracle
    row; the 13b case still does).  (Rounds 3-6 also forced row 0 with the split ids: a third sample of the one batched oracle pass,
    ~20 s of the suite's budget on the box's 16 host CPUs — dropped at the end of round 6; rows 0 of the batch are what the fixtures,
    the true-dims cases and the full-depth inexact case run.)"""
    cfg = vcfg.vicuna_7b("vcoder_ds")
    r = run_case(cfg, B=8, n_new=128, seed=42, oracle_rows=(7,), checkpoints=(2, 8, 16, 32), strict_tokens=8, fast_vs="oracle")
    assert r["err32"].max() < REL_TOL_VS_FP32 * max(1.0, r["scale"])
    assert r["e_split"] < 1e-3


def test_full_depth_7b_inexact_checkpoint():
    """VCoder-DS 7b at FULL depth (32 decoder + 23 ViT layers, the C2 prompt) on a checkpoint with the values the reference COMPUTES
    with — fp16-valued LLM / projector tensors (model/builder.py:25-40 loads torch_dtype=float16) and the CLIP tower as
    model/builder.py:142 casts it at load, fp16-valued too ("reference_loaded"; the fp32-valued tower of the hub file is the stress
    case of test_inexact_checkpoint_true_dims) — generated on the device (vc_model_synth_tensor_rounded), which bf16 cannot
    hold: every matrix keeps a lo plane, and hi + lo holds every value EXACTLY.  B = 2, 16 greedy tokens: SPLIT mode (three-segment prefill GEMMs, lo-plane decode GEMV)
    and STRICT mode against the fp32 oracle on the SAME values, 1e-3 absolute, ids identical at every step; the bf16 path (which
    rounds the weights) against the split path for the record."""
    cfg = vcfg.vicuna_7b("vcoder_ds")
    r = run_case(cfg, B=2, n_new=16, seed=43, oracle_rows=(1,), checkpoints=(), strict_tokens=4, split=True, pooled_calls=1,
                 fast_vs="split", dtypes="reference_loaded")
    assert r["e_split"] < 1e-3 and r["e_strict"] < 1e-3
    print(f"    bf16 path on the inexact checkpoint vs the split path: |dlogit|max {r['err32'].max():.4f} (rel {r['err32'].max() / r['scale']:.2e})")
    # measured on MI355X (profiles/r05_o_inexact_checkpoint_full_depth_7b.txt): 0.19 at |logit|max 6.42 = 3.0e-2 -> 2x (round 5
    # asserted 2 x REL_TOL_VS_FP32 = 8e-2 without a measurement behind it)
    assert r["err32"].max() < REL_TOL_VS_FP32_INEXACT * max(1.0, r["scale"])


def test_fp8_formats_vs_oracle_full_depth_13b():
    """BASELINE configs[4]'s model and format (13b, 40 layers, e4m3 weights, W8A8 prefill on the scaled fp8 MFMA, e4m3 KV cache) against
    the ORACLE in its fp8 mode on the EFFECTIVE weights (oracle/device_weights.effective_fp8_rows) — not against the device's own
    bf16 path (round 5's criterion, which cannot tell an implementation error from the format's error).  B = 1, the C2 prompt, 8
    tokens teacher-forced with the device's ids: oracle prefill over e4m3 activation rows + 7 cached steps over the e4m3 cache.
    Measured on MI355X (profiles/r06_j_fp8_full_depth_13b_vs_oracle.txt): |dlogit|max = 0.304 of |logit|max in the prefill, 0.151 over
    the cached steps, logit correlation min 0.934 / median 0.991 — BELOW what the oracle's own result moves when its arithmetic
    switches between fp32 and bf16 emulation at this depth (0.367 / 0.143, correlation min 0.915): with per-token e4m3 rows every
    rounding-level perturbation re-draws the quantisation noise downstream, and the device is as close to the oracle as the oracle
    is to itself.  (Against the bf16 PATH the same logits correlate at 0.874 / 0.915: that is the format.)  Tolerance: 1.5x the
    measured deviation, correlation floors 0.90 / 0.98."""
    cfg = vcfg.vicuna_13b("vcoder_ds")
    n_new = 8
    ids = synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=0)[None]
    imgs, segs, deps = synth.synth_batch(1, 336)
    eng = HipEngine(cfg)
    eng.load_synthetic(42)
    eng.set_weight_format("fp8")
    eng.finalize()
    dev_logits, dev_ids, S = _loop(eng, ids, imgs, segs, deps, n_new, [0])
    sd = device_state_dict(eng, cfg, 42, effective_fp8=True)
    eng.close()
    om = cpu_ref.OracleModel(cfg, sd, emu_bf16=True, act_fp8=True)
    t = torch.from_numpy
    o = []
    with torch.no_grad():
        lg, cache = om.forward(ids.tolist(), t(imgs), t(segs), t(deps), last_only=True)
        o.append(lg[0, -1].numpy())
        for s_ in range(1, n_new):
            o.append(om.decode_step([int(dev_ids[0, s_ - 1])], cache)[0, -1].numpy())
    o, d_ = np.stack(o, 0), dev_logits[0]
    scale = float(np.abs(o).max())
    dev = np.abs(d_ - o).max(-1) / scale
    a, b = d_ - d_.mean(-1, keepdims=True), o - o.mean(-1, keepdims=True)
    corr = (a * b).sum(-1) / np.sqrt((a * a).sum(-1) * (b * b).sum(-1))
    print(f"    13b fp8 (e4m3 weights, W8A8 prefill, e4m3 KV) vs the oracle's fp8 mode on the effective weights, 40 layers, S={S}: |dlogit|max / "
          f"|logit|max prefill {dev[0]:.3f}, cached steps {dev[1:].max():.3f}; logit correlation min {corr.min():.4f} median {np.median(corr):.4f}; "
          f"argmax(oracle) == device id at {int((np.argmax(o, -1) == dev_ids[0]).sum())}/{n_new} steps; |logit|max {scale:.2f}")
    assert dev[0] < 0.46 and dev[1:].max() < 0.23, dev
    assert corr.min() > 0.90 and np.median(corr) > 0.98, corr


def test_full_depth_7b_fp16_operand_library():
    """VCoder-DS 7b at FULL depth (32 + 23 layers, the C2 prompt, B = 2, 16 greedy tokens) on the fp16-operand library
    (libvcoder_hip_f16.so, round 6) with the checkpoint in the reference's own value classes (fp16-valued LLM / projectors: held
    exactly; fp32 tower): every step's logits against the SPLIT path of the bf16 library on the same checkpoint, teacher-forced with
    the fp16 library's ids — the split path is within 2.6e-4 of the fp32 oracle at this depth with identical ids
    (test_full_depth_7b_inexact_checkpoint), 20x below the tolerance here.  Oracle study at the same dims: 4.0e-3 of |logit|max at
    32 layers (profiles/r06_fp16_go_nogo.txt; bf16 operands: 3.4e-2) -> REL_TOL 6e-3."""
    cfg = vcfg.vicuna_7b("vcoder_ds")
    B, n_new, seed = 2, 16, 43
    ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=b) for b in range(B)])
    imgs, segs, deps = synth.synth_batch(B, cfg.vit_image_size)
    e16 = HipEngine(cfg, operands="fp16")
    e16.load_synthetic(seed, dtypes="reference")
    e16.finalize()
    n_inexact16 = e16.inexact_tensors()
    h_logits, h_ids, S = _loop(e16, ids, imgs, segs, deps, n_new, [0, 1])
    assert np.array_equal(e16.generate_greedy(ids, imgs, segs, deps, max_new_tokens=n_new), h_ids), "fp16 library: generate() differs from the session loop"
    e16.close()
    ref = HipEngine(cfg)
    ref.load_synthetic(seed, dtypes="reference")
    ref.finalize()
    assert 0 < n_inexact16 < ref.inexact_tensors()      # the fp16 library holds the fp16-valued tensors exactly
    f_logits = _loop_forced(ref, ids, imgs, segs, deps, h_ids)             # the bf16 fast path, forced with the same ids
    ref.set_precision("split")
    s_logits = _loop_forced(ref, ids, imgs, segs, deps, h_ids)             # the reference: split path (1e-3 of the fp32 oracle)
    ref.close()
    scale = float(np.abs(s_logits).max())
    e_h, e_f = np.abs(h_logits - s_logits).max(-1), np.abs(f_logits - s_logits).max(-1)
    same = np.argmax(s_logits, -1) == h_ids
    print(f"    fp16-operand library, 7b full depth, S={S}: |dlogit|max vs the split path {e_h.max():.4f} = {e_h.max() / scale:.2e} of |logit|max "
          f"{scale:.2f} (the bf16 library on the same ids: {e_f.max():.4f} = {e_f.max() / scale:.2e}); greedy ids equal the split path's choice at "
          f"{int(same.sum())}/{same.size} steps")
    assert e_h.max() < 6e-3 * scale
    for b, s_ in np.argwhere(~same):    # a differing choice must be a numerical near-tie of the reference
        o = s_logits[b, s_]
        assert float(o.max() - o[h_ids[b, s_]]) < 2.0 * e_h[b, s_], f"fp16 library: greedy id mismatch at row {b} step {s_} beyond a near-tie"


def test_full_size_13b_c3():
    """BASELINE configs[2] AS WRITTEN: VCoder-DS 13b (D 5120, 40 layers, 40 heads, F 13824), B = 16, 128 greedy tokens; 2
    concurrent calls through the pool; SPLIT mode against the fp32 oracle teacher-forced on row 15 (1e-3 absolute,
    128 / 128 ids, no near-tie criterion — BASELINE.json's bar at this size; rounds 4-5 checked rows {0, 15}); strict mode on 4
    tokens; the bf16 path against the split path."""
    cfg = vcfg.vicuna_13b("vcoder_ds")
    # (round 6: ONE oracle row — the last of the batch — instead of two: the 13b oracle pass is 45 s per row on the box's 16 host
    # CPUs and the suite has a time budget; the 7b case keeps two rows + the fast path's own)
    r = run_case(cfg, B=16, n_new=128, seed=42, oracle_rows=(15,), checkpoints=(40,), strict_tokens=4, split=True,
                 pooled_calls=2, fast_vs="split")
    assert r["err32"].max() < REL_TOL_VS_FP32 * max(1.0, r["scale"])
    assert r["e_split"] < 1e-3


def _device_run(cfg, fmt, ids, imgs, segs, deps, n_new, forced=None, fp8_kv=True):
    """prefill + n_new greedy (or teacher-forced) steps on the device: logits [B, n_new, V], ids [B, n_new]"""
    eng = HipEngine(cfg)
    eng.load_synthetic(42)
    if fmt != "bf16":
        eng.set_weight_format(fmt)
    if not fp8_kv:
        eng.set_fp8_kv(False)
    eng.finalize()
    last, _, _ = eng.prefill(ids, imgs, segs, deps)
    logits, toks = [last], [np.argmax(last, -1).astype(np.int32)]
    for s_ in range(1, n_new):
        feed = toks[-1] if forced is None else forced[:, s_ - 1].astype(np.int32)
        lg, _ = eng.decode_step(feed)
        logits.append(lg)
        toks.append(np.argmax(lg, -1).astype(np.int32))
    eng.close()
    return np.stack(logits, 1), np.stack(toks, 1)


# fp8 formats against the device's own bf16 path, all 40 layers, on the seeded RANDOM checkpoint.  What this measures is
# the quantisation (e4m3 weights: 3 mantissa bits, per-row scales; 'fp8' adds e4m3 activation rows in the prefill) pushed
# through 40 layers of a model with no trained structure — a worst case for any 8-bit format: measured on MI355X the
# W8A16 logits move by 0.34 (prefill) ... 0.42 of max|logit| and keep 13 of 32 greedy choices (the bf16 path's own median
# top-2 margin is 0.25 = 4 % of max|logit|), 'fp8' by 0.50 and 12 of 32.  The e4m3 activations therefore add little to
# what the e4m3 weights already cost, and neither number says anything about an implementation error: the kernels'
# exactness on quantised operands is pinned in test_gpu_kernels.py (GEMM vs float64: 4e-5; quantisers bit-identical to
# vcoder_amd/quant.py) and test_gpu_e2e.py.  Asserted here: the quantised paths still compute the SAME function (logit
# vectors correlated with the bf16 path's), and a changed greedy choice is one whose bf16 margin was within the shift.
MIN_LOGIT_CORRELATION = 0.75   # measured: w8a16 min 0.913, fp8 (bf16 KV rows) min 0.863 (median 0.92 both)
MIN_KV8_CORRELATION = 0.98     # fp8 + e4m3 KV against fp8 + bf16 KV on the same forced ids: measured min 0.996, median 0.997 (round 5)


def test_fp8_formats_vs_bf16_full_depth_13b():
    """BASELINE configs[4] model geometry (13b, 40 layers), B=2, C2 prompt, 16 tokens teacher-forced on the bf16 path's ids:
    how far the W8A16 and the fp8 (W8A8 prefill) configurations move the logits, and how many greedy choices they keep — the
    fp8 format with bf16 KV rows against the round-3 floor (0.75), and separately what its default e4m3 KV cache adds on top
    (fp8 + e4m3 KV against fp8 + bf16 KV on the same forced ids)."""
    cfg = vcfg.vicuna_13b("vcoder_ds")
    B, n_new = 2, 16
    ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=b) for b in range(B)])
    imgs, segs, deps = synth.synth_batch(B, 336)
    ref_logits, ref_ids = _device_run(cfg, "bf16", ids, imgs, segs, deps, n_new)
    scale = float(np.abs(ref_logits).max())
    margin = np.sort(ref_logits, -1)[..., -1] - np.sort(ref_logits, -1)[..., -2]

    def corr_of(lg, base):
        a, b = lg - lg.mean(-1, keepdims=True), base - base.mean(-1, keepdims=True)
        return (a * b).sum(-1) / np.sqrt((a * a).sum(-1) * (b * b).sum(-1))
    runs = {}
    for fmt, kv8 in (("w8a16", True), ("fp8", False), ("fp8", True)):
        lg, tk = _device_run(cfg, fmt, ids, imgs, segs, deps, n_new, forced=ref_ids, fp8_kv=kv8)
        runs[(fmt, kv8)] = lg
        dev = np.abs(lg - ref_logits).max(-1)              # [B, n]
        same = tk == ref_ids
        corr = corr_of(lg, ref_logits)
        tag = fmt + (" + e4m3 KV" if fmt == "fp8" and kv8 else (" + bf16 KV" if fmt == "fp8" else ""))
        print(f"    13b {tag:15s} vs bf16 path: |dlogit|max/|logit|max prefill {dev[:, 0].max() / scale:.3f}, over {n_new} steps "
              f"{dev.max() / scale:.3f}; logit correlation min {corr.min():.3f} median {np.median(corr):.3f}; greedy choices kept "
              f"{same.sum()}/{same.size} (bf16 top-2 margins: min {margin.min():.3f}, median {np.median(margin):.3f}; "
              f"|logit|max {scale:.2f})")
        if fmt == "fp8" and kv8:
            # the default e4m3 KV cache of the fp8 format: the cached steps read 3-mantissa-bit keys / values.  Its own cost is the
            # step below (against the SAME format on bf16 rows); against the bf16 path the floor is what round 4 measured for
            # the combination — the arithmetic is pinned by check_kv8 (bytes == torch's e4m3 cast, attention == the oracle on the
            # dequantised cache) and by the fixtures' oracle, which models the e4m3 cache (cpu_ref.llama_layer)
            base = runs[("fp8", False)]
            c2 = corr_of(lg, base)
            d2 = np.abs(lg - base).max(-1)
            print(f"    13b fp8: e4m3 KV against bf16 KV (same weights, same forced ids): |dlogit|max/|logit|max over the cached steps "
                  f"{d2[:, 1:].max() / scale:.3f}; logit correlation min {c2[:, 1:].min():.3f} median {np.median(c2[:, 1:]):.3f}; "
                  f"prefill logits identical: {bool(np.array_equal(lg[:, 0], base[:, 0]))}")
            assert np.array_equal(lg[:, 0], base[:, 0]), "the prefill does not read the cache: its logits must not depend on the KV format"
            assert c2[:, 1:].min() > MIN_KV8_CORRELATION, f"e4m3 KV decorrelates the fp8 format's own logits ({c2[:, 1:].min():.3f})"
        # (round 4 had lowered the floor of the e4m3-KV combination to 0.6 without a measurement; measured in round 5 over 16 steps:
        # min 0.874 with either KV format — the round-3 floor holds)
        floor = MIN_LOGIT_CORRELATION
        assert corr.min() > floor, f"{tag}: logits decorrelated from the bf16 path ({corr.min():.3f})"
        for r_, s_ in zip(*np.nonzero(~same)):
            assert margin[r_, s_] < 2.0 * dev[r_, s_], (f"{tag}: choice changed at row {r_} step {s_} although the bf16 margin "
                                                        f"{margin[r_, s_]:.3f} exceeds the shift")
