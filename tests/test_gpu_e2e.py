"""`-m gpu`: the hot path end to end through the C ABI against the committed reference fixtures
(inputs_embeds, prefill logits at every position, greedy ids, the quirks) on the tiny golden models."""
import numpy as np
import pytest

import e2e_cases
from vcoder_amd import config as vcfg, synth
from vcoder_amd.engine import HipEngine

pytestmark = pytest.mark.gpu

FIXTURES = ["ds_img_depth_seg", "ds_img_seg_depth", "ds_img_seg", "ds_img_only", "ds_zero_depth", "ds_img_text_seg",
            "vc_img_seg", "vc_img_text_seg", "llava_img", "ds_proj_linear_mlp3x", "ds_proj_identity"]


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture(name):
    r = e2e_cases.check_fixture(name)
    print(name, r)


def test_host_weight_load_equals_device_synth():
    """vc_model_load_tensor (host fp32 state dict) and vc_model_synth_tensor (device generator) give the same model."""
    cfg = vcfg.tiny("vcoder_ds")
    eng = HipEngine(cfg)
    used, dead = eng.load_state_dict(synth.synth_state_dict(cfg, 42))
    assert dead > 0  # depth_mm_projector / mm2_projector / vcoder_lm_emb / unused CLIP layer are accepted and ignored
    eng.finalize()
    g, _, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_depth_seg")
    a, _, _ = eng.prefill(ids, imgs, segs, deps)
    b, _, _ = e2e_cases.engine_for("vcoder_ds").prefill(ids, imgs, segs, deps)
    assert np.array_equal(a, b)
    eng.close()


def test_quirks():
    eng = e2e_cases.engine_for("vcoder_ds")
    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_depth_seg")
    base, _, _ = eng.prefill(ids, imgs, segs, deps)
    # quirk 4: with the reference's own token order the depth pixels never reach the logits
    alt, _, _ = eng.prefill(ids, imgs, segs, deps * 0.5 + 1.0)
    assert np.array_equal(base, alt)
    # batched rows == single-sample rows (what makes data-parallel sharding parity-safe)
    one, _, _ = eng.prefill(ids[1:], imgs[1:], segs[1:], deps[1:])
    assert np.array_equal(base[1:], one)
    # quirk 6: unequal spliced lengths + attention_mask -> the reference's UnboundLocalError
    ragged = ids.copy()
    ragged[1, ragged[1] == synth.SEG_TOKEN_INDEX] = 5
    with pytest.raises(UnboundLocalError):
        eng.prefill(ragged, imgs, segs, deps, has_attention_mask=True)
    _, _, S = eng.prefill(ragged, imgs, segs, deps, has_attention_mask=False)  # zero right-padding instead
    # row 0 drops <depth> and splices 2 blocks (42 rows); row 1 keeps its 3 text ids after <image> (43 rows) -> S = 43
    assert S == int(g["spliced_len"]) + 1
    # quirk 5: non-DS image-only prompt reaches the embedding lookup with -200 -> IndexError
    vc = e2e_cases.engine_for("vcoder")
    g2, _, ids2, imgs2, segs2, _ = e2e_cases.fixture_inputs("vc_img_seg")
    bad = ids2.copy()
    bad[bad == synth.SEG_TOKEN_INDEX] = 7
    with pytest.raises(IndexError):
        vc.prefill(bad, imgs2, segs2)


def test_eos_and_padding():
    eng = e2e_cases.engine_for("vcoder_ds")
    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_depth_seg")
    ref = g["greedy_ids"]
    free = eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=8)
    eos = int(free[0, 2])  # make row 0 finish at its 3rd token
    got = eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=8, eos_token_id=eos, pad_token_id=0)
    first = list(free[0]).index(eos)
    assert list(got[0, : first + 1]) == list(free[0, : first + 1])
    assert all(t == 0 for t in got[0, first + 1:])  # finished rows emit pad


def test_true_shape_7b_layer_smoke():
    """One real-dimension step: VCoder-DS 7b geometry with 2 decoder layers and 2 ViT layers (weights generated on
    device), B=2 — checks that the true tile shapes / strides run and that prefill == incremental decode."""
    cfg = vcfg.vicuna_7b("vcoder_ds")
    cfg.num_hidden_layers = 2
    cfg.vit_num_layers = 3
    eng = HipEngine(cfg)
    eng.load_synthetic(7)
    eng.finalize()
    B = 2
    ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=b) for b in range(B)])
    imgs, segs, deps = synth.synth_batch(B, 336)
    out = eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=4)
    assert out.shape == (B, 4) and eng.last_timings()["decode_ms"] > 0
    # teacher-forced consistency: feeding generated token t through decode_step reproduces token t+1
    last, _, S = eng.prefill(ids, imgs, segs, deps)
    assert S == 64 + 2 * 576
    assert np.array_equal(np.argmax(last, -1), out[:, 0])
    _, nxt = eng.decode_step(out[:, 0])
    assert np.array_equal(nxt, out[:, 1])
    eng.close()


def test_true_dims_against_oracle():
    """TRUE 7b / ViT-L dimensions (D 4096, F 11008, V 32000, hd 128; ViT 1024/4096, 577 tokens, 336 px), cut to
    2 decoder + 2 ViT layers so the CPU oracle finishes in about a minute: prefill logits (S = 1216) and the first
    decode step against oracle/cpu_ref.py with the same bf16 rounding points, on identical synthetic weights."""
    import torch
    import cpu_ref

    cfg = vcfg.vicuna_7b("vcoder_ds")
    cfg.num_hidden_layers = 2
    cfg.vit_num_layers = 3
    sd = synth.synth_state_dict(cfg, 11)
    eng = HipEngine(cfg)
    eng.load_synthetic(11)
    eng.finalize()
    ids = synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=3)[None]
    imgs, segs, deps = synth.synth_batch(1, 336, first=3)
    last, _, S = eng.prefill(ids, imgs, segs, deps)
    lg2, nxt2 = eng.decode_step(np.argmax(last, -1).astype(np.int32))
    om = cpu_ref.OracleModel(cfg, sd, emu_bf16=True)
    t = torch.from_numpy
    with torch.no_grad():
        o_last, cache = om.forward(ids.tolist(), t(imgs), t(segs), t(deps), last_only=True)
        o_lg2 = om.decode_step(np.argmax(last, -1).tolist(), cache)
    o_last, o_lg2 = o_last[:, -1].numpy(), o_lg2[:, -1].numpy()
    e1, e2 = np.abs(last - o_last).max(), np.abs(lg2 - o_lg2).max()
    scale = np.abs(o_last).max()
    print(f"true-dims parity: S={S} |logits|max={scale:.3f} prefill err={e1:.4f} decode err={e2:.4f}")
    assert S == 1216
    # relative to max|logits|; measured on MI355X 6.0e-3 (3.5e-2 absolute at |logits| <= 5.8): tolerance = 2x measured
    assert e1 < 1.2e-2 * scale and e2 < 1.2e-2 * scale
    m = np.sort(o_last[0])[-1] - np.sort(o_last[0])[-2]
    if m > 4 * e1:
        assert int(np.argmax(last)) == int(np.argmax(o_last))
    eng.close()


def test_concurrent_sessions_match_single_session():
    """3 sessions on shared weights driven from 3 host threads at once (what bench.py does with --inflight 3) produce
    exactly the ids of a lone session — the overlap changes timing only."""
    import threading

    eng = e2e_cases.engine_for("vcoder_ds")
    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_depth_seg")
    ref = eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=8)
    sessions = [eng, eng.fork(), eng.fork()]
    outs = [[None] * 4 for _ in sessions]

    def work(si):
        for j in range(4):
            outs[si][j] = sessions[si].generate_greedy(ids, imgs, segs, deps, max_new_tokens=8)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for si in range(3):
        for j in range(4):
            assert np.array_equal(outs[si][j], ref)
    for s in sessions[1:]:
        s.close()


def test_true_dims_13b_geometry():
    """VCoder-DS 13b geometry (D 5120, 40 heads, F 13824) cut to 2 layers, batch 16 (BASELINE config 3): runs the
    direct (non-LDS-staged) fused-norm GEMV path (16 x 5120 rows do not fit) and 16-row decode; prefill == incremental."""
    cfg = vcfg.vicuna_13b("vcoder_ds")
    cfg.num_hidden_layers = 2
    cfg.vit_num_layers = 2
    eng = HipEngine(cfg)
    eng.load_synthetic(5)
    eng.finalize()
    B = 16
    ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=b) for b in range(B)])
    imgs, segs, deps = synth.synth_batch(B, 336)
    out = eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=3)
    assert out.shape == (B, 3)
    last, _, S = eng.prefill(ids, imgs, segs, deps)
    assert S == 1216 and np.array_equal(np.argmax(last, -1), out[:, 0])
    _, nxt = eng.decode_step(out[:, 0])
    assert np.array_equal(nxt, out[:, 1])
    # rows of the batch are independent: sample 5 alone gives the same first tokens
    one = eng.generate_greedy(ids[5:6], imgs[5:6], segs[5:6], deps[5:6], max_new_tokens=3)
    assert np.array_equal(one[0], out[5])
    eng.close()


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_strict_mode(name):
    """The literal BASELINE.json bar through the HIP path: logits within 1e-3 of the reference's fp32 CPU outputs at
    every position and decode step, greedy ids bit-exact (vc_model_set_precision = strict, fp32 MFMA)."""
    print(name, e2e_cases.check_fixture_strict(name))


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_split_mode(name):
    """The BASELINE.json bar on the FAST kernels (vc_model_set_precision = split: every MFMA operand as bf16 hi + lo): logits
    within 1e-3 absolute of the reference's fp32 CPU outputs at every position and decode step, greedy ids bit-exact with no
    margin criterion; generate() runs through the decode pool (stacked groups of 16), the step-by-step check on the
    session's loop (groups of 8)."""
    print(name, e2e_cases.check_fixture_strict(name, mode="split"))


def test_true_dims_split_against_fp32_oracle():
    """True 7b / ViT-L dims (2+2 layers, B = 2 and a 13b-geometry variant), split mode vs the fp32 oracle: 1e-3 absolute, ids
    equal; and the bf16 path on the same inputs for scale."""
    import torch
    import cpu_ref

    for mk, seed in ((vcfg.vicuna_7b, 11), (vcfg.vicuna_13b, 12)):
        cfg = mk("vcoder_ds")
        cfg.num_hidden_layers = 2
        cfg.vit_num_layers = 3
        sd = synth.synth_state_dict(cfg, seed)
        eng = HipEngine(cfg)
        eng.load_synthetic(seed)
        eng.finalize()
        B = 2
        ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=3 + b) for b in range(B)])
        imgs, segs, deps = synth.synth_batch(B, 336, first=3)
        fast_last, _, _ = eng.prefill(ids, imgs, segs, deps)
        eng.set_precision("split")
        last, _, S = eng.prefill(ids, imgs, segs, deps)
        tok = np.argmax(last, -1).astype(np.int32)
        lg2, _ = eng.decode_step(tok)
        om = cpu_ref.OracleModel(cfg, sd, emu_bf16=False)
        t = torch.from_numpy
        with torch.no_grad():
            o_last, cache = om.forward(ids.tolist(), t(imgs), t(segs), t(deps), last_only=True)
            o_lg2 = om.decode_step(tok.tolist(), cache)
        e1 = np.abs(last - o_last[:, -1].numpy()).max()
        e2 = np.abs(lg2 - o_lg2[:, -1].numpy()).max()
        ef = np.abs(fast_last - o_last[:, -1].numpy()).max()
        print(f"true-dims SPLIT parity (D={cfg.hidden_size}): prefill err={e1:.2e} decode err={e2:.2e}  (bf16 path: {ef:.2e})")
        assert e1 < 1e-3 and e2 < 1e-3
        assert np.array_equal(tok, np.argmax(o_last[:, -1].numpy(), -1))
        assert np.array_equal(np.argmax(lg2, -1), np.argmax(o_lg2[:, -1].numpy(), -1))
        eng.close()


@pytest.mark.parametrize("name", ["ds_img_depth_seg", "vc_img_text_seg", "llava_img", "ds_proj_linear_mlp3x"])
def test_fixture_fp16_operand_library(name):
    """libvcoder_hip_f16.so (the kernels built with -DVC_F16: fp16 MFMA operands, the precision of the reference's own GPU path,
    model/builder.py:39,142) against the live reference's fp32 fixtures: all-position prefill logits and every cached step within
    1.2e-3 of |logit|max (measured 4.7e-4; the bf16 library 3.4e-3 ... 4.3e-3), greedy ids equal."""
    r = e2e_cases.check_fixture_fp16(name)
    print(name, r)
    assert r["ids_equal"]


@pytest.mark.parametrize("name,mode", [("ds_img_depth_seg", "split"), ("ds_img_depth_seg", "strict"), ("vc_img_seg", "split"),
                                       ("llava_img", "split"), ("llava_img", "strict")])
def test_inexact_checkpoint(name, mode):
    """Checkpoints bf16 cannot hold — fp16-valued LLM / projector tensors and an fp32-valued CLIP tower, the reference's own dtypes
    (model/builder.py:25-40, multimodal_encoder/clip_encoder.py:22-27): the loader keeps a lo plane per inexact matrix and the
    strict / split modes stay within 1e-3 of the fp32 oracle run on the ORIGINAL values, greedy ids bit-exact."""
    r = e2e_cases.check_inexact_checkpoint(name, mode=mode)
    print(name, mode, r)
    assert r["logits_err"] < 1e-4 and r["decode_logits_err"] < 1e-4, r


@pytest.mark.parametrize("dtypes", ["reference", "reference_loaded"])
def test_inexact_checkpoint_true_dims(dtypes):
    """("reference": the files' value classes, an fp32-valued tower; "reference_loaded": the tower as model/builder.py:142 casts it,
    fp16-valued — what the reference computes with.)
    The same at true 7b / ViT-L dims (2 + 2 layers): split and strict vs the fp32 oracle on the fp16- / fp32-valued weights
    (1e-3 absolute, ids equal), the split decode step through the lo-plane form of the workgroup-shared GEMV; the bf16 fast
    path on the same checkpoint: prefill within 1.7e-2 of |logit|max, its teacher-forced decode step within 4x its prefill's deviation."""
    import torch
    import cpu_ref

    cfg = vcfg.vicuna_7b("vcoder_ds")
    cfg.num_hidden_layers = 2
    cfg.vit_num_layers = 3
    sd = synth.synth_state_dict(cfg, 13, dtypes=dtypes)
    eng = HipEngine(cfg)
    eng.load_state_dict(sd)
    eng.finalize()
    assert eng.inexact_tensors() >= 2 + 7 * 2 + 4 + 1 + 6 * 2
    ids = synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=5)[None]
    imgs, segs, deps = synth.synth_batch(1, 336, first=5)
    om = cpu_ref.OracleModel(cfg, sd, emu_bf16=False)
    t = torch.from_numpy
    with torch.no_grad():
        o_last, cache = om.forward(ids.tolist(), t(imgs), t(segs), t(deps), last_only=True)
        tok = np.argmax(o_last[:, -1].numpy(), -1).astype(np.int32)
        o_lg2 = om.decode_step(tok.tolist(), cache)
    fast_last, _, _ = eng.prefill(ids, imgs, segs, deps)
    ef = np.abs(fast_last - o_last[:, -1].numpy()).max()
    fast_lg2, _ = eng.decode_step(tok)   # teacher-forced with the oracle's token: the bf16 path's cached step sees the oracle's prefix
    ef2 = np.abs(fast_lg2 - o_lg2[:, -1].numpy()).max()
    scale = float(np.abs(o_last.numpy()).max())
    print(f"true-dims inexact checkpoint, bf16 path: prefill err={ef:.2e} decode err={ef2:.2e} of |logit|max {scale:.2f}")
    # measured on MI355X: 5.6e-2 at |logit|max 6.58 = 8.5e-3 relative (3.4e-2 absolute on the bf16-exact checkpoint of the shape) -> 2x
    assert ef < 1.7e-2 * scale and ef2 <= 4 * ef, (ef, ef2, scale)
    for mode in ("split", "strict"):
        eng.set_precision(mode)
        last, _, _ = eng.prefill(ids, imgs, segs, deps)
        lg2, _ = eng.decode_step(tok)
        e1 = np.abs(last - o_last[:, -1].numpy()).max()
        e2 = np.abs(lg2 - o_lg2[:, -1].numpy()).max()
        print(f"true-dims inexact checkpoint, {mode}: prefill err={e1:.2e} decode err={e2:.2e}  (bf16 path: {ef:.2e}, |logit|max "
              f"{np.abs(o_last.numpy()).max():.2f}, {eng.inexact_tensors()} inexact tensors)")
        assert e1 < 1e-3 and e2 < 1e-3
        assert np.array_equal(np.argmax(last, -1), tok) and np.array_equal(np.argmax(lg2, -1), np.argmax(o_lg2[:, -1].numpy(), -1))
    eng.close()
    # the fp16-operand library (round 6) on the same checkpoint: it holds the fp16-valued LLM / projector tensors EXACTLY (only the
    # fp32 tower keeps lo planes) and rounds activations to 11 bits: measured 5.5e-3 at |logit|max 6.58 = 8.4e-4 relative -> 2x
    eng16 = HipEngine(cfg, operands="fp16")
    eng16.load_state_dict(sd)
    eng16.finalize()
    n_tower = sum(1 for k, v in sd.items() if "vision_tower" in k and np.asarray(v).ndim >= 2)
    if dtypes == "reference":
        assert 0 < eng16.inexact_tensors() <= n_tower, (eng16.inexact_tensors(), n_tower)
    else:   # the tower as the reference casts it: the whole checkpoint is fp16-valued and this library holds it exactly
        assert eng16.inexact_tensors() == 0
    h_last, _, _ = eng16.prefill(ids, imgs, segs, deps)
    h_lg2, _ = eng16.decode_step(tok)
    eh, eh2 = np.abs(h_last - o_last[:, -1].numpy()).max(), np.abs(h_lg2 - o_lg2[:, -1].numpy()).max()
    print(f"true-dims inexact checkpoint, fp16-operand library: prefill err={eh:.2e} decode err={eh2:.2e} ({eh / scale:.2e} of |logit|max; "
          f"the bf16 library: {ef / scale:.2e}); {eng16.inexact_tensors()} inexact tensors")
    assert eh < 1.7e-3 * scale and eh2 < 1.7e-3 * scale and eh < 0.25 * ef
    assert np.array_equal(np.argmax(h_last, -1), tok)
    eng16.close()


def test_true_dims_strict_against_fp32_oracle():
    """True 7b / ViT-L dims (2+2 layers), strict mode vs the fp32 oracle (= the reference's CPU path): 1e-3."""
    import torch
    import cpu_ref

    cfg = vcfg.vicuna_7b("vcoder_ds")
    cfg.num_hidden_layers = 2
    cfg.vit_num_layers = 3
    sd = synth.synth_state_dict(cfg, 11)
    eng = HipEngine(cfg)
    eng.load_synthetic(11)
    eng.finalize()
    eng.set_precision("strict")
    ids = synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=3)[None]
    imgs, segs, deps = synth.synth_batch(1, 336, first=3)
    last, _, S = eng.prefill(ids, imgs, segs, deps)
    lg2, _ = eng.decode_step(np.argmax(last, -1).astype(np.int32))
    om = cpu_ref.OracleModel(cfg, sd, emu_bf16=False)
    t = torch.from_numpy
    with torch.no_grad():
        o_last, cache = om.forward(ids.tolist(), t(imgs), t(segs), t(deps), last_only=True)
        o_lg2 = om.decode_step(np.argmax(last, -1).tolist(), cache)
    e1 = np.abs(last - o_last[:, -1].numpy()).max()
    e2 = np.abs(lg2 - o_lg2[:, -1].numpy()).max()
    print(f"true-dims STRICT parity: prefill err={e1:.2e} decode err={e2:.2e}")
    assert e1 < 1e-3 and e2 < 1e-3 and int(np.argmax(last)) == int(np.argmax(o_last[:, -1].numpy()))
    eng.close()


def test_load_pretrained_model_from_hf_layout(tmp_path):
    """The reference's loader entry point on real HF-layout directories: (a) VCoder checkpoint embedding the CLIP tower,
    (b) tower in a separate local CLIP directory named by config.mm_vision_tower (the reference's layout).  Same 6-tuple,
    name-substring dispatch and processor aliasing as builder.py:25-154; generate() returns [B, T+n] with the prompt."""
    import json
    import torch
    from vcoder_amd import checkpoint
    from vcoder_amd.model import load_pretrained_model

    cfg = vcfg.tiny("vcoder_ds")
    sd = synth.synth_state_dict(cfg, 42)
    g, _, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_depth_seg")
    # (b) first: separate CLIP directory
    clip_dir = str(tmp_path / "clip-tiny")
    vt = "model.vision_tower.vision_tower."
    checkpoint.save_checkpoint(clip_dir, {"model_type": "clip_vision_model"},
                               {k[len(vt):]: v for k, v in sd.items() if k.startswith(vt)})
    with open(f"{clip_dir}/preprocessor_config.json", "w") as f:
        json.dump({"crop_size": 56, "size": 56, "do_center_crop": True, "do_normalize": True, "do_resize": True,
                   "image_mean": synth.CLIP_MEAN.tolist(), "image_std": synth.CLIP_STD.tolist(), "resample": 3,
                   "image_processor_type": "CLIPImageProcessor"}, f)
    cfg.mm_vision_tower = clip_dir
    d = str(tmp_path / "vcoder_ds_llava-v1.5-tiny")
    checkpoint.save_checkpoint(d, cfg.to_hf_dict(), {k: v for k, v in sd.items() if not k.startswith(vt)})
    tok, model, ip, sip, dip, ctx = load_pretrained_model(d, None, "vcoder_ds_llava-v1.5-tiny")
    assert ip is not None and sip is ip and dip is ip and ctx == 2048
    assert type(model).__name__ == "VCoderDSLlavaLlamaForCausalLM" and model.get_vision_tower().is_loaded
    t = torch.from_numpy
    out = model.generate(t(ids).cuda(), images=t(imgs).cuda(), segs=t(segs).cuda(), depths=t(deps).cuda(), do_sample=False,
                         max_new_tokens=8, use_cache=True, eos_token_id=-1)
    assert out.is_cuda and tuple(out.shape) == (2, ids.shape[1] + 8)
    assert np.array_equal(out[:, ids.shape[1]:].cpu().numpy(), g["greedy_ids"])
    with pytest.raises(RuntimeError):
        tok("hello")   # no tokenizer files in a synthetic checkpoint: the error surfaces lazily, not at load
    # the plugin modules are LOADED modules after load_pretrained_model (vcoder_ds_llava_arch.py:34-49): standalone forward ==
    # the oracle's projector, and == what vc_encode applies to the tower's features inside the model
    import cpu_ref
    sd_t = cpu_ref.as_torch_state(sd)
    gm = model.get_model()
    feats = t(model.engine.vision_tower_forward(imgs)).cuda()                 # CLIPVisionTower.forward output [B, 16, 128]
    for name, mod_name in (("mm_projector", "img"), ("seg_mm_projector", "seg"), ("depth_mm_projector", None), ("mm2_projector", None)):
        mod = getattr(gm, name)
        assert mod.is_loaded(), name
        got = mod(feats).float().cpu().numpy()
        ref = cpu_ref.projector_forward(feats.float().cpu(), sd_t, f"model.{name}", mod.projector_type, emu_bf16=True).numpy()
        assert np.abs(got - ref).max() < 2 ** -7 * max(1.0, np.abs(ref).max()), name
        if mod_name == "img":
            enc = model.engine.encode(imgs, "img")
            assert np.abs(enc - got).max() < 2 ** -7 * max(1.0, np.abs(got).max()), "vc_encode != tower forward + mm_projector module"
    model.engine.close()
    # (a) tower embedded in the VCoder checkpoint; non-DS name dispatch
    cfg2 = vcfg.tiny("vcoder")
    cfg2.mm_vision_tower = clip_dir
    d2 = str(tmp_path / "vcoder_llava-v1.5-tiny")
    checkpoint.save_checkpoint(d2, cfg2.to_hf_dict(), synth.synth_state_dict(cfg2, 42))
    _, model2, ip2, sip2, dip2, _ = load_pretrained_model(d2, None, "vcoder_llava-v1.5-tiny")
    assert type(model2).__name__ == "VCoderLlavaLlamaForCausalLM" and sip2 is ip2 and dip2 is None
    g2, _, ids2, imgs2, segs2, _ = e2e_cases.fixture_inputs("vc_img_seg")
    out2 = model2.generate(t(ids2), images=t(imgs2), segs=t(segs2), do_sample=False, max_new_tokens=8, eos_token_id=-1)
    assert np.array_equal(out2[:, ids2.shape[1]:].numpy(), g2["greedy_ids"])
    model2.engine.close()
    with pytest.raises(NotImplementedError):
        load_pretrained_model(d, None, "vcoder_ds_llava-v1.5-tiny", load_4bit=True)
    # load_8bit -> the W8A16 weight format: same ids as an engine that was told set_weight_format("w8a16") directly
    _, model8, _, _, _, _ = load_pretrained_model(d, None, "vcoder_ds_llava-v1.5-tiny", load_8bit=True)
    out8 = model8.generate(t(ids), images=t(imgs), segs=t(segs), depths=t(deps), do_sample=False, max_new_tokens=4,
                           eos_token_id=-1)
    eng8 = HipEngine(cfg)
    eng8.load_synthetic(42)
    eng8.set_weight_format("w8a16")
    eng8.finalize()
    ref8 = eng8.generate_greedy(ids, imgs, segs, deps, max_new_tokens=4)
    assert np.array_equal(out8[:, ids.shape[1]:].numpy(), ref8)
    model8.engine.close()
    eng8.close()


def test_auto_model_for_causal_lm_reaches_this_backend(tmp_path):
    """AutoModelForCausalLM.from_pretrained(<vcoder_ds_llava checkpoint>) after vcoder_amd.hf_register.register() — what the
    reference's AutoConfig.register / AutoModelForCausalLM.register provide (vcoder_ds_llava_llama.py:144-145): the loaded model
    is this backend's class and generates the reference fixture's ids."""
    pytest.importorskip("transformers")
    import torch
    from transformers import AutoModelForCausalLM
    from vcoder_amd import checkpoint, hf_register

    hf_register.register()
    cfg = vcfg.tiny("vcoder_ds")
    d = str(tmp_path / "vcoder_ds_llava-v1.5-tiny")
    checkpoint.save_checkpoint(d, cfg.to_hf_dict(), synth.synth_state_dict(cfg, 42))
    model = AutoModelForCausalLM.from_pretrained(d)
    assert type(model).__name__ == "VCoderDSLlavaLlamaForCausalLM" and model.get_model().mm_projector.is_loaded()
    g, _, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_depth_seg")
    t = torch.from_numpy
    out = model.generate(t(ids), images=t(imgs), segs=t(segs), depths=t(deps), do_sample=False, max_new_tokens=8, eos_token_id=-1)
    assert np.array_equal(out[:, ids.shape[1]:].numpy(), g["greedy_ids"])
    model.engine.close()


def test_load_pretrained_model_lora_and_projector_only(tmp_path):
    """load_pretrained_model(model_path, model_base, name) on the reference's two overlay layouts (builder.py:42-92): a peft
    LoRA checkpoint ('lora' in the name) merged into the base LLM, and a projector-only checkpoint; both are plain LLaVA
    models; generate() ids equal the fp32 oracle's on the merged weights in split mode."""
    import torch
    import cpu_ref
    from test_model_api_emu import _write_lora_family
    from vcoder_amd.model import load_pretrained_model

    cfg = vcfg.tiny("llava")
    sd = synth.synth_state_dict(cfg, 42)
    paths, merged = _write_lora_family(tmp_path, cfg, sd, np.random.RandomState(5))
    g, _, ids, imgs, _, _ = e2e_cases.fixture_inputs("llava_img")
    t = torch.from_numpy
    for path, name, want in ((paths["lora"], "llava-v1.5-tiny-lora", merged), (paths["ponly"], "llava-v1.5-tiny-pretrain", sd)):
        tok, model, ip, sip, dip, ctx = load_pretrained_model(path, paths["base"], name)
        assert type(model).__name__ == "LlavaLlamaForCausalLM" and sip is None and dip is None
        model.engine.set_precision("split")
        out = model.generate(t(ids), images=t(imgs), do_sample=False, max_new_tokens=6, eos_token_id=-1)
        ref, _ = cpu_ref.OracleModel(model.config, want).generate_greedy(ids.tolist(), t(imgs), max_new_tokens=6, return_logits=True)
        assert np.array_equal(out[:, ids.shape[1]:].numpy(), ref.numpy()), name
        model.engine.close()
    with pytest.warns(UserWarning, match="no `model_base`"):
        try:
            load_pretrained_model(paths["lora"], None, "llava-v1.5-tiny-lora")   # the reference warns, then loads it as a full model
        except (FileNotFoundError, ValueError):
            pass                                                                  # ... which a LoRA directory is not (adapter keys)


def test_beam_search_on_device_rows():
    """generate(num_beams=n) on the GPU: rows expanded to beams, KV rows permuted by vc_reorder_cache after every step.  The
    beam scores come from the engine's logits; with the fp32-faithful split mode the result must equal a host restatement that
    re-scores EVERY candidate sequence with the fp32 oracle: the returned sequence is the best-scoring of the n * V one-step
    extensions chain (checked as: its total log-probability under the oracle >= the greedy sequence's, and each of its steps
    lies in the top 2n of the oracle's distribution given its prefix)."""
    import torch
    import cpu_ref
    from vcoder_amd.model import language_model as lm

    cfg = vcfg.tiny("vcoder_ds")
    sd = synth.synth_state_dict(cfg, 42)
    model = lm.VCoderDSLlavaLlamaForCausalLM(cfg)
    model.load_state_dict(sd)
    model.finalize_weights()
    model.engine.set_precision("split")
    g, _, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_depth_seg")
    t = torch.from_numpy
    kw = dict(images=t(imgs), segs=t(segs), depths=t(deps), max_new_tokens=6, eos_token_id=-1)
    greedy = model.generate(t(ids), do_sample=False, **kw)
    beams = model.generate(t(ids), num_beams=4, **kw)
    assert tuple(beams.shape) == tuple(greedy.shape) and torch.equal(beams[:, : ids.shape[1]], t(ids))
    om = cpu_ref.OracleModel(cfg, sd)

    def seq_logprob(new_ids):   # teacher-forced total log-probability of the generated part under the fp32 oracle
        tot, ranks = np.zeros(ids.shape[0]), []
        logits, cache = om.forward(ids.tolist(), t(imgs), t(segs), t(deps), last_only=True)
        for s_ in range(new_ids.shape[1]):
            lp = torch.log_softmax(logits[:, -1].float(), -1)
            tok = new_ids[:, s_]
            tot += lp[torch.arange(ids.shape[0]), tok].numpy()
            ranks.append((lp > lp[torch.arange(ids.shape[0]), tok][:, None]).sum(-1).numpy())
            logits = om.decode_step(tok.tolist(), cache)
        return tot, np.stack(ranks, 1)

    lp_g, _ = seq_logprob(greedy[:, ids.shape[1]:])
    lp_b, ranks = seq_logprob(beams[:, ids.shape[1]:])
    print("beam search: log-prob greedy", lp_g, "beams", lp_b, "ranks of the chosen tokens", ranks.tolist())
    assert (lp_b >= lp_g - 1e-4).all(), "4 beams returned a sequence the oracle scores below the greedy one"
    assert (ranks < 8).all(), "a chosen token lies outside the top 2n of the oracle's distribution for its prefix"
    model.engine.close()


def test_projector_types_standalone():
    """build_vision_projector / build_seg_projector / build_depth_projector for every type string the reference accepts
    (multimodal_projector/builder.py:33-51): 'linear', 'mlp2x_gelu', 'mlp3x_gelu', 'identity' — the module's device forward
    against oracle/cpu_ref.projector_forward at the true adapter dims (1024 -> 4096)."""
    import torch
    import cpu_ref
    from vcoder_amd.model import build_depth_projector, build_seg_projector, build_vision_projector

    rng = np.random.RandomState(1)
    for fn, ptype in ((build_vision_projector, "linear"), (build_seg_projector, "mlp2x_gelu"), (build_depth_projector, "mlp3x_gelu"),
                      (build_vision_projector, "identity")):
        c = vcfg.vicuna_7b("vcoder_ds")
        c.mm_projector_type = c.seg_mm_projector_type = c.depth_mm_projector_type = ptype
        if ptype == "identity":
            c.mm_hidden_size = c.hidden_size
        mod = fn(c)
        sdp = {k: torch.from_numpy(synth.round_to_bf16((rng.randn(*mod.shape_of(k)) * 0.03).astype(np.float32))) for k in mod.keys()}
        mod.load_state_dict(sdp)
        x = torch.from_numpy(synth.round_to_bf16(rng.randn(2, 576, mod.in_features).astype(np.float32)))
        got = mod(x.cuda()).float().cpu().numpy()
        ref = cpu_ref.projector_forward(x, {"p." + k: v for k, v in sdp.items()}, "p", ptype, emu_bf16=True).numpy()
        assert got.shape == ref.shape == (2, 576, c.hidden_size)
        e = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
        print(f"projector {ptype}: rel err {e:.2e}")
        assert e < 2 ** -7, ptype


def test_device_preprocessing_matches_pil(tmp_path):
    """§8(f) row 2 on the GPU at the real tower size (336): COCO-like shapes, both aspect modes, bit-exact uint8 stages
    (1e-6 after normalisation) against PIL / CLIPImageProcessor; plus the throughput next to host PIL."""
    import time
    import preprocess_cases as pc
    from PIL import Image

    cfg = vcfg.vicuna_7b("vcoder_ds")
    cfg.num_hidden_layers = 1
    cfg.vit_num_layers = 2
    eng = HipEngine(cfg)
    eng.load_synthetic(1)
    eng.finalize()
    pc.check_preprocess(eng, [(480, 640), (640, 427), (336, 336), (500, 375), (1024, 768), (200, 300)])
    pc.check_preprocess(eng, [(480, 640)], to_device=True)
    pc.check_against_hf_processor(eng, tmp_path)
    rng = np.random.RandomState(0)
    imgs = [Image.fromarray(rng.randint(0, 256, size=(480, 640, 3)).astype(np.uint8)) for _ in range(24)]
    eng.preprocess(imgs[:2], to_device=True)
    t0 = time.perf_counter()
    eng.preprocess(imgs, to_device=True)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    for im in imgs:
        pc.expected(im, 336, True)
    t_pil = time.perf_counter() - t0
    print(f"preprocess 24 x 480x640 -> 336: device path {t_dev * 1e3:.1f} ms, host PIL+numpy {t_pil * 1e3:.1f} ms")
    eng.close()


@pytest.mark.parametrize("name,fmt", [("ds_img_depth_seg", "w8a16"), ("vc_img_seg", "w8a16"), ("ds_img_depth_seg", "fp8"),
                                      ("vc_img_seg", "fp8")])
def test_fp8_weight_format(name, fmt):
    """fp8-e4m3 decoder weights (BASELINE configs[4]) on the tiny fixtures: strict + fast paths; 'fp8' = W8A8 prefill on
    the K=128 scaled MFMA against the oracle that quantises the same activation rows."""
    r = e2e_cases.check_fp8_weights(name, n_new=8, fmt=fmt)
    print(f"{fmt} weights {name}: {r}")


@pytest.mark.parametrize("fmt", ["w8a16", "fp8"])
def test_fp8_weights_true_dims_against_oracle(fmt):
    """13b geometry (D 5120, F 13824, 2 layers), B=1: the device quantisers + byte-streaming GEMV (+ for 'fp8' the
    e4m3 x e4m3 prefill GEMMs) against the bf16-emulating oracle run on the host-quantised effective weights
    (vcoder_amd/quant.py) and, for 'fp8', quantising the same activation rows."""
    import torch
    import cpu_ref

    cfg = vcfg.vicuna_13b("vcoder_ds")
    cfg.num_hidden_layers = 2
    cfg.vit_num_layers = 2
    sd = e2e_cases.host_state(cfg, 13, effective=True)   # shared by the two parametrisations
    eng = HipEngine(cfg)
    eng.load_synthetic(13)
    eng.set_weight_format(fmt)
    eng.finalize()
    ids = synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=1)[None]
    imgs, segs, deps = synth.synth_batch(1, 336, first=1)
    last, _, S = eng.prefill(ids, imgs, segs, deps)
    lg2, _ = eng.decode_step(np.argmax(last, -1).astype(np.int32))
    om = cpu_ref.OracleModel(cfg, sd, emu_bf16=True, act_fp8=(fmt == "fp8"))
    t = torch.from_numpy
    with torch.no_grad():
        o_last, cache = om.forward(ids.tolist(), t(imgs), t(segs), t(deps), last_only=True)
        o_lg2 = om.decode_step(np.argmax(last, -1).tolist(), cache)
    o_last, o_lg2 = o_last[:, -1].numpy(), o_lg2[:, -1].numpy()
    e1, e2 = np.abs(last - o_last).max(), np.abs(lg2 - o_lg2).max()
    scale = np.abs(o_last).max()
    print(f"{fmt} true-dims parity: |logits|max={scale:.3f} prefill err={e1 / scale:.2e} decode(fp8 gemv) err={e2 / scale:.2e} (relative)")
    if fmt == "w8a16":
        # measured on MI355X: 8.7e-3 (prefill) / 1.23e-2 (decode, fp8 GEMV) relative; tolerance = 2x measured
        assert e1 < 2.5e-2 * scale and e2 < 2.5e-2 * scale
    else:
        # With e4m3 ACTIVATIONS every rounding-level perturbation re-draws the quantisation noise of the rows behind it:
        # the oracle itself moves by 1.0e-1 (relative) between its fp32 and its bf16-emulating arithmetic on this model
        # (W8A16: 7.4e-3), which is also the distance between the W8A8 and the W8A16 oracles (8.9e-2).  The device
        # (8.6e-2 measured) is held to the oracle's own sensitivity, measured here; what pins the arithmetic itself is the
        # kernel test (GEMM on quantised operands vs float64: 4e-5) and the bit-exact quantisers.
        om32 = cpu_ref.OracleModel(cfg, sd, emu_bf16=False, act_fp8=True)
        with torch.no_grad():
            p_last, _ = om32.forward(ids.tolist(), t(imgs), t(segs), t(deps), last_only=True)
        sens = np.abs(p_last[:, -1].numpy() - o_last).max()
        print(f"   oracle fp32 vs bf16-emulating arithmetic under W8A8: {sens / scale:.2e} (relative)")
        assert e1 < 2.0 * sens and e2 < 2.0 * sens
    eng.close()


@pytest.mark.parametrize("mode", ["bf16", "strict", "split"])
def test_padded_batch_attention_mask(mode):
    """a 2-D attention_mask that hides positions (row 1 right-padded, row 2 left-padded) against the live reference's fixture
    tests/golden/ds_padded_mask.npz: masked prefill logits at all positions, the cached steps under the reference's all-ones
    mask (generate) and under the carried mask (forward without images), generate() ids"""
    print(mode, e2e_cases.check_masked_fixture(mode=mode))


@pytest.mark.parametrize("mode", ["bf16", "strict", "split"])
def test_output_hidden_states(mode):
    """forward(output_hidden_states=True): the L + 1 hidden states of the prefill vs the live reference's tuple"""
    print(mode, e2e_cases.check_hidden_states(mode=mode))


def test_padded_batch_true_dims():
    """the key mask at the true 7b head geometry (S = 1216, hd 128, 2 layers): a right-padded row against the fp32 oracle in
    split mode (1e-3) — ragged masked tiles in the flash kernel, the fp32-KV decode attention with hidden keys"""
    import torch
    import cpu_ref

    cfg = vcfg.vicuna_7b("vcoder_ds")
    cfg.num_hidden_layers = 2
    cfg.vit_num_layers = 2
    sd = synth.synth_state_dict(cfg, 19)
    eng = HipEngine(cfg)
    eng.load_synthetic(19)
    eng.finalize()
    eng.set_precision("split")
    B = 2
    ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=b) for b in range(B)])
    T = ids.shape[1]
    mask = np.ones((B, T), np.int64)
    ids[1, -5:] = 0
    mask[1, -5:] = 0
    mask[0, 3:6] = 0            # hidden in the middle of row 0 as well (lands behind the feature blocks after the left extension)
    imgs, segs, deps = synth.synth_batch(B, 336)
    last, _, S = eng.prefill(ids, imgs, segs, deps, attention_mask=mask, reserve=4)
    tok = np.argmax(last, -1).astype(np.int32)
    lg_keep, _ = eng.decode_step(tok)
    last2, _, _ = eng.prefill(ids, imgs, segs, deps, attention_mask=mask, reserve=4)
    eng.clear_attention_mask()
    lg_ones, _ = eng.decode_step(tok)
    om = cpu_ref.OracleModel(cfg, sd, emu_bf16=False)
    t = torch.from_numpy
    with torch.no_grad():
        o_last, cache = om.forward(ids.tolist(), t(imgs), t(segs), t(deps), last_only=True, attention_mask=mask)
        o_keep = om.decode_step(tok.tolist(), cache, keep_mask=True)[:, -1].numpy()
        _, cache2 = om.forward(ids.tolist(), t(imgs), t(segs), t(deps), last_only=True, attention_mask=mask)
        o_ones = om.decode_step(tok.tolist(), cache2, keep_mask=False)[:, -1].numpy()
    e = [float(np.abs(a - b).max()) for a, b in ((last, o_last[:, -1].numpy()), (lg_keep, o_keep), (lg_ones, o_ones))]
    print(f"padded batch, true dims, split mode vs fp32 oracle: prefill {e[0]:.2e} keep-step {e[1]:.2e} ones-step {e[2]:.2e}; "
          f"ones-vs-keep differ by {np.abs(o_keep - o_ones).max():.3f}")
    assert max(e) < 1e-3 and np.array_equal(last, last2)
    eng.close()


@pytest.mark.parametrize("fmt", ["w8a16", "fp8"])
def test_fp8_formats_per_layer_teacher_forced(fmt):
    """13b geometry (D 5120, F 13824, 40 heads; 3 layers), the C2 prompt length S = 1216, B = 2: every layer fed the ORACLE's
    own layer input (vc_debug_prefill_layers), output compared with the oracle's for that input — the quantisation noise of
    the layers in front cannot compound: W8A16 rms 6e-4 / max 3.6e-3 of max|x_out| (tolerance 2e-3 / 2e-2); with e4m3 activation
    rows ('fp8') rms 7e-3 / max 3.8e-2 (tolerance 1.5e-2 / 8e-2) — the format's own re-rounding quantum, see
    check_layers_teacher_forced.  BASELINE configs[4]'s arithmetic: e4m3 weights; 'fp8' adds e4m3 activation rows on the K=128
    scaled MFMA."""
    cfg = vcfg.vicuna_13b("vcoder_ds")
    cfg.num_hidden_layers = 3
    cfg.vit_num_layers = 2
    r = e2e_cases.check_layers_teacher_forced(cfg, 17, fmt, layers=(0, 1, 2), B=2)
    print(fmt, "per-layer deviation relative to |x_out|max (max, rms):", {l: (f"{a:.2e}", f"{b:.2e}") for l, (a, b) in r.items()})


@pytest.mark.parametrize("fmt", ["w8a16", "fp8"])
def test_fp8_formats_per_layer_with_massive_activation_channels(fmt):
    """The same at the 13b geometry with three hidden channels of the residual stream 300 x the rest — the "massive activation"
    profile of trained LLaMA-family checkpoints, which the seeded synthetic checkpoint lacks: a token row's e4m3 scale is set by
    channels that carry no information.  Device vs the oracle quantising the same rows, relative to the layer's UPDATE y - x
    (|y|max is the outlier itself): W8A16 rms 2e-3 / max 2e-2, 'fp8' rms 1.5e-2 / max 8e-2 (the tolerances of the plain profile);
    'format' = what the e4m3 activation rows themselves cost on these inputs (oracle vs oracle; profiles/r05_fp8_outlier_study.txt:
    4.7 % of the update without outliers, 5.1 % with)."""
    cfg = vcfg.vicuna_13b("vcoder_ds")
    cfg.num_hidden_layers = 2
    cfg.vit_num_layers = 2
    r = e2e_cases.check_layers_teacher_forced(cfg, 17, fmt, layers=(0, 1), B=2, S=640, outlier_gain=300.0)
    print(fmt, "massive-activation profile, deviation relative to the layer update (max, rms):",
          {l: (f"{a:.2e}", f"{b:.2e}") for l, (a, b) in r["update"].items()}, "format cost:", r.get("format"))
    if fmt == "fp8":
        assert max(r["format"].values()) < 0.08, r["format"]


@pytest.mark.parametrize("name", ["ds_img_depth_seg", "vc_img_seg"])
def test_device_side_stop_sequences(name):
    e2e_cases.check_stop_sequences(name)


@pytest.mark.parametrize("name", ["ds_list_two_each", "ds_list_uneven"])
def test_list_and_5d_image_inputs(name):
    """list / 5-D image form (vcoder_ds_llava_arch.py:135-169), several images per sample, vs the live reference's fixture"""
    print(name, e2e_cases.check_list_fixture(name))


def test_vision_tower_boundary():
    """a2 at its own boundary: vc_vision_tower_forward vs CLIPVisionTower.forward of the live reference (tiny tower fixture)
    and vs cpu_ref.vit_forward at the true ViT-L/14@336 dimensions (23 of 24 layers, 577 tokens)."""
    import torch
    import cpu_ref

    print("tower fixture: strict", e2e_cases.check_tower_fixture(strict=True), "bf16", e2e_cases.check_tower_fixture())
    cfg = vcfg.vicuna_7b("vcoder_ds")
    cfg.num_hidden_layers = 1
    eng = HipEngine(cfg)
    eng.load_synthetic(3)
    eng.finalize()
    imgs = synth.synth_batch(2, 336, first=5)[0]
    sd = cpu_ref.as_torch_state(synth.synth_state_dict(cfg, 3, only_prefix="model.vision_tower"))
    with torch.no_grad():
        ref32 = cpu_ref.vit_forward(torch.from_numpy(imgs), sd, cfg).numpy()
        ref_emu = cpu_ref.vit_forward(torch.from_numpy(imgs), sd, cfg, emu_bf16=True).numpy()
    got = eng.vision_tower_forward(imgs)
    assert got.shape == (2, 576, 1024)
    scale = float(np.abs(ref32).max())
    e32, eemu = float(np.abs(got - ref32).max()), float(np.abs(got - ref_emu).max())
    eng.set_precision("strict")
    es = float(np.abs(eng.vision_tower_forward(imgs) - ref32).max())
    print(f"ViT-L/14@336 tower, 23 layers: |feat|max={scale:.2f}  bf16 path vs fp32 oracle {e32:.4f} (rel {e32 / scale:.2e}), "
          f"vs bf16-emulating oracle {eemu:.4f}, strict vs fp32 oracle {es:.2e}")
    assert e32 < 3e-2 * scale and eemu < 2e-2 * scale and es < 1e-3 * max(1.0, scale)
    eng.close()


def test_sampling_on_device_true_vocab():
    """Device sampling at the real vocabulary (32000 logits staged in LDS): support and distribution vs the HF warpers."""
    import kernel_cases as kc

    be = kc.HipBackend()
    print("TV T=0.2 k=50:", kc.check_sampling(be, 32000, 0.2, 50, 1.0, draws=4096))
    print("TV T=1.0 p=0.7:", kc.check_sampling(be, 32000, 1.0, 0, 0.7, draws=4096))
    print("TV T=0.7 k=20 p=0.9:", kc.check_sampling(be, 32000, 0.7, 20, 0.9, draws=2048))
    kc.check_select_embed(be, 8, 32000, 4096)
    kc.check_uniform_extremes(be)


def test_cost_harness_batched_equals_per_sample(tmp_path):
    """SURVEY §8(f) row 3 on the GPU: the batched COST harness (eval_task over a synthetic image folder, device-side
    preprocessing, question-bucketed batches) writes the same answers as one-sample-at-a-time generation, in the
    reference's answers-file format (model_seg_loader.py:160-166, file names as keys)."""
    from PIL import Image
    from vcoder_amd.eval import cost_eval
    from vcoder_amd.model import language_model as lm

    cfg = vcfg.tiny("vcoder_ds")
    model = lm.VCoderDSLlavaLlamaForCausalLM(cfg)
    model.engine.load_synthetic(42)
    model.finalize_weights()

    class Tok:
        bos_token_id, eos_token_id = 1, 2

        def __call__(self, text):
            class R:
                pass
            r = R()
            r.input_ids = [1] + [3 + (b % (cfg.vocab_size - 3)) for b in text.encode()]
            return r

        def batch_decode(self, rows, skip_special_tokens=True):
            return [" ".join(str(int(t)) for t in r if not (skip_special_tokens and int(t) in (0, 1, 2))) for r in rows]

    rng = np.random.RandomState(0)
    for sub in ("images", "segs/semantic_inference", "depths"):
        (tmp_path / sub).mkdir(parents=True)
    for i in range(7):
        for sub in ("images", "segs/semantic_inference", "depths"):
            Image.fromarray(rng.randint(0, 256, size=(70 + i, 90, 3)).astype(np.uint8)).save(tmp_path / sub / f"{i:03d}.jpg")
    qs = ["What objects can be seen in the image?", "List the objects."]
    kw = dict(questions=qs, max_new_tokens=5, seed=3)
    out_b = cost_eval.eval_task(model, Tok(), "semantic", str(tmp_path / "images"), str(tmp_path / "segs"),
                                str(tmp_path / "ans_batched"), depth_image_folder=str(tmp_path / "depths"), batch_size=4, **kw)
    out_1 = cost_eval.eval_task(model, Tok(), "semantic", str(tmp_path / "images"), str(tmp_path / "segs"),
                                str(tmp_path / "ans_single"), depth_image_folder=str(tmp_path / "depths"), batch_size=1, **kw)
    a, b = open(out_b).read(), open(out_1).read()
    assert a == b and a.count("<<ANSWER>>:") == 7
    assert "Image: 000.jpg\n" in a and str(tmp_path) not in a          # bare file names, as the reference's scorers expect
    model.engine.close()


def test_cost_answers_equal_reference_loaders(tmp_path):
    """SURVEY §8(f) row 3 pinned to the reference: eval_task (batched, device preprocessing, device greedy loop) writes byte for
    byte the answers files the REFERENCE'S model_seg_loader.eval_model / model_depth_loader.eval_model wrote for the same
    folder / checkpoint / tokenizer / question seed (tests/golden/cost <- oracle/gen_cost_golden.py), in split mode (the mode
    that meets the 1e-3 bar) and in strict mode; batch sizes 4 and 1 agree."""
    done = e2e_cases.check_cost_answers(str(tmp_path / "split"), mode="split")
    assert done == {"semantic_1_0": 6, "panoptic_2_1": 3, "semantic_noseg_1_0": 6, "depth_1_0": 6, "depth_2_0": 3,
                    "depth_noseg_1_0": 6}
    e2e_cases.check_cost_answers(str(tmp_path / "strict"), mode="strict", runs=["depth_1_0", "semantic_noseg_1_0"])
    e2e_cases.check_cost_answers(str(tmp_path / "b1"), mode="split", runs=["depth_1_0"], batch_size=1)


def test_bench_two_ranks_self_launched(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts its own two ranks (torch.distributed.run on 127.0.0.1), shards
    the global batch contiguously, and the gathered token stream equals the single-process one.  On this 1-GPU box both
    ranks share device 0 and the id gather runs over gloo (VC_BENCH_FORCE_DEVICE / VC_BENCH_BACKEND: RCCL refuses two ranks
    on one GPU); on a multi-GPU node the same command uses one GPU per rank and RCCL."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = str(tmp_path / "ids.npy")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(VC_BENCH_FORCE_DEVICE="0", VC_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--batch",
                        "2", "--new-tokens", "6", "--inflight", "2", "--dump-ids", dump], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 4 and res["config"]["parallelism"] == "dp2"
    assert res["value"] > 0 and "roofline" in res and "composite_roofline" in res and "pcie_inclusive" in res
    # a world > 1 line never times the CPU baseline: it carries the cached N = 1 measurement of the box or says that there is none
    assert "cpu_baseline" in res and ("measured_at" in res["cpu_baseline"] or res["cpu_baseline"]["value"] is None)
    got = np.load(dump)
    assert got.shape == (4, 6)
    cfg = vcfg.vicuna_7b("vcoder_ds")
    eng = HipEngine(cfg)
    eng.load_synthetic(42)
    eng.finalize()
    ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=b) for b in range(4)])
    imgs, segs, deps = synth.synth_batch(4, 336)
    # the same shards the ranks ran (2 samples each): a GEMM's split-K remainder round depends on the number of tiles, so a
    # prefill of 4 samples is not bit-for-bit two prefills of 2 at these dimensions, and greedy ids on random weights
    # have near-ties that notice
    ref = np.concatenate([eng.generate_greedy(ids[r:r + 2], imgs[r:r + 2], segs[r:r + 2], deps[r:r + 2], max_new_tokens=6)
                          for r in (0, 2)], axis=0)
    eng.close()
    assert np.array_equal(got, ref), "gathered ids of the 2-rank run differ from the single-process ids"


def test_batch_invariant_mode_true_dims():
    """vc_model_set_batch_invariant at true 7b dims (2 layers): one batch of 4 gives every sample the BITS — prefill logits and
    greedy ids — of two batches of 2 and of four batches of 1 (SURVEY §4 test 4 / §0 quirk 6: a rank's shard of a global batch
    equals the single-GPU run of that batch), in the bf16 and the split mode.  (Without it the split-K remainder round of a
    prefill GEMM depends on the tile count: the comparison below is expected to find differing low-order bits at these sizes.)"""
    cfg = vcfg.vicuna_7b("vcoder_ds")
    cfg.num_hidden_layers = 2
    cfg.vit_num_layers = 3
    eng = HipEngine(cfg)
    eng.load_synthetic(21)
    eng.finalize()
    ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=b) for b in range(4)])
    imgs, segs, deps = synth.synth_batch(4, 336)

    def run(lo, hi):
        last, _, _ = eng.prefill(ids[lo:hi], imgs[lo:hi], segs[lo:hi], deps[lo:hi])
        return last, eng.generate_greedy(ids[lo:hi], imgs[lo:hi], segs[lo:hi], deps[lo:hi], max_new_tokens=6)
    base4, _ = run(0, 4)
    differs_by_default = not np.array_equal(base4[:2], run(0, 2)[0])
    for mode in ("bf16", "split"):
        eng.set_precision(mode)
        eng.set_batch_invariant(True)
        l4, g4 = run(0, 4)
        for lo, hi in ((0, 2), (2, 4), (1, 2), (3, 4)):
            l, g_ = run(lo, hi)
            assert np.array_equal(l, l4[lo:hi]), f"{mode}: prefill logits of samples {lo}..{hi - 1} depend on the batch"
            assert np.array_equal(g_, g4[lo:hi]), f"{mode}: greedy ids of samples {lo}..{hi - 1} depend on the batch"
        eng.set_batch_invariant(False)
    eng.set_precision("bf16")
    print("default mode: a batch of 4 and a batch of 2 differ in the low-order bits of the prefill logits:", differs_by_default)
    eng.close()


@pytest.mark.parametrize("gather", ["torch", "cabi"])
def test_bench_force_dist_one_gpu(tmp_path, gather):
    """`bench.py --gpus 1 --force-dist`: the multi-GPU code path on ONE GPU with the REAL backend — init_process_group("nccl",
    world_size = 1) (= RCCL), the per-step id all-gather (all_gather_into_tensor on the device, or the library's one-rank RCCL
    communicator with --gather cabi), barrier(device_ids=...) fences and the MAX all-reduce of the timing — so that the
    driver's 8-GPU run is not the first execution of this code.  The gathered ids equal the plain single-process ids."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = str(tmp_path / "ids.npy")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "VC_COMM_FORCE_RCCL")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--gather", gather, "--steps",
                        "2", "--warmup", "1", "--batch", "2", "--new-tokens", "6", "--inflight", "2", "--no-cpu-baseline",
                        "--dump-ids", dump], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["config"]["force_dist"] is True and res["ids_checked"] is True
    assert ("RCCL via the C ABI)" in res["config"]["token_gather"]) if gather == "cabi" else ("(nccl)" in res["config"]["token_gather"])
    got = np.load(dump)
    cfg = vcfg.vicuna_7b("vcoder_ds")
    eng = HipEngine(cfg)
    eng.load_synthetic(42)
    eng.finalize()
    ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=b) for b in range(2)])
    imgs, segs, deps = synth.synth_batch(2, 336)
    ref = eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=6)
    eng.close()
    assert np.array_equal(got, ref)


def test_token_comm_c_abi_world1():
    """vc_comm_create / vc_allgather_tokens at world 1 (identity, no RCCL needed); the multi-rank form needs one GPU per
    rank and runs in the driver's multi-GPU bench (`bench.py --gather cabi`)."""
    from vcoder_amd.parallel import TokenComm

    eng = e2e_cases.engine_for("vcoder_ds")
    c = TokenComm(eng, 0, 1)
    x = np.arange(24, dtype=np.int32).reshape(3, 8)
    assert np.array_equal(c.allgather(x), x)
    c.close()


def test_token_comm_rccl_forced_world1(monkeypatch):
    """VC_COMM_FORCE_RCCL=1: a communicator of ONE rank that still dlopens librccl, takes a unique id (by-value ncclUniqueId
    ABI), runs ncclCommInitRank(nranks = 1) and serves vc_allgather_tokens with ncclAllGather on the engine's stream — the code
    the multi-GPU run executes (scripts/v1_5/eval/cost_depth.sh:10-34 is the reference's counterpart), exercised here on one
    GPU, interleaved with kernels on the same stream, including a buffer growth and the explicit-unique-id form."""
    import ctypes as C

    from vcoder_amd.parallel import TokenComm

    monkeypatch.setenv("VC_COMM_FORCE_RCCL", "1")
    eng = e2e_cases.engine_for("vcoder_ds")
    g, cfg, ids, imgs, segs, deps = e2e_cases.fixture_inputs("ds_img_depth_seg")
    c = TokenComm(eng, 0, 1)
    assert c.uses_rccl
    new = eng.generate_greedy(ids, imgs, segs, deps, max_new_tokens=8)       # kernels on the stream the gather uses
    assert np.array_equal(c.allgather(new), new)
    big = np.arange(5 * 4096, dtype=np.int32).reshape(5, 4096)              # grows the device buffers
    assert np.array_equal(c.allgather(big), big)
    assert np.array_equal(c.allgather(new), new)
    c.close()
    # the explicit unique-id form the ranks of a real world use (rank 0: vc_comm_unique_id -> vc_comm_create)
    uid = C.create_string_buffer(128)
    eng._check(eng.lib.vc_comm_unique_id(eng._ctx, uid))
    assert any(uid.raw)
    comm = C.c_void_p()
    eng._check(eng.lib.vc_comm_create(eng._ctx, 0, 1, uid, C.byref(comm)))
    assert eng.lib.vc_comm_uses_rccl(comm) == 1
    out = np.empty_like(new)
    eng._check(eng.lib.vc_allgather_tokens(comm, new.ctypes.data_as(C.c_void_p), int(new.size), out.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(out, new)
    eng.lib.vc_comm_destroy(comm)
    monkeypatch.delenv("VC_COMM_FORCE_RCCL")
    c = TokenComm(eng, 0, 1)
    assert not c.uses_rccl and np.array_equal(c.allgather(new), new)         # default: no RCCL in a world of one
    c.close()


def test_decode_pool_tiny():
    """the decode pool on the GPU (tiny model): the three scenarios of tests/test_pool_emu.py — concurrent requests of
    different shapes, mixed EOS / stop / sampling parameters, more requests than rows — each equal to its lone run"""
    import test_pool_emu as tp

    tp.test_concurrent_requests_share_steps_and_keep_their_ids(None)
    tp.test_pool_profile_counts_every_launch_and_changes_no_id(None)      # round 5: in-situ timing slots
    tp.test_pool_hold_policy_changes_scheduling_not_ids(None)             # round 5: vc_pool_set_hold
    tp.test_pool_mixes_eos_stops_and_sampling(None)
    tp.test_pool_queues_requests_beyond_its_rows(None)


def test_decode_pool_true_dims():
    """TRUE 7b dimensions (4 decoder layers): three concurrent generate() calls of batch 8 — the bench's configuration —
    share 32-row decode steps (two MFMA row groups per weight pass) and each gets bit-for-bit the ids of the session's own
    16-row loop; a lone call (16-row steps) gets them too."""
    import threading
    import test_pool_emu as tp

    cfg = vcfg.vicuna_7b("vcoder_ds")
    cfg.num_hidden_layers = 4
    cfg.vit_num_layers = 3
    root = HipEngine(cfg)
    root.load_synthetic(9)
    root.finalize()
    B, n = 8, 12
    cases, refs = [], []
    for k in range(3):
        ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=8 * k + b) for b in range(B)])
        px = synth.synth_batch(B, 336, first=8 * k)
        cases.append((ids, *px))
        refs.append(tp.session_loop_ids(root, ids, *px, n))
    assert np.array_equal(root.generate_greedy(*cases[0], max_new_tokens=n), refs[0])      # lone request: 16-row steps
    sessions = [root, root.fork(), root.fork()]
    outs, errs = [[None] * 2 for _ in sessions], []

    def work(si):
        try:
            for j in range(2):
                outs[si][j] = sessions[si].generate_greedy(*cases[si], max_new_tokens=n)
        except BaseException as e:
            errs.append(e)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for si in range(3):
        for j in range(2):
            assert np.array_equal(outs[si][j], refs[si]), f"session {si} run {j}: pooled ids differ from the session loop"
    for s in sessions[1:]:
        s.close()
    root.close()


@pytest.mark.parametrize("variant,seed,mode", [("vcoder_ds", 31, "split"), ("vcoder_ds", 32, "strict"), ("vcoder", 33, "split"),
                                               ("llava", 34, "split")])
def test_random_prompt_structures_and_configs(variant, seed, mode):
    """The randomised differential check of tests/test_fuzz_emu.py on the device: random prompt structures (placeholder orders,
    missing modalities, list-form images, masks with holes, unequal lengths) on the tiny architecture and on a random variation
    of it, engine against the fp32 oracle — same outcome class, all-position logits within 1e-3, greedy ids equal."""
    import test_fuzz_emu as fz

    rng = np.random.RandomState(seed)
    print(variant, mode, fz._run_cases(e2e_cases.engine_for(variant), 24, rng, mode))
    over = fz.fuzz_cases.random_overrides(rng, variant)
    eng = HipEngine(e2e_cases.tiny_cfg(variant, over))
    try:
        eng.load_synthetic(42)
        eng.finalize()
        print(variant, mode, over, fz._run_cases(eng, 16, rng, mode))
    finally:
        eng.close()


def test_pool_random_schedule_and_limits_on_the_device():
    """The CPU suite's randomised pool schedule (five sessions, random EOS / stop / sampling / masks / delays, one failing call),
    the context-limit wall and the empty-input refusals, on the real device: same assertions, true concurrency."""
    import test_engine_emu as te
    import test_pool_emu as tp

    for seed in (2024, 7, 77, 777, 7777, 77777):      # a second on the device each
        tp.test_pool_random_schedule_stress(None, seed)
    te.test_context_limit_is_a_clean_error(None)
    te.test_empty_inputs_are_refused(None)
