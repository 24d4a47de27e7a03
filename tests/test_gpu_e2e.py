"""`-m gpu`: the hot path end to end through the C ABI against the committed reference fixtures
(inputs_embeds, prefill logits at every position, greedy ids, the quirks) on the tiny golden models."""
import numpy as np
import pytest

import e2e_cases
from vcoder_amd import config as vcfg, synth
from vcoder_amd.engine import HipEngine

pytestmark = pytest.mark.gpu

FIXTURES = ["ds_img_depth_seg", "ds_img_seg_depth", "ds_img_seg", "ds_img_only", "ds_zero_depth", "ds_img_text_seg",
            "vc_img_seg", "vc_img_text_seg", "llava_img"]


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
    assert e1 < 2e-2 * max(1.0, scale) and e2 < 2e-2 * max(1.0, scale)
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
    # load_8bit -> the W8A16 weight format: same ids as an engine that was told set_weight_format("fp8") directly
    _, model8, _, _, _, _ = load_pretrained_model(d, None, "vcoder_ds_llava-v1.5-tiny", load_8bit=True)
    out8 = model8.generate(t(ids), images=t(imgs), segs=t(segs), depths=t(deps), do_sample=False, max_new_tokens=4,
                           eos_token_id=-1)
    eng8 = HipEngine(cfg)
    eng8.load_synthetic(42)
    eng8.set_weight_format("fp8")
    eng8.finalize()
    ref8 = eng8.generate_greedy(ids, imgs, segs, deps, max_new_tokens=4)
    assert np.array_equal(out8[:, ids.shape[1]:].numpy(), ref8)
    model8.engine.close()
    eng8.close()


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


@pytest.mark.parametrize("name", ["ds_img_depth_seg", "vc_img_seg"])
def test_fp8_weight_format(name):
    """W8A16 decoder weights (BASELINE configs[4] weight format) on the tiny fixtures: strict + fast paths."""
    r = e2e_cases.check_fp8_weights(name, n_new=8)
    print(f"fp8 weights {name}: {r}")


def test_fp8_weights_true_dims_against_oracle():
    """13b geometry (D 5120, F 13824, 2 layers), B=2: the device quantiser + byte-streaming GEMV against the bf16-emulating
    oracle run on the host-quantised effective weights (vcoder_amd/quant.py)."""
    import torch
    import cpu_ref
    from vcoder_amd import quant

    cfg = vcfg.vicuna_13b("vcoder_ds")
    cfg.num_hidden_layers = 2
    cfg.vit_num_layers = 2
    sd = quant.effective_state_dict(synth.synth_state_dict(cfg, 13))
    eng = HipEngine(cfg)
    eng.load_synthetic(13)
    eng.set_weight_format("fp8")
    eng.finalize()
    ids = synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=1)[None]
    imgs, segs, deps = synth.synth_batch(1, 336, first=1)
    last, _, S = eng.prefill(ids, imgs, segs, deps)
    lg2, _ = eng.decode_step(np.argmax(last, -1).astype(np.int32))
    om = cpu_ref.OracleModel(cfg, sd, emu_bf16=True)
    t = torch.from_numpy
    with torch.no_grad():
        o_last, cache = om.forward(ids.tolist(), t(imgs), t(segs), t(deps), last_only=True)
        o_lg2 = om.decode_step(np.argmax(last, -1).tolist(), cache)
    o_last, o_lg2 = o_last[:, -1].numpy(), o_lg2[:, -1].numpy()
    e1, e2 = np.abs(last - o_last).max(), np.abs(lg2 - o_lg2).max()
    scale = np.abs(o_last).max()
    print(f"fp8 true-dims parity: |logits|max={scale:.3f} prefill err={e1:.4f} decode(fp8 gemv) err={e2:.4f}")
    assert e1 < 2e-2 * max(1.0, scale) and e2 < 2e-2 * max(1.0, scale)
    eng.close()


@pytest.mark.parametrize("name", ["ds_img_depth_seg", "vc_img_seg"])
def test_device_side_stop_sequences(name):
    e2e_cases.check_stop_sequences(name)
