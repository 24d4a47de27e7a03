"""TEST INFRASTRUCTURE — generates tests/golden/*.npz by running the REAL reference
(/root/reference, imported through oracle/ref_shim.py, CPU fp32) on seeded synthetic checkpoints,
and verifies oracle/cpu_ref.py against it while doing so.

Run in the build container only:   python oracle/gen_golden.py
Fixtures hold inputs + the reference's outputs (data only); weights are NOT stored — they are
regenerated bit-exactly from (name, seed) by vcoder_amd/synth.py.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
import cpu_ref  # noqa: E402
from vcoder_amd import config as vcfg  # noqa: E402
from vcoder_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SEED = 42


def make_clip_dir(cfg, d):
    from transformers import CLIPVisionConfig, CLIPVisionModel

    c = CLIPVisionConfig(hidden_size=cfg.mm_hidden_size, intermediate_size=cfg.vit_intermediate_size,
                         num_hidden_layers=cfg.vit_num_layers, num_attention_heads=cfg.vit_num_heads,
                         image_size=cfg.vit_image_size, patch_size=cfg.vit_patch_size,
                         layer_norm_eps=cfg.vit_layer_norm_eps, hidden_act="quick_gelu")
    CLIPVisionModel(c).save_pretrained(d)
    with open(os.path.join(d, "preprocessor_config.json"), "w") as f:
        json.dump({"crop_size": cfg.vit_image_size, "size": cfg.vit_image_size, "do_center_crop": True,
                   "do_normalize": True, "do_resize": True, "image_mean": synth.CLIP_MEAN.tolist(),
                   "image_std": synth.CLIP_STD.tolist(), "resample": 3,
                   "image_processor_type": "CLIPImageProcessor"}, f)


def build_reference_model(cfg, sd_np, clip_dir):
    ref = ref_shim.load_reference()
    common = dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                  num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                  num_key_value_heads=cfg.num_attention_heads, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                  rms_norm_eps=cfg.rms_norm_eps, max_position_embeddings=cfg.max_position_embeddings,
                  mm_vision_tower=clip_dir, mm_projector_type=cfg.mm_projector_type,
                  mm_hidden_size=cfg.mm_hidden_size, mm_vision_select_layer=cfg.mm_vision_select_layer,
                  mm_vision_select_feature=cfg.mm_vision_select_feature, attn_implementation="eager")
    if cfg.variant == "vcoder_ds":
        c = ref.VCoderDSLlavaConfig(**common, seg_mm_projector_type=cfg.seg_mm_projector_type,
                                    seg_mm_hidden_size=cfg.seg_mm_hidden_size,
                                    depth_mm_projector_type=cfg.depth_mm_projector_type,
                                    depth_mm_hidden_size=cfg.depth_mm_hidden_size,
                                    mm_vcoder_lm_emb=True, use_mm2_proj=cfg.use_mm2_proj)
        model = ref.VCoderDSLlavaLlamaForCausalLM(c)
    elif cfg.variant == "vcoder":
        c = ref.VCoderLlavaConfig(**common, seg_mm_projector_type=cfg.seg_mm_projector_type,
                                  seg_mm_hidden_size=cfg.seg_mm_hidden_size, mm_vcoder_lm_emb=True,
                                  use_mm2_proj=cfg.use_mm2_proj)
        model = ref.VCoderLlavaLlamaForCausalLM(c)
    else:
        c = ref.LlavaConfig(**common)
        model = ref.LlavaLlamaForCausalLM(c)
    model = model.eval().float()
    model.get_vision_tower().load_model()
    # load the synthetic checkpoint by key (accept both CLIP prefix generations)
    tgt = model.state_dict()
    missing = []
    with torch.no_grad():
        for k, t in tgt.items():
            cand = [k, k.replace("vision_tower.vision_tower.", "vision_tower.vision_tower.vision_model.")]
            src = next((sd_np[c_] for c_ in cand if c_ in sd_np), None)
            if src is None:
                missing.append(k)
                continue
            t.copy_(torch.from_numpy(src).reshape(t.shape))
    missing = [k for k in missing if "rotary" not in k and "position_ids" not in k]
    assert not missing, missing
    return model


@torch.no_grad()
def ref_greedy(model, ids, images, segs, depths, n_new, variant):
    """Harness-side greedy loop: model.generate() of the reference breaks under Transformers 5.x
    (vcoder_ds_llava_arch.py:132 indexes a DynamicCache), so run the no-cache form: re-run forward on
    growing ids, argmax of the last row."""
    cur = ids.clone()
    kw = {"images": images}
    if variant != "llava":
        kw["segs"] = segs
    if variant == "vcoder_ds":
        kw["depths"] = depths
    toks, lg, first = [], [], None
    for _ in range(n_new):
        out = model(input_ids=cur, use_cache=False, **kw)
        if first is None:
            first = out.logits.float().numpy()
        last = out.logits[:, -1].float()
        lg.append(last.numpy())
        nxt = last.argmax(-1)
        toks.append(nxt.numpy())
        cur = torch.cat([cur, nxt[:, None]], 1)
    return np.stack(toks, 1), np.stack(lg, 1), first


def top2_margin(lg):
    s = np.sort(lg, axis=-1)
    return s[..., -1] - s[..., -2]


def run_case(model, oracle, cfg, name, ids_list, use_seg=True, use_depth=True, n_new=8, B=None, zero_depth=False,
             cfg_overrides=None):
    B = len(ids_list)
    size = cfg.vit_image_size
    imgs, segs, deps = synth.synth_batch(B, size)
    if zero_depth:
        deps = np.zeros_like(deps)
    ids = torch.tensor(np.stack(ids_list), dtype=torch.long)
    ti, ts, td = (torch.from_numpy(a) for a in (imgs, segs, deps))
    toks, lg, full = ref_greedy(model, ids, ti, ts if use_seg else None, td if use_depth else None, n_new, cfg.variant)
    # reference-side inputs_embeds of the prefill
    args = [ids, torch.ones_like(ids), None, None, ti]
    if cfg.variant != "llava":
        args.append(ts if use_seg else None)
    if cfg.variant == "vcoder_ds":
        args.append(td if use_depth else None)
    emb = model.prepare_inputs_labels_for_multimodal(*args)[3].detach().float().numpy()
    # ---- pin the oracle against the live reference
    o_emb, _ = oracle.prepare_inputs(ids.tolist(), ti, ts if use_seg else None, td if use_depth else None)
    o_ids, o_lg = oracle.generate_greedy(ids.tolist(), ti, ts if use_seg else None, td if use_depth else None,
                                         max_new_tokens=n_new, return_logits=True)
    o_full, _ = oracle.forward(ids.tolist(), ti, ts if use_seg else None, td if use_depth else None)
    e1 = float(np.abs(o_emb.numpy() - emb).max())
    e2 = float(np.abs(o_lg.numpy() - lg).max())
    e3 = float(np.abs(o_full.numpy() - full).max())
    same = bool((o_ids.numpy() == toks).all())
    print(f"[{name}] S={emb.shape[1]} embeds|d|={e1:.2e} logits|d|={e2:.2e} full|d|={e3:.2e} ids_equal={same} "
          f"min top2 margin={top2_margin(lg).min():.3e}")
    assert e1 < 1e-5 and e2 < 2e-4 and e3 < 2e-4 and same, "oracle/cpu_ref.py disagrees with the reference"
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), variant=cfg.variant, seed=SEED, input_ids=ids.numpy(),
                        use_seg=use_seg, use_depth=use_depth, zero_depth=zero_depth,
                        spliced_len=emb.shape[1], embeds_rowsum=emb.sum(-1).astype(np.float32),
                        embeds_sample=emb[:, ::7, ::16].astype(np.float32),
                        prefill_logits=full.astype(np.float32), greedy_ids=toks.astype(np.int64),
                        step_logits=lg.astype(np.float32), top2_margin=top2_margin(lg).astype(np.float32),
                        cfg_overrides=json.dumps(cfg_overrides or {}))


def run_list_case(model, oracle, cfg, name, ids_list, counts):
    """The list / 5-D image input form (vcoder_ds_llava_arch.py:135-169): sample b owns counts[k][b] images of modality k
    (k = img, seg, depth); their features are spliced as ONE block at the placeholder.  Stores the reference's spliced
    length, inputs_embeds sample and prefill logits; the pixel data are regenerated from synth.synth_batch."""
    B = len(ids_list)
    size = cfg.vit_image_size
    ids = torch.tensor(np.stack(ids_list), dtype=torch.long)
    tot = [int(sum(c)) for c in counts]
    pools = synth.synth_batch(max(tot), size)                       # (images, segs, depths) pools of max(tot) images
    lists = []
    for k in range(3):
        off, items = 0, []
        for b in range(B):
            items.append(torch.from_numpy(pools[k][off:off + counts[k][b]]))
            off += counts[k][b]
        lists.append(items)
    with torch.no_grad():
        out = model(input_ids=ids, images=lists[0], segs=lists[1], depths=lists[2], use_cache=False)
        full = out.logits.float().numpy()
        emb = model.prepare_inputs_labels_for_multimodal(ids, torch.ones_like(ids), None, None, lists[0], lists[1],
                                                         lists[2])[3].detach().float().numpy()
        # the 5-D tensor form of the same inputs (equal counts only) must agree with the list form
        if all(len(set(c)) == 1 for c in counts):
            five = [torch.stack(l, 0) for l in lists]
            full5 = model(input_ids=ids, images=five[0], segs=five[1], depths=five[2], use_cache=False).logits.float().numpy()
            assert np.array_equal(full, full5), "5-D form differs from the list form"
    o_emb, _ = oracle.prepare_inputs(ids.tolist(), lists[0], lists[1], lists[2])
    o_full, _ = oracle.forward(ids.tolist(), lists[0], lists[1], lists[2])
    e1, e3 = float(np.abs(o_emb.numpy() - emb).max()), float(np.abs(o_full.numpy() - full).max())
    print(f"[{name}] S={emb.shape[1]} embeds|d|={e1:.2e} full|d|={e3:.2e}")
    assert e1 < 1e-5 and e3 < 2e-4, "oracle/cpu_ref.py disagrees with the reference on list-form images"
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), variant=cfg.variant, seed=SEED, input_ids=ids.numpy(),
                        counts=np.asarray(counts, dtype=np.int64), spliced_len=emb.shape[1],
                        embeds_sample=emb[:, ::7, ::16].astype(np.float32), prefill_logits=full.astype(np.float32))


def tower_vectors(model, cfg, name):
    """a2: CLIPVisionTower.forward of the live reference (clip_encoder.py:39-51) on the tiny tower — un-projected
    hidden_states[-2] without CLS — for 3 synthetic images; also pins cpu_ref.vit_forward."""
    imgs = torch.from_numpy(synth.synth_batch(3, cfg.vit_image_size)[0])
    with torch.no_grad():
        feats = model.get_vision_tower()(imgs).float().numpy()
    sd = cpu_ref.as_torch_state(synth.synth_state_dict(cfg, SEED, only_prefix="model.vision_tower"))
    o = cpu_ref.vit_forward(imgs, sd, cfg).numpy()
    e = float(np.abs(o - feats).max())
    print(f"[{name}] tower features {feats.shape} oracle|d|={e:.2e}")
    assert e < 1e-5
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), variant=cfg.variant, seed=SEED, features=feats.astype(np.float32))


def per_op_vectors():
    """G3: per-op vectors at TRUE inner dims from the third-party modules the reference calls
    (HF CLIP / Llama building blocks + torch.nn), a few rows each."""
    from transformers.models.llama.modeling_llama import LlamaRMSNorm, LlamaRotaryEmbedding, apply_rotary_pos_emb
    from transformers import LlamaConfig
    from transformers.activations import ACT2FN

    g = torch.Generator().manual_seed(1234)
    out = {}
    x = torch.randn(4, 1024, generator=g) * 2 + 0.3
    w, b = torch.rand(1024, generator=g) + 0.5, torch.randn(1024, generator=g) * 0.1
    out["ln_x"], out["ln_w"], out["ln_b"] = x.numpy(), w.numpy(), b.numpy()
    out["ln_y"] = torch.nn.functional.layer_norm(x, (1024,), w, b, 1e-5).numpy()
    x = torch.randn(4, 4096, generator=g) * 3
    w = torch.rand(4096, generator=g) + 0.5
    n = LlamaRMSNorm(4096, eps=1e-5)
    n.weight.data = w
    out["rms_x"], out["rms_w"], out["rms_y"] = x.numpy(), w.numpy(), n(x).detach().numpy()
    x = torch.randn(4, 512, generator=g) * 3
    out["act_x"] = x.numpy()
    out["quick_gelu_y"] = ACT2FN["quick_gelu"](x).numpy()
    out["gelu_y"] = torch.nn.GELU()(x).numpy()
    out["silu_y"] = torch.nn.functional.silu(x).numpy()
    cfg = LlamaConfig(hidden_size=4096, num_attention_heads=32, max_position_embeddings=4096)
    rot = LlamaRotaryEmbedding(config=cfg)
    pos = torch.tensor([[0, 1, 1215, 1343]])
    q = torch.randn(1, 2, 4, 128, generator=g)
    k = torch.randn(1, 2, 4, 128, generator=g)
    cos, sin = rot(q, pos)
    qe, ke = apply_rotary_pos_emb(q, k, cos, sin)
    out["rope_pos"], out["rope_q"], out["rope_k"] = pos.numpy(), q.numpy(), k.numpy()
    out["rope_qe"], out["rope_ke"] = qe.numpy(), ke.numpy()
    s = torch.randn(3, 577, generator=g) * 4
    out["softmax_x"], out["softmax_y"] = s.numpy(), torch.softmax(s, -1).numpy()
    np.savez_compressed(os.path.join(GOLD, "per_op.npz"), **out)
    # check the oracle's restatements right here
    assert np.allclose(cpu_ref.layer_norm(torch.from_numpy(out["ln_x"]), torch.from_numpy(out["ln_w"]),
                                          torch.from_numpy(out["ln_b"]), 1e-5).numpy(), out["ln_y"], atol=2e-6)
    assert np.allclose(cpu_ref.rms_norm(torch.from_numpy(out["rms_x"]), torch.from_numpy(out["rms_w"]), 1e-5).numpy(),
                       out["rms_y"], atol=2e-6)
    c, s_ = cpu_ref.rope_cos_sin(pos[0], 128, 10000.0)
    assert np.allclose(cpu_ref.apply_rope(q, c, s_).numpy(), out["rope_qe"], atol=2e-6)
    assert np.allclose(cpu_ref.quick_gelu(torch.from_numpy(out["act_x"])).numpy(), out["quick_gelu_y"], atol=2e-6)
    print("[per_op] oracle restatements match HF/torch building blocks")


def tokenizer_orders():
    """G6: placeholder id orders produced by the reference's own tokenizer helpers with a fake tokenizer."""
    ref_shim.load_reference()
    from vcoder_llava import mm_utils

    class Fake:
        bos_token_id = 1

        def __call__(self, text):
            class R:
                pass
            r = R()
            r.input_ids = [1] + [3 + (ord(c) % 50) for c in text]
            return r

    class NoBos(Fake):   # a tokenizer that prepends nothing: the reference's offset == 0 paths (mm_utils.py:50-54,73-82)
        bos_token_id = 1

        def __call__(self, text):
            r = Fake.__call__(self, text)
            r.input_ids = r.input_ids[1:]
            return r

    tk, nb = Fake(), NoBos()
    out = {
        "ds": mm_utils.tokenizer_depth_seg_token("ab <depth>\n<seg>\n<image>\ncd", tk),
        "seg": mm_utils.tokenizer_depth_seg_token("ab <seg>\n<image>\ncd", tk),
        "img": mm_utils.tokenizer_image_token("ab <image>\ncd", tk),
        "ds_nobos": mm_utils.tokenizer_depth_seg_token("ab <depth>\n<seg>\n<image>\ncd", nb),
        "seg_nobos": mm_utils.tokenizer_depth_seg_token("ab <seg>\n<image>\ncd", nb),
        "img_nobos": mm_utils.tokenizer_image_token("ab <image>\ncd", nb),
    }
    with open(os.path.join(GOLD, "tokenizer_orders.json"), "w") as f:
        json.dump({k: [int(t) for t in v] for k, v in out.items()}, f)
    print("[tokenizer]", {k: [t for t in v if t < 0] for k, v in out.items()})


def main_round2():
    """Fixtures added in round 2 (list-form images, the tower boundary, BOS-less tokenizer orders); the round-1 fixtures
    are left untouched (`python oracle/gen_golden.py --round2`)."""
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tokenizer_orders()
    I, S, D = synth.IMAGE_TOKEN_INDEX, synth.SEG_TOKEN_INDEX, synth.DEPTH_TOKEN_INDEX
    with tempfile.TemporaryDirectory() as tmp:
        cfg = vcfg.tiny("vcoder_ds")
        clip_dir = os.path.join(tmp, "clip")
        make_clip_dir(cfg, clip_dir)
        sd = synth.synth_state_dict(cfg, SEED)
        model = build_reference_model(cfg, sd, clip_dir)
        oracle = cpu_ref.OracleModel(cfg, sd)
        V = cfg.vocab_size
        p = lambda s, ph: np.concatenate([[1], synth.synth_prompt_ids(V, "llava", 5, 4, s)[1:6], ph,
                                          synth.synth_prompt_ids(V, "llava", 5, 4, s)[7:]]).astype(np.int64)
        tower_vectors(model, cfg, "tower_tiny")
        # two images per sample for every modality, hand order [IMG, SEG, DEPTH] so that depth blocks are spliced too
        run_list_case(model, oracle, cfg, "ds_list_two_each", [p(0, [I, S, D]), p(1, [I, S, D])],
                      [[2, 2], [2, 2], [2, 2]])
        # uneven image counts with equal totals per sample (equal spliced lengths): 2+1+1 and 1+2+1
        run_list_case(model, oracle, cfg, "ds_list_uneven", [p(2, [I, S, D]), p(3, [I, S, D])],
                      [[2, 1], [1, 2], [1, 1]])
    print("round-2 fixtures written to", GOLD)


@torch.no_grad()
def run_masked_case(model, oracle, cfg, name, ids_list, mask, n_new=6):
    """A PADDED batch through the live reference: prefill with a 2-D attention_mask that hides positions
    (prepare_inputs_labels_for_multimodal left-extends it by position, vcoder_ds_llava_arch.py:305-311; LlamaModel hides those
    keys), then cached greedy steps in the two forms a caller can produce:
      'ones'  the all-ones mask of the reference's multimodal decode path (vcoder_ds_llava_arch.py:130-133) — what
              model.generate() runs.  Under Transformers 5.x that line crashes on the DynamicCache, so the step is issued as
              SURVEY.md section 8(c) prescribes: forward(input_ids=[[t]], attention_mask=ones(B, L + 1), past_key_values=pkv)
              WITHOUT images (the early return then leaves exactly that mask in place);
      'keep'  the caller's own mask carried through: cat(extended prefill mask, ones) — forward() without images."""
    B = len(ids_list)
    imgs, segs, deps = (torch.from_numpy(a) for a in synth.synth_batch(B, cfg.vit_image_size))
    ids = torch.tensor(np.stack(ids_list), dtype=torch.long)
    am = torch.tensor(np.asarray(mask), dtype=torch.long)
    mask_ext = model.prepare_inputs_labels_for_multimodal(ids, am, None, None, imgs, segs, deps)[1]
    res = {}
    for variant in ("ones", "keep"):
        out = model(input_ids=ids, attention_mask=am, images=imgs, segs=segs, depths=deps, use_cache=True)
        full = out.logits.float().numpy()
        pkv = out.past_key_values
        L = full.shape[1]
        last = out.logits[:, -1].float()
        toks, lgs = [], []
        for step in range(n_new):
            lgs.append(last.numpy())
            nxt = last.argmax(-1)
            toks.append(nxt.numpy())
            if step + 1 == n_new:
                break
            step_mask = torch.ones(B, L + step + 1, dtype=torch.long)
            if variant == "keep":
                step_mask[:, :L] = mask_ext.long()
            o = model(input_ids=nxt[:, None], attention_mask=step_mask, past_key_values=pkv, use_cache=True)
            pkv = o.past_key_values
            last = o.logits[:, -1].float()
        res[variant] = (np.stack(toks, 1), np.stack(lgs, 1), full)
    assert np.array_equal(res["ones"][2], res["keep"][2])
    full = res["ones"][2]
    # ---- pin the oracle
    o_full, cache = oracle.forward(ids.tolist(), imgs, segs, deps, attention_mask=am.numpy())
    assert np.array_equal(oracle.mask_ext.numpy(), mask_ext.bool().numpy()), "mask extension differs from the reference"
    e_full = float(np.abs(o_full.numpy() - full).max())
    errs = {}
    for variant in ("ones", "keep"):
        o_full2, cache = oracle.forward(ids.tolist(), imgs, segs, deps, attention_mask=am.numpy(), last_only=True)
        last = o_full2[:, -1]
        e, same = 0.0, True
        for step in range(n_new):
            e = max(e, float(np.abs(last.numpy() - res[variant][1][:, step]).max()))
            nxt = last.argmax(-1)
            same = same and bool((nxt.numpy() == res[variant][0][:, step]).all())
            if step + 1 < n_new:
                last = oracle.decode_step(nxt.tolist(), cache, keep_mask=(variant == "keep"))[:, -1]
        errs[variant] = (e, same)
    unmasked = model(input_ids=ids, attention_mask=torch.ones_like(am), images=imgs, segs=segs, depths=deps).logits.float().numpy()
    print(f"[{name}] S={full.shape[1]} full|d|={e_full:.2e} steps ones {errs['ones']} keep {errs['keep']}; the mask moves the prefill "
          f"logits by {np.abs(unmasked - full).max():.3f}; ones-vs-keep step logits differ by "
          f"{np.abs(res['ones'][1] - res['keep'][1]).max():.3f}")
    assert e_full < 2e-4 and all(e < 2e-4 and same for e, same in errs.values()), "oracle/cpu_ref.py disagrees with the reference"
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), variant=cfg.variant, seed=SEED, input_ids=ids.numpy(),
                        attention_mask=am.numpy().astype(np.int64), mask_ext=mask_ext.bool().numpy(), spliced_len=full.shape[1],
                        prefill_logits=full.astype(np.float32), ids_ones=res["ones"][0].astype(np.int64),
                        step_logits_ones=res["ones"][1].astype(np.float32), ids_keep=res["keep"][0].astype(np.int64),
                        step_logits_keep=res["keep"][1].astype(np.float32))


def main_round3():
    """Fixtures added in round 3 (`python oracle/gen_golden.py --round3`): the other projector types of the plugin factories
    (multimodal_projector/builder.py:33-51) through the live reference — 'linear' for <image>, 'mlp3x_gelu' for <seg> / <depth>;
    and 'identity' (mm_hidden_size == hidden_size) for <image> with the usual mlp2x_gelu seg adapter."""
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    I, S, D = synth.IMAGE_TOKEN_INDEX, synth.SEG_TOKEN_INDEX, synth.DEPTH_TOKEN_INDEX
    cases = [("ds_proj_linear_mlp3x", dict(mm_projector_type="linear", seg_mm_projector_type="mlp3x_gelu",
                                           depth_mm_projector_type="mlp3x_gelu")),
             ("ds_proj_identity", dict(mm_projector_type="identity", mm_hidden_size=256, seg_mm_hidden_size=256,
                                       depth_mm_hidden_size=256, vit_num_heads=4))]
    for name, ov in cases:
        with tempfile.TemporaryDirectory() as tmp:
            cfg = vcfg.tiny("vcoder_ds")
            for k, v in ov.items():
                setattr(cfg, k, v)
            cfg.validate()
            clip_dir = os.path.join(tmp, "clip")
            make_clip_dir(cfg, clip_dir)
            sd = synth.synth_state_dict(cfg, SEED)
            model = build_reference_model(cfg, sd, clip_dir)
            oracle = cpu_ref.OracleModel(cfg, sd)
            V = cfg.vocab_size
            p = lambda s_, ph: np.concatenate([[1], synth.synth_prompt_ids(V, "llava", 5, 4, s_)[1:6], ph,
                                               synth.synth_prompt_ids(V, "llava", 5, 4, s_)[7:]]).astype(np.int64)
            run_case(model, oracle, cfg, name, [p(0, [I, S, D]), p(1, [I, S, D])], cfg_overrides=ov)
            del model
    # padded batch (attention_mask with hidden positions): row 0 unpadded, row 1 right-padded, row 2 LEFT-padded (the
    # reference's by-position left extension then hides three rows in the middle of the spliced sequence, not the pads)
    with tempfile.TemporaryDirectory() as tmp:
        cfg = vcfg.tiny("vcoder_ds")
        clip_dir = os.path.join(tmp, "clip")
        make_clip_dir(cfg, clip_dir)
        sd = synth.synth_state_dict(cfg, SEED)
        model = build_reference_model(cfg, sd, clip_dir)
        oracle = cpu_ref.OracleModel(cfg, sd)
        V = cfg.vocab_size
        p = lambda s_, ph: np.concatenate([[1], synth.synth_prompt_ids(V, "llava", 5, 4, s_)[1:6], ph,
                                           synth.synth_prompt_ids(V, "llava", 5, 4, s_)[7:]]).astype(np.int64)
        r0, r1, r2 = p(0, [I, D, S]), p(1, [I, D, S]), p(2, [I, D, S])
        T = len(r0)
        rows = [r0, np.concatenate([r1[:-3], [0, 0, 0]]), np.concatenate([[0, 0, 0], r2[:-3]])]
        mask = np.ones((3, T), dtype=np.int64)
        mask[1, -3:] = 0
        mask[2, :3] = 0
        run_masked_case(model, oracle, cfg, "ds_padded_mask", rows, mask)
        # output_hidden_states of the prefill forward (vcoder_ds_llava_llama.py:81-90,117): LlamaModel's tuple of L + 1 tensors
        ids = torch.tensor(np.stack([r0, r1]), dtype=torch.long)
        imgs, segs, deps = (torch.from_numpy(a) for a in synth.synth_batch(2, cfg.vit_image_size))
        with torch.no_grad():
            out = model(input_ids=ids, images=imgs, segs=segs, depths=deps, output_hidden_states=True, output_attentions=True,
                        use_cache=False)
        hs = torch.stack([h.float() for h in out.hidden_states], 0).numpy()           # [L + 1, B, S, D]
        assert hs.shape[0] == cfg.num_hidden_layers + 1
        ho, ao = [], []
        oracle.forward(ids.tolist(), imgs, segs, deps, hidden_out=ho, attn_out=ao)
        e = float(np.abs(torch.stack(ho, 0).numpy() - hs).max())
        print(f"[ds_hidden_states] {hs.shape} oracle|d|={e:.2e} |h|max={np.abs(hs).max():.2f}")
        assert e < 2e-5 * max(1.0, float(np.abs(hs).max()))
        at = torch.stack([a.float() for a in out.attentions], 0).numpy()               # [L, B, H, S, S] (eager attention)
        assert at.shape == (cfg.num_hidden_layers, 2, cfg.num_attention_heads, hs.shape[2], hs.shape[2])
        e_at = float(np.abs(torch.stack(ao, 0).numpy() - at).max())
        print(f"[ds_hidden_states] attentions {at.shape} oracle|d|={e_at:.2e}")
        assert e_at < 1e-6
        # ... and of a CACHED decode step behind it (form B of SURVEY.md section 8(c): no images, a mask of ones over past + 1).
        # On a FRESH reference model: under Transformers 5.15 a forward with output_attentions=True leaves the model in a state
        # in which later use_cache=True forwards differ (by 6e-2 here) from the same call on a fresh model and from its own
        # use_cache=False forward — third-party state, not the reference's arithmetic.
        model2 = build_reference_model(cfg, sd, clip_dir)
        with torch.no_grad():
            pre = model2(input_ids=ids, images=imgs, segs=segs, depths=deps, use_cache=True)
            assert float((pre.logits - out.logits).abs().max()) < 1e-5, "cached prefill differs from the no-cache forward"
            nxt = pre.logits[:, -1].float().argmax(-1)
            L_ = pre.logits.shape[1]
            st = model2(input_ids=nxt[:, None], attention_mask=torch.ones(2, L_ + 1, dtype=torch.long),
                        past_key_values=pre.past_key_values, use_cache=True, output_hidden_states=True, output_attentions=True)
        step_hs = torch.stack([h.float() for h in st.hidden_states], 0).numpy()        # [L + 1, B, 1, D]
        step_at = torch.stack([a.float() for a in st.attentions], 0).numpy()           # [L, B, H, 1, S + 1]
        assert step_hs.shape == (cfg.num_hidden_layers + 1, 2, 1, cfg.hidden_size)
        assert step_at.shape == (cfg.num_hidden_layers, 2, cfg.num_attention_heads, 1, L_ + 1)
        _, cache = oracle.forward(ids.tolist(), imgs, segs, deps, last_only=True)
        sh, sa = [], []
        o_step = oracle.decode_step(nxt.tolist(), cache, hidden_out=sh, attn_out=sa)
        e_sh = float(np.abs(torch.stack(sh, 0).numpy() - step_hs).max())
        e_sa = float(np.abs(torch.stack(sa, 0).numpy() - step_at).max())
        e_sl = float(np.abs(o_step[:, -1].numpy() - st.logits[:, -1].float().numpy()).max())
        print(f"[ds_hidden_states] cached step: hidden oracle|d|={e_sh:.2e} attentions oracle|d|={e_sa:.2e} logits|d|={e_sl:.2e}")
        assert e_sh < 2e-5 * max(1.0, float(np.abs(step_hs).max())) and e_sa < 1e-6 and e_sl < 2e-4
        np.savez_compressed(os.path.join(GOLD, "ds_hidden_states.npz"), variant=cfg.variant, seed=SEED, input_ids=ids.numpy(),
                            hidden_sample=hs[:, :, ::3, ::8].astype(np.float32), hidden_rowsum=hs.sum(-1).astype(np.float32),
                            logits_last=out.logits[:, -1].float().numpy(), attentions=at.astype(np.float32),
                            step_token=nxt.numpy().astype(np.int64), step_hidden=step_hs.astype(np.float32),
                            step_attentions=step_at.astype(np.float32), step_logits=st.logits[:, -1].float().numpy())
    print("round-3 fixtures written to", GOLD)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    per_op_vectors()
    tokenizer_orders()
    I, S, D = synth.IMAGE_TOKEN_INDEX, synth.SEG_TOKEN_INDEX, synth.DEPTH_TOKEN_INDEX
    with tempfile.TemporaryDirectory() as tmp:
        # ---------------- DS
        cfg = vcfg.tiny("vcoder_ds")
        clip_dir = os.path.join(tmp, "clip")
        make_clip_dir(cfg, clip_dir)
        sd = synth.synth_state_dict(cfg, SEED)
        model = build_reference_model(cfg, sd, clip_dir)
        oracle = cpu_ref.OracleModel(cfg, sd)
        V = cfg.vocab_size
        p = lambda s, ph: np.concatenate([[1], synth.synth_prompt_ids(V, "llava", 5, 4, s)[1:6], ph,
                                          synth.synth_prompt_ids(V, "llava", 5, 4, s)[7:]]).astype(np.int64)
        run_case(model, oracle, cfg, "ds_img_depth_seg", [p(0, [I, D, S]), p(1, [I, D, S])])
        run_case(model, oracle, cfg, "ds_img_seg_depth", [p(0, [I, S, D]), p(1, [I, S, D])])
        run_case(model, oracle, cfg, "ds_img_seg", [p(2, [I, S])], use_depth=False)
        run_case(model, oracle, cfg, "ds_img_only", [p(3, [I])], use_seg=False, use_depth=False)
        run_case(model, oracle, cfg, "ds_zero_depth", [p(4, [I, D, S])], zero_depth=True)
        run_case(model, oracle, cfg, "ds_img_text_seg", [p(5, [I, 17, 18, S])])
        # quirk assertions (G5): dead weights / depth pixels do not change logits
        ids = torch.tensor(np.stack([p(0, [I, D, S])]))
        imgs, segs, deps = (torch.from_numpy(a) for a in synth.synth_batch(1, cfg.vit_image_size))
        with torch.no_grad():
            base = model(input_ids=ids, images=imgs, segs=segs, depths=deps).logits
            alt = model(input_ids=ids, images=imgs, segs=segs, depths=deps * 0.5 + 1.0).logits
            assert torch.equal(base, alt), "depth pixels changed logits with reference token order"
            for dead in ("depth_mm_projector", "mm2_projector"):
                for prm in getattr(model.model, dead).parameters():
                    prm.add_(1.0)
            model.model.vcoder_lm_emb.weight.add_(1.0)
            alt = model(input_ids=ids, images=imgs, segs=segs, depths=deps).logits
            assert torch.equal(base, alt), "dead tensors changed logits"
        print("[quirks] depth pixels / depth_mm_projector / mm2_projector / vcoder_lm_emb are dead: verified")
        # batched rows == single-sample rows
        with torch.no_grad():
            ids2 = torch.tensor(np.stack([p(0, [I, D, S]), p(1, [I, D, S])]))
            i2, s2, d2 = (torch.from_numpy(a) for a in synth.synth_batch(2, cfg.vit_image_size))
            two = model(input_ids=ids2, images=i2, segs=s2, depths=d2).logits
            one = model(input_ids=ids2[1:], images=i2[1:], segs=s2[1:], depths=d2[1:]).logits
            print("[quirks] batched-vs-single max|d| =", float((two[1:] - one).abs().max()))
        del model
        # ---------------- non-DS
        cfg = vcfg.tiny("vcoder")
        sd = synth.synth_state_dict(cfg, SEED)
        model = build_reference_model(cfg, sd, clip_dir)
        oracle = cpu_ref.OracleModel(cfg, sd)
        run_case(model, oracle, cfg, "vc_img_seg", [p(0, [I, S]), p(1, [I, S])])
        run_case(model, oracle, cfg, "vc_img_text_seg", [p(2, [I, 11, 12, S])])
        try:
            with torch.no_grad():
                model(input_ids=torch.tensor(np.stack([p(3, [I])])),
                      images=torch.from_numpy(synth.synth_batch(1, cfg.vit_image_size)[0]),
                      segs=torch.from_numpy(synth.synth_batch(1, cfg.vit_image_size)[1]))
            raise AssertionError("expected IndexError (quirk 5)")
        except IndexError:
            print("[quirks] non-DS image-only prompt raises IndexError: verified")
        del model
        # ---------------- llava
        cfg = vcfg.tiny("llava")
        sd = synth.synth_state_dict(cfg, SEED)
        model = build_reference_model(cfg, sd, clip_dir)
        oracle = cpu_ref.OracleModel(cfg, sd)
        run_case(model, oracle, cfg, "llava_img", [p(0, [I]), p(1, [I])], use_seg=False, use_depth=False)
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    if "--round2" in sys.argv:
        main_round2()
    elif "--round3" in sys.argv:
        main_round3()
    else:
        main()
        main_round2()
        main_round3()
