"""Deterministic synthetic checkpoints and COST-shaped inputs.

There is no network in the build/bench environment, so every parity test and the benchmark run on
seeded synthetic weights with the real architecture (SURVEY.md §8(d) "Value distributions / seeds").
The generator is a pure 32-bit integer hash (murmur3 finaliser) followed by ONE fp32 multiply-add
free affine map and a round-to-nearest-even bf16 truncation, so that

  * numpy here,
  * the device-side generator `vc_model_synth_tensor` (csrc/misc_kernels.hip, same integer ops), and
  * any other box

produce bit-identical tensors: nothing depends on a library RNG stream.

Every tensor value is bf16-representable (the "checkpoint dtype"), so the fp32 oracle and the bf16
HIP path see *identical* weights.

Tensor names are the HF state-dict keys the reference's checkpoints carry.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Tuple

import numpy as np

GOLDEN = np.uint32(0x9E3779B1)
UNIFORM_STD_TO_HALFWIDTH = math.sqrt(3.0)


def fnv1a32(name: str) -> int:
    h = 0x811C9DC5
    for ch in name.encode("utf-8"):
        h ^= ch
        h = (h * 0x01000193) & 0xFFFFFFFF
    return h


def tensor_seed(name: str, seed: int) -> int:
    """32-bit per-tensor seed: FNV-1a(name) mixed with the global seed."""
    x = (fnv1a32(name) ^ ((seed * 0x9E3779B1) & 0xFFFFFFFF)) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 13
    return x


def hash_u24(idx: np.ndarray, tseed: int) -> np.ndarray:
    """murmur3-finaliser of (idx * GOLDEN + tseed); returns the top 24 bits as uint32."""
    with np.errstate(over="ignore"):
        x = idx.astype(np.uint32) * GOLDEN + np.uint32(tseed)
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x85EBCA6B)
        x ^= x >> np.uint32(13)
        x *= np.uint32(0xC2B2AE35)
        x ^= x >> np.uint32(16)
    return x >> np.uint32(8)


def round_to_bf16(x: np.ndarray) -> np.ndarray:
    """fp32 -> nearest-even bf16, returned as fp32 (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 (already bf16-representable or not) -> uint16 bf16 bit patterns, RNE."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    return r.astype(np.uint16)


def from_bf16_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int, offset: float, halfwidth: float, rounding: str = "bf16") -> np.ndarray:
    """value[i] = bf16( offset + (u24[i] - 8388607.5) * (halfwidth / 8388608) ), fp32 array.

    (u24 - 8388607.5) is exact in fp32; the product and the sum are one IEEE fp32 op each, computed
    without contraction both here and in the device kernel (which uses __fmul_rn / __fadd_rn).

    rounding: "bf16" (default: what the device-side generator produces), "fp16" (the value an fp16 checkpoint holds: 11
    significant bits, NOT bf16-representable in general) or "fp32" (unrounded) — see synth_state_dict(dtypes="reference").
    """
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint32)
    u = hash_u24(idx, tensor_seed(name, seed)).astype(np.float32)
    v = (u - np.float32(8388607.5)) * np.float32(halfwidth / 8388608.0)
    if offset != 0.0:
        v = v + np.float32(offset)
    if rounding == "fp32":
        return v.reshape(shape)
    if rounding == "fp16":
        return v.astype(np.float16).astype(np.float32).reshape(shape)
    return round_to_bf16(v).reshape(shape)


# ---------------------------------------------------------------------------------------------
# architecture description -> list of (key, shape, offset, halfwidth)
# ---------------------------------------------------------------------------------------------

W_STD = 0.02


def _hw(std: float) -> float:
    return std * UNIFORM_STD_TO_HALFWIDTH


def tensor_specs(cfg) -> Iterable[Tuple[str, Tuple[int, ...], float, float]]:
    """Yield (hf_key, shape, offset, halfwidth) for every tensor of a VCoder checkpoint.

    `cfg` is a vcoder_amd.config.VCoderConfig (or anything with the same attributes).  Dead tensors
    (depth_mm_projector, mm2_projector, vcoder_lm_emb) are generated too so
    that state-dict round trips and the "dead weights do not change logits" tests have data.
    """
    D, F, V, L = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.num_hidden_layers
    w = _hw(W_STD)
    yield "model.embed_tokens.weight", (V, D), 0.0, w
    yield "lm_head.weight", (V, D), 0.0, w
    yield "model.norm.weight", (D,), 1.0, 0.1
    for i in range(L):
        p = f"model.layers.{i}."
        yield p + "input_layernorm.weight", (D,), 1.0, 0.1
        yield p + "post_attention_layernorm.weight", (D,), 1.0, 0.1
        for nm in ("q", "k", "v", "o"):
            yield p + f"self_attn.{nm}_proj.weight", (D, D), 0.0, w
        yield p + "mlp.gate_proj.weight", (F, D), 0.0, w
        yield p + "mlp.up_proj.weight", (F, D), 0.0, w
        yield p + "mlp.down_proj.weight", (D, F), 0.0, w
    Dv = cfg.mm_hidden_size
    projs = []
    if cfg.variant in ("llava", "vcoder", "vcoder_ds"):
        projs.append(("model.mm_projector", cfg.mm_projector_type))
    if cfg.variant in ("vcoder", "vcoder_ds"):
        projs.append(("model.seg_mm_projector", cfg.seg_mm_projector_type))
        if cfg.use_mm2_proj:
            projs.append(("model.mm2_projector", cfg.mm_projector_type))
        if cfg.mm_vcoder_lm_emb:
            yield "model.vcoder_lm_emb.weight", (V, D), 0.0, w
    if cfg.variant == "vcoder_ds":
        projs.append(("model.depth_mm_projector", cfg.depth_mm_projector_type))
    for prefix, ptype in projs:
        depth = projector_depth(ptype)
        if depth == 0:
            continue
        yield f"{prefix}.0.weight" if depth > 1 else f"{prefix}.weight", (D, Dv), 0.0, w
        yield f"{prefix}.0.bias" if depth > 1 else f"{prefix}.bias", (D,), 0.0, w
        for j in range(1, depth):
            yield f"{prefix}.{2 * j}.weight", (D, D), 0.0, w
            yield f"{prefix}.{2 * j}.bias", (D,), 0.0, w
    # CLIP vision tower (Transformers-5.x key names)
    vt = "model.vision_tower.vision_tower.vision_model."
    Fv, Lv, P = cfg.vit_intermediate_size, cfg.vit_num_layers, cfg.vit_patch_size
    T = (cfg.vit_image_size // P) ** 2 + 1
    yield vt + "embeddings.class_embedding", (Dv,), 0.0, w
    yield vt + "embeddings.patch_embedding.weight", (Dv, 3, P, P), 0.0, w
    yield vt + "embeddings.position_embedding.weight", (T, Dv), 0.0, w
    yield vt + "pre_layrnorm.weight", (Dv,), 1.0, 0.1
    yield vt + "pre_layrnorm.bias", (Dv,), 0.0, w
    for j in range(Lv):
        p = vt + f"encoder.layers.{j}."
        for ln in ("layer_norm1", "layer_norm2"):
            yield p + ln + ".weight", (Dv,), 1.0, 0.1
            yield p + ln + ".bias", (Dv,), 0.0, w
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            yield p + f"self_attn.{nm}.weight", (Dv, Dv), 0.0, w
            yield p + f"self_attn.{nm}.bias", (Dv,), 0.0, w
        yield p + "mlp.fc1.weight", (Fv, Dv), 0.0, w
        yield p + "mlp.fc1.bias", (Fv,), 0.0, w
        yield p + "mlp.fc2.weight", (Dv, Fv), 0.0, w
        yield p + "mlp.fc2.bias", (Dv,), 0.0, w
    yield vt + "post_layernorm.weight", (Dv,), 1.0, 0.1
    yield vt + "post_layernorm.bias", (Dv,), 0.0, w


def projector_depth(ptype: str) -> int:
    """'linear' -> 1, 'mlpNx_gelu' -> N, 'identity' -> 0 (multimodal_projector/builder.py:33-51)."""
    import re

    if ptype == "linear":
        return 1
    if ptype == "identity":
        return 0
    m = re.match(r"^mlp(\d+)x_gelu$", ptype)
    if m:
        return int(m.group(1))
    raise ValueError(f"Unknown projector type: {ptype}")


ROUNDING_CODE = {"bf16": 0, "fp16": 1, "fp32": 2}   # vc_model_synth_tensor_rounded / vck_synth_f32_rounded


REFERENCE_CLASSES = ("reference", "reference_loaded")


def reference_rounding(key: str, dtypes: str = "reference") -> str:
    """the value class of a tensor in the reference's own checkpoints.  "reference": as the files hold them — the CLIP tower fp32 (hub
    checkpoint), everything else fp16.  "reference_loaded": as the reference COMPUTES with them — model/builder.py:142 casts the
    loaded tower to fp16 (`vision_tower.to(device=device, dtype=torch.float16)`), so every tensor is fp16-valued (and a bf16 hi + lo
    pair holds each of them exactly)."""
    if dtypes == "reference_loaded":
        return "fp16"
    return "fp32" if "vision_tower" in key else "fp16"


def synth_state_dict(cfg, seed: int = 42, only_prefix: str | None = None, dtypes: str = "bf16") -> Dict[str, np.ndarray]:
    """dtypes="bf16": every value bf16-representable (the default checkpoint of the tests and the benchmark).
    dtypes="reference": the value classes of the reference's own checkpoints — the LLM, its head, embeddings and the projectors
    fp16-valued (model/builder.py:25-40 loads them with torch_dtype=float16), the CLIP tower fp32-valued (the hub checkpoint,
    multimodal_encoder/clip_encoder.py:22-27) — which bf16 cannot hold: vc_model_inexact_tensors() > 0, and the strict / split
    precision modes must add the lo planes back to stay within 1e-3 of an fp32 oracle run on THESE values.
    dtypes="reference_loaded": the values the reference computes with — the tower fp16-valued too (model/builder.py:142)."""
    assert dtypes in ("bf16",) + REFERENCE_CLASSES, dtypes
    out = {}
    for key, shape, off, hw in tensor_specs(cfg):
        if only_prefix is not None and not key.startswith(only_prefix):
            continue
        rounding = reference_rounding(key, dtypes) if dtypes in REFERENCE_CLASSES else "bf16"
        out[key] = synth_tensor(key, shape, seed, off, hw, rounding)
    return out


# ---------------------------------------------------------------------------------------------
# COST-shaped inputs: RGB noise, panoptic-like blocks, smooth depth gradient
# ---------------------------------------------------------------------------------------------

CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)


def _u8_hash(n: int, tseed: int) -> np.ndarray:
    return (hash_u24(np.arange(n, dtype=np.uint32), tseed) >> np.uint32(16)).astype(np.uint8)


def _normalize(u8_hwc: np.ndarray) -> np.ndarray:
    x = u8_hwc.astype(np.float32) * np.float32(1.0 / 255.0)
    x = (x - CLIP_MEAN) / CLIP_STD
    return np.ascontiguousarray(x.transpose(2, 0, 1))  # CHW fp32


def synth_rgb(i: int, size: int) -> np.ndarray:
    """uniform uint8 noise, 2x2 box low-pass, CLIP-normalised, [3,size,size] fp32."""
    raw = _u8_hash(3 * (size + 1) * (size + 1), tensor_seed("rgb", 1000 + i)).reshape(size + 1, size + 1, 3)
    r = raw.astype(np.uint16)
    lp = ((r[:-1, :-1] + r[1:, :-1] + r[:-1, 1:] + r[1:, 1:]) // 4).astype(np.uint8)
    return _normalize(lp)


def synth_seg(i: int, size: int, block: int = 28) -> np.ndarray:
    """piecewise-constant random colour blocks (panoptic-map-like)."""
    nb = (size + block - 1) // block
    cols = _u8_hash(3 * nb * nb, tensor_seed("seg", 2000 + i)).reshape(nb, nb, 3)
    img = np.repeat(np.repeat(cols, block, axis=0), block, axis=1)[:size, :size]
    return _normalize(img)


def synth_depth(i: int, size: int) -> np.ndarray:
    """smooth monotone grey gradient replicated to 3 channels (non-zero mean)."""
    a = int(_u8_hash(2, tensor_seed("depth", 3000 + i))[0]) % 64
    yy, xx = np.meshgrid(np.arange(size, dtype=np.int64), np.arange(size, dtype=np.int64), indexing="ij")
    g = (32 + a + (yy * 3 + xx * 2) * 150 // (5 * size)).astype(np.uint8)
    img = np.stack([g, g, g], axis=-1)
    return _normalize(img)


def synth_batch(B: int, size: int, first: int = 0):
    """(images, segs, depths) each [B,3,size,size] fp32 for global sample indices first..first+B-1."""
    imgs = np.stack([synth_rgb(first + b, size) for b in range(B)])
    segs = np.stack([synth_seg(first + b, size) for b in range(B)])
    deps = np.stack([synth_depth(first + b, size) for b in range(B)])
    return imgs, segs, deps


IMAGE_TOKEN_INDEX = -200
SEG_TOKEN_INDEX = -300
DEPTH_TOKEN_INDEX = -400


def synth_prompt_ids(vocab: int, variant: str, n_before: int = 34, n_after: int = 29, sample: int = 0,
                     seed: int = 7) -> np.ndarray:
    """ids = [bos=1] + n_before text + placeholders + n_after text.

    Placeholder order is the one the reference's own tokenizer helpers emit
    (mm_utils.py:43-117): DS -> [IMG, DEPTH, SEG]; non-DS -> [IMG, SEG]; llava -> [IMG].
    """
    n = n_before + n_after
    h = hash_u24(np.arange(n, dtype=np.uint32), tensor_seed("ids%d" % sample, seed))
    txt = (3 + (h.astype(np.int64) % (vocab - 3))).astype(np.int64)
    ph = {"vcoder_ds": [IMAGE_TOKEN_INDEX, DEPTH_TOKEN_INDEX, SEG_TOKEN_INDEX],
          "vcoder": [IMAGE_TOKEN_INDEX, SEG_TOKEN_INDEX],
          "llava": [IMAGE_TOKEN_INDEX]}[variant]
    return np.concatenate([[1], txt[:n_before], ph, txt[n_before:]]).astype(np.int64)
