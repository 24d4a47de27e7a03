"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (plain torch fp32 ops, no HF Transformers import, no reference import) of the
algorithm on the VCoder inference hot path.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this module; the product (`vcoder_amd/`) never does
and fails loudly when its HIP library is missing.

PINNING: this restatement is checked against the live reference (imported from /root/reference
through oracle/ref_shim.py) by `oracle/gen_golden.py`, which also writes the committed fixtures in
`tests/golden/`; `tests/test_oracle_golden.py` re-checks it against those fixtures on every run
(and against the live reference when it is present).  The reference itself has no tests or golden
vectors, so the fixtures generated from the reference's own forward are the pin.

Each function cites the reference / third-party file:line it restates.  `[HF]` =
transformers/models (the reference's arithmetic lives in HF `CLIPVisionModel` and `LlamaModel`,
which are not vendored in /root/reference; reference pin transformers==4.31.0, pyproject.toml:23).

`emu_bf16=True` rounds activations to bf16 at exactly the points where the HIP path stores bf16
(DESIGN.md "rounding points") so that the bf16 kernels can be checked to a tight tolerance; with
`emu_bf16=False` this is the reference's fp32 CPU path.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

IGNORE_INDEX = -100          # vcoder_llava/constants.py:4
IMAGE_TOKEN_INDEX = -200     # constants.py:5
SEG_TOKEN_INDEX = -300       # constants.py:8
DEPTH_TOKEN_INDEX = -400     # constants.py:11


def _bf16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def quantize_rows_e4m3(x: torch.Tensor) -> torch.Tensor:
    """The device's W8A8 activation format (vc_model_set_weight_format(m, 2), csrc/decode.hip quant_act_rows_kernel):
    every row (last dim) gets the power-of-two scale s = 2^e with the smallest e such that max|row| <= 448 * 2^e, is
    rounded to OCP e4m3 (RNE) as row / s, and enters the matmul as q * s.  Returns the effective fp32 values."""
    amax = x.abs().amax(-1, keepdim=True)
    m, ex = torch.frexp(amax)                       # amax = m * 2^ex, m in [0.5, 1);  448 = 0.875 * 2^9
    e = torch.where(m <= 0.875, ex - 9, ex - 8)
    e = torch.where(amax > 0, e, torch.zeros_like(e))
    s = torch.ldexp(torch.ones_like(amax), e)
    return (x / s).to(torch.float8_e4m3fn).to(torch.float32) * s


def quantize_e4m3(x: torch.Tensor) -> torch.Tensor:
    """OCP e4m3fn without a scale, saturating at +-448 (the device's f2fp8): the KV-cache element of the fp8 weight format"""
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


class Rounder:
    """Identity in fp32 mode, bf16 round-trip in emulation mode.  act_fp8: additionally the device's `fp8` weight format — the
    inputs of the decoder linears are quantised per token row (`q8`) while the pass is a prefill (W8A8), and the KV cache the
    cached decode steps read holds e4m3 values (llama_layer)."""

    def __init__(self, emu_bf16, act_fp8: bool = False):
        # emu_bf16: False = fp32; True / "bf16" = the device's bf16 rounding points; "fp16" = the same points in IEEE fp16, saturating
        # at 65504 (libvcoder_hip_f16.so, the -DVC_F16 build of the kernels: vcoder_amd/csrc/vc_device.h)
        self.emu = emu_bf16
        self.act_fp8 = act_fp8
        self.prefill = False

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if self.emu == "fp16":
            return x.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)
        return _bf16(x) if self.emu else x

    def q8(self, x: torch.Tensor) -> torch.Tensor:
        return quantize_rows_e4m3(x) if (self.act_fp8 and self.prefill) else x


def usable_cpus() -> int:
    """CPUs this process can actually keep busy: its affinity mask capped by the cgroup CPU quota (cgroup v2 `cpu.max`, v1
    `cpu.cfs_quota_us / cpu.cfs_period_us`).  torch sizes its pool from the machine (128 threads on the 2 x 64-core MI355X
    hosts), but the GPU boxes run under a 16-CPU quota: 128 threads there are throttled to HALF the fp32 matmul rate of 16
    (profiles/r04_o_host_matmul.txt: 0.84 vs 1.61 TFLOP/s)."""
    import os

    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, math.ceil(quota)))
    return n


def fit_threads() -> int:
    """torch's intra-op pool never wider than usable_cpus(); -> the thread count now in effect (what a timing reports as `cores`)"""
    n = min(torch.get_num_threads(), usable_cpus())
    torch.set_num_threads(n)
    return n


def as_torch_state(sd: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in sd.items()}


# =============================================================================================
# CLIP vision tower  (a2-a4 of SURVEY.md §8)
# =============================================================================================

def _vt_prefix(sd) -> str:
    for p in ("model.vision_tower.vision_tower.vision_model.", "model.vision_tower.vision_tower.",
              "vision_model.", ""):
        if p + "embeddings.class_embedding" in sd:
            return p
    raise KeyError("no CLIP vision tower weights in state dict")


def layer_norm(x, w, b, eps):
    """nn.LayerNorm: biased variance over the last dim ([HF] clip/modeling_clip.py:370,379,642)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def quick_gelu(x):
    """x * sigmoid(1.702 x)  ([HF] activations.py QuickGELUActivation; CLIPMLP :346-350)."""
    return x * torch.sigmoid(1.702 * x)


def softmax_attention(q, k, v, scale, causal, r: Rounder, q_pos0: int = 0, key_mask=None, probs_out: Optional[list] = None):
    """q [N,H,Tq,hd], k/v [N,H,Tk,hd].  softmax in fp32 ([HF] clip :272, llama eager_attention :191-214).
    key_mask [N,Tk] bool (padded batches): a False key is hidden from every query of its sequence — the additive form of the
    2-D attention_mask HF's LlamaModel combines with the causal mask.

    Emulation: P = exp(s - rowmax) is rounded to bf16 before P.V, the row sum uses the unrounded
    fp32 p, and the normalised output is rounded to bf16 (what the flash kernel does)."""
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if causal:
        Tq, Tk = q.shape[-2], k.shape[-2]
        qi = torch.arange(Tq).unsqueeze(1) + q_pos0
        ki = torch.arange(Tk).unsqueeze(0)
        s = s.masked_fill(ki > qi, float("-inf"))
    if key_mask is not None:
        s = s.masked_fill(~key_mask[:, None, None, : k.shape[-2]].bool(), float("-inf"))
    m = s.max(-1, keepdim=True).values
    p = torch.exp(s - m)
    l = p.sum(-1, keepdim=True)
    if probs_out is not None:   # attn_weights of the eager attention ([HF] llama eager_attention_forward: softmax in fp32)
        probs_out.append(p / l)
    o = torch.matmul(r(p), v) / l
    return r(o)


def vit_embed(pixels, sd, vp, cfg):
    """CLIPVisionEmbeddings.forward ([HF] clip/modeling_clip.py:202-218): conv(k=s=P, no bias) as an
    im2col GEMM, prepend CLS, add learned position embeddings."""
    N = pixels.shape[0]
    P = cfg.vit_patch_size
    g = cfg.vit_image_size // P
    Wp = sd[vp + "embeddings.patch_embedding.weight"]  # [Dv,3,P,P]
    Dv = Wp.shape[0]
    # im2col: rows = (n, gy, gx), cols = (c, py, px)  == conv2d weight flatten order
    cols = pixels.reshape(N, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(N * g * g, 3 * P * P)
    return cols, Wp.reshape(Dv, 3 * P * P)


def vit_forward(pixels: torch.Tensor, sd: Dict[str, torch.Tensor], cfg, emu_bf16: bool = False,
                return_all: bool = False):
    """CLIPVisionTower.forward + feature_select (multimodal_encoder/clip_encoder.py:29-51):
    hidden_states[select_layer], CLS dropped for 'patch'.  Only `vit_layers_used` encoder layers are
    evaluated (hidden_states[-2] never needs the last layer / post_layernorm)."""
    r = Rounder(emu_bf16)
    vp = _vt_prefix(sd)
    eps = cfg.vit_layer_norm_eps
    N = pixels.shape[0]
    H = cfg.vit_num_heads
    cols, Wp = vit_embed(pixels.float(), sd, vp, cfg)
    patches = torch.matmul(r(cols), Wp.t())  # fp32 accumulate, fp32 out
    Dv = Wp.shape[0]
    g2 = patches.shape[0] // N
    cls = sd[vp + "embeddings.class_embedding"].reshape(1, 1, Dv).expand(N, 1, Dv)
    x = torch.cat([cls, patches.reshape(N, g2, Dv)], dim=1) + sd[vp + "embeddings.position_embedding.weight"].unsqueeze(0)
    x = layer_norm(x, sd[vp + "pre_layrnorm.weight"], sd[vp + "pre_layrnorm.bias"], eps)  # hidden_states[0]
    T = x.shape[1]
    hd = Dv // H
    hs = [x]
    for j in range(cfg.vit_layers_used):
        p = vp + f"encoder.layers.{j}."
        h = r(layer_norm(x, sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], eps))
        q = r(F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]))
        k = r(F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"]))
        v = r(F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"]))
        q, k, v = (t.reshape(N, T, H, hd).transpose(1, 2) for t in (q, k, v))
        a = softmax_attention(q, k, v, hd ** -0.5, False, r)  # [HF] clip :259-277,320-330
        a = a.transpose(1, 2).reshape(N, T, Dv)
        x = x + F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        h = r(layer_norm(x, sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], eps))
        h = r(quick_gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])))
        x = x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        hs.append(x)
    feats = x[:, 1:] if cfg.mm_vision_select_feature == "patch" else x
    feats = r(feats)  # `.to(images.dtype)` cast at clip_encoder.py:49
    return (feats, hs) if return_all else feats


# =============================================================================================
# adapters (a5)  multimodal_projector/builder.py:33-51 (+ adapter / depth_adapter twins)
# =============================================================================================

def projector_depth(ptype: str) -> int:
    if ptype == "linear":
        return 1
    if ptype == "identity":
        return 0
    m = re.match(r"^mlp(\d+)x_gelu$", ptype)
    if m:
        return int(m.group(1))
    raise ValueError(f"Unknown projector type: {ptype}")


def projector_forward(x, sd, prefix: str, ptype: str, emu_bf16: bool = False):
    """Linear | Sequential(Linear, [GELU(erf), Linear]*(N-1)) | identity."""
    r = Rounder(emu_bf16)
    depth = projector_depth(ptype)
    if depth == 0:
        return x
    if depth == 1:
        return r(F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"]))
    h = F.linear(x, sd[prefix + ".0.weight"], sd[prefix + ".0.bias"])
    for j in range(1, depth):
        h = r(F.gelu(h))  # nn.GELU() default = exact erf form
        h = F.linear(h, sd[f"{prefix}.{2 * j}.weight"], sd[f"{prefix}.{2 * j}.bias"])
    return r(h)


# =============================================================================================
# splice plan (a7)  — host logic; mirrors the reference loops statement by statement
# =============================================================================================

@dataclass
class Segment:
    kind: str          # 'text' | 'img' | 'seg' | 'depth'
    ids: Optional[List[int]] = None   # for text
    index: int = -1                   # feature batch index for img/seg/depth


def _text(ids: Sequence[int]) -> Segment:
    ids = [int(t) for t in ids]
    for t in ids:
        if t < 0:
            # nn.Embedding on a negative id: the reference dies with IndexError
            raise IndexError(f"index out of range in self (placeholder id {t} reached the embedding lookup)")
    return Segment("text", ids=ids)


def splice_plan(input_ids: Sequence[Sequence[int]], variant: str, has_seg: bool, depth_is_zero: Optional[Sequence[bool]],
                n_img_feats: int, n_seg_feats: int = 0, n_depth_feats: int = 0) -> List[List[Segment]]:
    """prepare_inputs_labels_for_multimodal, per-sample loop only.

    DS:     vcoder_ds_llava_arch.py:175-276     (text between <image> and <seg> is DROPPED, :233-244)
    non-DS: vcoder_llava_arch.py:181-260        (text before <seg> is kept, :236; `or` guard :187)
    llava:  llava_arch.py:117-160
    `depth_is_zero` is None when no depth images were passed (-> [True]*B, :171).
    """
    B = len(input_ids)
    if variant == "vcoder_ds" and depth_is_zero is None:
        depth_is_zero = [True] * B
    plans: List[List[Segment]] = []
    cur_image_idx = cur_seg_idx = cur_depth_idx = 0

    def feat(kind, idx, n):
        if idx >= n:
            raise IndexError(f"{kind} feature index {idx} out of range ({n} given)")
        return Segment(kind, index=idx)

    for b in range(B):
        cur = [int(t) for t in input_ids[b]]
        n_img = sum(1 for t in cur if t == IMAGE_TOKEN_INDEX)
        n_seg = sum(1 for t in cur if t == SEG_TOKEN_INDEX)
        if variant == "llava":
            hack = n_img == 0
        elif variant == "vcoder":
            hack = (n_img == 0) or (n_seg == 0)          # vcoder_llava_arch.py:187
        else:
            hack = (n_img == 0) and (n_seg == 0)         # vcoder_ds_llava_arch.py:181
        if hack:
            feat("img", cur_image_idx, n_img_feats)      # image_features[cur_image_idx] is indexed
            if variant != "llava" and has_seg:
                feat("seg", cur_seg_idx, n_seg_feats)
            half = len(cur) // 2
            plans.append([_text(cur[:half]), _text(cur[half:])])
            cur_image_idx += 1
            cur_seg_idx += 1
            cur_depth_idx += 1
            continue
        segs: List[Segment] = []
        while IMAGE_TOKEN_INDEX in cur:
            start = cur.index(IMAGE_TOKEN_INDEX)
            f = feat("img", cur_image_idx, n_img_feats)
            segs.append(_text(cur[:start]))
            segs.append(f)
            cur_image_idx += 1
            cur = cur[start + 1:]
        if variant != "llava" and has_seg:
            while SEG_TOKEN_INDEX in cur:
                start = cur.index(SEG_TOKEN_INDEX)
                f = feat("seg", cur_seg_idx, n_seg_feats)
                if variant == "vcoder":
                    segs.append(_text(cur[:start]))      # vcoder_llava_arch.py:236
                segs.append(f)                           # DS appends ONLY the seg features (:238)
                cur_seg_idx += 1
                cur = cur[start + 1:]
        if variant == "vcoder_ds":
            if not depth_is_zero[cur_depth_idx]:
                while DEPTH_TOKEN_INDEX in cur:
                    start = cur.index(DEPTH_TOKEN_INDEX)
                    f = feat("depth", cur_depth_idx, n_depth_feats)
                    segs.append(_text(cur[:start]))
                    segs.append(f)
                    cur_depth_idx += 1
                    cur = cur[start + 1:]
            else:
                cur_depth_idx += 1
        if len(cur) > 0:
            segs.append(_text(cur))
        plans.append(segs)
    return plans


def plan_length(plan: List[Segment], n_patches: int) -> int:
    return sum(len(s.ids) if s.kind == "text" else n_patches for s in plan)


# =============================================================================================
# Llama decoder (a9)   [HF] models/llama/modeling_llama.py
# =============================================================================================

def rms_norm(x, w, eps):
    """LlamaRMSNorm :53-70 — fp32 variance, x*rsqrt(var+eps), then * weight."""
    var = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def rope_cos_sin(positions: torch.Tensor, hd: int, theta: float):
    """LlamaRotaryEmbedding :73-127 — inv_freq = theta^(-2i/hd), emb = cat(freqs, freqs)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    freqs = positions.float().unsqueeze(-1) * inv_freq.unsqueeze(0)
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    """:130-135"""
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def apply_rope(x, cos, sin):
    """apply_rotary_pos_emb :138-160;  x [B,H,T,hd], cos/sin [T,hd]."""
    return x * cos + rotate_half(x) * sin


class KVCache:
    def __init__(self, n_layers: int):
        self.k: List[Optional[torch.Tensor]] = [None] * n_layers
        self.v: List[Optional[torch.Tensor]] = [None] * n_layers

    @property
    def length(self) -> int:
        return 0 if self.k[0] is None else self.k[0].shape[2]


def llama_layer(x, sd, i: int, cfg, cache: KVCache, pos0: int, r: Rounder, key_mask=None, probs_out: Optional[list] = None):
    """LlamaDecoderLayer :284-325 with LlamaAttention :217-281 and LlamaMLP :163-176."""
    B, T, D = x.shape
    H = cfg.num_attention_heads
    hd = D // H
    p = f"model.layers.{i}."
    h = r.q8(r(rms_norm(x, sd[p + "input_layernorm.weight"], cfg.rms_norm_eps)))
    q = r(F.linear(h, sd[p + "self_attn.q_proj.weight"]))
    k = r(F.linear(h, sd[p + "self_attn.k_proj.weight"]))
    v = r(F.linear(h, sd[p + "self_attn.v_proj.weight"]))
    q, k, v = (t.reshape(B, T, H, hd).transpose(1, 2) for t in (q, k, v))
    cos, sin = rope_cos_sin(torch.arange(pos0, pos0 + T), hd, cfg.rope_theta)
    q = r(apply_rope(q, cos, sin))
    k = r(apply_rope(k, cos, sin))
    if r.act_fp8:
        # the fp8 weight format keeps its KV cache in e4m3 (no scale, saturating): a cached decode step reads the cache — its own
        # new row included, appended before the attention — while the flash attention of a PREFILL reads this pass's bf16 rows
        kq, vq = quantize_e4m3(k), quantize_e4m3(v)
        if cache.k[i] is not None:
            k = torch.cat([cache.k[i], kq], dim=2)
            v = torch.cat([cache.v[i], vq], dim=2)
            cache.k[i], cache.v[i] = k, v
        else:
            cache.k[i], cache.v[i] = kq, vq
    else:
        if cache.k[i] is not None:
            k = torch.cat([cache.k[i], k], dim=2)
            v = torch.cat([cache.v[i], v], dim=2)
        cache.k[i], cache.v[i] = k, v
    a = softmax_attention(q, k, v, 1.0 / math.sqrt(hd), True, r, q_pos0=pos0, key_mask=key_mask, probs_out=probs_out)
    a = r.q8(a.transpose(1, 2).reshape(B, T, D))
    x = x + F.linear(a, sd[p + "self_attn.o_proj.weight"])
    h = r.q8(r(rms_norm(x, sd[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)))
    g = F.linear(h, sd[p + "mlp.gate_proj.weight"])
    u = F.linear(h, sd[p + "mlp.up_proj.weight"])
    m = r.q8(r(F.silu(g) * u))
    x = x + F.linear(m, sd[p + "mlp.down_proj.weight"])
    return x


def llama_forward(x, sd, cfg, cache: KVCache, emu_bf16: bool = False, last_only: bool = False, act_fp8: bool = False,
                  key_mask=None, hidden_out: Optional[list] = None, attn_out: Optional[list] = None):
    """LlamaModel.forward :367-418 on inputs_embeds + lm_head (vcoder_ds_llava_llama.py:81-93).
    position_ids = arange(T) + past_len.  act_fp8: the device's fp8 weight format quantises the decoder linears'
    activation rows in the PREFILL (a pass that starts an empty cache); cached decode steps keep bf16 activations."""
    r = Rounder(emu_bf16, act_fp8)
    pos0 = cache.length
    r.prefill = pos0 == 0
    # hidden_out (a list): receives LlamaModel's all_hidden_states ([HF] llama/modeling_llama.py: inputs_embeds, every layer's
    # output, the last entry AFTER the final norm); attn_out (a list): all_self_attns, one [B, H, T, past + T] per layer
    for i in range(cfg.num_hidden_layers):
        if hidden_out is not None:
            hidden_out.append(x.clone())
        x = llama_layer(x, sd, i, cfg, cache, pos0, r, key_mask=key_mask, probs_out=attn_out)   # key_mask [B, past + T]: 2-D attention_mask
    if hidden_out is not None:
        hidden_out.append(rms_norm(x, sd["model.norm.weight"], cfg.rms_norm_eps))
    if last_only:
        x = x[:, -1:]
    h = r(rms_norm(x, sd["model.norm.weight"], cfg.rms_norm_eps))
    return F.linear(h, sd["lm_head.weight"])  # fp32 logits


# =============================================================================================
# the model: forward / greedy generate  (a6-a11)
# =============================================================================================

class OracleModel:
    def __init__(self, cfg, state_dict: Dict[str, np.ndarray], emu_bf16: bool = False, act_fp8: bool = False):
        self.cfg = cfg
        self.act_fp8 = act_fp8
        self.sd = as_torch_state(state_dict) if not isinstance(next(iter(state_dict.values())), torch.Tensor) else state_dict
        self.emu = emu_bf16

    # -- a6: encode_images / encode_seg_images / encode_depth_images (vcoder_ds_llava_arch.py:106-119)
    def encode(self, pixels: torch.Tensor, modality: str) -> torch.Tensor:
        feats = vit_forward(pixels, self.sd, self.cfg, self.emu)
        if modality == "img":
            return projector_forward(feats, self.sd, "model.mm_projector", self.cfg.mm_projector_type, self.emu)
        # seg AND depth both go through seg_mm_projector (quirk 1, :111-114)
        return projector_forward(feats, self.sd, "model.seg_mm_projector", self.cfg.seg_mm_projector_type, self.emu)

    def embed_tokens(self, ids: Sequence[int]) -> torch.Tensor:
        # vcoder_lm_emb is overwritten with embed_tokens on every multimodal forward (quirk 3, :173)
        return self.sd["model.embed_tokens.weight"][torch.tensor(list(ids), dtype=torch.long)]

    def _encode_any(self, images, modality: str):
        """The three input forms of vcoder_ds_llava_arch.py:135-169: a 4-D tensor [B,3,S,S] -> features [B,P,D]; a list
        of [n_b,3,S,S] tensors or a 5-D tensor -> concat, encode, split by n_b and `flatten(0, 1)`: a LIST of
        [n_b*P, D] blocks, one per sample."""
        if type(images) is list or images.ndim == 5:
            items = [im for im in images]
            feats = self.encode(torch.cat(items, dim=0), modality)
            parts = torch.split(feats, [im.shape[0] for im in items], dim=0)
            return [x.flatten(0, 1) for x in parts]
        return self.encode(images, modality)

    def prepare_inputs(self, input_ids, images, segs=None, depths=None, attention_mask_given: bool = False):
        """Returns (inputs_embeds [B,S,D], plans).  Equal spliced lengths are stacked; unequal lengths are
        right-padded with zero rows when no attention_mask was passed (:278-285) and raise when one was
        (quirk 6: the reference dies with UnboundLocalError at :297)."""
        cfg = self.cfg
        if images is None:
            # the early return of prepare_inputs_labels_for_multimodal (vcoder_ds_llava_arch.py:129-133 and siblings): the ids go
            # to LlamaModel's embed_tokens as they are — a placeholder id (negative) is then torch's embedding IndexError
            rows = [list(map(int, r)) for r in input_ids]
            if any(t < 0 or t >= cfg.vocab_size for r in rows for t in r):
                raise IndexError("index out of range in self")
            return torch.stack([self.embed_tokens(r) for r in rows], 0), None
        img_f = self._encode_any(images, "img")
        seg_f = self._encode_any(segs, "seg") if (segs is not None and cfg.variant != "llava") else None
        dep_f, dz = None, None
        if cfg.variant == "vcoder_ds" and depths is not None:
            dz = [bool(torch.mean(d) == 0) for d in depths]          # :161
            dep_f = self._encode_any(depths, "depth")                # computed even if never spliced
        n_of = lambda f: 0 if f is None else (len(f) if isinstance(f, list) else f.shape[0])
        plans = splice_plan(input_ids, cfg.variant, seg_f is not None, dz, n_of(img_f), n_of(seg_f), n_of(dep_f))
        rows = []
        for plan in plans:
            parts = []
            for s in plan:
                if s.kind == "text":
                    parts.append(self.embed_tokens(s.ids))
                else:
                    parts.append({"img": img_f, "seg": seg_f, "depth": dep_f}[s.kind][s.index])
            rows.append(torch.cat(parts, dim=0))
        lens = [t.shape[0] for t in rows]
        if len(set(lens)) > 1:
            if attention_mask_given:
                raise UnboundLocalError("local variable '_new_labels' referenced before assignment "
                                        "(reference quirk: unequal spliced lengths with attention_mask and no labels)")
            S = max(lens)
            rows = [torch.cat([t, torch.zeros(S - t.shape[0], t.shape[1])], 0) for t in rows]
        return torch.stack(rows, 0), plans

    def forward(self, input_ids, images, segs=None, depths=None, cache: Optional[KVCache] = None,
                last_only: bool = False, attention_mask=None, hidden_out: Optional[list] = None,
                attn_out: Optional[list] = None):
        """VCoder[DS]LlavaLlamaForCausalLM.forward (vcoder_ds_llava_llama.py:57-118), prefill.  attention_mask [B,T]: left-
        extended with True over the S - T rows the splice added, by position (vcoder_ds_llava_arch.py:305-311), then the 2-D
        key mask of LlamaModel; kept in self.mask_ext for decode_step(keep_mask=True)."""
        x, _ = self.prepare_inputs(input_ids, images, segs, depths, attention_mask_given=attention_mask is not None)
        self.mask_ext = None
        if attention_mask is not None:
            am = torch.as_tensor(np.asarray(attention_mask)).bool()
            self.mask_ext = torch.cat([torch.ones(am.shape[0], x.shape[1] - am.shape[1], dtype=torch.bool), am], dim=1)
        cache = cache if cache is not None else KVCache(self.cfg.num_hidden_layers)
        logits = llama_forward(x, self.sd, self.cfg, cache, self.emu, last_only, self.act_fp8, key_mask=self.mask_ext,
                               hidden_out=hidden_out, attn_out=attn_out)
        return logits, cache

    def decode_step(self, tokens: Sequence[int], cache: KVCache, keep_mask: bool = False, hidden_out: Optional[list] = None,
                    attn_out: Optional[list] = None):
        """one cached step.  keep_mask False: the all-ones mask the reference's multimodal decode path builds
        (vcoder_ds_llava_arch.py:130-133); True: the prefill's mask extended with ones (a caller carrying its mask, no images)."""
        x = self.embed_tokens(tokens).unsqueeze(1)
        km = None
        if keep_mask and getattr(self, "mask_ext", None) is not None:
            n = cache.length + 1 - self.mask_ext.shape[1]
            km = torch.cat([self.mask_ext, torch.ones(self.mask_ext.shape[0], n, dtype=torch.bool)], dim=1)
        return llama_forward(x, self.sd, self.cfg, cache, self.emu, last_only=True, act_fp8=self.act_fp8, key_mask=km,
                             hidden_out=hidden_out, attn_out=attn_out)

    def generate_greedy(self, input_ids, images, segs=None, depths=None, max_new_tokens: int = 8,
                        eos_token_id: Optional[int] = None, pad_token_id: int = 0, return_logits: bool = False):
        """HF GenerationMixin greedy loop: logits[:, -1] in fp32, argmax with
        lowest-index ties, finished rows emit pad, stop when every row has produced EOS."""
        logits, cache = self.forward(input_ids, images, segs, depths, last_only=True)
        B = logits.shape[0]
        unfinished = torch.ones(B, dtype=torch.long)
        out, all_logits = [], []
        for step in range(max_new_tokens):
            last = logits[:, -1].float()
            all_logits.append(last.clone())
            nxt = torch.argmax(last, dim=-1)
            if eos_token_id is not None:
                nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
            out.append(nxt)
            if eos_token_id is not None:
                unfinished = unfinished * (nxt != eos_token_id).long()
                if int(unfinished.max()) == 0:
                    break
            if step + 1 < max_new_tokens:
                logits = self.decode_step(nxt.tolist(), cache)
        ids = torch.stack(out, 1)
        return (ids, torch.stack(all_logits, 1)) if return_logits else ids
