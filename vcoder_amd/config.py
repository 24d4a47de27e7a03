"""Model configuration for the VCoder hot path.

Mirrors the keys the reference's constructors read from the HF `config.json`
(vcoder_ds_llava_arch.py:30-49, clip_encoder.py:13-15,
multimodal_projector/builder.py:33-46, multimodal_adapter/builder.py:31-44,
multimodal_depth_adapter/builder.py:32-45) plus the CLIP vision dims that the reference reads from the
separately-downloaded CLIP repo (clip_encoder.py:20-24).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field, asdict
from typing import Any, Dict, Optional

# model_type strings carried by checkpoints (vcoder_ds_llava_llama.py:30-31,144 etc.)
MODEL_TYPE_TO_VARIANT = {"vcoder_ds_llava": "vcoder_ds", "vcoder_llava": "vcoder", "llava": "llava"}
VARIANT_TO_MODEL_TYPE = {v: k for k, v in MODEL_TYPE_TO_VARIANT.items()}


@dataclass
class VCoderConfig:
    # --- which reference class this is: 'llava' | 'vcoder' | 'vcoder_ds'
    variant: str = "vcoder_ds"
    # --- LlamaConfig fields
    vocab_size: int = 32000
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: Optional[int] = None
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    max_position_embeddings: int = 4096
    pad_token_id: Optional[int] = 0
    bos_token_id: int = 1
    eos_token_id: int = 2
    max_sequence_length: Optional[int] = None
    # --- multimodal keys (consumed via hasattr/getattr in the reference)
    mm_vision_tower: str = "openai/clip-vit-large-patch14-336"
    mm_vision_select_layer: int = -2
    mm_vision_select_feature: str = "patch"
    mm_projector_type: str = "mlp2x_gelu"
    mm_hidden_size: int = 1024
    seg_mm_projector_type: str = "mlp2x_gelu"
    seg_mm_hidden_size: int = 1024
    depth_mm_projector_type: str = "mlp2x_gelu"
    depth_mm_hidden_size: int = 1024
    use_mm2_proj: bool = False
    mm_vcoder_lm_emb: bool = True
    image_aspect_ratio: str = "pad"
    # --- CLIP ViT dims (openai/clip-vit-large-patch14-336 defaults)
    vit_intermediate_size: int = 4096
    vit_num_layers: int = 24
    vit_num_heads: int = 16
    vit_image_size: int = 336
    vit_patch_size: int = 14
    vit_layer_norm_eps: float = 1e-5
    extra: Dict[str, Any] = field(default_factory=dict)

    # ---- derived
    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def vit_head_dim(self) -> int:
        return self.mm_hidden_size // self.vit_num_heads

    @property
    def num_patches(self) -> int:
        return (self.vit_image_size // self.vit_patch_size) ** 2

    @property
    def vit_layers_used(self) -> int:
        """hidden_states[select_layer]: index k of the (Lv+1)-tuple = output of encoder layer k.

        select_layer=-2 with 24 layers -> index 23 -> 23 layers are computed, the 24th and
        post_layernorm never are."""
        sl = self.mm_vision_select_layer
        k = sl if sl >= 0 else self.vit_num_layers + 1 + sl
        if not (0 <= k <= self.vit_num_layers):
            raise ValueError(f"mm_vision_select_layer={sl} out of range for {self.vit_num_layers} layers")
        return k

    @property
    def model_type(self) -> str:
        return VARIANT_TO_MODEL_TYPE[self.variant]

    def validate(self) -> None:
        if self.variant not in VARIANT_TO_MODEL_TYPE:
            raise ValueError(f"unknown variant {self.variant}")
        kv = self.num_key_value_heads
        if kv is not None and kv != self.num_attention_heads:
            raise ValueError("grouped-query attention is not part of the VCoder hot path (Vicuna-1.5 uses MHA)")
        if self.hidden_size % self.num_attention_heads:
            raise ValueError("hidden_size must be divisible by num_attention_heads")
        if self.mm_vision_select_feature not in ("patch", "cls_patch"):
            raise ValueError(f"Unexpected select feature: {self.mm_vision_select_feature}")  # clip_encoder.py:36

    # ---- (de)serialisation in the HF config.json shape
    def to_hf_dict(self) -> Dict[str, Any]:
        d = asdict(self)
        d.pop("extra")
        variant = d.pop("variant")
        d["model_type"] = VARIANT_TO_MODEL_TYPE[variant]
        d["architectures"] = [{"vcoder_ds": "VCoderDSLlavaLlamaForCausalLM", "vcoder": "VCoderLlavaLlamaForCausalLM",
                               "llava": "LlavaLlamaForCausalLM"}[variant]]
        if variant == "llava":
            for k in ("seg_mm_projector_type", "seg_mm_hidden_size", "use_mm2_proj", "mm_vcoder_lm_emb"):
                d.pop(k)
        if variant != "vcoder_ds":
            for k in ("depth_mm_projector_type", "depth_mm_hidden_size"):
                d.pop(k)
        d["vision_config"] = {"hidden_size": self.mm_hidden_size, "intermediate_size": self.vit_intermediate_size,
                              "num_hidden_layers": self.vit_num_layers, "num_attention_heads": self.vit_num_heads,
                              "image_size": self.vit_image_size, "patch_size": self.vit_patch_size,
                              "layer_norm_eps": self.vit_layer_norm_eps}
        for k in list(d):
            if k.startswith("vit_"):
                d.pop(k)
        d.update(self.extra)
        return d

    @classmethod
    def from_hf_dict(cls, d: Dict[str, Any], model_name: Optional[str] = None) -> "VCoderConfig":
        d = dict(d)
        mt = d.pop("model_type", None)
        variant = MODEL_TYPE_TO_VARIANT.get(mt)
        if variant is None and model_name is not None:
            # name-substring dispatch of the reference loader (builder.py:93-108)
            low = model_name.lower()
            variant = "vcoder_ds" if "vcoder_ds_llava" in low else "vcoder" if "vcoder_llava" in low else "llava"
        if variant is None:
            raise ValueError(f"cannot infer model variant from model_type={mt!r}")
        kw: Dict[str, Any] = {"variant": variant}
        vc = d.pop("vision_config", None) or {}
        vmap = {"intermediate_size": "vit_intermediate_size", "num_hidden_layers": "vit_num_layers",
                "num_attention_heads": "vit_num_heads", "image_size": "vit_image_size", "patch_size": "vit_patch_size",
                "layer_norm_eps": "vit_layer_norm_eps"}
        for k, v in vc.items():
            if k in vmap:
                kw[vmap[k]] = v
        names = {f for f in cls.__dataclass_fields__ if f not in ("variant", "extra")}
        extra = {}
        for k, v in d.items():
            if k in names:
                kw[k] = v
            else:
                extra[k] = v
        kw["extra"] = extra
        cfg = cls(**kw)
        if cfg.pad_token_id is None:
            cfg.pad_token_id = 0
        cfg.validate()
        return cfg

    @classmethod
    def from_pretrained(cls, path: str, model_name: Optional[str] = None) -> "VCoderConfig":
        with open(os.path.join(path, "config.json")) as f:
            return cls.from_hf_dict(json.load(f), model_name)

    def __getattr__(self, name):  # config.<unknown hf key> falls through to extra (getattr(cfg, k, default) users)
        extra = self.__dict__.get("extra")
        if extra is not None and name in extra:
            return extra[name]
        raise AttributeError(name)


def vicuna_7b(variant: str = "vcoder_ds") -> VCoderConfig:
    return VCoderConfig(variant=variant)


def vicuna_13b(variant: str = "vcoder_ds") -> VCoderConfig:
    return VCoderConfig(variant=variant, hidden_size=5120, intermediate_size=13824, num_hidden_layers=40,
                        num_attention_heads=40)


def tiny(variant: str = "vcoder_ds") -> VCoderConfig:
    """Tiny architecture used by the golden fixtures: same head dims as the real models
    (ViT hd 64, LLM hd 128) so the true-shape attention kernels are exercised."""
    return VCoderConfig(variant=variant, vocab_size=320, hidden_size=256, intermediate_size=384, num_hidden_layers=2,
                        num_attention_heads=2, mm_hidden_size=128, seg_mm_hidden_size=128, depth_mm_hidden_size=128,
                        vit_intermediate_size=256, vit_num_layers=3, vit_num_heads=2, vit_image_size=56,
                        vit_patch_size=14, use_mm2_proj=True, mm_vcoder_lm_emb=True, max_position_embeddings=512)
