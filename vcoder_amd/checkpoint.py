"""Checkpoint I/O: HF-layout directories (config.json + *.safetensors | pytorch_model*.bin) <-> state dicts.

Replaces the `from_pretrained` plumbing of vcoder_llava/model/builder.py:93-108 and
multimodal_encoder/clip_encoder.py:22-27 for the inference path (no bitsandbytes / LoRA paths)."""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Iterator, Tuple

import numpy as np


def iter_checkpoint_tensors(path: str) -> Iterator[Tuple[str, object]]:
    """Yields (key, tensor) from every weight shard in `path`; tensors are torch tensors (any float dtype)."""
    st = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if st:
        from safetensors import safe_open

        for f in st:
            with safe_open(f, framework="pt", device="cpu") as sf:
                for k in sf.keys():
                    yield k, sf.get_tensor(k)
        return
    bins = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin"))) + sorted(glob.glob(os.path.join(path, "mm_projector.bin")))
    if not bins:
        raise FileNotFoundError(f"no *.safetensors or pytorch_model*.bin under {path}")
    import torch

    for f in bins:
        sd = torch.load(f, map_location="cpu", weights_only=True)
        for k, v in sd.items():
            yield k, v


def has_weights(path: str) -> bool:
    return bool(glob.glob(os.path.join(path, "*.safetensors")) or glob.glob(os.path.join(path, "pytorch_model*.bin")))


def save_checkpoint(path: str, config_dict: dict, state: Dict[str, np.ndarray], bf16: bool = True) -> None:
    """Writes an HF-layout checkpoint (used by tests / synthetic model export)."""
    import torch
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(config_dict, f, indent=1)
    tens = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}
    if bf16:
        tens = {k: v.to(torch.bfloat16) for k, v in tens.items()}
    save_file(tens, os.path.join(path, "model.safetensors"))


# ---- the reference loader's overlay paths (vcoder_llava/model/builder.py:42-92) -----------------------------------------------
def _torch_load(path):
    import torch

    return torch.load(path, map_location="cpu", weights_only=True)


def _strip_peft_prefixes(sd: dict) -> dict:
    """builder.py:66-68: drop a leading 'base_model.' and, if keys then start with 'model.model.', one 'model.'"""
    sd = {(k[11:] if k.startswith("base_model.") else k): v for k, v in sd.items()}
    if any(k.startswith("model.model.") for k in sd):
        sd = {(k[6:] if k.startswith("model.") else k): v for k, v in sd.items()}
    return sd


def load_lora_adapter(path: str):
    """adapter_config.json + adapter_model.{safetensors,bin} of a peft LoRA checkpoint ->
    ({target state-dict key: (A [r, in], B [out, r])}, scale = lora_alpha / r, fan_in_fan_out).

    Plain LoRA only: anything that would change what `PeftModel.merge_and_unload` computes (rsLoRA scaling, DoRA magnitudes,
    per-module rank / alpha patterns, saved extra modules, LoRA on embeddings) is refused instead of being merged wrongly."""
    with open(os.path.join(path, "adapter_config.json")) as f:
        ac = json.load(f)
    if ac.get("peft_type", "LORA") != "LORA":
        raise ValueError(f"unsupported peft_type {ac.get('peft_type')}")
    for flag in ("use_rslora", "use_dora"):
        if ac.get(flag):
            raise NotImplementedError(f"LoRA adapter with {flag}=true: merged weights would differ from peft's (not supported)")
    for pat in ("rank_pattern", "alpha_pattern"):
        if ac.get(pat):
            raise NotImplementedError(f"LoRA adapter with a non-empty {pat}: per-module scales are not supported")
    if ac.get("modules_to_save"):
        raise NotImplementedError(f"LoRA adapter with modules_to_save={ac['modules_to_save']}: saved modules are not supported")
    st = os.path.join(path, "adapter_model.safetensors")
    if os.path.exists(st):
        from safetensors import safe_open

        with safe_open(st, framework="pt", device="cpu") as sf:
            raw = {k: sf.get_tensor(k) for k in sf.keys()}
    else:
        raw = _torch_load(os.path.join(path, "adapter_model.bin"))
    pairs, unknown = {}, []
    for k, v in raw.items():
        for tag, idx in ((".lora_A.", 0), (".lora_B.", 1)):
            if tag in k:
                base = k.split(tag)[0] + ".weight"                       # ...q_proj.lora_A[.default].weight -> ...q_proj.weight
                base = base[len("base_model.model."):] if base.startswith("base_model.model.") else base
                pairs.setdefault(base, [None, None])[idx] = v.float()
                break
        else:
            unknown.append(k)   # lora_embedding_A/B, lora_magnitude_vector, modules_to_save copies, ...
    if unknown:
        raise NotImplementedError(f"LoRA adapter holds tensors that are not lora_A / lora_B matrices: {sorted(unknown)[:3]}")
    missing = [k for k, (a, b) in pairs.items() if a is None or b is None]
    if missing:
        raise ValueError(f"LoRA adapter lacks a lora_A / lora_B partner for {missing[:3]}")
    return {k: (a, b) for k, (a, b) in pairs.items()}, float(ac["lora_alpha"]) / float(ac["r"]), bool(ac.get("fan_in_fan_out", False))


def iter_lora_merged(model_base: str, model_path: str) -> Iterator[Tuple[str, object]]:
    """builder.py:42-77 in the reference's order: the base LLM's tensors, overridden by the non-LoRA trainables of the LoRA
    checkpoint (`load_state_dict(non_lora_trainables, strict=False)`, :72 — projector etc., and embed_tokens / lm_head when the
    fine-tune saved them, possibly with another vocabulary size: the model is built from the LoRA checkpoint's config), THEN
    the LoRA deltas merged in (W + alpha / r * B @ A — what PeftModel.merge_and_unload computes, :74-77)."""
    pairs, scale, fifo = load_lora_adapter(model_path)
    nl = os.path.join(model_path, "non_lora_trainables.bin")
    if not os.path.exists(nl):
        raise FileNotFoundError(f"{nl} (no network here: the reference would download it from the hub)")
    over = _strip_peft_prefixes(_torch_load(nl))
    used = set()

    def merged(k, v):
        if k in pairs:
            a, b = pairs[k]
            delta = (b @ a) * scale
            v = v.float() + (delta.t() if fifo else delta)
            used.add(k)
        return v

    for k, v in iter_checkpoint_tensors(model_base):
        if k in over:
            continue   # the checkpoint's own copy wins (yielded below; its shape follows the LoRA checkpoint's config)
        yield k, merged(k, v)
    for k, v in over.items():
        yield k, merged(k, v)
    left = set(pairs) - used
    if left:
        raise KeyError(f"LoRA targets not found in the base checkpoint: {sorted(left)[:3]}")


def iter_base_with_projector(model_base: str, model_path: str) -> Iterator[Tuple[str, object]]:
    """builder.py:78-88: the base LLM's tensors, then the projector weights of a projector-only checkpoint"""
    for k, v in iter_checkpoint_tensors(model_base):
        yield k, v
    for k, v in _torch_load(os.path.join(model_path, "mm_projector.bin")).items():
        yield k, v

