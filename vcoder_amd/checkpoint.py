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
