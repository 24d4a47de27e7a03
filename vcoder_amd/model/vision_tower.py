"""CLIPVisionTower counterpart (vcoder_llava/model/multimodal_encoder/{builder.py:5-11, clip_encoder.py:7-77}).

The tower's arithmetic (K1-K8 of SURVEY.md §2) runs inside libvcoder_hip.so; this object carries the attributes the
reference's callers read: is_loaded / load_model() / image_processor / num_patches / hidden_size / config / to()."""
from __future__ import annotations

import json
import os
from types import SimpleNamespace


class HipCLIPVisionTower:
    def __init__(self, vision_tower: str, args, delay_load: bool = False):
        self.is_loaded = False
        self.vision_tower_name = vision_tower
        self.select_layer = args.mm_vision_select_layer
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch")
        self._args = args
        self.image_processor = None
        self._engine = None
        if not delay_load:
            self.load_model()

    # -- the CLIP repo is needed for two things: the preprocessor config and (if the VCoder checkpoint does not embed
    #    them) the tower weights
    def load_model(self, engine=None):
        path = self.vision_tower_name
        if self.image_processor is None:
            self.image_processor = load_image_processor(path, self._args)
        if engine is not None:
            from .. import checkpoint

            if not (os.path.isdir(path) and checkpoint.has_weights(path)):
                raise FileNotFoundError(
                    f"CLIP tower weights not found: '{path}' is not a local directory with weights and this build has "
                    "no network access (the reference downloads it at clip_encoder.py:24)")
            for k, v in checkpoint.iter_checkpoint_tensors(path):
                engine.load_tensor(k, v)
            self._engine = engine
        self.is_loaded = True

    def to(self, *a, **k):
        return self

    def requires_grad_(self, flag=False):
        return self

    @property
    def config(self):
        a = self._args
        return SimpleNamespace(hidden_size=a.mm_hidden_size, image_size=a.vit_image_size, patch_size=a.vit_patch_size,
                               num_hidden_layers=a.vit_num_layers, num_attention_heads=a.vit_num_heads,
                               intermediate_size=a.vit_intermediate_size)

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2

    def feature_select(self, image_forward_outs):
        """clip_encoder.py:29-37 on a tower output.  The engine evaluates exactly hidden_states[select_layer] (later
        layers are never computed), so the object forward() works with carries that one tensor (CLS row included);
        an HF-style output with a `hidden_states` sequence is accepted too."""
        hs = getattr(image_forward_outs, "hidden_states", None)
        feats = hs[self.select_layer] if isinstance(hs, (list, tuple)) else image_forward_outs.selected_hidden_state
        if self.select_feature == "patch":
            return feats[:, 1:]
        if self.select_feature == "cls_patch":
            return feats
        raise ValueError(f"Unexpected select feature: {self.select_feature}")

    def forward(self, images):
        """clip_encoder.py:39-51: [B,3,S,S] tensor (or a list of [3,S,S] images) -> un-projected features
        [B, num_patches, hidden_size] in the dtype / on the device of the input (vc_vision_tower_forward)."""
        import numpy as np
        import torch

        if self._engine is None or not self._engine.finalized:
            raise RuntimeError("vision tower weights are not loaded (load_model() / finalize_weights() first)")
        if type(images) is list:
            return [self.forward(im.unsqueeze(0)) for im in images]
        t = images if isinstance(images, torch.Tensor) else torch.as_tensor(np.asarray(images))
        feats = torch.from_numpy(self._engine.vision_tower_forward(t))
        return feats.to(device=t.device, dtype=t.dtype if t.dtype.is_floating_point else torch.float32)

    __call__ = forward

    @property
    def dummy_feature(self):
        import torch

        return torch.zeros(1, self.hidden_size)


def load_image_processor(path: str, args):
    """CLIPImageProcessor for `path` if HF Transformers + a preprocessor_config.json are available; otherwise a
    dependency-free processor with the CLIP constants (mean/std, square size) from the config."""
    cfg_file = os.path.join(path, "preprocessor_config.json") if os.path.isdir(path) else None
    if cfg_file and os.path.exists(cfg_file):
        try:
            from transformers import CLIPImageProcessor

            return CLIPImageProcessor.from_pretrained(path)
        except Exception:
            with open(cfg_file) as f:
                pc = json.load(f)
            size = pc.get("crop_size", args.vit_image_size)
            size = size["height"] if isinstance(size, dict) else size
            return SimpleClipProcessor(size, pc.get("image_mean"), pc.get("image_std"))
    return SimpleClipProcessor(args.vit_image_size)


class SimpleClipProcessor:
    """resize(shortest side, bicubic) -> center crop -> /255 -> normalise; `preprocess(img, return_tensors='pt')`."""

    def __init__(self, size: int, mean=None, std=None):
        from ..synth import CLIP_MEAN, CLIP_STD

        self.size = size
        self.crop_size = {"height": size, "width": size}
        self.image_mean = list(mean) if mean is not None else CLIP_MEAN.tolist()
        self.image_std = list(std) if std is not None else CLIP_STD.tolist()

    def preprocess(self, images, return_tensors="pt"):
        import numpy as np
        from PIL import Image

        if not isinstance(images, (list, tuple)):
            images = [images]
        out = []
        for im in images:
            im = im.convert("RGB")
            w, h = im.size
            s = self.size / min(w, h)
            im = im.resize((max(self.size, round(w * s)), max(self.size, round(h * s))), Image.BICUBIC)
            w, h = im.size
            l, t = (w - self.size) // 2, (h - self.size) // 2
            im = im.crop((l, t, l + self.size, t + self.size))
            a = np.asarray(im, dtype=np.float32) / 255.0
            a = (a - np.array(self.image_mean, np.float32)) / np.array(self.image_std, np.float32)
            out.append(a.transpose(2, 0, 1))
        arr = np.stack(out)
        if return_tensors == "pt":
            import torch

            arr = torch.from_numpy(arr)
        return {"pixel_values": arr}

    __call__ = preprocess


def build_vision_tower(vision_tower_cfg, **kwargs):
    """multimodal_encoder/builder.py:5-11 — same acceptance rule and error."""
    vision_tower = getattr(vision_tower_cfg, "mm_vision_tower", getattr(vision_tower_cfg, "vision_tower", None))
    if vision_tower is not None and (os.path.exists(vision_tower) or vision_tower.startswith("openai")
                                     or vision_tower.startswith("laion")):
        return HipCLIPVisionTower(vision_tower, args=vision_tower_cfg, **kwargs)
    raise ValueError(f"Unknown vision tower: {vision_tower}")
