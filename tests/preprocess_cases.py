"""Parity of the device-side image preprocessing (vc_preprocess_image) with what the reference does on the host:
expand2square (vcoder_llava/mm_utils.py:14-25) + PIL bicubic resize / center crop + rescale + normalise (HF
CLIPImageProcessor).  The expected values come from PIL itself (the third-party implementation the reference calls);
the uint8 stages must match bit for bit, the float stage to 1e-6."""
from __future__ import annotations

import numpy as np
from PIL import Image

from vcoder_amd import mm_utils, synth


def expected(img: Image.Image, S: int, pad: bool) -> np.ndarray:
    if pad:
        img = mm_utils.expand2square(img, tuple(int(x * 255) for x in synth.CLIP_MEAN))
    w, h = img.size
    if h <= w:
        nh, nw = S, int(S * w / h)
    else:
        nw, nh = S, int(S * h / w)
    img = img.resize((nw, nh), resample=Image.BICUBIC)
    top, left = (nh - S) // 2, (nw - S) // 2
    a = np.asarray(img)[top:top + S, left:left + S].astype(np.float64) * 0.00392156862745098
    a = a.astype(np.float32)
    a = (a - synth.CLIP_MEAN) / synth.CLIP_STD
    return np.ascontiguousarray(a.transpose(2, 0, 1))


def check_preprocess(eng, sizes, to_device=False):
    rng = np.random.RandomState(0)
    S = eng.cfg.vit_image_size
    for (h, w) in sizes:
        # smooth + noisy content so both the antialiasing and the rounding paths are exercised
        yy, xx = np.mgrid[0:h, 0:w]
        base = (np.stack([xx * 255 // max(w - 1, 1), yy * 255 // max(h - 1, 1), (xx + yy) % 256], -1)).astype(np.int64)
        a = np.clip(base + rng.randint(-40, 40, size=(h, w, 3)), 0, 255).astype(np.uint8)
        img = Image.fromarray(a)
        for pad in (True, False):
            got = eng.preprocess([img], pad=pad, to_device=to_device)
            got = got.cpu().numpy() if to_device else got
            ref = expected(img, S, pad)
            err = np.abs(got[0] - ref).max()
            assert err < 1e-6, f"preprocess {h}x{w} pad={pad}: max err {err}"


def check_against_hf_processor(eng, tmp_path):
    """Same numbers as the real CLIPImageProcessor + the reference's own process_images glue (pad mode)."""
    import json
    from types import SimpleNamespace
    from transformers import CLIPImageProcessor

    S = eng.cfg.vit_image_size
    d = str(tmp_path)
    with open(f"{d}/preprocessor_config.json", "w") as f:
        json.dump({"crop_size": S, "size": S, "do_center_crop": True, "do_normalize": True, "do_resize": True,
                   "image_mean": synth.CLIP_MEAN.tolist(), "image_std": synth.CLIP_STD.tolist(), "resample": 3,
                   "image_processor_type": "CLIPImageProcessor"}, f)
    proc = CLIPImageProcessor.from_pretrained(d)
    rng = np.random.RandomState(1)
    img = Image.fromarray(rng.randint(0, 256, size=(90, 130, 3)).astype(np.uint8))
    ref = mm_utils.process_images([img], proc, SimpleNamespace(image_aspect_ratio="pad")).numpy()
    got = eng.preprocess([img], pad=True)
    assert np.abs(got - ref).max() < 1e-6
