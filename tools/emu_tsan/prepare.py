"""TEST TOOL: writes the inputs of tools/emu_tsan/stress.cpp — the tiny VCoder-DS config as a vc_model_cfg, the synthetic tensor
specs (key, shape, seed, offset, half width: what HipEngine.load_synthetic feeds vc_model_synth_tensor) and three prompts."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402

import e2e_cases  # noqa: E402
from vcoder_amd import config as vcfg, synth  # noqa: E402
from vcoder_amd.engine import HipEngine  # noqa: E402

out = sys.argv[1]
cfg = vcfg.tiny("vcoder_ds")
c = HipEngine._model_cfg(cfg) if hasattr(HipEngine, "_model_cfg") else None
with open(out, "w") as f:
    fields = ["variant", "vit_hidden", "vit_heads", "vit_ffn", "vit_layers", "vit_layers_used", "vit_image", "vit_patch",
              "vit_keep_cls", "vit_ln_eps", "hidden", "heads", "ffn", "layers", "vocab", "max_positions", "rms_eps", "rope_theta",
              "mm_proj_depth", "seg_proj_depth", "pad_token_id"]
    assert c is not None, "HipEngine._model_cfg missing"
    f.write(" ".join(repr(getattr(c, k)) for k in fields) + "\n")
    specs = list(synth.tensor_specs(cfg))
    f.write("%d\n" % len(specs))
    for key, shape, off, hw in specs:
        f.write("%s %d %s %d %r %r\n" % (key, len(shape), " ".join(str(int(d)) for d in shape), synth.tensor_seed(key, 42), float(off), float(hw)))
    names = ["ds_img_depth_seg", "ds_img_only", "ds_img_seg"]
    f.write("%d\n" % len(names))
    for n in names:
        g, _, ids, imgs, segs, deps = e2e_cases.fixture_inputs(n)
        f.write("%d %d %d %d %s\n" % (ids.shape[0], ids.shape[1], int(segs is not None), int(deps is not None),
                                      " ".join(str(int(t)) for t in ids.reshape(-1))))
    f.write("%d\n" % cfg.vit_image_size)
print("wrote", out)
