"""Makes the reference's entry points (vcoder_llava.serve.cli, eval loaders) run on this backend unchanged:

    import vcoder_amd.dropin; vcoder_amd.dropin.install()      # before `import vcoder_llava.serve.cli`

registers vcoder_amd's `load_pretrained_model` / model classes / projector builders under the module names the
reference imports (`vcoder_llava.model.builder`, `vcoder_llava.model`), leaving the reference's pure-Python glue
(conversation templates, mm_utils, constants) untouched.  See INTEGRATION.md."""
from __future__ import annotations

import sys
import types


def install() -> None:
    from . import constants, model
    from .model import builder, projector, vision_tower

    pkg = sys.modules.get("vcoder_llava")
    if pkg is None:
        try:
            import vcoder_llava as pkg  # the reference's glue package, if it is on sys.path
        except Exception:
            pkg = types.ModuleType("vcoder_llava")
            pkg.__path__ = []
            sys.modules["vcoder_llava"] = pkg
            from . import mm_utils

            sys.modules["vcoder_llava.constants"] = constants
            sys.modules["vcoder_llava.mm_utils"] = mm_utils
    sys.modules["vcoder_llava.model"] = model
    sys.modules["vcoder_llava.model.builder"] = builder
    sys.modules["vcoder_llava.model.multimodal_projector.builder"] = projector
    sys.modules["vcoder_llava.model.multimodal_adapter.builder"] = projector
    sys.modules["vcoder_llava.model.multimodal_depth_adapter.builder"] = projector
    sys.modules["vcoder_llava.model.multimodal_encoder.builder"] = vision_tower
    pkg.model = model
