"""Makes the reference's entry points (vcoder_llava.serve.cli, serve.chat, the eval loaders) run on this backend
unchanged:

    import vcoder_amd.dropin; vcoder_amd.dropin.install()      # before `import vcoder_llava.serve.cli`

What install() does — and deliberately does NOT do:

* it never executes the reference's `vcoder_llava/__init__.py` (that file imports the torch/HF model classes, whose
  `AutoConfig.register("llava", ...)` raises under Transformers >= 4.36 and which are exactly what this backend replaces);
* it pre-seeds `sys.modules["vcoder_llava"]` with a package object whose `__path__` is the reference's package directory
  (found with `importlib.util.find_spec`, which does not import anything, or passed as `reference_root`), so the
  reference's pure-Python glue — `serve/`, `eval/`, `vcoder_conversation`, `constants`, `questions`, `mm_utils`,
  `utils` — keeps resolving to the reference's own files;
* it aliases the model layer (`vcoder_llava.model`, `.model.builder`, the three projector builders, the vision-tower
  builder and the three `language_model.*` modules) to vcoder_amd's counterparts BEFORE anything can import them.

Without the reference on the path the package is a stub and `constants` / `mm_utils` fall back to vcoder_amd's own
restatements, so `from vcoder_llava.model.builder import load_pretrained_model` still works.  See INTEGRATION.md."""
from __future__ import annotations

import importlib.util
import os
import sys
import types
from typing import Optional

_MODEL_ALIASES = (
    "vcoder_llava.model.multimodal_projector.builder",
    "vcoder_llava.model.multimodal_adapter.builder",
    "vcoder_llava.model.multimodal_depth_adapter.builder",
)


def _reference_dir(reference_root: Optional[str]) -> Optional[str]:
    if reference_root is not None:
        d = os.path.join(reference_root, "vcoder_llava")
        if not os.path.isdir(d):
            raise FileNotFoundError(f"{d} is not a directory")
        return d
    try:
        spec = importlib.util.find_spec("vcoder_llava")  # top-level lookup: nothing is executed
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        return list(spec.submodule_search_locations)[0]
    return None


def _namespace(name: str, path: Optional[str], parent: Optional[types.ModuleType]) -> types.ModuleType:
    """A package object that is never executed: submodule imports go through its __path__."""
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    m.__package__ = name
    sys.modules[name] = m
    if parent is not None:
        setattr(parent, name.rsplit(".", 1)[1], m)
    return m


def install(reference_root: Optional[str] = None) -> types.ModuleType:
    from . import constants, mm_utils, model
    from .model import builder, language_model, projector, vision_tower

    ref = _reference_dir(reference_root)
    old = sys.modules.get("vcoder_llava")
    if old is not None and getattr(old, "_vcoder_amd_dropin", False):
        return old
    # drop anything of the reference's model layer that was imported before us (it must not shadow the aliases)
    for k in [k for k in sys.modules if k == "vcoder_llava" or k.startswith("vcoder_llava.")]:
        del sys.modules[k]
    pkg = _namespace("vcoder_llava", ref, None)
    pkg._vcoder_amd_dropin = True
    pkg.__file__ = os.path.join(ref, "__init__.py") if ref else None

    # ---- the model layer: ours, under the reference's module names
    sys.modules["vcoder_llava.model"] = model
    pkg.model = model
    sys.modules["vcoder_llava.model.builder"] = builder
    for name in _MODEL_ALIASES:
        parent = name.rsplit(".", 1)[0]
        if parent not in sys.modules:
            _namespace(parent, None, None)
        sys.modules[name] = projector
    if "vcoder_llava.model.multimodal_encoder" not in sys.modules:
        _namespace("vcoder_llava.model.multimodal_encoder", None, None)
    sys.modules["vcoder_llava.model.multimodal_encoder.builder"] = vision_tower
    sys.modules["vcoder_llava.model.multimodal_encoder.clip_encoder"] = vision_tower
    _namespace("vcoder_llava.model.language_model", None, None)
    for name in ("llava_llama", "vcoder_llava_llama", "vcoder_ds_llava_llama"):
        sys.modules["vcoder_llava.model.language_model." + name] = language_model
    # the reference's package __init__ re-exports the three model classes
    for cls in ("LlavaLlamaForCausalLM", "VCoderLlavaLlamaForCausalLM", "VCoderDSLlavaLlamaForCausalLM"):
        setattr(pkg, cls, getattr(language_model, cls))

    # the import side effect of the reference's model modules: AutoConfig / AutoModelForCausalLM know the VCoder model types
    # (vcoder_ds_llava_llama.py:144-145).  Optional glue: skipped when Transformers is not installed.
    try:
        from . import hf_register

        hf_register.register()
    except (ImportError, TypeError, ValueError):   # optional glue: an absent or older Transformers must not break the install
        pass

    # ---- pure-Python glue: the reference's own files when present, ours otherwise
    if ref is None:
        sys.modules["vcoder_llava.constants"] = constants
        sys.modules["vcoder_llava.mm_utils"] = mm_utils
        pkg.constants, pkg.mm_utils = constants, mm_utils
    return pkg


def uninstall() -> None:
    for k in [k for k in sys.modules if k == "vcoder_llava" or k.startswith("vcoder_llava.")]:
        del sys.modules[k]
