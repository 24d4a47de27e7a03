"""Registration with HF Transformers' Auto classes — the counterpart of the reference's

    AutoConfig.register("vcoder_ds_llava", VCoderDSLlavaConfig)
    AutoModelForCausalLM.register(VCoderDSLlavaConfig, VCoderDSLlavaLlamaForCausalLM)

(vcoder_llava/model/language_model/vcoder_ds_llava_llama.py:144-145, vcoder_llava_llama.py:141-142, llava_llama.py:139-140),
so that `AutoModelForCausalLM.from_pretrained(<vcoder checkpoint>)` resolves the checkpoint's `model_type` to THIS backend.

Transformers is third-party glue here, not part of the hot path: nothing in vcoder_amd imports it unless `register()` is
called (explicitly, or by `vcoder_amd.dropin.install()`, which reproduces the import side effects of the reference's model
package).  The Auto classes need `PretrainedConfig` subclasses, so each model_type gets a thin one that only carries the
checkpoint's config.json keys; the model classes convert it to `vcoder_amd.config.VCoderConfig` in from_pretrained.

`llava` is registered only where Transformers does not already own that model_type (>= 4.36 ships its own LlavaConfig;
the reference's own registration raises ValueError there) — VCoder checkpoints carry `vcoder_llava` / `vcoder_ds_llava`."""
from __future__ import annotations

_registered = {}


def register(exist_ok: bool = True):
    """-> {model_type: (hf config class, model class)} of what is registered with AutoConfig / AutoModelForCausalLM"""
    if _registered:
        return dict(_registered)
    import inspect

    from transformers import AutoConfig, AutoModelForCausalLM, PretrainedConfig
    from transformers.models.auto.configuration_auto import CONFIG_MAPPING

    # `exist_ok` only exists from Transformers 4.32 on; the reference pins 4.31 (pyproject.toml:23), whose register()
    # takes (model_type, config) / (config_class, model_class) and raises ValueError for a key it already has
    def _kw(fn):
        try:
            return {"exist_ok": exist_ok} if "exist_ok" in inspect.signature(fn).parameters else {}
        except (TypeError, ValueError):
            return {}

    from .model import language_model as lm

    def make(model_type: str, name: str):
        def __init__(self, **kwargs):
            PretrainedConfig.__init__(self, **kwargs)   # every config.json key becomes an attribute

        return type(name, (PretrainedConfig,), {"model_type": model_type, "__init__": __init__, "__module__": __name__})

    for model_type, name, model_cls in (("vcoder_ds_llava", "VCoderDSLlavaHFConfig", lm.VCoderDSLlavaLlamaForCausalLM),
                                        ("vcoder_llava", "VCoderLlavaHFConfig", lm.VCoderLlavaLlamaForCausalLM),
                                        ("llava", "LlavaHFConfig", lm.LlavaLlamaForCausalLM)):
        if model_type == "llava" and model_type in CONFIG_MAPPING:
            continue   # Transformers' own Llava: not ours to replace
        cfg_cls = make(model_type, name)
        kw = _kw(AutoConfig.register)
        if not kw and model_type in CONFIG_MAPPING:
            if not exist_ok:
                raise ValueError(f"'{model_type}' is already used by a Transformers config")
            continue   # pre-4.32: no way to replace an entry; whoever registered it first keeps it
        AutoConfig.register(model_type, cfg_cls, **kw)
        model_cls.config_class = cfg_cls
        AutoModelForCausalLM.register(cfg_cls, model_cls, **_kw(AutoModelForCausalLM.register))
        _registered[model_type] = (cfg_cls, model_cls)
    return dict(_registered)
