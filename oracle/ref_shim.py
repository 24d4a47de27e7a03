"""TEST INFRASTRUCTURE ONLY — imports the SHI-Labs/VCoder reference (read-only, /root/reference) on CPU.

Only usable in the build container (the GPU box has no /root/reference).  Used by
oracle/gen_golden.py to produce the committed fixtures in tests/golden/ and by the
`-m "not gpu"` tests that pin oracle/cpu_ref.py against the live reference when present.

The reference registers model_type "llava" with transformers.AutoConfig
(vcoder_llava/model/language_model/llava_llama.py:139), which modern Transformers already
ships; the wrappers below pass exist_ok=True from the harness side so the reference tree is
imported unmodified.
"""
import os
import sys

REFERENCE_ROOT = os.environ.get("VCODER_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "vcoder_llava"))


_loaded = None


def load_reference():
    """Returns the imported `vcoder_llava.model` module of the reference."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from transformers import AutoConfig, AutoModelForCausalLM

    _cfg_register = AutoConfig.register
    AutoConfig.register = staticmethod(lambda mt, cfg, exist_ok=False: _cfg_register(mt, cfg, exist_ok=True))
    _mdl_register = AutoModelForCausalLM.register.__func__
    AutoModelForCausalLM.register = classmethod(
        lambda cls, c, m, exist_ok=False: _mdl_register(cls, c, m, exist_ok=True))
    import vcoder_llava.model as ref_model  # noqa: E402

    _loaded = ref_model
    return ref_model
