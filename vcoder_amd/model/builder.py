"""load_pretrained_model — drop-in for vcoder_llava/model/builder.py:25-154 (inference paths).

Same signature and 6-tuple; same name-substring dispatch ('vcoder_ds_llava' / 'vcoder_llava' / else llava,
builder.py:93-108), same processor aliasing (:145-151) and context_len rule (:133-136).  `load_8bit=True` (the
reference: bitsandbytes LLM.int8, builder.py:31-33) selects this build's 8-bit weight format instead: W8A16, decoder
linears as fp8-e4m3 with per-row power-of-two scales (vcoder_amd/quant.py).  `model_base` selects the reference's two
overlay paths (builder.py:42-92): LoRA adapters are merged on the host (W + alpha / r * B A, what peft's merge_and_unload
computes), projector-only checkpoints overlay `mm_projector.bin`.  4-bit (NF4) raises."""
from __future__ import annotations

from .language_model import LlavaLlamaForCausalLM, VCoderDSLlavaLlamaForCausalLM, VCoderLlavaLlamaForCausalLM


def load_pretrained_model(model_path, model_base, model_name, load_8bit=False, load_4bit=False, device_map="auto",
                          device="cuda", weight_format=None, operands="bf16"):
    """Signature and return value of the reference's loader (builder.py:25,154).  load_8bit=True selects the 8-bit
    weight format of this build, 'w8a16' (e4m3 decoder weights, bf16 activations); `weight_format` (an addition, keyword
    only in practice) names a format directly: 'bf16' | 'w8a16' | 'fp8' (the latter also runs the prefill linears on
    e4m3 activations — BASELINE configs[4]; faster, and noisier: DESIGN.md section 4.2b).  `operands` (an addition): 'bf16' — the
    benchmarked library; 'fp16' — libvcoder_hip_f16.so, IEEE fp16 MFMA operands, what the reference's torch_dtype=torch.float16
    (builder.py:39) and its fp16 tower (:142) compute with: an fp16 checkpoint is held exactly."""
    if load_4bit:
        raise NotImplementedError("bitsandbytes NF4 loading is a CUDA-only path of the reference; the MI355X build runs "
                                  "bf16 weights or, with load_8bit=True, fp8-e4m3 decoder weights (W8A16)")
    name = model_name.lower()
    if "llava" not in name:
        raise ValueError(f"'{model_name}': only LLaVA-family checkpoints (llava / vcoder_llava / vcoder_ds_llava) are "
                         "on the hot path")
    if "vcoder_it" in name:
        # the reference dispatches these names to VCoderITLlavaLlamaForCausalLM (builder.py:93), an unreleased variant that
        # is not on the hot path — refuse rather than load it as a plain LLaVA
        raise NotImplementedError("vcoder_it_llava checkpoints (VCoderITLlavaLlamaForCausalLM) are outside the MI355X hot path")
    fmt = weight_format or ("w8a16" if load_8bit else "bf16")
    if model_base is not None:
        # builder.py:42-92: a LoRA checkpoint ('lora' in the name: base LLM + non_lora_trainables.bin + the adapter, merged) or
        # a projector-only checkpoint (base LLM + mm_projector.bin).  Both build a plain LlavaLlamaForCausalLM in the
        # reference, whatever else the name says; the tokenizer comes from the base.
        from ..config import VCoderConfig
        from .. import checkpoint

        tokenizer = _load_tokenizer(model_base)
        cfg = VCoderConfig.from_pretrained(model_path, "llava")
        cfg.variant = "llava"
        tensors = (checkpoint.iter_lora_merged(model_base, model_path) if "lora" in name
                   else checkpoint.iter_base_with_projector(model_base, model_path))
        model = LlavaLlamaForCausalLM.from_tensors(cfg, tensors, device=device, weight_format=fmt, operands=operands)
    else:
        if "lora" in name:
            import warnings

            warnings.warn("There is `lora` in model name but no `model_base` is provided. If you are loading a LoRA model, please "
                          "provide the `model_base` argument.")   # builder.py:41-42: the reference warns and loads it as a full model
        tokenizer = _load_tokenizer(model_path)
        if "vcoder_ds_llava" in name:
            cls = VCoderDSLlavaLlamaForCausalLM
        elif "vcoder_llava" in name:
            cls = VCoderLlavaLlamaForCausalLM
        else:
            cls = LlavaLlamaForCausalLM
        model = cls.from_pretrained(model_path, low_cpu_mem_usage=True, device=device, weight_format=fmt, operands=operands)
    context_len = model.config.max_sequence_length if getattr(model.config, "max_sequence_length", None) else 2048
    vision_tower = model.get_vision_tower()
    if not vision_tower.is_loaded or vision_tower.image_processor is None:
        vision_tower.load_model()
    image_processor = vision_tower.image_processor
    seg_image_processor = image_processor if "vcoder" in name else None
    depth_image_processor = image_processor if "ds" in name else None
    model.requires_grad_(False)
    return tokenizer, model, image_processor, seg_image_processor, depth_image_processor, context_len


def _load_tokenizer(model_path):
    try:
        from transformers import AutoTokenizer

        return AutoTokenizer.from_pretrained(model_path, use_fast=False)
    except Exception as e:  # tokenizer files are not part of the tensor hot path; surface a clear error lazily
        return _MissingTokenizer(model_path, e)


class _MissingTokenizer:
    def __init__(self, path, err):
        self._path, self._err = path, err

    def __getattr__(self, name):
        raise RuntimeError(f"no tokenizer could be loaded from {self._path}: {self._err}")

    def __call__(self, *a, **k):
        raise RuntimeError(f"no tokenizer could be loaded from {self._path}: {self._err}")
