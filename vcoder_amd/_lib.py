"""ctypes binding of libvcoder_hip.so (include/vcoder_hip.h + include/vcoder_kernels.h).

The product has exactly one compute path: the HIP library.  There is no CPU fallback — if the
library is missing or there is no GPU, loading fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# VCODER_HIP_LIB: another build of the same library (kernel A/B experiments, tools/experiments/); default = the in-tree build
LIB_PATH = os.environ.get("VCODER_HIP_LIB") or os.path.join(HERE, "lib", "libvcoder_hip.so")

VC_OK, VC_IGNORED = 0, 1
VC_ERR_INVALID, VC_ERR_HIP, VC_ERR_STATE, VC_ERR_INDEX, VC_ERR_UNEQUAL = -1, -2, -3, -4, -5
VC_F32, VC_BF16 = 0, 1
VARIANTS = {"llava": 0, "vcoder": 1, "vcoder_ds": 2}
MODALITY = {"img": 0, "seg": 1, "depth": 2}


class ModelCfg(C.Structure):
    _fields_ = [("variant", C.c_int32),
                ("vit_hidden", C.c_int32), ("vit_heads", C.c_int32), ("vit_ffn", C.c_int32), ("vit_layers", C.c_int32),
                ("vit_layers_used", C.c_int32), ("vit_image", C.c_int32), ("vit_patch", C.c_int32),
                ("vit_keep_cls", C.c_int32), ("vit_ln_eps", C.c_float),
                ("hidden", C.c_int32), ("heads", C.c_int32), ("ffn", C.c_int32), ("layers", C.c_int32),
                ("vocab", C.c_int32), ("max_positions", C.c_int32), ("rms_eps", C.c_float), ("rope_theta", C.c_float),
                ("mm_proj_depth", C.c_int32), ("seg_proj_depth", C.c_int32), ("pad_token_id", C.c_int32)]


class Sampling(C.Structure):
    _fields_ = [("do_sample", C.c_int32), ("temperature", C.c_float), ("top_k", C.c_int32), ("top_p", C.c_float),
                ("seed", C.c_uint64)]


TOKEN_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32))


def declare(lib: C.CDLL) -> C.CDLL:
    """Attach argtypes/restypes of the model-level ABI (the vck_* kernel entry points are called with
    explicit ctypes values by the tests)."""
    vp, i32, f32p = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    i64p, i32p = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    lib.vc_init.argtypes = [i32, C.POINTER(vp)]
    lib.vc_shutdown.argtypes = [vp]
    lib.vc_shutdown.restype = None
    lib.vc_last_error.argtypes = [vp]
    lib.vc_last_error.restype = C.c_char_p
    lib.vc_synchronize.argtypes = [vp]
    lib.vc_stream.argtypes = [vp]
    lib.vc_stream.restype = vp
    lib.vc_model_create.argtypes = [vp, C.POINTER(ModelCfg), C.POINTER(vp)]
    lib.vc_model_create_shared.argtypes = [vp, vp, C.POINTER(vp)]
    lib.vc_model_create_shared.restype = C.c_int
    lib.vc_model_destroy.argtypes = [vp]
    lib.vc_model_destroy.restype = None
    lib.vc_model_load_tensor.argtypes = [vp, C.c_char_p, vp, i32, i64p, i32]
    lib.vc_model_synth_tensor.argtypes = [vp, C.c_char_p, i64p, i32, C.c_uint32, C.c_float, C.c_float]
    lib.vc_model_synth_tensor_rounded.argtypes = [vp, C.c_char_p, i64p, i32, C.c_uint32, C.c_float, C.c_float, i32]
    lib.vc_model_synth_tensor_rounded.restype = C.c_int
    lib.vc_model_finalize.argtypes = [vp]
    lib.vc_model_set_precision.argtypes = [vp, i32]
    lib.vc_model_set_precision.restype = C.c_int
    lib.vc_model_set_weight_format.argtypes = [vp, i32]
    lib.vc_model_set_weight_format.restype = C.c_int
    lib.vc_encode.argtypes = [vp, i32, vp, i32, i32, vp]
    lib.vc_prefill.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32, i32, vp, vp, C.POINTER(C.c_int)]
    lib.vc_prefill_embeds_only.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32, i32, vp, C.POINTER(C.c_int)]
    lib.vc_plan_spliced_len.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32, i32, C.POINTER(C.c_int)]
    lib.vc_decode_step.argtypes = [vp, vp, vp, vp]
    lib.vc_generate_greedy.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp, C.POINTER(C.c_int)]
    lib.vc_generate_greedy_stop.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, vp,
                                            C.POINTER(C.c_int)]
    lib.vc_generate_greedy_stop.restype = C.c_int
    lib.vc_generate.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, C.POINTER(Sampling), TOKEN_CB,
                                vp, i32, vp, C.POINTER(C.c_int)]
    lib.vc_generate.restype = C.c_int
    lib.vc_vision_tower_forward.argtypes = [vp, vp, i32, i32, vp]
    lib.vc_vision_tower_forward.restype = C.c_int
    lib.vc_set_image_counts.argtypes = [vp, vp, vp, vp, i32]
    lib.vc_set_image_counts.restype = C.c_int
    lib.vc_model_reserve_decode.argtypes = [vp, i32]
    lib.vc_model_reserve_decode.restype = C.c_int
    lib.vc_comm_unique_id.argtypes = [vp, vp]
    lib.vc_comm_unique_id.restype = C.c_int
    lib.vc_comm_create.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
    lib.vc_comm_create.restype = C.c_int
    lib.vc_allgather_tokens.argtypes = [vp, vp, i32, vp]
    lib.vc_allgather_tokens.restype = C.c_int
    lib.vc_comm_uses_rccl.argtypes = [vp]
    lib.vc_comm_uses_rccl.restype = C.c_int
    lib.vc_comm_destroy.argtypes = [vp]
    lib.vc_comm_destroy.restype = None
    lib.vc_model_set_layer_limit.argtypes = [vp, i32]
    lib.vc_model_set_layer_limit.restype = C.c_int
    lib.vc_debug_prefill_layers.argtypes = [vp, i32, i32, vp, i32, i32, vp]
    lib.vc_debug_prefill_layers.restype = C.c_int
    lib.vc_reorder_cache.argtypes = [vp, vp, i32]
    lib.vc_reorder_cache.restype = C.c_int
    lib.vc_request_attentions.argtypes = [vp, vp, C.c_size_t]
    lib.vc_request_attentions.restype = C.c_int
    lib.vc_request_hidden_states.argtypes = [vp, vp, C.c_size_t]
    lib.vc_request_hidden_states.restype = C.c_int
    lib.vc_set_attention_mask.argtypes = [vp, vp, i32, i32]
    lib.vc_set_attention_mask.restype = C.c_int
    lib.vc_clear_attention_mask.argtypes = [vp]
    lib.vc_clear_attention_mask.restype = C.c_int
    lib.vc_last_spliced_len.argtypes = [vp]
    lib.vc_last_spliced_len.restype = C.c_int
    lib.vc_profile_decode_gemv.argtypes = [vp, i32, i32, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.vc_profile_decode_attention.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.vc_profile_decode_attention.restype = C.c_int
    lib.vc_pool_step_counts.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    lib.vc_pool_step_counts.restype = C.c_int
    lib.vc_model_set_batch_invariant.argtypes = [vp, i32]
    lib.vc_model_set_batch_invariant.restype = C.c_int
    lib.vc_pool_set_rows.argtypes = [vp, i32]
    lib.vc_pool_set_rows.restype = C.c_int
    lib.vc_model_set_qkv_fused.argtypes = [vp, i32]
    lib.vc_model_set_qkv_fused.restype = C.c_int
    lib.vc_model_set_fp8_kv.argtypes = [vp, i32]
    lib.vc_model_set_fp8_kv.restype = C.c_int
    lib.vc_model_inexact_tensors.argtypes = [vp]
    lib.vc_model_inexact_tensors.restype = C.c_int
    lib.vc_pool_set_hold.argtypes = [vp, i32]
    lib.vc_pool_set_hold.restype = C.c_int
    lib.vc_pool_profile.argtypes = [vp, i32]
    lib.vc_pool_profile.restype = C.c_int
    lib.vc_pool_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_ulonglong), i32]
    lib.vc_pool_profile_read.restype = C.c_int
    lib.vc_last_timings.argtypes = [vp, f32p, f32p, f32p]
    lib.vc_preprocess_image.argtypes = [vp, vp, i32, i32, i32, f32p, f32p, vp, i32]
    lib.vc_preprocess_image.restype = C.c_int
    for name in ("vc_init", "vc_synchronize", "vc_model_create", "vc_model_load_tensor", "vc_model_synth_tensor",
                 "vc_model_finalize", "vc_encode", "vc_prefill", "vc_prefill_embeds_only", "vc_plan_spliced_len", "vc_decode_step",
                 "vc_generate_greedy", "vc_profile_decode_gemv", "vc_last_timings"):
        getattr(lib, name).restype = C.c_int
    return lib


_lib = None
_lib_f16 = None
LIB_PATH_F16 = os.environ.get("VCODER_HIP_LIB_F16") or os.path.join(HERE, "lib", "libvcoder_hip_f16.so")


def load(operands: str = "bf16") -> C.CDLL:
    """operands="bf16": libvcoder_hip.so (the benchmarked path); "fp16": libvcoder_hip_f16.so — the same kernels and C ABI with IEEE
    fp16 MFMA operands (include/vcoder_hip.h vc_operand_format), the precision of the reference's own GPU path.  No fallback of one
    to the other, and none to a CPU path: a missing library is an error."""
    global _lib, _lib_f16
    if operands not in ("bf16", "fp16"):
        raise ValueError("operands: 'bf16' or 'fp16'")
    f16 = operands == "fp16"
    cached = _lib_f16 if f16 else _lib
    if cached is not None:
        return cached
    path = LIB_PATH_F16 if f16 else LIB_PATH
    if not os.path.exists(path):
        try:  # not a fallback: the same HIP library, compiled now (hipcc, gfx950)
            from . import build as _build

            _build.build(verbose=False, operands=operands)
        except Exception as e:
            raise RuntimeError(
                f"{path} is missing and could not be built ({e}): run `python -m vcoder_amd.build{' --fp16' if f16 else ''}` "
                "(hipcc, gfx950).  vcoder_amd has no CPU fallback.") from e
    lib = declare(C.CDLL(path))
    lib.vc_operand_format.restype = C.c_int
    if lib.vc_operand_format() != (1 if f16 else 0):
        raise RuntimeError(f"{path} was built for the other operand format")
    if f16:
        _lib_f16 = lib
    else:
        _lib = lib
    return lib
