"""Thin Python owner of a `vc_model` (include/vcoder_hip.h).  All arithmetic happens in libvcoder_hip.so;
this file only marshals pointers, shapes and errors."""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Dict, Optional, Tuple

import numpy as np

from . import _lib, synth
from .config import VCoderConfig

_ERRORS = {_lib.VC_ERR_INVALID: ValueError, _lib.VC_ERR_HIP: RuntimeError, _lib.VC_ERR_STATE: RuntimeError,
           _lib.VC_ERR_INDEX: IndexError, _lib.VC_ERR_UNEQUAL: UnboundLocalError}


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class HipEngine:
    """One model replica on one GPU (one HIP stream).  `lib` is injectable ONLY for the CPU emulator tests;
    product code always goes through `_lib.load()`, which has no fallback."""

    MAX_BATCH = 16   # sequences a replica prefills / decodes at a time (csrc/engine.hip)

    def __init__(self, cfg: VCoderConfig, device_index: int = 0, lib: Optional[C.CDLL] = None, _parent=None, operands: str = "bf16"):
        """operands: "bf16" — the benchmarked library; "fp16" — libvcoder_hip_f16.so, the same engine and kernels with IEEE fp16 MFMA
        operands (the precision of the reference's own GPU path, vcoder_llava/model/builder.py:39,142; same throughput)."""
        cfg.validate()
        self.cfg = cfg
        self.operands = _parent.operands if _parent is not None else operands
        self.lib = lib if lib is not None else _lib.load(self.operands)
        self._lib_arg = lib
        if lib is not None:
            _lib.declare(self.lib)
        self.device_index = device_index
        self._ctx = C.c_void_p()
        self._model = C.c_void_p()
        self._parent = _parent
        self.last_S = 0
        rc = self.lib.vc_init(device_index, C.byref(self._ctx))
        if rc != 0 or not self._ctx:
            raise RuntimeError(f"vc_init(device {device_index}) failed with status {rc}: no usable HIP device")
        if _parent is not None:  # a session sharing the parent's weights (own stream, KV cache, workspaces, graph)
            self._check(self.lib.vc_model_create_shared(self._ctx, _parent._model, C.byref(self._model)))
            self.finalized = True
            return
        c = self._model_cfg(cfg)
        self._check(self.lib.vc_model_create(self._ctx, C.byref(c), C.byref(self._model)))
        self.finalized = False
        self.last_S = 0

    @staticmethod
    def _model_cfg(cfg: VCoderConfig) -> "_lib.ModelCfg":
        """the vc_model_cfg of a configuration (include/vcoder_hip.h)"""
        return _lib.ModelCfg(
            variant=_lib.VARIANTS[cfg.variant], vit_hidden=cfg.mm_hidden_size, vit_heads=cfg.vit_num_heads,
            vit_ffn=cfg.vit_intermediate_size, vit_layers=cfg.vit_num_layers, vit_layers_used=cfg.vit_layers_used,
            vit_image=cfg.vit_image_size, vit_patch=cfg.vit_patch_size,
            vit_keep_cls=int(cfg.mm_vision_select_feature == "cls_patch"), vit_ln_eps=cfg.vit_layer_norm_eps,
            hidden=cfg.hidden_size, heads=cfg.num_attention_heads, ffn=cfg.intermediate_size,
            layers=cfg.num_hidden_layers, vocab=cfg.vocab_size,
            max_positions=min(int(cfg.max_position_embeddings), 4096), rms_eps=cfg.rms_norm_eps,
            rope_theta=cfg.rope_theta, mm_proj_depth=synth.projector_depth(cfg.mm_projector_type),
            seg_proj_depth=synth.projector_depth(cfg.seg_mm_projector_type) if cfg.variant != "llava" else 0,
            pad_token_id=int(cfg.pad_token_id or 0))

    def fork(self) -> "HipEngine":
        """A new session on the same (finalized) weights: own HIP stream, KV cache and hipGraph.  Sessions can be
        driven from different host threads concurrently (ctypes releases the GIL during the C calls)."""
        if not self.finalized:
            raise RuntimeError("fork() needs a finalized engine")
        root = self._parent if self._parent is not None else self
        return HipEngine(self.cfg, self.device_index, lib=self._lib_arg, _parent=root)

    # ---- plumbing -----------------------------------------------------------------------------------
    def _check(self, rc: int) -> int:
        if rc < 0:
            msg = self.lib.vc_last_error(self._ctx)
            msg = msg.decode("utf-8", "replace") if msg else f"status {rc}"
            raise _ERRORS.get(rc, RuntimeError)(msg)
        return rc

    def close(self):
        if getattr(self, "_model", None):
            self.lib.vc_model_destroy(self._model)
            self._model = C.c_void_p()
        if getattr(self, "_ctx", None):
            self.lib.vc_shutdown(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream_ptr(self) -> int:
        return int(self.lib.vc_stream(self._ctx) or 0)

    def synchronize(self):
        self._check(self.lib.vc_synchronize(self._ctx))

    # ---- weights ------------------------------------------------------------------------------------
    def load_tensor(self, key: str, value) -> bool:
        """Returns False when the key is accepted but dead at inference (depth_mm_projector, mm2_projector,
        vcoder_lm_emb, unused CLIP layers)."""
        if _is_torch(value):
            import torch

            t = value.detach()
            if t.dtype == torch.bfloat16:
                arr = t.contiguous().cpu().view(torch.int16).numpy()
                dt = _lib.VC_BF16
            else:
                arr = t.to(torch.float32).contiguous().cpu().numpy()
                dt = _lib.VC_F32
        else:
            arr = np.ascontiguousarray(value, dtype=np.float32)
            dt = _lib.VC_F32
        shape = (C.c_int64 * max(arr.ndim, 1))(*(arr.shape if arr.ndim else (1,)))
        rc = self._check(self.lib.vc_model_load_tensor(self._model, key.encode(), arr.ctypes.data_as(C.c_void_p), dt,
                                                       shape, max(arr.ndim, 1)))
        return rc == _lib.VC_OK

    def load_state_dict(self, sd: Dict[str, object]) -> Tuple[int, int]:
        used = dead = 0
        for k, v in sd.items():
            if self.load_tensor(k, v):
                used += 1
            else:
                dead += 1
        return used, dead

    def load_synthetic(self, seed: int = 42, dtypes: str = "bf16"):
        """Seeded synthetic checkpoint generated ON THE DEVICE, bit-identical to synth.synth_state_dict(cfg, seed, dtypes=dtypes):
        "bf16" (default), or "reference" — fp16-valued LLM / projector tensors and an fp32-valued CLIP tower, the value classes of
        the reference's own checkpoints, which keep weight lo planes (inexact_tensors() > 0)."""
        for key, shape, off, hw in synth.tensor_specs(self.cfg):
            sh = (C.c_int64 * len(shape))(*shape)
            rounding = 0 if dtypes == "bf16" else synth.ROUNDING_CODE[synth.reference_rounding(key, dtypes)]
            self._check(self.lib.vc_model_synth_tensor_rounded(self._model, key.encode(), sh, len(shape),
                                                               C.c_uint32(synth.tensor_seed(key, seed)), C.c_float(off),
                                                               C.c_float(hw), rounding))

    def set_precision(self, mode: str):
        """'bf16' (default: bf16 MFMA operands, what the benchmark runs); 'strict' (fp32 activations on fp32 MFMA —
        fp32-faithful to the reference's CPU path, slow); 'split' (fp32 activations in HBM, every MFMA operand as bf16 hi + lo
        fragments on the FAST kernels: the same 1e-3 / bit-exact-ids bar at about half the fast path's MFMA rate)."""
        self._check(self.lib.vc_model_set_precision(self._model, {"bf16": 0, "fast": 0, "strict": 1, "fp32": 1, "split": 2}[mode]))

    def set_weight_format(self, fmt: str):
        """'bf16' (default); 'w8a16': decoder linears as e4m3 + per-row power-of-two scales, quantised at finalize and
        streamed as bytes by the decode steps, bf16 activations everywhere; 'fp8' (BASELINE configs[4]): the same weights,
        and the prefill's decoder linears also quantise their activation rows to e4m3 and run on the K=128 scaled fp8 MFMA
        (W8A8).  vcoder_amd/quant.py is the host restatement of both quantisers.  Before finalize()."""
        code = {"bf16": 0, "w8a16": 1, "fp8": 2}[fmt]
        self._check(self.lib.vc_model_set_weight_format(self._model, code))
        self.weight_format = fmt

    def set_batch_invariant(self, on: bool = True):
        """a sample's bits independent of the batch / shard it runs in (vc_model_set_batch_invariant): N ranks x B == one rank x N B"""
        self._check(self.lib.vc_model_set_batch_invariant(self._model, 1 if on else 0))

    def set_qkv_fused(self, on=True):
        """RoPE + head split + KV write in the prefill's QKV GEMM epilogue (default on from 1024 token rows); False / 0 = the separate
        launches (A/B, regression), 2 = the fused form for every problem size"""
        self._check(self.lib.vc_model_set_qkv_fused(self._model, int(on)))

    def set_fp8_kv(self, on: bool):
        """'fp8' weight format: e4m3 KV cache for the decode steps (default) or bf16 rows.  Before finalize()."""
        self._check(self.lib.vc_model_set_fp8_kv(self._model, 1 if on else 0))
        self.fp8_kv = bool(on)

    def set_layer_limit(self, n_layers: int):
        """parity diagnostic: prefills evaluate only the first n decoder layers (0 = all)"""
        self._check(self.lib.vc_model_set_layer_limit(self._model, int(n_layers)))

    def run_layers(self, l0: int, l1: int, x: np.ndarray) -> np.ndarray:
        """parity diagnostic: decoder layers [l0, l1) of a prefill applied to the residual stream x [B, S, hidden] (fp32)"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        B, S, D = x.shape
        out = np.empty_like(x)
        self._check(self.lib.vc_debug_prefill_layers(self._model, int(l0), int(l1), x.ctypes.data_as(C.c_void_p), B, S,
                                                     out.ctypes.data_as(C.c_void_p)))
        return out

    def inexact_tensors(self) -> int:
        """loaded tensors whose values bf16 cannot hold (an fp16 / fp32 checkpoint): each keeps a bf16 lo plane that the 'strict' and
        'split' precision modes add back (w = hi + lo); the bf16 fast path runs on the rounded weights (vc_model_inexact_tensors)"""
        return self._check(self.lib.vc_model_inexact_tensors(self._model))

    def finalize(self):
        self._check(self.lib.vc_model_finalize(self._model))
        self.finalized = True

    # ---- inputs -------------------------------------------------------------------------------------
    def _pixels(self, *arrs):
        """-> (ctypes pointers, on_device flag, keep-alive list).  All given arrays share one residency."""
        present = [a for a in arrs if a is not None]
        on_dev = bool(present) and all(_is_torch(a) and a.is_cuda for a in present)
        keep, ptrs = [], []
        for a in arrs:
            if a is None:
                ptrs.append(None)
                continue
            if on_dev:
                import torch

                t = a.to(torch.float32).contiguous()
                keep.append(t)
                ptrs.append(C.c_void_p(t.data_ptr()))
                # the engine's stream is non-blocking: make whatever produced the tensor on torch's stream visible
                torch.cuda.current_stream(t.device).synchronize()
            else:
                if _is_torch(a):
                    a = a.detach().float().cpu().numpy()
                n = np.ascontiguousarray(a, dtype=np.float32)
                keep.append(n)
                ptrs.append(n.ctypes.data_as(C.c_void_p))
        return ptrs, int(on_dev), keep

    def _check_pixels(self, a, B):
        if a is None:
            return
        s = self.cfg.vit_image_size
        if tuple(a.shape) != (B, 3, s, s):
            raise ValueError(f"expected pixel tensor [{B},3,{s},{s}], got {tuple(a.shape)}")

    def _image_block(self, a, B):
        """One modality of a batch -> (pixels [N,3,S,S], counts or None).  Accepts the reference's three input forms
        (vcoder_ds_llava_arch.py:135-169): a 4-D tensor [B,3,S,S]; a list of B tensors [n_b,3,S,S]; a 5-D tensor
        [B,n,3,S,S] — in the last two, sample b owns n_b images whose features are spliced as one block."""
        if a is None:
            return None, None
        if not isinstance(a, (list, tuple)) and a.ndim != 5:
            self._check_pixels(a, B)
            return a, None
        items = list(a)
        if len(items) != B:
            raise ValueError(f"expected {B} per-sample image groups, got {len(items)}")
        s = self.cfg.vit_image_size
        for it in items:
            if it.ndim != 4 or tuple(it.shape[1:]) != (3, s, s):
                raise ValueError(f"expected per-sample image groups [n,3,{s},{s}], got {tuple(it.shape)}")
        counts = [int(it.shape[0]) for it in items]
        if _is_torch(items[0]):
            import torch

            cat = torch.cat(items, dim=0)
        else:
            cat = np.concatenate([np.asarray(it) for it in items], axis=0)
        return cat, (None if all(c == 1 for c in counts) else counts)

    def _image_blocks(self, B, images, segs, depths):
        """-> ([img, seg, depth] concatenated pixel blocks, keep-alive counts arrays); announces per-sample image counts
        to the library (one-shot, consumed by the next prefill / generate call)."""
        blocks, counts = zip(*(self._image_block(a, B) for a in (images, segs, depths)))
        # images one placeholder can expand to: the largest per-sample image group of any modality (1 in the 4-D form)
        self._max_images_per_block = max([1] + [max(c) for c in counts if c is not None])
        if any(c is not None for c in counts):
            arrs = [None if blk is None else np.ascontiguousarray(c if c is not None else [1] * B, dtype=np.int32)
                    for blk, c in zip(blocks, counts)]
            self._check(self.lib.vc_set_image_counts(self._model, *(None if a is None else a.ctypes.data_as(C.c_void_p)
                                                                    for a in arrs), B))
        return list(blocks)

    def _announce_mask(self, attention_mask, B: int, T: int) -> bool:
        """hands a caller's 2-D attention_mask to the library for the next prefill / generate call (one-shot) when it hides
        positions; -> whether a mask was given at all (the reference's behaviour for unequal spliced lengths depends on it)"""
        if attention_mask is None:
            return False
        m = attention_mask.detach().cpu().numpy() if _is_torch(attention_mask) else np.asarray(attention_mask)
        if m.shape != (B, T):
            raise ValueError(f"attention_mask must be [{B},{T}], got {tuple(m.shape)}")
        if not bool(np.all(m != 0)):
            mk = np.ascontiguousarray(m != 0, dtype=np.uint8)
            self._check(self.lib.vc_set_attention_mask(self._model, mk.ctypes.data_as(C.c_void_p), B, T))
        return True

    def reorder_cache(self, beam_idx):
        """KV rows of the current prefill / decode_step loop permuted: row r <- old row beam_idx[r] (beam search)"""
        idx = np.ascontiguousarray(np.asarray(beam_idx).reshape(-1), dtype=np.int32)
        self._check(self.lib.vc_reorder_cache(self._model, idx.ctypes.data_as(C.c_void_p), int(idx.shape[0])))

    def clear_attention_mask(self):
        """the cached decode steps behind the current prefill see every key again (the reference's multimodal decode path)"""
        self._check(self.lib.vc_clear_attention_mask(self._model))

    @staticmethod
    def _ids(input_ids) -> np.ndarray:
        if _is_torch(input_ids):
            input_ids = input_ids.detach().cpu().numpy()
        ids = np.ascontiguousarray(input_ids, dtype=np.int64)
        if ids.ndim != 2:
            raise ValueError("input_ids must be [B,T]")
        return ids

    # ---- hot path -----------------------------------------------------------------------------------
    def encode(self, pixels, modality: str = "img") -> np.ndarray:
        B = int(pixels.shape[0])
        self._check_pixels(pixels, B)
        (p,), on_dev, keep = self._pixels(pixels)
        rows = self.cfg.num_patches + (1 if self.cfg.mm_vision_select_feature == "cls_patch" else 0)
        out = np.empty((B, rows, self.cfg.hidden_size), dtype=np.float32)
        self._check(self.lib.vc_encode(self._model, _lib.MODALITY[modality], p, on_dev, B,
                                       out.ctypes.data_as(C.c_void_p)))
        return out

    def inputs_embeds(self, input_ids, images, segs=None, depths=None, has_attention_mask: bool = False) -> np.ndarray:
        ids = self._ids(input_ids)
        B, T = ids.shape
        (pi, ps, pd), on_dev, keep = self._pixels(*self._image_blocks(B, images, segs, depths))
        S = C.c_int(0)
        # first call sizes the output; lengths are only known after the splice plan, so run twice is avoided by
        # allocating for the worst case: every placeholder expands to a feature block
        rows = self.cfg.num_patches + (1 if self.cfg.mm_vision_select_feature == "cls_patch" else 0)
        worst = T + rows * self._max_feature_blocks(ids)
        out = np.empty((B * worst * self.cfg.hidden_size,), dtype=np.float32)
        self._check(self.lib.vc_prefill_embeds_only(self._model, ids.ctypes.data_as(C.c_void_p), B, T, pi, ps, pd, on_dev,
                                                    int(has_attention_mask), out.ctypes.data_as(C.c_void_p), C.byref(S)))
        self.last_S = S.value
        return out[: B * S.value * self.cfg.hidden_size].reshape(B, S.value, self.cfg.hidden_size).copy()

    def planned_len(self, input_ids, images, segs=None, depths=None, has_attention_mask: bool = False) -> int:
        """spliced length S of a prefill / generate call with these arguments — the splice plan alone (vc_plan_spliced_len):
        no tower pass, no KV / mask state touched; raises what the real call's plan would raise"""
        ids = self._ids(input_ids)
        B, T = ids.shape
        (pi, ps, pd), on_dev, keep = self._pixels(*self._image_blocks(B, images, segs, depths))
        S = C.c_int(0)
        self._check(self.lib.vc_plan_spliced_len(self._model, ids.ctypes.data_as(C.c_void_p), B, T, pi, ps, pd, on_dev,
                                                 int(has_attention_mask), C.byref(S)))
        return S.value

    def prefill(self, input_ids, images, segs=None, depths=None, has_attention_mask: bool = False,
                all_logits: bool = False, reserve: Optional[int] = None, attention_mask=None, hidden_states: bool = False,
                attentions: bool = False):
        """-> (logits_last [B,V], logits_all [B,S,V] or None, S).  reserve: decode_step calls the caller intends to make
        (sizes the KV cache up front; a longer loop still works — the cache grows).  attention_mask [B,T]: padded batches —
        hidden positions are hidden as keys in this prefill and in the decode_step loop behind it (clear_attention_mask() ends
        that)."""
        ids = self._ids(input_ids)
        B, T = ids.shape
        # output_attentions is sized exactly: the spliced length comes from the splice plan alone (no tower pass), asked with
        # the mask-or-not of the real call so that an unequal-length batch takes the same error path in both
        S_att = self.planned_len(ids, images, segs, depths, has_attention_mask or attention_mask is not None) if attentions else 0
        if reserve is not None:
            self._check(self.lib.vc_model_reserve_decode(self._model, int(reserve)))
        (pi, ps, pd), on_dev, keep = self._pixels(*self._image_blocks(B, images, segs, depths))
        # the mask is announced LAST: anything above may raise, and an armed one-shot mask would hit the next call
        has_attention_mask = self._announce_mask(attention_mask, B, T) or has_attention_mask
        V = self.cfg.vocab_size
        last = np.empty((B, V), dtype=np.float32)
        S = C.c_int(0)
        self.last_hidden_states = self.last_attentions = None
        hid = att = None
        rows = self.cfg.num_patches + (1 if self.cfg.mm_vision_select_feature == "cls_patch" else 0)
        worst = T + rows * self._max_feature_blocks(ids)
        full = np.empty((B * worst * V,), dtype=np.float32) if all_logits else None
        try:
            if attentions:      # [L, B, H, S, S]
                att = np.empty((self.cfg.num_hidden_layers, B, self.cfg.num_attention_heads, S_att, S_att), dtype=np.float32)
                self._check(self.lib.vc_request_attentions(self._model, att.ctypes.data_as(C.c_void_p), C.c_size_t(att.size)))
            if hidden_states:   # [(L + 1), B, S, D] for the worst-case S; trimmed below
                hid = np.empty(((self.cfg.num_hidden_layers + 1) * B * worst * self.cfg.hidden_size,), dtype=np.float32)
                self._check(self.lib.vc_request_hidden_states(self._model, hid.ctypes.data_as(C.c_void_p), C.c_size_t(hid.size)))
            self._check(self.lib.vc_prefill(self._model, ids.ctypes.data_as(C.c_void_p), B, T, pi, ps, pd, on_dev,
                                            int(has_attention_mask), last.ctypes.data_as(C.c_void_p),
                                            full.ctypes.data_as(C.c_void_p) if all_logits else None, C.byref(S)))
        except BaseException:
            # the library forgets its one-shot requests when vc_prefill returns (error or not); a failure BEFORE that call must
            # not leave it holding pointers into buffers that are about to be freed — or an armed mask
            self.lib.vc_request_attentions(self._model, None, C.c_size_t(0))
            self.lib.vc_request_hidden_states(self._model, None, C.c_size_t(0))
            self.lib.vc_clear_attention_mask(self._model)
            raise
        self.last_S = self._step_pos = S.value
        self._cur_batch = B
        self._keep_hidden(hid, B, S.value)
        self.last_attentions = att
        return last, (full[: B * S.value * V].reshape(B, S.value, V).copy() if all_logits else None), S.value

    last_attentions = None

    def _keep_hidden(self, hid, B, S):
        """hidden states of the last prefill(hidden_states=True): [(L + 1), B, S, D] (the device packs them for the true S)"""
        if hid is not None:
            L1, D = self.cfg.num_hidden_layers + 1, self.cfg.hidden_size
            self.last_hidden_states = hid[: L1 * B * S * D].reshape(L1, B, S, D).copy()

    _max_images_per_block = 1

    def _max_feature_blocks(self, ids) -> int:
        """upper bound of the image blocks (of `rows` feature rows) ONE sample can splice: each of its placeholders expands
        to the images of one per-sample group (bounded by the largest group of the call) — per sample, not multiplied by
        the batch: the host buffers sized from it are B * (T + rows * this) rows"""
        return max(1, int((ids < 0).sum(axis=1).max())) * self._max_images_per_block

    def vision_tower_forward(self, pixels) -> np.ndarray:
        """CLIPVisionTower.forward (clip_encoder.py:39-51): [N,3,S,S] -> un-projected features [N, R, mm_hidden_size]."""
        N = int(pixels.shape[0])
        self._check_pixels(pixels, N)
        (p,), on_dev, keep = self._pixels(pixels)
        rows = self.cfg.num_patches + (1 if self.cfg.mm_vision_select_feature == "cls_patch" else 0)
        out = np.empty((N, rows, self.cfg.mm_hidden_size), dtype=np.float32)
        self._check(self.lib.vc_vision_tower_forward(self._model, p, on_dev, N, out.ctypes.data_as(C.c_void_p)))
        return out

    def decode_step(self, tokens=None, want_logits: bool = True, hidden_states: bool = False, attentions: bool = False):
        """-> (logits [B,V] or None, next_tok [B] int32).  hidden_states / attentions: this cached step's output_hidden_states
        [(L + 1), B, 1, D] / output_attentions [L, B, H, 1, pos + 1] in last_hidden_states / last_attentions (the step then runs
        outside its hipGraph)."""
        B = self._cur_batch
        self.last_hidden_states = self.last_attentions = None
        tok_p = None
        if tokens is not None:
            tk = np.ascontiguousarray(np.asarray(tokens).reshape(-1), dtype=np.int32)
            if tk.shape[0] != B:
                raise ValueError(f"expected {B} tokens")
            tok_p = tk.ctypes.data_as(C.c_void_p)
        lg = np.empty((B, self.cfg.vocab_size), dtype=np.float32) if want_logits else None
        nxt = np.empty((B,), dtype=np.int32)
        hid = att = None
        try:
            if hidden_states:
                hid = np.empty((self.cfg.num_hidden_layers + 1, B, 1, self.cfg.hidden_size), dtype=np.float32)
                self._check(self.lib.vc_request_hidden_states(self._model, hid.ctypes.data_as(C.c_void_p), C.c_size_t(hid.size)))
            if attentions:
                att = np.empty((self.cfg.num_hidden_layers, B, self.cfg.num_attention_heads, 1, self._step_pos + 1), dtype=np.float32)
                self._check(self.lib.vc_request_attentions(self._model, att.ctypes.data_as(C.c_void_p), C.c_size_t(att.size)))
            self._check(self.lib.vc_decode_step(self._model, tok_p, lg.ctypes.data_as(C.c_void_p) if want_logits else None,
                                                nxt.ctypes.data_as(C.c_void_p)))
        except BaseException:
            self.lib.vc_request_attentions(self._model, None, C.c_size_t(0))
            self.lib.vc_request_hidden_states(self._model, None, C.c_size_t(0))
            raise
        self._step_pos += 1
        self.last_hidden_states, self.last_attentions = hid, att
        return lg, nxt

    _step_pos = 0   # position of the token the next decode_step processes (prefill: S)

    _cur_batch = 0
    last_hidden_states = None

    def generate_greedy(self, input_ids, images, segs=None, depths=None, max_new_tokens: int = 128,
                        eos_token_id: Optional[int] = None, pad_token_id: Optional[int] = None,
                        stop_sequences: Optional[Sequence[Sequence[int]]] = None) -> np.ndarray:
        """-> new token ids [B, n_generated] int32 (prompt not included).  stop_sequences: up to 8 token-id sequences of
        up to 8 ids; a row is finished (pads afterwards) once its ids end with one of them — checked on the device."""
        return self.generate(input_ids, images, segs, depths, max_new_tokens=max_new_tokens, eos_token_id=eos_token_id,
                             pad_token_id=pad_token_id, stop_sequences=stop_sequences)

    def generate(self, input_ids, images, segs=None, depths=None, max_new_tokens: int = 128,
                 eos_token_id: Optional[int] = None, pad_token_id: Optional[int] = None,
                 stop_sequences: Optional[Sequence[Sequence[int]]] = None, do_sample: bool = False,
                 temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, seed: int = 0,
                 on_tokens=None, stream_every: int = 1, attention_mask=None) -> np.ndarray:
        """generate() on the device (vc_generate): greedy, or temperature / top-k / top-p sampling with a counter-based
        generator (same seed -> same tokens).  on_tokens(first_step, ids [B, n]) is called with every `stream_every` new
        columns while the hipGraph-replayed decode loop keeps running in between.  -> new ids [B, n_generated] int32."""
        ids = self._ids(input_ids)
        B, T = ids.shape
        pad = int(self.cfg.pad_token_id or 0) if pad_token_id is None else int(pad_token_id)
        if B > self.MAX_BATCH:
            # a replica prefills at most 16 sequences at a time: larger batches run as consecutive pieces.  Rows are
            # independent, so the result equals the whole-batch one: finished rows pad to the longest piece, and unequal
            # spliced lengths ACROSS pieces fail like they do inside one (the reference's quirk 6)
            if on_tokens is not None:
                raise ValueError(f"streaming needs a batch of at most {self.MAX_BATCH} sequences")
            parts, lens = [], set()
            for b0 in range(0, B, self.MAX_BATCH):
                sl = slice(b0, b0 + self.MAX_BATCH)
                parts.append(self.generate(ids[sl], *(None if a is None else a[sl] for a in (images, segs, depths)),
                                           attention_mask=None if attention_mask is None else attention_mask[sl],
                                           max_new_tokens=max_new_tokens, eos_token_id=eos_token_id,
                                           pad_token_id=pad_token_id, stop_sequences=stop_sequences, do_sample=do_sample,
                                           temperature=temperature, top_k=top_k, top_p=top_p, seed=seed + b0))
                lens.add(int(self.lib.vc_last_spliced_len(self._model)))
            if len(lens) > 1:
                raise UnboundLocalError("local variable '_new_labels' referenced before assignment")
            n = max(p.shape[1] for p in parts)
            return np.concatenate([np.pad(p, ((0, 0), (0, n - p.shape[1])), constant_values=pad) for p in parts], axis=0)
        (pi, ps, pd), on_dev, keep = self._pixels(*self._image_blocks(B, images, segs, depths))
        out = np.empty((B, max_new_tokens), dtype=np.int32)
        n = C.c_int(0)
        eos = -1 if eos_token_id is None else int(eos_token_id)
        stops = [list(map(int, q)) for q in (stop_sequences or [])]
        flat = np.ascontiguousarray([t for q in stops for t in q], dtype=np.int32)
        lens = np.ascontiguousarray([len(q) for q in stops], dtype=np.int32)
        samp = None
        if do_sample:
            samp = _lib.Sampling(1, float(temperature), int(top_k or 0), float(1.0 if top_p is None else top_p),
                                 int(seed) & 0xFFFFFFFFFFFFFFFF)
        errs = []

        def _cb(_user, first, nsteps, nb, ptr):
            try:
                on_tokens(first, np.ctypeslib.as_array(ptr, shape=(nb, nsteps)).copy())
            except BaseException as e:   # never unwind through the C frames
                errs.append(e)

        cb = _lib.TOKEN_CB(_cb) if on_tokens is not None else _lib.TOKEN_CB()
        # hides keys in the prefill; the cached steps see every key (the reference).  Announced last: the one-shot mask is
        # consumed by the call right below, nothing in between can raise and leave it armed
        self._announce_mask(attention_mask, B, T)
        self._check(self.lib.vc_generate(self._model, ids.ctypes.data_as(C.c_void_p), B, T, pi, ps, pd, on_dev,
                                         int(max_new_tokens), eos, pad,
                                         flat.ctypes.data_as(C.c_void_p) if stops else None,
                                         lens.ctypes.data_as(C.c_void_p) if stops else None, len(stops),
                                         C.byref(samp) if samp is not None else None, cb, None, int(stream_every),
                                         out.ctypes.data_as(C.c_void_p), C.byref(n)))
        if errs:
            raise errs[0]
        self._cur_batch = B
        return out[:, : n.value].copy()

    # bookkeeping used by decode_step
    def note_prefill(self, B: int):
        self._cur_batch = B

    def preprocess(self, images, pad: bool = True, mean=None, std=None, to_device: bool = False):
        """process_images() on the device (vcoder_llava/mm_utils.py:28-40): list of PIL images / uint8 [h,w,3] arrays ->
        fp32 [N,3,S,S] CLIP-normalised pixels (numpy, or a torch CUDA tensor when to_device).  PIL-exact bicubic."""
        S = self.cfg.vit_image_size
        mean = np.ascontiguousarray(synth.CLIP_MEAN if mean is None else mean, dtype=np.float32)
        std = np.ascontiguousarray(synth.CLIP_STD if std is None else std, dtype=np.float32)
        f32p = C.POINTER(C.c_float)
        if to_device:
            import torch

            out = torch.empty((len(images), 3, S, S), dtype=torch.float32, device=f"cuda:{self.device_index}")
        else:
            out = np.empty((len(images), 3, S, S), dtype=np.float32)
        for i, im in enumerate(images):
            a = np.ascontiguousarray(np.asarray(im.convert("RGB") if hasattr(im, "convert") else im), dtype=np.uint8)
            if a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("expected an RGB image [h,w,3]")
            dst = C.c_void_p(out[i].data_ptr()) if to_device else out[i].ctypes.data_as(C.c_void_p)
            self._check(self.lib.vc_preprocess_image(self._model, a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1],
                                                     int(pad), mean.ctypes.data_as(f32p), std.ctypes.data_as(f32p), dst,
                                                     int(to_device)))
        return out

    def last_timings(self):
        e, p, d = C.c_float(), C.c_float(), C.c_float()
        self._check(self.lib.vc_last_timings(self._model, C.byref(e), C.byref(p), C.byref(d)))
        return {"encode_ms": e.value, "prefill_ms": p.value, "decode_ms": d.value}

    def profile_decode_attention(self, B: int, ctx: int, reps: int = 3):
        n, us, by = C.c_int(), C.c_double(), C.c_double()
        self._check(self.lib.vc_profile_decode_attention(self._model, B, ctx, reps, C.byref(n), C.byref(us), C.byref(by)))
        return {"launches_per_step": n.value, "avg_us": us.value, "avg_bytes": by.value}

    def pool_step_counts(self):
        """cumulative pooled decode steps launched over 8 / 16 / 24 / 32 rows (zeros when generate() never used the pool)"""
        c = (C.c_ulonglong * 4)()
        self._check(self.lib.vc_pool_step_counts(self._model, c))
        return [int(x) for x in c]

    PROFILE_KINDS = ("qkv", "attention", "o_proj", "gate_up", "down", "lm_head")

    def pool_set_hold(self, on: bool = True):
        """pool policy (vc_pool_set_hold): True (default) = no decode step while a call that holds rows is still prefilling"""
        self._check(self.lib.vc_pool_set_hold(self._model, 1 if on else 0))

    def pool_set_rows(self, rows: int = 32):
        """rows of the shared decode pool (32 default; 64 = two weight passes per step, measurement) — when the pool is next built"""
        self._check(self.lib.vc_pool_set_rows(self._model, int(rows)))

    def pool_profile(self, on: bool = True):
        """in-situ timing of the pool's decode-step launches (vc_pool_profile): takes effect when the pool is next (re)built"""
        self._check(self.lib.vc_pool_profile(self._model, 1 if on else 0))

    def pool_profile_read(self, reset: bool = True):
        """{rows: {kind: {"us": period microseconds, "exec_us": ..., "launches": n}}} since the last reset, rows in (8, 16, 24, 32).
        "us" = latest workgroup end of the previous launch of the step -> latest end of this one (what the dependency chain pays per
        launch); "exec_us" = earliest start -> latest end of the launch's own workgroups.  The pool must be idle."""
        ex, pe, n = (C.c_double * 24)(), (C.c_double * 24)(), (C.c_ulonglong * 24)()
        self._check(self.lib.vc_pool_profile_read(self._model, ex, pe, n, 1 if reset else 0))
        return {8 * (s + 1): {k: {"us": pe[s * 6 + i], "exec_us": ex[s * 6 + i], "launches": int(n[s * 6 + i])}
                              for i, k in enumerate(self.PROFILE_KINDS)} for s in range(4)}

    def profile_decode_gemv(self, B: int, reps: int = 3):
        n, us, by = C.c_int(), C.c_double(), C.c_double()
        self._check(self.lib.vc_profile_decode_gemv(self._model, B, reps, C.byref(n), C.byref(us), C.byref(by)))
        return {"launches_per_step": n.value, "avg_us": us.value, "avg_bytes": by.value}
