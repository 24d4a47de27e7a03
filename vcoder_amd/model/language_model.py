"""The reference's model API on top of the HIP engine.

Mirrors (same names, arguments, return conventions and error behaviour):
  VCoderDSLlavaLlamaForCausalLM   vcoder_llava/model/language_model/vcoder_ds_llava_llama.py:41-142
  VCoderLlavaLlamaForCausalLM     vcoder_llava/model/language_model/vcoder_llava_llama.py:40-139
  LlavaLlamaForCausalLM           vcoder_llava/model/language_model/llava_llama.py
and the parts of HF `GenerationMixin.generate` the reference's callers use (SURVEY.md Appendix C;
callers: serve/cli.py:122-132, serve/chat.py:141-151, eval/model_seg_loader.py:129-139).
No torch.nn / HF Transformers module is involved: forward/generate marshal to libvcoder_hip.so.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np

from ..config import VCoderConfig
from ..engine import HipEngine
from .projector import build_depth_projector, build_seg_projector, build_vision_projector
from .vision_tower import build_vision_tower


class CausalLMOutputWithPast(dict):
    """Attribute + mapping access like transformers.modeling_outputs.CausalLMOutputWithPast."""

    def __init__(self, loss=None, logits=None, past_key_values=None, hidden_states=None, attentions=None):
        super().__init__(loss=loss, logits=logits, past_key_values=past_key_values, hidden_states=hidden_states,
                         attentions=attentions)
        self.__dict__ = self

    def __getitem__(self, k):
        if isinstance(k, int):
            return [v for v in (self.loss, self.logits, self.past_key_values) if v is not None][k]
        return dict.__getitem__(self, k)


class KVCacheHandle:
    """Opaque `past_key_values`: the cache lives in the engine (K key-major, V transposed, per layer); this handle
    only proves which prefill it belongs to.  `[-1][-1].shape[-2]` (how vcoder_ds_llava_arch.py:132 reads the past
    length) is supported."""

    def __init__(self, model, generation: int, length: int, batch: int):
        self._model, self.generation, self.length, self.batch = model, generation, length, batch

    def get_seq_length(self) -> int:
        return self.length

    def __bool__(self):
        return True

    def __getitem__(self, i):
        shape = (self.batch, self._model.config.num_attention_heads, self.length, self._model.config.head_dim)
        t = SimpleNamespace(shape=shape)
        return (t, t)

    def __len__(self):
        return self._model.config.num_hidden_layers


class _InnerModel:
    """What `get_model()` returns: the plugin modules of VCoder[DS]LlavaMetaModel (vcoder_ds_llava_arch.py:30-49)."""

    def __init__(self, config: VCoderConfig):
        self.config = config
        if hasattr(config, "mm_vision_tower"):
            self.vision_tower = build_vision_tower(config, delay_load=True)
            self.mm_projector = build_vision_projector(config)
        if config.variant != "llava":
            self.seg_mm_projector = build_seg_projector(config)
            if config.use_mm2_proj:
                self.mm2_projector = build_vision_projector(config)      # dead at inference (quirk 2)
        if config.variant == "vcoder_ds":
            self.depth_mm_projector = build_depth_projector(config)      # dead at inference (quirk 1)
        self.embed_tokens = SimpleNamespace(num_embeddings=config.vocab_size, embedding_dim=config.hidden_size)

    def get_vision_tower(self):
        return getattr(self, "vision_tower", None)


class _HipCausalLMBase:
    variant = "llava"
    model_type = "llava"

    def __init__(self, config: VCoderConfig, device="cuda", _lib_override=None, operands="bf16"):
        """operands (an addition of this build): "bf16" — libvcoder_hip.so, the benchmarked path; "fp16" — libvcoder_hip_f16.so, the
        same kernels with IEEE fp16 MFMA operands: the operand precision of the reference's own GPU path (builder.py:39,142)."""
        if config.variant != self.variant:
            config.variant = self.variant
        self.config = config
        self.model = _InnerModel(config)
        self._device_str = device if isinstance(device, str) else str(device)
        idx = 0
        if ":" in self._device_str:
            idx = int(self._device_str.split(":")[1])
        self.engine = HipEngine(config, device_index=idx, lib=_lib_override, operands=operands)
        tower = self.model.get_vision_tower()
        if tower is not None:
            tower._engine = self.engine          # CLIPVisionTower.forward runs on the engine's tower kernels
        self._generation = 0
        self.training = False
        self.generation_config = SimpleNamespace(pad_token_id=config.pad_token_id, eos_token_id=config.eos_token_id,
                                                 bos_token_id=config.bos_token_id)

    # ---- nn.Module-ish surface the reference's callers touch ------------------------------------------
    @property
    def device(self):
        import torch

        return torch.device(self._device_str if ":" in self._device_str else self._device_str + ":0")

    @property
    def dtype(self):
        import torch

        return torch.bfloat16

    def get_model(self):
        return self.model

    def get_vision_tower(self):
        return self.model.get_vision_tower()

    def eval(self):
        return self

    def requires_grad_(self, flag: bool = False):
        return self

    def to(self, *a, **k):
        return self

    def half(self):
        return self

    def cuda(self, *a):
        return self

    def parameters(self):
        return iter(())

    # ---- weights ----------------------------------------------------------------------------------------
    _PLUGINS = ("mm_projector", "seg_mm_projector", "depth_mm_projector", "mm2_projector")

    def _load_tensor(self, key: str, value) -> bool:
        """one checkpoint tensor: into the engine's inference layout AND — for the adapter plugins — into the module object
        get_model() hands out, so that `get_model().mm_projector` is a loaded, callable module as in the reference
        (vcoder_ds_llava_arch.py:34-49; depth_mm_projector / mm2_projector are dead in forward but still state-dict
        complete).  -> False when the engine ignores the key (dead at inference)."""
        for name in self._PLUGINS:
            pre = f"model.{name}."
            if key.startswith(pre) and hasattr(self.model, name):
                getattr(self.model, name).load_state_dict({key[len(pre):]: value}, strict=False)
        return self.engine.load_tensor(key, value)

    def load_state_dict(self, sd, strict: bool = True):
        used = dead = 0
        for k, v in sd.items():
            if self._load_tensor(k, v):
                used += 1
            else:
                dead += 1
        return SimpleNamespace(missing_keys=[], unexpected_keys=[], used=used, dead=dead)

    def finalize_weights(self):
        self.engine.finalize()
        # tensors whose checkpoint values bf16 cannot hold (an fp16 LLM / fp32 CLIP tower, the reference's own dtypes): the engine
        # kept a lo plane for each; the bf16 fast path computes with the rounded weights, 'strict' / 'split' with hi + lo
        self.inexact_tensors = self.engine.inexact_tensors()
        if self.inexact_tensors:
            import logging

            logging.getLogger("vcoder_amd").info(
                "%d checkpoint tensors are not bf16-representable: the bf16 fast path rounds them to bf16; precision modes "
                "'split' / 'strict' compute with bf16 hi + lo planes (the checkpoint's values to ~16 mantissa bits)", self.inexact_tensors)
        tower = self.get_vision_tower()
        if tower is not None:
            tower.is_loaded = True
        return self

    @classmethod
    def from_pretrained(cls, model_path: str, low_cpu_mem_usage=True, device="cuda", config=None, weight_format="bf16",
                        operands="bf16", **kwargs):
        """Loads config.json + weight shards; the CLIP tower comes from the checkpoint if present, else from the
        local directory `config.mm_vision_tower` (the reference downloads it: clip_encoder.py:22-27 — there is no
        network here, so a hub name that is not a local directory is an error)."""
        import os

        from .. import checkpoint

        if config is not None and not isinstance(config, VCoderConfig):
            # an HF PretrainedConfig handed over by AutoModelForCausalLM.from_pretrained (vcoder_amd/hf_register.py)
            config = VCoderConfig.from_hf_dict(config.to_dict(), os.path.basename(os.path.normpath(str(model_path))))
        cfg = config if config is not None else VCoderConfig.from_pretrained(model_path, os.path.basename(model_path))
        return cls.from_tensors(cfg, checkpoint.iter_checkpoint_tensors(model_path), device=device, weight_format=weight_format,
                                operands=operands)

    @classmethod
    def from_tensors(cls, cfg: VCoderConfig, tensors, device="cuda", weight_format="bf16", _lib_override=None, operands="bf16"):
        """A model from an iterator of (HF state-dict key, tensor) pairs — one checkpoint, or a base checkpoint overlaid with
        merged LoRA deltas / projector weights (model/builder.py).  The CLIP tower comes from the stream when it carries
        `vision_tower` keys, else from the local directory `config.mm_vision_tower`."""
        model = cls(cfg, device=device, _lib_override=_lib_override, operands=operands)
        saw_tower = False
        for k, v in tensors:
            saw_tower |= "vision_tower" in k
            model._load_tensor(k, v)
        if not saw_tower:
            model.get_vision_tower().load_model(engine=model.engine)
        if weight_format != "bf16":
            model.engine.set_weight_format(weight_format)   # W8A16: quantised on the device at finalize
        model.finalize_weights()
        return model

    # ---- forward (vcoder_ds_llava_llama.py:57-118) ----------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, images=None, segs=None, depths=None,
                return_dict=None):
        import torch

        if labels is not None:
            raise NotImplementedError("training loss is outside the inference hot path (SURVEY.md §8: train/* out of scope)")
        # The reference OVERWRITES a caller's inputs_embeds with what prepare_inputs_labels_for_multimodal returns
        # (vcoder_ds_llava_llama.py:79: the spliced embeddings, or None) — the argument is dead; without input_ids the call
        # then dies in LlamaModel ("You must specify exactly one of input_ids or inputs_embeds").  Same here.
        if input_ids is None:
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds (the reference discards a caller's "
                             "inputs_embeds: vcoder_ds_llava_llama.py:79)")
        ids = input_ids
        B = ids.shape[0]
        if past_key_values is not None and ids.shape[1] == 1:
            # cached decode step: the input_ids.shape[1]==1 fast path (vcoder_ds_llava_arch.py:130-133)
            if not isinstance(past_key_values, KVCacheHandle) or past_key_values.generation != self._generation:
                raise RuntimeError("past_key_values does not belong to the engine's current KV cache")
            tok = ids.reshape(-1).detach().cpu().numpy() if hasattr(ids, "detach") else np.asarray(ids).reshape(-1)
            # With images the reference REPLACES the mask by ones here (vcoder_ds_llava_arch.py:130-133): keys a padded
            # prefill hid become visible.  Without images the caller's mask goes through to LlamaModel: the engine keeps the
            # prefill's hidden keys hidden (new positions are visible, as HF's generate loop appends ones).
            if images is not None or attention_mask is None:
                self.engine.clear_attention_mask()
            # output_hidden_states / output_attentions of a cached step: [B, 1, D] per entry, [B, H, 1, past + 1] per layer
            lg, _ = self.engine.decode_step(tok, hidden_states=bool(output_hidden_states), attentions=bool(output_attentions))
            past_key_values.length += 1
            logits = torch.from_numpy(lg).unsqueeze(1)
            pkv = past_key_values
        else:
            if past_key_values is not None:
                raise NotImplementedError("multi-token continuation of a cached sequence is not on the reference's path")
            # images None: prepare_inputs_labels_for_multimodal returns early (vcoder_ds_llava_arch.py:129-133) and the
            # call is a plain Llama forward over the text ids
            if images is None and attention_mask is not None and not _all_ones(attention_mask):
                # (before the prefill: a refused call must leave the engine's cache and the live KVCacheHandles as they were)
                raise NotImplementedError("a padded TEXT-ONLY batch (attention_mask with zeros, images=None) is outside the "
                                          "VCoder hot path")
            _, full, S = self.engine.prefill(ids, images, (segs if self.variant != "llava" else None) if images is not None else None,
                                             (depths if self.variant == "vcoder_ds" else None) if images is not None else None,
                                             all_logits=True, reserve=self._decode_reserve,
                                             attention_mask=attention_mask if images is not None else None,
                                             hidden_states=bool(output_hidden_states), attentions=bool(output_attentions))
            self._generation += 1
            logits = torch.from_numpy(full)
            pkv = KVCacheHandle(self, self._generation, S, B)
        if hasattr(ids, "device") and getattr(ids, "is_cuda", False):
            logits = logits.to(ids.device)
        hs = None
        if output_hidden_states and self.engine.last_hidden_states is not None:
            # the tuple LlamaModel.forward returns: inputs_embeds, every layer's output, the last one after the final norm
            hs = tuple(torch.from_numpy(h) for h in self.engine.last_hidden_states)
        at = None
        if output_attentions and self.engine.last_attentions is not None:
            at = tuple(torch.from_numpy(a) for a in self.engine.last_attentions)     # L tensors [B, H, S, S]
        out = CausalLMOutputWithPast(loss=None, logits=logits, past_key_values=pkv if use_cache is not False else None,
                                     hidden_states=hs, attentions=at)
        if return_dict is False:
            return (out.logits,) + ((out.past_key_values,) if out.past_key_values is not None else ())
        return out

    __call__ = forward

    # KV slots a forward() prefill keeps free for the caller's cached decode steps (an HF-style external generate loop
    # drives forward once per token); the cache grows if the loop runs longer
    _decode_reserve = 512

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None,
                                      **kwargs):
        """vcoder_ds_llava_llama.py:120-142"""
        if past_key_values:
            input_ids = input_ids[:, -1:]
        model_inputs = {"inputs_embeds": inputs_embeds} if (inputs_embeds is not None and past_key_values is None) \
            else {"input_ids": input_ids}
        model_inputs.update({"past_key_values": past_key_values, "use_cache": kwargs.get("use_cache"),
                             "attention_mask": attention_mask, "images": kwargs.get("images", None)})
        if self.variant != "llava":
            model_inputs["segs"] = kwargs.get("segs", None)
        if self.variant == "vcoder_ds":
            model_inputs["depths"] = kwargs.get("depths", None)
        return model_inputs

    # ---- generate (HF GenerationMixin subset) ---------------------------------------
    def generate(self, input_ids=None, inputs=None, images=None, segs=None, depths=None, do_sample: bool = False,
                 temperature: float = 1.0, top_p: Optional[float] = None, top_k: Optional[int] = None, num_beams: int = 1,
                 max_new_tokens: Optional[int] = None, max_length: Optional[int] = None, streamer=None,
                 use_cache: bool = True, stopping_criteria=None, eos_token_id=None, pad_token_id=None,
                 attention_mask=None, generator=None, seed: Optional[int] = None, **kwargs):
        """Returns cat(input_ids, new_ids) [B, T+n] int64 — the prompt part keeps its negative placeholder ids,
        callers slice `[:, T:]` (serve/cli.py:135).

        Everything the reference's callers ask for runs on the device inside the hipGraph-replayed decode loop
        (vc_generate): greedy argmax; sampling with HF's warper order temperature -> top-k -> top-p -> multinomial
        (`top_k` defaults to 50 when sampling, HF's GenerationConfig default, which is what serve/cli.py:122-132 gets);
        EOS / pad bookkeeping; stopping criteria that reduce to token-id suffixes (KeywordsStoppingCriteria(["</s>"]));
        the streamer, fed from a callback every `stream_every` tokens.  Only a stopping criterion that needs host code
        (a text match over several tokens) falls back to the per-token decode_step loop, like the reference's HF loop.
        Sampling draws from a counter-based generator: pass `seed` (or a torch `generator`, whose initial seed is used)
        for reproducible output; bit-equality with torch.multinomial's stream is not defined."""
        import torch

        if input_ids is None:
            input_ids = inputs
        if temperature is not None and float(temperature) <= 0.0:
            do_sample = False
        if num_beams != 1:
            if streamer is not None:
                raise ValueError("`streamer` cannot be used with beam search")   # HF's own check
            sample = None
            if do_sample:   # beam-sample (the eval loaders forward --temperature and --num_beams independently)
                if generator is None:
                    generator = torch.Generator().manual_seed(int(seed if seed is not None else torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF)
                sample = dict(temperature=float(temperature or 1.0), top_k=50 if top_k is None else int(top_k),
                              top_p=1.0 if top_p is None else float(top_p), generator=generator)
            return self._beam_search(input_ids, images, segs, depths, int(num_beams), max_new_tokens, max_length, eos_token_id,
                                     pad_token_id, attention_mask, stopping_criteria,
                                     float(kwargs.get("length_penalty", 1.0)), kwargs.get("early_stopping", False),
                                     kwargs.get("_beam_len_counts_prompt", True), sample)
        T = input_ids.shape[1]
        B = input_ids.shape[0]
        if max_new_tokens is None:
            max_new_tokens = (max_length - T) if max_length is not None else 20
        eos = self.config.eos_token_id if eos_token_id is None else eos_token_id
        # HF's GenerationConfig takes one EOS id or a list: any of them finishes a row.  The first is the device loop's EOS, the
        # others end a row the same way as single-token stop sequences (token kept, pads afterwards)
        eos_more = []
        if isinstance(eos, (list, tuple)):
            eos_more = [int(e) for e in eos[1:]]
            eos = int(eos[0]) if len(eos) else None
        pad = pad_token_id if pad_token_id is not None else (self.config.pad_token_id if self.config.pad_token_id is not None else eos)
        segs = segs if self.variant != "llava" else None
        depths = depths if self.variant == "vcoder_ds" else None
        ids_cpu = input_ids.detach().cpu() if hasattr(input_ids, "detach") else torch.as_tensor(np.asarray(input_ids))
        if do_sample:
            top_k = 50 if top_k is None else int(top_k)
            top_p = 1.0 if top_p is None else float(top_p)
            if seed is None:
                seed = int(generator.initial_seed()) if generator is not None else int(torch.initial_seed())
                seed = (seed + 0x9E3779B97F4A7C15 * self._sample_calls) & 0xFFFFFFFFFFFFFFFF   # successive calls differ
                self._sample_calls += 1
        # stopping criteria that reduce to a token-id suffix match run on the device: the
        # reference's KeywordsStoppingCriteria(["</s>"]) of cli.py / the eval loaders is of that kind
        stops = []
        for crit in (stopping_criteria or []):
            seqs = crit.device_stop_sequences() if hasattr(crit, "device_stop_sequences") else _reference_keyword_stop(crit)
            if seqs is None:
                stops = None
                break
            stops += seqs
        if stops is not None:
            stops = stops + [[e] for e in eos_more]
        on_device = stops is not None and len(stops) <= 8
        if streamer is not None:
            streamer.put(ids_cpu)
        if on_device:
            on_tokens = None
            if streamer is not None:
                def on_tokens(first, new_ids):   # HF feeds the streamer one [B] tensor per step
                    for j in range(new_ids.shape[1]):
                        streamer.put(torch.from_numpy(new_ids[:, j].astype(np.int64)))
            new = self.engine.generate(ids_cpu.numpy(), images, segs, depths, max_new_tokens=max_new_tokens,
                                       eos_token_id=eos, pad_token_id=pad, stop_sequences=stops or None,
                                       do_sample=bool(do_sample), temperature=float(temperature or 1.0),
                                       top_k=top_k or 0, top_p=1.0 if top_p is None else top_p, seed=seed or 0,
                                       on_tokens=on_tokens, stream_every=kwargs.get("stream_every", 1),
                                       attention_mask=attention_mask)
            self._generation += 1
            out = torch.cat([ids_cpu, torch.from_numpy(new.astype(np.int64))], dim=1)
            if streamer is not None:
                streamer.end()
        else:
            last, _, S = self.engine.prefill(ids_cpu.numpy(), images, segs, depths, has_attention_mask=True,
                                             reserve=max_new_tokens, attention_mask=attention_mask)
            self.engine.clear_attention_mask()   # generate(): the cached steps run under an all-ones mask (:130-133)
            self._generation += 1
            logits = torch.from_numpy(last)
            unfinished = torch.ones(B, dtype=torch.long)
            cur = ids_cpu
            if generator is None and do_sample:
                generator = torch.Generator().manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
            for step in range(max_new_tokens):
                scores = logits.float()
                if do_sample:
                    scores = scores / float(temperature)
                    if top_k and 0 < top_k < scores.shape[-1]:
                        kth = torch.topk(scores, top_k)[0][..., -1, None]
                        scores = scores.masked_fill(scores < kth, float("-inf"))
                    if top_p is not None and top_p < 1.0:
                        scores = _top_p_filter(scores, float(top_p))
                    probs = torch.softmax(scores, dim=-1)
                    nxt = torch.multinomial(probs, 1, generator=generator).squeeze(1)
                else:
                    nxt = torch.argmax(scores, dim=-1)
                if eos is not None:
                    nxt = nxt * unfinished + pad * (1 - unfinished)
                cur = torch.cat([cur, nxt[:, None]], dim=1)
                if streamer is not None:
                    streamer.put(nxt.cpu())
                if eos is not None:
                    unfinished = unfinished * (~torch.isin(nxt, torch.tensor([eos] + eos_more))).long()
                stop = bool(unfinished.max() == 0) if eos is not None else False
                if stopping_criteria:
                    for crit in stopping_criteria:
                        r = crit(cur, scores)
                        stop = stop or bool(r.all() if hasattr(r, "all") else r)
                if stop or step + 1 == max_new_tokens:
                    break
                lg, _ = self.engine.decode_step(nxt.numpy().astype(np.int32))
                logits = torch.from_numpy(lg)
            out = cur
            if streamer is not None:
                streamer.end()
        if hasattr(input_ids, "device"):
            out = out.to(input_ids.device)
        return out

    _sample_calls = 0

    # ---- beam search (HF GenerationMixin.beam_search + BeamSearchScorer of the reference's pinned Transformers 4.31) -----------
    def _beam_search(self, input_ids, images, segs, depths, num_beams, max_new_tokens, max_length, eos_token_id, pad_token_id,
                     attention_mask, stopping_criteria, length_penalty, early_stopping, len_counts_prompt, sample=None):
        """`generate(num_beams=n)` as the reference's eval loaders can ask for it (eval/model_seg_loader.py:129-139 forwards
        args.num_beams): every sequence is expanded to n beams (rows b*n .. b*n+n-1, as `_expand_inputs_for_generation`), the
        prefill and one cached decode step per token run on the engine, scoring is HF's — log-softmax of the fp32 logits plus
        the running beam score, the 2n best continuations per sequence, BeamSearchScorer.process / finalize (hypotheses
        scored by sum_logprobs / len ** length_penalty; 4.31 counts the PROMPT ids in len) — and the KV rows are reordered by
        beam_idx after every step (vc_reorder_cache).  Returns [B, T + n_new] int64, shorter rows padded with pad_token_id
        (after one EOS), like HF.

        sample (do_sample=True: `beam_sample` of Transformers 4.31, generation/utils.py): the 2n candidates per sequence are DRAWN
        instead of taken — the warpers (temperature -> top-k -> top-p) act on log-softmax + running beam score, as 4.31 applies
        them, `torch.multinomial(softmax(.), 2n)` over the n * V continuations, the draws sorted by score — everything behind it
        (EOS handling, hypotheses, cache reorder) is the beam search's.  The draws come from the caller's torch generator
        (or `seed`), so a seed reproduces them; bit-equality with a given HF version's stream is not defined (its warper
        order changed after 4.31)."""
        import torch

        ids_cpu = input_ids.detach().cpu() if hasattr(input_ids, "detach") else torch.as_tensor(np.asarray(input_ids))
        B, T = ids_cpu.shape
        nb = num_beams
        if B * nb > self.engine.MAX_BATCH:
            raise ValueError(f"batch {B} x num_beams {nb} exceeds the {self.engine.MAX_BATCH} rows a replica decodes at a time")
        if max_new_tokens is None:
            max_new_tokens = (max_length - T) if max_length is not None else 20
        eos = self.config.eos_token_id if eos_token_id is None else eos_token_id
        # an int or a list of ids (HF's BeamSearchScorer takes both and tests membership); negative = none
        eos_list = [int(e) for e in (eos if isinstance(eos, (list, tuple)) else ([] if eos is None else [eos])) if int(e) >= 0]
        eos_set = set(eos_list)
        eos = eos_list[0] if eos_list else None          # what pads / terminates a finished row (HF: eos_token_id[0])
        pad = pad_token_id if pad_token_id is not None else (self.config.pad_token_id if self.config.pad_token_id is not None else eos)
        segs = segs if self.variant != "llava" else None
        depths = depths if self.variant == "vcoder_ds" else None
        rep = np.repeat(np.arange(B), nb)

        def expand(a):
            if a is None:
                return None
            if isinstance(a, (list, tuple)):
                return [a[i] for i in rep]
            return a[torch.as_tensor(rep, device=a.device)] if hasattr(a, "detach") else np.asarray(a)[rep]

        ids_x = ids_cpu[torch.as_tensor(rep)]
        mask_x = expand(attention_mask)
        last, _, S = self.engine.prefill(ids_x.numpy(), expand(images), expand(segs), expand(depths), has_attention_mask=True,
                                         reserve=max_new_tokens, attention_mask=mask_x)
        self.engine.clear_attention_mask()          # cached steps under the all-ones mask (vcoder_ds_llava_arch.py:130-133)
        self._generation += 1
        V = last.shape[-1]
        seqs = ids_x.clone()                                                  # [B*nb, T + generated]
        beam_scores = torch.zeros(B, nb)
        beam_scores[:, 1:] = -1e9
        beam_scores = beam_scores.view(-1)
        hyps = [[] for _ in range(B)]                                         # per sequence: (score, ids), at most nb kept
        worst = [1e9] * B
        done = [False] * B

        def hyp_len(n_ids):                                                    # 4.31: the whole id row; >= 4.38: generated part only
            return n_ids if len_counts_prompt else n_ids - T

        def add_hyp(b, ids_row, sum_logprobs):
            score = sum_logprobs / (max(hyp_len(ids_row.shape[-1]), 1) ** length_penalty)
            if len(hyps[b]) < nb or score > worst[b]:
                hyps[b].append((score, ids_row))
                if len(hyps[b]) > nb:
                    srt = sorted((s_, i) for i, (s_, _) in enumerate(hyps[b]))
                    del hyps[b][srt[0][1]]
                    worst[b] = srt[1][0]
                else:
                    worst[b] = min(score, worst[b])

        def is_done(b, best_sum_logprobs, cur_len):
            if len(hyps[b]) < nb:
                return False
            if early_stopping is True:
                return True
            if early_stopping is False:
                return worst[b] >= best_sum_logprobs / (max(hyp_len(cur_len), 1) ** length_penalty)
            # "never"
            hl = hyp_len(cur_len) if length_penalty <= 0.0 else hyp_len(T + max_new_tokens)
            return worst[b] >= best_sum_logprobs / (max(hl, 1) ** length_penalty)

        logits = torch.from_numpy(last).float()
        n_steps = 0
        for step in range(max_new_tokens):
            scores = torch.log_softmax(logits, dim=-1) + beam_scores[:, None]
            if sample is not None:
                w = scores / sample["temperature"]
                if 0 < sample["top_k"] < V:
                    kth = torch.topk(w, sample["top_k"])[0][..., -1, None]
                    w = w.masked_fill(w < kth, float("-inf"))
                if sample["top_p"] < 1.0:
                    w = _top_p_filter(w, sample["top_p"], min_tokens_to_keep=2)   # 4.31: min_tokens_to_keep = 2 when num_beams > 1
                flat = w.view(B, nb * V)
                probs = torch.softmax(flat, dim=-1)
                # torch.multinomial without replacement needs >= 2n non-zero entries (step 0: only beam 0 is alive, and a tight
                # top_p / top_k can leave fewer): the finite-score continuations are topped up with the next-best ones by score
                short = (probs > 0).sum(-1) < 2 * nb
                if bool(short.any()):
                    fill = torch.topk(scores.view(B, nb * V), 2 * nb, dim=1).indices
                    tiny = torch.zeros_like(probs).scatter(1, fill, torch.finfo(probs.dtype).tiny)
                    probs = torch.where(short[:, None] & (probs == 0), tiny, probs)
                    flat = torch.where(short[:, None] & torch.isinf(flat) & (tiny > 0), scores.view(B, nb * V), flat)
                draw = torch.multinomial(probs, 2 * nb, generator=sample["generator"])
                top_s, order = torch.sort(torch.gather(flat, -1, draw), descending=True, dim=1)
                top_i = torch.gather(draw, -1, order)
                scores = w
            else:
                top_s, top_i = torch.topk(scores.view(B, nb * V), 2 * nb, dim=1, largest=True, sorted=True)
            next_idx, next_tok = top_i // V, top_i % V
            nscore, ntok, nsrc = torch.zeros(B, nb), torch.zeros(B, nb, dtype=torch.long), torch.zeros(B, nb, dtype=torch.long)
            cur_len = seqs.shape[1]
            for b in range(B):
                if done[b]:
                    ntok[b] = pad if pad is not None else 0
                    continue
                k = 0
                for rank in range(2 * nb):
                    tok, sc, src = int(next_tok[b, rank]), float(top_s[b, rank]), b * nb + int(next_idx[b, rank])
                    if tok in eos_set:
                        if rank >= nb:
                            continue
                        add_hyp(b, seqs[src].clone(), sc)
                    else:
                        nscore[b, k], ntok[b, k], nsrc[b, k] = sc, tok, src
                        k += 1
                    if k == nb:
                        break
                if k < nb:
                    raise ValueError(f"At most {nb} tokens in {next_tok[b].tolist()} can be equal to `eos_token_id: {eos}`.")
                done[b] = done[b] or is_done(b, float(top_s[b].max()), cur_len)
            beam_scores = nscore.view(-1)
            beam_idx = nsrc.view(-1)
            seqs = torch.cat([seqs[beam_idx], ntok.view(-1, 1)], dim=1)
            n_steps = step + 1
            stop = all(done)
            if stopping_criteria:
                for crit in stopping_criteria:
                    r = crit(seqs, scores)
                    stop = stop or bool(r.all() if hasattr(r, "all") else r)
            if stop or step + 1 == max_new_tokens:
                break
            self.engine.reorder_cache(beam_idx.numpy())
            lg, _ = self.engine.decode_step(ntok.view(-1).numpy().astype(np.int32))
            logits = torch.from_numpy(lg).float()
        # finalize: open beams become hypotheses, the best one per sequence is returned
        out_rows = []
        for b in range(B):
            if not done[b]:
                for j in range(nb):
                    add_hyp(b, seqs[b * nb + j], float(beam_scores[b * nb + j]))
            out_rows.append(max(hyps[b], key=lambda t_: t_[0])[1])
        L = max(r.shape[0] for r in out_rows)
        max_len = T + max_new_tokens
        L_out = min(L + 1, max_len)            # BeamSearchScorer.finalize: room for the EOS a finished hypothesis does not carry
        if any(r.shape[0] < L_out for r in out_rows) and pad is None:
            raise ValueError("`pad_token_id` has to be defined")
        out = torch.full((B, L_out), int(pad) if pad is not None else 0, dtype=torch.long)
        for b, r in enumerate(out_rows):
            out[b, : r.shape[0]] = r
            if r.shape[0] < L_out and eos is not None:
                out[b, r.shape[0]] = int(eos)
        if hasattr(input_ids, "device"):
            out = out.to(input_ids.device)
        return out


def _all_ones(attention_mask) -> bool:
    m = attention_mask.detach().cpu().numpy() if hasattr(attention_mask, "detach") else np.asarray(attention_mask)
    return bool(np.all(m != 0))


def _reference_keyword_stop(crit):
    """The reference's own KeywordsStoppingCriteria object (vcoder_llava/mm_utils.py:128-151): usable on the device when
    every keyword is ONE special token — its text check decodes with skip_special_tokens=True and can never match one, and
    its id check only works for single-token keywords in the first place.  Anything else -> None (host loop)."""
    kws, tok = getattr(crit, "keyword_ids", None), getattr(crit, "tokenizer", None)
    if kws is None or tok is None or not hasattr(crit, "keywords"):
        return None
    special = set(getattr(tok, "all_special_ids", []) or [])
    out = []
    for k in kws:
        ids = [int(t) for t in (k.tolist() if hasattr(k, "tolist") else k)]
        if len(ids) != 1 or ids[0] not in special:
            return None
        out.append(ids)
    return out


def _top_p_filter(scores, top_p: float, min_tokens_to_keep: int = 1):
    import torch

    s, idx = torch.sort(scores, descending=False, dim=-1)
    cum = s.softmax(-1).cumsum(-1)
    remove = cum <= (1 - top_p)
    remove[..., -min_tokens_to_keep:] = False
    mask = remove.scatter(1, idx, remove)
    return scores.masked_fill(mask, float("-inf"))


class LlavaLlamaForCausalLM(_HipCausalLMBase):
    variant, model_type = "llava", "llava"


class VCoderLlavaLlamaForCausalLM(_HipCausalLMBase):
    variant, model_type = "vcoder", "vcoder_llava"


class VCoderDSLlavaLlamaForCausalLM(_HipCausalLMBase):
    variant, model_type = "vcoder_ds", "vcoder_ds_llava"


class LlavaConfig(VCoderConfig):
    def __init__(self, **kw):
        kw.setdefault("variant", "llava")
        super().__init__(**kw)


class VCoderLlavaConfig(VCoderConfig):
    def __init__(self, **kw):
        kw.setdefault("variant", "vcoder")
        super().__init__(**kw)


class VCoderDSLlavaConfig(VCoderConfig):
    def __init__(self, **kw):
        kw.setdefault("variant", "vcoder_ds")
        super().__init__(**kw)
