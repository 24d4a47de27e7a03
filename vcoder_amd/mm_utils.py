"""Prompt -> ids glue and image padding (the step right before the hot path).

Behavioural counterparts of vcoder_llava/mm_utils.py:14-151: the placeholder id ORDER these helpers emit
([IMG, DEPTH, SEG] for a '<depth>\\n<seg>\\n<image>' prompt, [IMG, SEG] for '<seg>\\n<image>') is part of the
parity contract of the splice (tests/golden/tokenizer_orders.json pins it against the reference)."""
from __future__ import annotations

from typing import List, Sequence

from .constants import DEPTH_TOKEN_INDEX, IMAGE_TOKEN_INDEX, SEG_TOKEN_INDEX


def _chunk_ids(prompt: str, tokenizer, marker: str) -> List[List[int]]:
    return [list(tokenizer(piece).input_ids) for piece in prompt.split(marker)]


def _join_with(chunks: List[List[int]], sep: Sequence[int], tokenizer, keep_sep_prefix_only: bool = False) -> List[int]:
    """Concatenate tokenised chunks with `sep` ids in between.  Every chunk starts with BOS when the tokenizer adds
    one; it is kept once at the very beginning and stripped elsewhere."""
    bos = getattr(tokenizer, "bos_token_id", None)
    has_bos = bool(chunks) and bool(chunks[0]) and chunks[0][0] == bos
    out: List[int] = [chunks[0][0]] if has_bos else []
    skip = 1 if has_bos else 0
    for i, ch in enumerate(chunks):
        if i > 0:
            out.extend(sep)
        out.extend(ch[skip:])
    return out


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    ids = _join_with(_chunk_ids(prompt, tokenizer, "<image>"), [image_token_index], tokenizer)
    return _ret(ids, return_tensors)


def tokenizer_seg_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, seg_token_index=SEG_TOKEN_INDEX,
                        return_tensors=None):
    # the reference inserts [SEG, IMG]*(offset+1) between the chunks and keeps x[offset:-1] of it (mm_utils.py:78-82):
    # with a BOS-prepending tokenizer (offset 1; every Llama tokenizer) that is the pair [IMG, SEG] observed in
    # tests/golden/tokenizer_orders.json; with a tokenizer that adds no BOS (offset 0) only [SEG] survives — the <image>
    # placeholder is lost.  Reproduced as is: the id order is part of the splice's parity contract.
    chunks = _chunk_ids(prompt, tokenizer, "<seg>\n<image>")
    bos = getattr(tokenizer, "bos_token_id", None)
    has_bos = bool(chunks) and bool(chunks[0]) and chunks[0][0] == bos
    sep = [image_token_index, seg_token_index] if has_bos else [seg_token_index]
    return _ret(_join_with(chunks, sep, tokenizer), return_tensors)


def _tokenizer_depth_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, seg_token_index=SEG_TOKEN_INDEX,
                           depth_token_index=DEPTH_TOKEN_INDEX, return_tensors=None):
    ids = _join_with(_chunk_ids(prompt, tokenizer, "<depth>\n<seg>\n<image>"),
                     [image_token_index, depth_token_index, seg_token_index], tokenizer)
    return _ret(ids, return_tensors)


def tokenizer_depth_seg_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, seg_token_index=SEG_TOKEN_INDEX,
                              depth_token_index=DEPTH_TOKEN_INDEX, return_tensors=None):
    if "<depth>" in prompt:
        return _tokenizer_depth_token(prompt, tokenizer, image_token_index, seg_token_index, depth_token_index,
                                      return_tensors)
    return tokenizer_seg_token(prompt, tokenizer, image_token_index, seg_token_index, return_tensors)


def _ret(ids, return_tensors):
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        import torch

        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


def expand2square(pil_img, background_color):
    from PIL import Image

    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    canvas = Image.new(pil_img.mode, (side, side), background_color)
    canvas.paste(pil_img, ((side - w) // 2, (side - h) // 2))
    return canvas


def process_images(images, image_processor, model_cfg):
    import torch

    if getattr(model_cfg, "image_aspect_ratio", None) != "pad":
        return image_processor(images, return_tensors="pt")["pixel_values"]
    fill = tuple(int(x * 255) for x in image_processor.image_mean)
    out = [image_processor.preprocess(expand2square(im, fill), return_tensors="pt")["pixel_values"][0] for im in images]
    if all(x.shape == out[0].shape for x in out):
        out = torch.stack(out, dim=0)
    return out


def get_model_name_from_path(model_path: str) -> str:
    parts = model_path.strip("/").split("/")
    return parts[-2] + "_" + parts[-1] if parts[-1].startswith("checkpoint-") else parts[-1]


def process_images_device(images, model, model_cfg=None, to_device: bool = True):
    """process_images() with the resize / pad / normalise work done by libvcoder_hip (vc_preprocess_image) instead of
    PIL + numpy on the host: PIL-exact bicubic, same expand2square fill colour, same rescale/normalise arithmetic.
    `model` is a vcoder_amd model (or engine); returns fp32 [N,3,S,S] on the model's GPU."""
    engine = getattr(model, "engine", model)
    cfg = model_cfg if model_cfg is not None else getattr(model, "config", None)
    pad = getattr(cfg, "image_aspect_ratio", None) == "pad"
    proc = getattr(getattr(model, "get_vision_tower", lambda: None)(), "image_processor", None)
    mean = getattr(proc, "image_mean", None)
    std = getattr(proc, "image_std", None)
    return engine.preprocess(list(images), pad=pad, mean=mean, std=std, to_device=to_device)



class KeywordsStoppingCriteria:
    """Counterpart of vcoder_llava/mm_utils.py:128-151 with batch > 1 support.

    Reference behaviour kept for batch size 1: stop when the sequence ends with the ids of a keyword, or when the text of
    the last <= 3 generated tokens (special tokens skipped) contains a keyword.  Differences, both documented: (i) any
    batch size — the call returns True only when EVERY row has met a keyword (rows are checked independently);
    (ii) multi-token keywords are compared element-wise (the reference's `if tensor == tensor:` raises for them).

    `device_stop_sequences()` returns the id sequences when the id match alone decides (every keyword is a single special
    token, which the text check can never see because it skips special tokens) — `generate()` then runs the stop on the
    device inside the hipGraph-replayed decode loop (vc_generate_greedy_stop) instead of syncing with the host every token."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = list(keywords)
        self.keyword_ids = []
        for keyword in self.keywords:
            cur = list(tokenizer(keyword).input_ids)
            if len(cur) > 1 and cur[0] == tokenizer.bos_token_id:
                cur = cur[1:]
            self.keyword_ids.append(cur)
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def _row_done(self, row) -> bool:
        ids = [int(t) for t in row]
        for kw in self.keyword_ids:
            if len(kw) <= len(ids) and ids[len(ids) - len(kw):] == kw:
                return True
        offset = min(len(ids) - self.start_len, 3)
        if offset > 0:
            text = self.tokenizer.batch_decode([ids[-offset:]], skip_special_tokens=True)[0]
            if any(k in text for k in self.keywords):
                return True
        return False

    def __call__(self, output_ids, scores=None, **kwargs) -> bool:
        return all(self._row_done(row) for row in output_ids)

    def device_stop_sequences(self):
        special = set(getattr(self.tokenizer, "all_special_ids", []) or [])
        if self.keyword_ids and all(len(kw) == 1 and kw[0] in special for kw in self.keyword_ids) and len(self.keyword_ids) <= 8:
            return [list(kw) for kw in self.keyword_ids]
        return None
