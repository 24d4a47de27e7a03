"""Batched COST generation harness — the consumer of the data-parallel throughput.

Counterpart of vcoder_llava/eval/model_seg_loader.py:35-166 / model_depth_loader.py:36-185 and of the per-GPU process
loop in scripts/v1_5/eval/cost*.sh: same dataset chunking (`--num-chunks / --chunk-idx`, ceil-sized contiguous chunks),
same prompt (`<seg>\\n<image>\\n{question}` in the llava_v1 template; `<depth>\\n` prepended for the depth task), same
answer-file format (`Image: / <<QUESTION>>: / <<ANSWER>>:` blocks in `{output}_{task}_{num_chunks}_{chunk_idx}.txt`).

What changes: the reference asserts batch_size == 1 (model_seg_loader.py:93).  Here samples are bucketed by question —
identical prompt text tokenises to identical ids, so a bucket splices to EQUAL lengths, which is exactly the condition
under which batched rows are independent — and each bucket is generated B samples at a time with
the device-side greedy loop; images are preprocessed on the GPU (vc_preprocess_image)."""
from __future__ import annotations

import glob
import math
import os
import random
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .. import mm_utils
from ..constants import DEFAULT_DEPTH_TOKEN, DEFAULT_IMAGE_TOKEN, DEFAULT_SEG_TOKEN

# the two templates the reference's eval scripts use (vcoder_conversation.py:340-362): SeparatorStyle.TWO, sep " ", sep2 "</s>"
_SYSTEM = {
    "llava_v1": "A chat between a curious human and an artificial intelligence assistant. "
                "The assistant gives helpful, detailed, and polite answers to the human's questions.",
    "vicuna_v1": "A chat between a curious user and an artificial intelligence assistant. "
                 "The assistant gives helpful, detailed, and polite answers to the user's questions.",
}
STOP_STR = "</s>"
# appended to the question when a sample has no segmentation map (model_seg_loader.py:73) ...
PARAGRAPH_INSTRUCTION = (" Return the answer in the paragraph format: 'The objects present in the image are: ...' and then "
                         "list the objects with their count in word format (if greater than 1) in front of them, like "
                         "'two people'.")
# ... and by the depth loader when it runs without --use_depth_seg (model_depth_loader.py:90)
DEPTH_PARAGRAPH_INSTRUCTION = (' Return answer in the paragraph format: "The depth order for the objects present in the image '
                               'is: ..." and then list the objects with their order number (if greater than 1) separated by a '
                               'hyphen like "person-2". For example, an acceptable response is "The depth order for objects '
                               'present in the image is: bicycle, bicycle-2, bicycle-3, pavement, road, bus, tree, sky, building."')


def build_prompt(question: str, conv_mode: str = "llava_v1") -> str:
    """conv.append_message(USER, q); conv.append_message(ASSISTANT, None); conv.get_prompt()"""
    return _SYSTEM[conv_mode] + " " + "USER: " + question + " " + "ASSISTANT:"


def split_list(lst: Sequence, n: int) -> List[list]:
    size = math.ceil(len(lst) / n) if len(lst) else 1
    return [list(lst[i:i + size]) for i in range(0, len(lst), size)]


def get_chunk(lst: Sequence, n: int, k: int) -> list:
    chunks = split_list(lst, n)
    return chunks[k] if k < len(chunks) else []


@dataclass
class Sample:
    image_file: str
    seg_file: Optional[str]
    depth_file: Optional[str]
    question: str


def load_questions(task: str) -> List[str]:
    """The question banks are the reference's data (vcoder_llava/questions.py); they are read from the reference package
    when it is importable (it is unchanged glue), otherwise the caller must pass `questions=`."""
    try:
        from vcoder_llava.questions import QUESTIONS  # type: ignore

        return list(QUESTIONS[task])
    except Exception as e:
        raise RuntimeError(f"no question bank for task '{task}': pass questions=[...] or put the reference's "
                           f"vcoder_llava package on sys.path ({e})")


def build_samples(image_folder: str, seg_folder: Optional[str], depth_folder: Optional[str], questions: Sequence[str],
                  num_chunks: int = 1, chunk_idx: int = 0, seed: Optional[int] = None, pattern: str = "*.jpg") -> List[Sample]:
    """One sample per image of this rank's chunk; the question is drawn at random per sample like the reference
    (model_seg_loader.py: random.choice)."""
    rng = random.Random(seed)
    images = get_chunk(sorted(glob.glob(os.path.join(image_folder, pattern))), num_chunks, chunk_idx)
    segs = get_chunk(sorted(glob.glob(os.path.join(seg_folder, pattern))), num_chunks, chunk_idx) if seg_folder else None
    deps = get_chunk(sorted(glob.glob(os.path.join(depth_folder, pattern))), num_chunks, chunk_idx) if depth_folder else None
    if segs is not None and len(segs) != len(images):
        raise AssertionError(f"Number of images ({len(images)}) and seg images ({len(segs)}) must be the same")
    if deps is not None and len(deps) != len(images):
        raise AssertionError(f"Number of images ({len(images)}) and depth images ({len(deps)}) must be the same")
    return [Sample(im, segs[i] if segs else None, deps[i] if deps else None, rng.choice(list(questions)))
            for i, im in enumerate(images)]


def _load_rgb(path: str):
    from PIL import Image

    return Image.open(path).convert("RGB")


def _eos_ids(tokenizer, model) -> List[int]:
    """EOS id(s) of the run: the tokenizer's, else the config's; HF's GenerationConfig takes one id or a list"""
    eos = getattr(tokenizer, "eos_token_id", None)
    eos = model.config.eos_token_id if eos is None else eos
    if eos is None:
        return []
    return [int(e) for e in (eos if isinstance(eos, (list, tuple)) else [eos])]


def generate_answers(model, tokenizer, samples: Sequence[Sample], batch_size: int = 8, max_new_tokens: int = 512,
                     conv_mode: str = "llava_v1", load_image: Callable = _load_rgb, pixels_on_device: bool = True,
                     task: str = "semantic") -> List[str]:
    """Greedy answers for `samples`, in order.  Buckets by prompt so every batch splices to equal lengths.  task == "depth":
    the depth loader's wording for samples without maps (model_depth_loader.py:90)."""
    engine = model.engine
    buckets: Dict[str, List[int]] = {}
    prompts = []
    for i, s in enumerate(samples):
        q = s.question
        if s.seg_file is not None:
            q = DEFAULT_SEG_TOKEN + "\n" + DEFAULT_IMAGE_TOKEN + "\n" + q
            if s.depth_file is not None:
                q = DEFAULT_DEPTH_TOKEN + "\n" + q
        else:
            q = DEFAULT_IMAGE_TOKEN + "\n" + q + (DEPTH_PARAGRAPH_INSTRUCTION if task == "depth" else PARAGRAPH_INSTRUCTION)
        p = build_prompt(q, conv_mode)
        prompts.append(p)
        buckets.setdefault(p, []).append(i)
    answers: List[Optional[str]] = [None] * len(samples)
    eos_ids = _eos_ids(tokenizer, model)
    # one EOS id goes to the device loop as such; further ids of an EOS list end a row the same way as single-token stop
    # sequences (the token is kept, the row pads afterwards — what HF's generate does for every id of the list)
    eos, more = (eos_ids[0] if eos_ids else None), [[e] for e in eos_ids[1:]]
    for prompt, idxs in buckets.items():
        if "<seg>" in prompt:
            ids = mm_utils.tokenizer_depth_seg_token(prompt, tokenizer)
        else:
            ids = mm_utils.tokenizer_image_token(prompt, tokenizer)
        ids = np.asarray(ids, dtype=np.int64)
        for b0 in range(0, len(idxs), batch_size):
            group = idxs[b0:b0 + batch_size]
            B = len(group)
            imgs = mm_utils.process_images_device([load_image(samples[i].image_file) for i in group], model, to_device=pixels_on_device)
            segs = deps = None
            if samples[group[0]].seg_file is not None:
                segs = mm_utils.process_images_device([load_image(samples[i].seg_file) for i in group], model, to_device=pixels_on_device)
            if samples[group[0]].depth_file is not None:
                deps = mm_utils.process_images_device([load_image(samples[i].depth_file) for i in group], model, to_device=pixels_on_device)
            new = engine.generate_greedy(np.tile(ids, (B, 1)), imgs, segs, deps, max_new_tokens=max_new_tokens,
                                         eos_token_id=eos, pad_token_id=model.config.pad_token_id, stop_sequences=more or None)
            texts = tokenizer.batch_decode(new.tolist(), skip_special_tokens=True)
            for i, t in zip(group, texts):
                t = t.strip()
                if t.endswith(STOP_STR):
                    t = t[:-len(STOP_STR)]
                answers[i] = t.strip().strip("\n")
    return answers  # type: ignore[return-value]


def write_answers(path: str, samples: Sequence[Sample], answers: Sequence[str]) -> None:
    """Appends in the reference's format (model_seg_loader.py:160-166)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "a") as f:
        for s, a in zip(samples, answers):
            # the scorers key predictions by the bare file name (model_seg_loader.py:85 `image_file.split("/")[-1]`;
            # eval_seg_accuracy.py:165, eval_depth_accuracy.py:42)
            f.write(f"Image: {s.image_file.split('/')[-1]}\n")
            f.write(f"<<QUESTION>>: {s.question}\n")
            f.write(f"<<ANSWER>>: {a}\n")
            f.write("-------------------------------------------------------\n")


def eval_task(model, tokenizer, task: str, image_folder: str, seg_image_folder: Optional[str], output_file: str,
              depth_image_folder: Optional[str] = None, num_chunks: int = 1, chunk_idx: int = 0, batch_size: int = 8,
              questions: Optional[Sequence[str]] = None, conv_mode: str = "llava_v1", max_new_tokens: int = 512,
              seed: Optional[int] = None, pixels_on_device: bool = True) -> str:
    """One COST task for this rank's chunk; returns the answers file.

    task in {"semantic", "instance", "panoptic"}: model_seg_loader.py:99-166 — maps from `{seg_image_folder}/{task}_inference`,
    file `{output}_{task}_{num_chunks}_{chunk_idx}.txt`.  task == "depth": model_depth_loader.py:116-185 — maps from
    `{seg_image_folder}/panoptic_inference` + `depth_image_folder` (both or neither, :54), file
    `{output}_{num_chunks}_{chunk_idx}.txt`.  seg_image_folder None = the loaders without --use_seg / --use_depth_seg."""
    questions = list(questions) if questions is not None else load_questions(task)
    if task == "depth":
        if seg_image_folder and not depth_image_folder:
            raise ValueError("Depth image folder must be provided if seg image folder is provided")
        seg_folder = os.path.join(seg_image_folder, "panoptic_inference") if seg_image_folder else None
        depth_folder = depth_image_folder if seg_image_folder else None
        out = os.path.expanduser(output_file) + f"_{num_chunks}_{chunk_idx}.txt"
    else:
        seg_folder = os.path.join(seg_image_folder, f"{task}_inference") if seg_image_folder else None
        depth_folder = None   # the seg loader passes depths=None (model_seg_loader.py:133)
        out = os.path.expanduser(output_file) + f"_{task}_{num_chunks}_{chunk_idx}.txt"
    samples = build_samples(image_folder, seg_folder, depth_folder, questions, num_chunks, chunk_idx, seed)
    answers = generate_answers(model, tokenizer, samples, batch_size, max_new_tokens, conv_mode,
                               pixels_on_device=pixels_on_device, task=task)
    if os.path.exists(out):
        os.remove(out)
    write_answers(out, samples, answers)
    return out
