"""TEST INFRASTRUCTURE — COST answers fixture: runs the REFERENCE'S OWN eval loaders

    vcoder_llava/eval/model_seg_loader.py:99-166   (eval_model(args, task): dataset -> prompt -> batch-1 greedy -> answers file)
    vcoder_llava/eval/model_depth_loader.py:116-185 (eval_model(args))

unmodified, on CPU fp32, over a small synthetic COST-shaped folder with the tiny VCoder-DS checkpoint and the fake tokenizer
(tests/fake_tokenizer.py), and commits what they wrote as tests/golden/cost/answers_*.txt together with the input images
(data only).  tests/test_cost_eval_emu.py / tests/test_gpu_e2e.py then require vcoder_amd.eval.cost_eval — batched, device
preprocessing, device greedy loop — to produce the same files byte for byte.

Build container only (needs /root/reference):   python oracle/gen_cost_golden.py

What the harness supplies around the reference's code, and why (nothing in the reference tree is touched):
  * `load_pretrained_model` of the loader modules -> the reference's VCoderDSLlavaLlamaForCausalLM holding the synthetic tiny
    checkpoint + the fake tokenizer + the HF CLIPImageProcessor of the tiny tower (no checkpoint / tokenizer exists offline);
  * `model.generate` -> a greedy loop over the reference model's own `forward` with HF's EOS semantics:
    the reference's generate() dies under the installed Transformers 5.x at vcoder_ds_llava_arch.py:132;
  * `.to(device='cuda', dtype=float16)` of the loaders' tensors -> no-op: this is the reference's CPU fp32 path;
  * `glob.glob` sorted and the DataLoader without worker processes + `random.seed`: the loaders take files in directory order
    and draw the question with the global `random` inside worker processes — neither is reproducible as written;
  * the module-global `args` (the loaders' `__main__` block defines it; `CustomDataset.__getitem__` reads it);
  * `shortuuid` (imported by the loaders, never used, not installed): an empty module.
EOS: a random tiny model never emits id 2, and 512 free-running greedy steps per image hold numerical near-ties no two fp32
evaluations agree on (measured: 31 of 15 360 decisions with a top-2 margin below 1e-4, the smallest 3e-6).  The fixture's config
therefore carries a LIST of `eos_token_id`s (HF's GenerationConfig takes one) chosen so that every answer ends after >= 1 token
and before its first decision with a margin below MIN_MARGIN; the generator asserts that margin on every decision it commits —
"byte for byte" is then a statement about the harness, not about luck."""
from __future__ import annotations

import argparse
import glob as _glob
import json
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_shim  # noqa: E402
import gen_golden  # noqa: E402
from fake_tokenizer import FakeTokenizer  # noqa: E402
from vcoder_amd import config as vcfg  # noqa: E402
from vcoder_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "cost")
N_IMAGES = 6
QUESTION_SEED = 11
WEIGHT_SEED = 42
MIN_MARGIN = 1e-3       # every greedy decision of the fixture; split / strict mode are ~1e-5 from the fp32 reference here
DRY_STEPS = 160
MAX_EOS = 6             # the engine takes one EOS id + up to 8 device-side stop sequences
MAX_POS = 1024          # prompt + the loaders' hard-coded max_new_tokens = 512 must fit the engine's position cap


def make_folder(root):
    """COST-shaped inputs: photo-like noise, piecewise-constant 'panoptic maps', smooth grey 'depth maps'; odd sizes so that
    expand2square and the bicubic resize both do work.  JPEG files: the loaders glob '*.jpg'."""
    from PIL import Image

    rng = np.random.RandomState(5)
    for sub in ("images", "segs/semantic_inference", "segs/panoptic_inference", "depths"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    for i in range(N_IMAGES):
        h, w = 48 + 5 * i, 72 - 4 * i
        photo = rng.randint(0, 256, size=(h // 4 + 1, w // 4 + 1, 3)).astype(np.uint8)
        photo = np.asarray(Image.fromarray(photo).resize((w, h), Image.BILINEAR))

        def blocks():
            cols = rng.randint(0, 256, size=(4, 3)).astype(np.uint8)
            lab = (np.arange(h)[:, None] * 3 // h) + (np.arange(w)[None, :] * 2 // w) * 2
            return cols[lab % 4]

        depth = np.clip(np.linspace(30 + 10 * i, 220, h)[:, None] + np.linspace(0, 20, w)[None, :], 0, 255).astype(np.uint8)
        depth = np.repeat(depth[:, :, None], 3, axis=2)
        for sub, arr in (("images", photo), ("segs/semantic_inference", blocks()), ("segs/panoptic_inference", blocks()),
                         ("depths", depth)):
            Image.fromarray(arr).save(os.path.join(root, sub, f"{i:06d}.jpg"), quality=92)


class RefGenerate:
    """what the loaders call `model`: .config + .generate(...) -> cat(input_ids, new ids), greedy, stops after EOS"""

    def __init__(self, model, eos_token_ids):
        self.model, self.config, self.eos = model, model.config, set(eos_token_ids or ())
        self.answers = []   # per generate() call: (new token ids, top-2 margin of every decision)

    @torch.no_grad()
    def generate(self, input_ids, images=None, segs=None, depths=None, do_sample=False, temperature=0.0, top_p=None,
                 num_beams=1, max_new_tokens=512, use_cache=True):
        assert not do_sample and num_beams == 1 and input_ids.shape[0] == 1
        cur = input_ids.clone()
        toks, margins = [], []
        for _ in range(max_new_tokens):
            out = self.model(input_ids=cur, use_cache=False, images=images.float(), segs=None if segs is None else segs.float(),
                             depths=None if depths is None else depths.float())
            last = out.logits[:, -1].float()
            s = torch.sort(last[0]).values
            margins.append(float(s[-1] - s[-2]))
            nxt = last.argmax(-1)
            toks.append(int(nxt[0]))
            cur = torch.cat([cur, nxt[:, None]], 1)
            if toks[-1] in self.eos:   # HF: the EOS token is appended, then the (batch-1) loop ends
                break
        self.answers.append((toks, margins))
        return cur


def run_reference_loaders(folder, out_dir, model, tok, proc, eos, max_new_cap=None):
    """-> {fixture name: answers text}; drives both loader modules' eval_model with the patches of the module docstring"""
    ref_shim.load_reference()
    sys.modules.setdefault("shortuuid", types.ModuleType("shortuuid"))
    import vcoder_llava.eval.model_depth_loader as depth_loader
    import vcoder_llava.eval.model_seg_loader as seg_loader
    from torch.utils.data import DataLoader

    wrapper = RefGenerate(model, eos)
    if max_new_cap is not None:   # the EOS search only: bound the dry run
        g0 = wrapper.generate
        wrapper.generate = lambda *a, **k: g0(*a, **{**k, "max_new_tokens": max_new_cap})
    sorted_glob = types.SimpleNamespace(glob=lambda p: sorted(_glob.glob(p)))
    orig_to = torch.Tensor.to

    def to_cpu_fp32(self, *a, **k):
        if k.get("device") == "cuda":
            return self
        return orig_to(self, *a, **k)

    results = {}
    torch.Tensor.to = to_cpu_fp32
    try:
        for mod in (seg_loader, depth_loader):
            mod.load_pretrained_model = lambda *a, **k: (tok, wrapper, proc, proc, proc, 2048)
            mod.glob = sorted_glob
            mod.DataLoader = lambda ds, batch_size=1, num_workers=0, shuffle=False: DataLoader(ds, batch_size=batch_size,
                                                                                                 num_workers=0, shuffle=shuffle)
            mod.tqdm = lambda it, **k: it
        base = dict(model_path="shi-labs/vcoder_ds_llava-v1.5-tiny", model_base=None, image_folder=os.path.join(folder, "images"),
                    seg_image_folder=os.path.join(folder, "segs"), conv_mode="vicuna_v1", temperature=0, top_p=None, num_beams=1)
        runs = [  # (fixture name, module, task, extra args)
            ("semantic_1_0", seg_loader, "semantic", dict(use_seg=True, num_chunks=1, chunk_idx=0)),
            ("panoptic_2_1", seg_loader, "panoptic", dict(use_seg=True, num_chunks=2, chunk_idx=1)),
            ("semantic_noseg_1_0", seg_loader, "semantic", dict(use_seg=False, num_chunks=1, chunk_idx=0, conv_mode="llava_v1")),
            ("depth_1_0", depth_loader, None, dict(use_depth_seg=True, depth_image_folder=os.path.join(folder, "depths"),
                                                   num_chunks=1, chunk_idx=0)),
            ("depth_2_0", depth_loader, None, dict(use_depth_seg=True, depth_image_folder=os.path.join(folder, "depths"),
                                                   num_chunks=2, chunk_idx=0)),
            ("depth_noseg_1_0", depth_loader, None, dict(use_depth_seg=False, depth_image_folder="", num_chunks=1, chunk_idx=0)),
        ]
        for name, mod, task, extra in runs:
            d = os.path.join(out_dir, name)
            a = argparse.Namespace(**{**base, **extra, "output_file": os.path.join(d, "answers")})
            random.seed(QUESTION_SEED)
            mod.args = a   # CustomDataset.__getitem__ reads the module-global `args` the loaders' __main__ block defines
            if task is None:
                mod.eval_model(a)
            else:
                mod.eval_model(a, task)
            files = sorted(_glob.glob(os.path.join(d, "*.txt")))
            assert len(files) == 1, files
            results[name] = (os.path.basename(files[0]), open(files[0]).read())
    finally:
        torch.Tensor.to = orig_to
    return results, wrapper


def main():
    cfg = vcfg.tiny("vcoder_ds")
    cfg.max_position_embeddings = MAX_POS
    cfg.image_aspect_ratio = "pad"    # what the v1.5 checkpoints carry (process_images: expand2square, mm_utils.py:28-40)
    sd = synth.synth_state_dict(cfg, WEIGHT_SEED)
    work = tempfile.mkdtemp()
    clip_dir = os.path.join(work, "clip")
    os.makedirs(clip_dir)
    gen_golden.make_clip_dir(cfg, clip_dir)
    model = gen_golden.build_reference_model(cfg, sd, clip_dir)
    model.config.image_aspect_ratio = "pad"
    proc = model.get_vision_tower().image_processor
    folder = os.path.join(work, "cost")
    make_folder(folder)
    # ---- pick the EOS ids: a dry run without EOS, then a small set of tokens such that every answer ends (i) after at least one
    # token and (ii) before its first greedy decision with a top-2 margin below MIN_MARGIN
    tok = FakeTokenizer(cfg.vocab_size, eos_token_id=2)
    res, w = run_reference_loaders(folder, os.path.join(work, "dry"), model, tok, proc, eos=None, max_new_cap=DRY_STEPS)
    ans = w.answers
    first_bad = [next((i for i, m in enumerate(mg) if m < MIN_MARGIN), len(mg)) for _, mg in ans]

    def first_hit(toks, chosen):
        return next((i for i, t in enumerate(toks) if t in chosen), None)

    def ok(chosen):
        hits = [first_hit(t, chosen) for t, _ in ans]
        return [h is not None and 1 <= h < fb for h, fb in zip(hits, first_bad)], hits

    if os.environ.get("COST_DEBUG"):
        for (t, mg), fb in zip(ans, first_bad):
            print(fb, t[:fb + 1])
    chosen = set()
    while not all(ok(chosen)[0]):
        assert len(chosen) < MAX_EOS, f"no set of {MAX_EOS} stop tokens ends every answer before a near-tie: change WEIGHT_SEED"
        best = None
        for cand in sorted({t for toks, _ in ans for t in toks} - chosen):
            good, hits = ok(chosen | {cand})
            if any(h == 0 for h in hits):   # an empty answer cannot be repaired by a later choice (a late hit can)
                continue
            score = (sum(good), -max((h for h in hits if h is not None), default=0))
            if best is None or score > best[0]:
                best = (score, cand)
        assert best is not None, "no admissible stop token: change WEIGHT_SEED"
        chosen.add(best[1])
    eos = sorted(chosen)
    print(f"eos_token_id = {eos}")
    tok = FakeTokenizer(cfg.vocab_size, eos_token_id=eos)
    res, w = run_reference_loaders(folder, os.path.join(work, "real"), model, tok, proc, eos=eos)
    lengths = [len(t) for t, _ in w.answers]
    margins = [m for _, mg in w.answers for m in mg]
    print(f"{len(lengths)} answers, {len(margins)} greedy decisions, lengths {min(lengths)}..{max(lengths)}, "
          f"min top-2 margin {min(margins):.3e}")
    assert min(margins) >= MIN_MARGIN and all(t[-1] in chosen for t, _ in w.answers)
    # ---- commit: images (data), answers, meta
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    shutil.copytree(folder, os.path.join(OUT, "folder"))
    for name, (fname, text) in res.items():
        with open(os.path.join(OUT, f"answers_{name}.txt"), "w") as f:
            f.write(text)
    from vcoder_llava.questions import DEPTH_QUESTIONS, QUESTIONS

    # The loaders draw `random.choice(bank)`: the fixture keeps each bank's LENGTH and the entries that were drawn (they are in
    # the answers files anyway) — enough for the test to rebuild a bank that gives the same draws from the same seed; the
    # reference's question file itself is not copied.
    banks = {}
    for name, (_, text) in res.items():
        task = "depth" if name.startswith("depth") else name.split("_")[0]
        bank = DEPTH_QUESTIONS if task == "depth" else QUESTIONS[task]
        b = banks.setdefault(task, {"len": len(bank), "drawn": {}})
        for line in text.splitlines():
            if line.startswith("<<QUESTION>>: "):
                q = line[len("<<QUESTION>>: "):]
                b["drawn"][str(bank.index(q))] = q
    meta = {"eos_token_id": eos, "weight_seed": WEIGHT_SEED, "question_seed": QUESTION_SEED, "max_position_embeddings": MAX_POS,
            "image_aspect_ratio": "pad", "min_top2_margin": min(margins), "max_answer_tokens": max(lengths),
            "file_names": {k: v[0] for k, v in res.items()}, "question_banks": banks}
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    shutil.rmtree(work)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
