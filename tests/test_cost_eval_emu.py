"""Batched COST harness on the emulator: bucketing-by-question batches give exactly the per-sample
(batch-1, what the reference does) answers; chunking and the answers-file format follow the reference's loader; the
prompt string equals the reference's conversation template when the reference tree is present."""
import os

import numpy as np
import pytest
from PIL import Image

import kernel_cases as kc
import ref_shim
from vcoder_amd import config as vcfg, synth
from vcoder_amd.eval import cost_eval as ce
from vcoder_amd.model import language_model as lm


class FakeTokenizer:
    bos_token_id, eos_token_id = 1, 2

    def __call__(self, text):
        class R:
            pass
        r = R()
        r.input_ids = [1] + [3 + (ord(c) * 7) % 300 for c in text][:12]   # short prompts keep the emulator fast
        return r

    def batch_decode(self, rows, skip_special_tokens=True):
        return [" ".join(f"t{t}" for t in r if not (skip_special_tokens and t in (0, 1, 2))) for r in rows]


@pytest.fixture(scope="module")
def model():
    cfg = vcfg.tiny("vcoder_ds")
    m = lm.VCoderDSLlavaLlamaForCausalLM(cfg, device="cuda", _lib_override=kc.EmuBackend().lib)
    m.engine.load_synthetic(42)
    m.finalize_weights()
    return m


@pytest.fixture(scope="module")
def folders(tmp_path_factory):
    root = tmp_path_factory.mktemp("cost")
    rng = np.random.RandomState(0)
    for sub in ("images", "segs/semantic_inference", "depths"):
        os.makedirs(root / sub)
        for i in range(5):
            Image.fromarray(rng.randint(0, 256, size=(40 + 3 * i, 64, 3)).astype(np.uint8)).save(root / sub / f"{i:03d}.jpg")
    return root


def test_chunking_matches_reference_semantics():
    assert ce.split_list(list(range(10)), 4) == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9]]
    assert ce.get_chunk(list(range(10)), 4, 3) == [9] and ce.get_chunk(list(range(3)), 4, 3) == []


def test_batched_equals_per_sample_and_file_format(model, folders, tmp_path):
    qs = ["What objects are here?", "List the things."]
    tok = FakeTokenizer()
    samples = ce.build_samples(str(folders / "images"), str(folders / "segs/semantic_inference"), str(folders / "depths"),
                               qs, seed=3)
    assert len(samples) == 5 and {s.question for s in samples} <= set(qs)
    batched = ce.generate_answers(model, tok, samples, batch_size=4, max_new_tokens=3, pixels_on_device=False)
    single = ce.generate_answers(model, tok, samples, batch_size=1, max_new_tokens=3, pixels_on_device=False)
    assert batched == single and all(a for a in batched)
    out = ce.eval_task(model, tok, "semantic", str(folders / "images"), str(folders / "segs"), str(tmp_path / "out/answers"),
                       depth_image_folder=str(folders / "depths"), num_chunks=2, chunk_idx=1, batch_size=4, questions=qs,
                       max_new_tokens=3, seed=3, pixels_on_device=False)
    assert out.endswith("answers_semantic_2_1.txt")
    lines = open(out).read().splitlines()
    assert len(lines) == 4 * 2 and lines[0].startswith("Image: ") and lines[1].startswith("<<QUESTION>>: ")
    assert lines[2].startswith("<<ANSWER>>: ") and set(lines[3]) == {"-"}


@pytest.mark.reference
@pytest.mark.skipif(not ref_shim.reference_available(), reason="needs /root/reference (build container only)")
def test_prompt_equals_reference_template():
    ref_shim.load_reference()
    from vcoder_llava.vcoder_conversation import conv_templates

    for mode in ("llava_v1", "vicuna_v1"):
        conv = conv_templates[mode].copy()
        q = "<depth>\n<seg>\n<image>\nWhat is in the image?"
        conv.append_message(conv.roles[0], q)
        conv.append_message(conv.roles[1], None)
        assert ce.build_prompt(q, mode) == conv.get_prompt()


def test_answers_equal_reference_loaders(tmp_path):
    """The committed answers files are what the REFERENCE'S loaders wrote (oracle/gen_cost_golden.py); the batched harness
    reproduces them byte for byte (emulator, split mode; all six runs also under -m gpu)."""
    import e2e_cases

    done = e2e_cases.check_cost_answers(str(tmp_path), lib=kc.EmuBackend().lib, mode="split", pixels_on_device=False,
                                        runs=["semantic_1_0", "depth_2_0", "depth_noseg_1_0", "panoptic_2_1"])
    assert done == {"semantic_1_0": 6, "depth_2_0": 3, "depth_noseg_1_0": 6, "panoptic_2_1": 3}
