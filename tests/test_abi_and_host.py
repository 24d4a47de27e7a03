"""CPU tests: the C-ABI library loads and exports every symbol the headers declare (no compute calls without a GPU),
config / synthetic-checkpoint host logic, and the splice planner (through the emulator build of the same engine
source) against the oracle's independent restatement."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import cpu_ref
import kernel_cases as kc
from vcoder_amd import _lib, config as vcfg, synth
from vcoder_amd.engine import HipEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vck?_[a-z0-9_]+)\s*\(", src)))


def test_product_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from vcoder_amd import build

        build.build(verbose=False)
    lib = _lib.load()          # dlopen works without a GPU
    names = _declared("vcoder_hip.h") + _declared("vcoder_kernels.h")
    assert len(names) > 35
    for n in names:
        assert hasattr(lib, n), f"libvcoder_hip.so does not export {n}"


def test_no_cpu_fallback_without_gpu():
    """The product must fail loudly, never compute on the CPU."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        HipEngine(vcfg.tiny("vcoder_ds"))
    # and nothing in the product package imports the oracle or the emulator
    out = subprocess.run(["grep", "-rlE", "cpu_ref|libvcoder_emu|hip_emu", os.path.join(ROOT, "vcoder_amd"),
                          "--include=*.py"], capture_output=True, text=True).stdout.strip()
    assert out == "", f"product code references test infrastructure: {out}"


def test_config_roundtrip_and_select_layer():
    cfg = vcfg.vicuna_7b("vcoder_ds")
    assert cfg.vit_layers_used == 23 and cfg.num_patches == 576 and cfg.head_dim == 128
    d = cfg.to_hf_dict()
    assert d["model_type"] == "vcoder_ds_llava"
    back = vcfg.VCoderConfig.from_hf_dict(d)
    assert back.hidden_size == 4096 and back.vit_num_layers == 24 and back.variant == "vcoder_ds"
    assert vcfg.VCoderConfig.from_hf_dict({**d, "model_type": None}, "my-vcoder_llava-7b").variant == "vcoder"
    with pytest.raises(ValueError):
        vcfg.VCoderConfig(mm_vision_select_feature="bogus").validate()
    assert getattr(back, "image_aspect_ratio", None) == "pad"


def test_synth_generator_is_bf16_exact_and_stable():
    a = synth.synth_tensor("model.layers.0.mlp.up_proj.weight", (4, 8), 42, 0.0, 0.02 * 3 ** 0.5)
    assert np.array_equal(a, synth.round_to_bf16(a))
    # frozen values: any change to the generator silently invalidates tests/golden
    assert a.view(np.uint32)[0, 0] == synth.synth_tensor("model.layers.0.mlp.up_proj.weight", (1,), 42, 0.0,
                                                         0.02 * 3 ** 0.5).view(np.uint32)[0]
    ids = synth.synth_prompt_ids(32000, "vcoder_ds")
    assert ids.shape == (67,) and list(ids[35:38]) == [-200, -400, -300] and ids[0] == 1
    sd = synth.synth_state_dict(vcfg.tiny("vcoder_ds"), 42)
    assert "model.depth_mm_projector.0.weight" in sd and "model.vcoder_lm_emb.weight" in sd


# ---- splice planner of the engine (C++, vc_prefill_embeds_only) vs the oracle's plan --------------------------
@pytest.fixture(scope="module")
def emu_engines():
    lib = kc.EmuBackend().lib
    out = {}
    for v in ("vcoder_ds", "vcoder", "llava"):
        cfg = vcfg.tiny(v)
        e = HipEngine(cfg, lib=lib)
        e.load_synthetic(42)
        e.finalize()
        out[v] = (e, cpu_ref.OracleModel(cfg, synth.synth_state_dict(cfg, 42), emu_bf16=True))
    return out


I, S, D = synth.IMAGE_TOKEN_INDEX, synth.SEG_TOKEN_INDEX, synth.DEPTH_TOKEN_INDEX
CASES = [
    ("vcoder_ds", [[1, 5, I, D, S, 6, 7]], True, True),
    ("vcoder_ds", [[1, 5, I, S, D, 6, 7]], True, True),          # hand order: depth IS spliced
    ("vcoder_ds", [[1, 5, I, 8, 9, S, 6, 7]], True, True),       # text between <image> and <seg> dropped
    ("vcoder_ds", [[1, 5, I, 6, 7]], False, False),
    ("vcoder_ds", [[1, 5, 6, 7, 8, 9]], True, True),             # no placeholders: ZeRO-3 branch
    ("vcoder_ds", [[1, I, S, 6], [1, 4, I, S]], True, False),    # batch of 2, equal lengths
    ("vcoder", [[1, 5, I, 8, 9, S, 6, 7]], True, False),         # non-DS keeps the text
    ("llava", [[1, 5, I, 6, 7, 8]], False, False),
    ("llava", [[1, 5, 6, 7]], False, False),
]


@pytest.mark.parametrize("variant,ids,use_seg,use_depth", CASES)
def test_splice_planner_matches_oracle(emu_engines, variant, ids, use_seg, use_depth):
    eng, om = emu_engines[variant]
    B = len(ids)
    imgs, segs, deps = synth.synth_batch(B, eng.cfg.vit_image_size)
    segs, deps = (segs if use_seg else None), (deps if use_depth else None)
    t = lambda a: None if a is None else torch.from_numpy(a)
    ref, _ = om.prepare_inputs(ids, t(imgs), t(segs), t(deps))
    got = eng.inputs_embeds(np.array(ids), imgs, segs, deps)
    assert got.shape == tuple(ref.shape)
    assert np.abs(got - ref.numpy()).max() < 4e-3   # text rows exact; feature rows differ by bf16 rounding flips only


@pytest.mark.parametrize("variant,ids,err", [
    ("vcoder", [[1, 5, I, 6, 7]], IndexError),             # quirk 5: `or` guard -> embed(-200)
    ("vcoder_ds", [[1, 5, D, 6, 7]], IndexError),          # lone <depth>: ZeRO-3 branch embeds -400
    ("vcoder_ds", [[1, I, I, S, 6]], IndexError),          # second <image> indexes image_features[1] of a batch of 1
    ("vcoder", [[1, 5, S, I, 6]], IndexError),             # <seg> before <image>: the text chunk before <image> holds -300
])
def test_splice_errors_match_reference_behaviour(emu_engines, variant, ids, err):
    eng, om = emu_engines[variant]
    imgs, segs, deps = synth.synth_batch(1, eng.cfg.vit_image_size)
    t = torch.from_numpy
    with pytest.raises(err):
        om.prepare_inputs(ids, t(imgs), t(segs), t(deps) if variant == "vcoder_ds" else None)
    with pytest.raises(err):
        eng.inputs_embeds(np.array(ids), imgs, segs, deps if variant == "vcoder_ds" else None)


def test_zero_depth_and_unequal_lengths(emu_engines):
    eng, om = emu_engines["vcoder_ds"]
    imgs, segs, deps = synth.synth_batch(2, eng.cfg.vit_image_size)
    ids = np.array([[1, 5, I, S, D, 6], [1, 5, I, S, D, 6]])
    deps0 = deps.copy()
    deps0[1] = 0                                            # sample 1: is_depth_zero -> <depth> stays -> IndexError
    with pytest.raises(IndexError):
        eng.inputs_embeds(ids, imgs, segs, deps0)
    ragged = np.array([[1, 5, I, S, 6, 7], [1, 5, I, 8, 9, 6]])
    with pytest.raises(UnboundLocalError):
        eng.inputs_embeds(ragged, imgs, segs, None, has_attention_mask=True)
    got = eng.inputs_embeds(ragged, imgs, segs, None, has_attention_mask=False)
    ref, _ = om.prepare_inputs(ragged.tolist(), torch.from_numpy(imgs), torch.from_numpy(segs), None)
    assert got.shape == tuple(ref.shape) and not got[1, -1].any()
    assert np.abs(got - ref.numpy()).max() < 4e-3


def test_load_errors(emu_engines):
    lib = kc.EmuBackend().lib
    cfg = vcfg.tiny("vcoder_ds")
    e = HipEngine(cfg, lib=lib)
    with pytest.raises(ValueError):
        e.load_tensor("model.layers.0.self_attn.q_proj.weight", np.zeros((3, 3), np.float32))   # wrong shape
    with pytest.raises(ValueError):
        e.load_tensor("model.not_a_tensor", np.zeros((3,), np.float32))
    assert e.load_tensor("model.depth_mm_projector.0.weight", np.zeros((256, 128), np.float32)) is False  # dead, accepted
    with pytest.raises(RuntimeError):
        e.finalize()                                                                              # missing tensors
    e.close()


def test_checkpoint_reader_handles_sharded_bin_layout(tmp_path):
    """The released VCoder / Vicuna checkpoints ship `pytorch_model-0000i-of-0000n.bin` shards (fp16); the reader yields
    every tensor of every shard under its HF key, and safetensors take precedence when both are present."""
    import torch
    from vcoder_amd import checkpoint

    d = str(tmp_path / "ckpt")
    os.makedirs(d)
    a = {"model.layers.0.self_attn.q_proj.weight": torch.randn(4, 4).half(), "model.norm.weight": torch.ones(4).half()}
    b = {"lm_head.weight": torch.randn(8, 4).half(), "model.mm_projector.0.bias": torch.zeros(4).half()}
    torch.save(a, os.path.join(d, "pytorch_model-00001-of-00002.bin"))
    torch.save(b, os.path.join(d, "pytorch_model-00002-of-00002.bin"))
    assert checkpoint.has_weights(d)
    got = dict(checkpoint.iter_checkpoint_tensors(d))
    assert set(got) == set(a) | set(b)
    for k, v in {**a, **b}.items():
        assert got[k].dtype == torch.float16 and torch.equal(got[k], v)
    with pytest.raises(FileNotFoundError):
        list(checkpoint.iter_checkpoint_tensors(str(tmp_path)))
    checkpoint.save_checkpoint(d, {"model_type": "llava"}, {"only.safetensors": np.ones((2, 2), np.float32)})
    assert list(dict(checkpoint.iter_checkpoint_tensors(d))) == ["only.safetensors"]


def test_bench_refuses_to_run_without_a_gpu():
    """bench.py is a GPU measurement: without a HIP device it must exit loudly, not fall back to anything."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
    assert not any(line.startswith("{") for line in r.stdout.splitlines()), "no result line may be printed"
