#!/usr/bin/env python
"""bench.py — images/sec of the VCoder-DS LLaVA-1.5-7b hot path on MI355X (BASELINE.json metric).

One "step" = one full pass of the hot path over one batch per GPU:
    3x CLIP ViT-L/14@336 encode (RGB, seg, depth) + adapters + splice + Llama prefill (S=1216)
    + 128 greedy tokens (1 from the prefill + 127 hipGraph-replayed decode steps), EOS disabled.
Synthetic COST-shaped inputs, seeded synthetic weights (no network), bf16 MFMA compute, fp32 accumulate.
Pixels are resident in HBM before the timed region.  N>1: one process per GPU (torchrun), batch sharded
data-parallel with no collective on the data path; the only exchange is one RCCL all-gather of the
generated token ids per step.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16


def cpu_baseline(cfg, B_for_rate: int, new_tokens: int):
    """Oracle (oracle/cpu_ref.py, torch fp32, kind='port') timed on the host cores on a bounded sample:
    one ViT encoder layer (1 image), one adapter, one Llama decoder layer prefill (B=1, S=1216) and one decoder
    layer decode step (B=1, ctx=1216), extrapolated linearly to the full per-sample work."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_ref

    torch.manual_seed(0)
    cores = torch.get_num_threads()
    g = lambda *s: torch.randn(*s) * 0.02
    Dv, Fv, D, F, V = cfg.mm_hidden_size, cfg.vit_intermediate_size, cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    vp = "vision_model."
    sd = {vp + "embeddings.class_embedding": g(Dv),
          vp + "embeddings.patch_embedding.weight": g(Dv, 3, cfg.vit_patch_size, cfg.vit_patch_size),
          vp + "embeddings.position_embedding.weight": g(cfg.num_patches + 1, Dv),
          vp + "pre_layrnorm.weight": torch.ones(Dv), vp + "pre_layrnorm.bias": torch.zeros(Dv)}
    p = vp + "encoder.layers.0."
    for ln in ("layer_norm1", "layer_norm2"):
        sd[p + ln + ".weight"], sd[p + ln + ".bias"] = torch.ones(Dv), torch.zeros(Dv)
    for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
        sd[p + f"self_attn.{nm}.weight"], sd[p + f"self_attn.{nm}.bias"] = g(Dv, Dv), g(Dv)
    sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = g(Fv, Dv), g(Fv)
    sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = g(Dv, Fv), g(Dv)
    sd["model.mm_projector.0.weight"], sd["model.mm_projector.0.bias"] = g(D, Dv), g(D)
    sd["model.mm_projector.2.weight"], sd["model.mm_projector.2.bias"] = g(D, D), g(D)
    lp = "model.layers.0."
    sd[lp + "input_layernorm.weight"] = torch.ones(D)
    sd[lp + "post_attention_layernorm.weight"] = torch.ones(D)
    for nm in ("q", "k", "v", "o"):
        sd[lp + f"self_attn.{nm}_proj.weight"] = g(D, D)
    sd[lp + "mlp.gate_proj.weight"], sd[lp + "mlp.up_proj.weight"], sd[lp + "mlp.down_proj.weight"] = g(F, D), g(F, D), g(D, F)
    sd["model.norm.weight"], sd["lm_head.weight"] = torch.ones(D), g(V, D)

    class C1:  # one-layer views of the config
        pass
    c1 = C1()
    for k in ("vit_patch_size", "vit_image_size", "vit_num_heads", "vit_layer_norm_eps", "mm_vision_select_feature",
              "num_attention_heads", "rms_norm_eps", "rope_theta"):
        setattr(c1, k, getattr(cfg, k))
    c1.vit_layers_used, c1.num_hidden_layers = 1, 1
    S = 64 + 2 * cfg.num_patches
    px = torch.randn(1, 3, cfg.vit_image_size, cfg.vit_image_size)
    def best_of(fn, reps):  # min over repetitions: excludes first-touch / thread-pool warm-up
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return min(ts), r

    with torch.no_grad():
        t_vit1, feats = best_of(lambda: cpu_ref.vit_forward(px, sd, c1), 5)            # embed + 1 layer
        t_ad, _ = best_of(lambda: cpu_ref.projector_forward(feats, sd, "model.mm_projector", "mlp2x_gelu"), 5)
        x = torch.randn(1, S, D)
        t_pre, _ = best_of(lambda: cpu_ref.llama_layer(x, sd, 0, c1, cpu_ref.KVCache(1), 0, cpu_ref.Rounder(False)), 4)
        cache = cpu_ref.KVCache(1)
        cpu_ref.llama_layer(x, sd, 0, c1, cache, 0, cpu_ref.Rounder(False))
        xd = torch.randn(1, 1, D)
        t0 = time.perf_counter()
        for i in range(24):
            cpu_ref.llama_layer(xd, sd, 0, c1, cache, S + i, cpu_ref.Rounder(False))
        t_dec = (time.perf_counter() - t0) / 24
        t_head, _ = best_of(lambda: torch.nn.functional.linear(xd, sd["lm_head.weight"]), 8)
    per_sample = 3 * (cfg.vit_layers_used * t_vit1 + t_ad) + cfg.num_hidden_layers * t_pre + \
        (new_tokens - 1) * (cfg.num_hidden_layers * t_dec + t_head)
    return {"value": 1.0 / per_sample, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": (f"oracle/cpu_ref.py fp32 at true 7b dims, B=1: 1 ViT layer+embed ({t_vit1:.2f}s), adapter "
                       f"({t_ad:.2f}s), 1 decoder layer prefill S={S} ({t_pre:.2f}s), 1 decoder layer decode step "
                       f"({t_dec * 1e3:.0f}ms), lm_head ({t_head * 1e3:.0f}ms); linearly extrapolated to "
                       f"3x{cfg.vit_layers_used} ViT layers + {cfg.num_hidden_layers} layers prefill + "
                       f"{new_tokens - 1} decode steps per image")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU (BASELINE configs[1]: 8)")
    ap.add_argument("--new-tokens", type=int, default=128)
    ap.add_argument("--model", default="7b", choices=["7b", "13b"])
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8"],
                    help="decoder weight storage: bf16 (BASELINE configs[1], the default metric) or fp8 = W8A16 e4m3 with "
                         "per-row power-of-two scales (the weight format of BASELINE configs[4])")
    ap.add_argument("--inflight", type=int, default=3,
                    help="batches in flight per GPU: independent sessions (own stream + KV cache, shared weights) driven by "
                         "host threads, so one batch's MFMA-bound prefill and per-launch ramps overlap another's "
                         "HBM-bound decode.  1 = strictly one batch at a time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-pixels", action="store_true",
                    help="hand the pixel tensors over as HOST buffers (fp32, pageable): the PCIe-inclusive rate noted in "
                         "DESIGN.md; the default (and `value`) has the inputs resident in HBM")
    ap.add_argument("--pmc-traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc pass "
                         "(default: profiles/r01_pmc_traffic.json, the committed FETCH_SIZE pass of this kernel)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False and there is no CPU fallback")
    # debugging knobs for exercising the multi-process path on a 1-GPU box: all ranks on one device, gloo gather
    if os.environ.get("VC_BENCH_FORCE_DEVICE") is not None:
        local = int(os.environ["VC_BENCH_FORCE_DEVICE"])
    backend = os.environ.get("VC_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from vcoder_amd import config as vcfg, synth
    from vcoder_amd.engine import HipEngine
    from vcoder_amd.parallel import gather_token_ids, shard_range

    cfg = vcfg.vicuna_7b("vcoder_ds") if args.model == "7b" else vcfg.vicuna_13b("vcoder_ds")
    eng = HipEngine(cfg, device_index=local)
    eng.load_synthetic(42)
    if args.weights == "fp8":
        eng.set_weight_format("fp8")
    eng.finalize()
    B, N_new = args.batch, args.new_tokens
    first, _ = shard_range(world * B, rank, world)   # contiguous shard of the global batch
    ids = np.stack([synth.synth_prompt_ids(cfg.vocab_size, "vcoder_ds", sample=first + b) for b in range(B)])
    imgs, segs, deps = synth.synth_batch(B, cfg.vit_image_size, first)
    if not args.host_pixels:
        imgs, segs, deps = (torch.from_numpy(a).cuda() for a in (imgs, segs, deps))

    import threading

    n_sess = max(1, min(args.inflight, args.steps))
    sessions = [eng] + [eng.fork() for _ in range(n_sess - 1)]

    def run_steps(k: int):
        """k steps (= k batches of B per GPU), distributed round-robin over the in-flight sessions; every step is the
        complete hot path for its batch.  Token ids are all-gathered across ranks once per step, in step order."""
        outs = [None] * k
        errs = []

        def worker(si):
            try:
                for j in range(si, k, n_sess):
                    outs[j] = sessions[si].generate_greedy(ids, imgs, segs, deps, max_new_tokens=N_new, eos_token_id=None)
            except BaseException as e:  # surface failures of a session thread in the main thread
                errs.append(e)

        ths = [threading.Thread(target=worker, args=(si,)) for si in range(n_sess)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]
        # the one exchange: all-gather of the token stream (RCCL over xGMI), once per batch; no-op for N=1
        return [gather_token_ids(o, dist, device="cuda" if backend == "nccl" else None) for o in outs]

    def fence():
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run_steps(max(args.warmup, n_sess))  # every session captures its decode graph before the timed region
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    timings = eng.last_timings()
    # transparency leg (outside the timed region): the same step strictly one batch at a time on this rank
    solo = None
    if n_sess > 1:
        fence()
        t1 = time.perf_counter()
        for _ in range(2):
            sessions[0].generate_greedy(ids, imgs, segs, deps, max_new_tokens=N_new, eos_token_id=None)
        torch.cuda.synchronize()
        solo = (time.perf_counter() - t1) / 2
        timings = sessions[0].last_timings()
    prof = eng.profile_decode_gemv(min(B, 16), reps=3)

    if rank == 0:
        traffic = args.pmc_traffic_bytes
        pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if traffic is None and args.model == "7b" and B == 8 and args.weights == "bf16" and os.path.exists(pmc_file):
            with open(pmc_file) as f:   # separate --pmc pass, gfx950 x2 correction applied (see the file)
                traffic = json.load(f)["hbm_read_bytes_per_launch"]
        S = 64 + 2 * cfg.num_patches
        ach = prof["avg_bytes"] / (prof["avg_us"] * 1e-6) / 1e9
        res = {
            "metric": "images/sec (3xViT encode + 128-tok decode), VCoder-DS-7b" if args.model == "7b"
                      else "images/sec (3xViT encode + 128-tok decode), VCoder-DS-13b",
            "value": world * B * args.steps / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"VCoder-DS LLaVA-1.5-{args.model} {'bf16' if args.weights == 'bf16' else 'bf16 activations / fp8-e4m3 decoder weights (W8A16)'}, batch={B}/GPU RGB+seg+depth 336x336, "
                                   f"prefill S={S}, {N_new}-token greedy decode", "global_batch": world * B,
                       "parallelism": f"dp{world}", "weights": "seeded synthetic (vcoder_amd/synth.py)",
                       "in_flight_batches_per_gpu": n_sess,
                       "inputs": "host buffers (PCIe inclusive)" if args.host_pixels else "resident in HBM"},
            "phase_ms_one_session": timings,  # encode / prefill / decode wall time of one batch run alone
            "one_batch_at_a_time": None if solo is None else {"value": B / solo, "unit": "images/s per GPU",
                                                              "ms_per_step": solo * 1e3},
            "roofline": {"bound": "hbm", "kernel": "gemv_dma_kernel (decode weight streaming, all 129 GEMV launches of a step)", "achieved": ach,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "avg_launch_us": prof["avg_us"],
                         "algorithmic_bytes_per_launch": prof["avg_bytes"],
                         "launches_per_decode_step": prof["launches_per_step"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, B, N_new)
        print(json.dumps(res), flush=True)
    if world > 1:
        fence()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
