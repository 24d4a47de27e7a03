"""Data-parallel sharding of the hot path: one process per GPU, full model replica per GPU,
contiguous batch shards, NO collective on the data path; the only exchange is one all-gather of the generated
token ids per batch (RCCL over xGMI on GPUs, gloo in the CPU tests).  Counterpart of the reference's process-level
sharding in scripts/v1_5/eval/cost_depth.sh:10-34 + eval/model_depth_loader.py:24-33 (files + `cat`)."""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous chunk of `n_items` for `rank` — ceil-sized chunks like the reference's get_chunk()
    (eval/model_seg_loader.py:24-32): the last ranks may get fewer (or zero) items."""
    per = -(-n_items // world)
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def gather_token_ids(local_ids: np.ndarray, dist=None, device=None) -> np.ndarray:
    """all-gather int32 [B_local, N] -> [world*B_local, N] (equal shard sizes).  `dist` = torch.distributed (already
    initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.ascontiguousarray(local_ids, dtype=np.int32)
    import torch

    t = torch.from_numpy(np.ascontiguousarray(local_ids, dtype=np.int32))
    if device is not None:
        t = t.to(device)
    out = torch.empty((dist.get_world_size() * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy()
