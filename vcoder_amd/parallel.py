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


def gather_token_ids(local_ids: np.ndarray, dist=None, device=None, force: bool = False) -> np.ndarray:
    """all-gather int32 [B_local, N] -> [world*B_local, N] (equal shard sizes).  `dist` = torch.distributed (already
    initialised) or None for a single process.  force: run the collective even in a world of one (bench.py --force-dist: the
    multi-GPU code path on a one-GPU box)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return np.ascontiguousarray(local_ids, dtype=np.int32)
    import torch

    t = torch.from_numpy(np.ascontiguousarray(local_ids, dtype=np.int32))
    if device is not None:
        t = t.to(device)
    out = torch.empty((dist.get_world_size() * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy()


class TokenComm:
    """The C-ABI communicator (include/vcoder_hip.h: vc_comm_create / vc_allgather_tokens): RCCL over xGMI called directly
    by libvcoder_hip.so on the engine's stream.  The 128-byte RCCL unique id travels from rank 0 to the others through
    `exchange` (a callable bytes -> bytes that broadcasts rank 0's value; bench.py uses torch.distributed's store for it).
    world 1 needs neither RCCL nor an exchange."""

    def __init__(self, engine, rank: int = 0, world: int = 1, exchange=None):
        import ctypes as C

        self.lib, self.engine, self.rank, self.world = engine.lib, engine, rank, world
        self._comm = C.c_void_p()
        uid = None
        if world > 1:
            buf = C.create_string_buffer(128)
            if rank == 0:
                engine._check(self.lib.vc_comm_unique_id(engine._ctx, buf))
            if exchange is None:
                raise ValueError("world > 1 needs an `exchange` to distribute the RCCL unique id")
            uid = C.create_string_buffer(exchange(buf.raw if rank == 0 else b""), 128)
        engine._check(self.lib.vc_comm_create(engine._ctx, rank, world, uid, C.byref(self._comm)))

    @property
    def uses_rccl(self) -> bool:
        return bool(self.lib.vc_comm_uses_rccl(self._comm))

    def allgather(self, local_ids: np.ndarray) -> np.ndarray:
        """int32 [B_local, N] -> [world * B_local, N]"""
        import ctypes as C

        loc = np.ascontiguousarray(local_ids, dtype=np.int32)
        out = np.empty((self.world * loc.shape[0],) + loc.shape[1:], dtype=np.int32)
        self.engine._check(self.lib.vc_allgather_tokens(self._comm, loc.ctypes.data_as(C.c_void_p), int(loc.size),
                                                        out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if self._comm:
            self.lib.vc_comm_destroy(self._comm)
            self._comm = None
