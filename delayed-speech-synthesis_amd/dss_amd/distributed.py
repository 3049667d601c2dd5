"""Utterance-sharded synthesis across the GPUs of one node: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI), a contiguous block of the batch per rank, weights replicated, and ONE collective at
the end -- a gather of the int16 PCM shards on rank 0.  Utterances share no state, so there is
no other exchange step.  The same code runs under gloo on CPU tensors (tests use it with a stand-in synthesiser).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block split; the first (n_items % world_size) ranks get one extra item."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_pcm(local_pcm: torch.Tensor, n_total: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Gather ragged (n_local, T) int16 shards to `dst` in rank order.  Returns (n_total, T) on dst, else None.
    Shards are padded to the largest shard so a single gather collective suffices."""
    world, rank = dist.get_world_size(), dist.get_rank()
    T = local_pcm.shape[1]
    largest = shard_bounds(n_total, world, 0)[1]
    padded = local_pcm
    if local_pcm.shape[0] < largest:
        padded = torch.zeros((largest, T), dtype=local_pcm.dtype, device=local_pcm.device)
        padded[: local_pcm.shape[0]] = local_pcm
    # neither RCCL/NCCL nor gloo has a 16-bit integer type: move the shard as raw bytes
    wire = padded.contiguous().view(torch.uint8)
    bufs: Optional[List[torch.Tensor]] = None
    if rank == dst:
        bufs = [torch.empty_like(wire) for _ in range(world)]
    dist.gather(wire, bufs, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, world, r)
        parts.append(bufs[r].view(local_pcm.dtype)[: hi - lo])
    return torch.cat(parts, dim=0)


def synthesize_sharded(features: np.ndarray, synth: Callable[[np.ndarray], torch.Tensor], dst: int = 0):
    """features: host (N, F, 20), identical on every rank (or only this rank's rows are read).  `synth` maps this
    rank's (n_local, F, 20) block to an (n_local, F*160) int16 tensor on the rank's device.  Returns the full
    (N, F*160) tensor on `dst`."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(features.shape[0], world, rank)
    local = synth(features[lo:hi])
    return gather_pcm(local, features.shape[0], dst=dst)


def lpcnet_synth_fn(max_utts: int, n_frames: int, device: Optional[int] = None):
    """The production `synth`: a per-rank LPCNetBatch on the rank's GPU (LOCAL_RANK unless `device` is given).
    torch's current device, the library's device and every tensor handed to it are pinned to the same index."""
    import os
    from .lpcnet import LPCNetBatch
    dev = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)
    torch.cuda.set_device(dev)
    dec = LPCNetBatch(max_utts, n_frames, device=dev)

    def run(block: np.ndarray) -> torch.Tensor:
        with torch.cuda.device(dev):
            dec.reset_async()
            d = torch.from_numpy(np.ascontiguousarray(block, dtype=np.float32)).to(f"cuda:{dev}")
            return dec.synthesize_torch(d)
    return run
