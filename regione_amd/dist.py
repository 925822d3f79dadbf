"""Multi-GPU plumbing for the image-sharded denoise path.

The hot path shards per image: every image owns its edited-token set, K/V cache and velocity
cache, so ranks never exchange anything during denoising.  One process per GPU (RCCL = backend
"nccl" on ROCm; "gloo" in the CPU tests), weights replicated.  The only collectives are the barrier
around the timed region, a MAX-reduce of the elapsed time and an all_gather of the final latents.
"""
from __future__ import annotations

import os
import time
from typing import Callable, List, Optional, Sequence

import torch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_images(n_images: int, world: int, rank: int) -> List[int]:
    """image j -> rank j mod world (round-robin keeps ragged batches balanced)."""
    return [j for j in range(n_images) if j % world == rank]


def init(backend: str, device: Optional[torch.device] = None):
    import torch.distributed as dist
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return dist


def gather_latents(local: Sequence[torch.Tensor], image_ids: Sequence[int], n_images: int, dist=None) -> List[Optional[torch.Tensor]]:
    """Every rank ends up with the latents of all images, in image order.  Ranks may hold different
    numbers of images (ragged batch): shorter ranks pad with a dummy that is dropped again."""
    if dist is None or dist.get_world_size() == 1:
        out: List[Optional[torch.Tensor]] = [None] * n_images
        for j, t in zip(image_ids, local):
            out[j] = t
        return out
    world = dist.get_world_size()
    per = (n_images + world - 1) // world
    out = [None] * n_images
    for slot in range(per):
        have = slot < len(local)
        mine = local[slot] if have else torch.zeros_like(local[0])
        dev = mine.device
        if dist.get_backend() == "gloo" and mine.is_cuda:      # debugging path: gloo collectives run on host tensors
            mine = mine.cpu()
        bucket = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(bucket, mine.contiguous())
        bucket = [b.to(dev) for b in bucket]
        for r in range(world):
            j = slot * world + r
            if j < n_images:
                out[j] = bucket[r]
    return out


def timed(fn: Callable[[], None], sync: Callable[[], None], dist=None) -> float:
    """barrier + sync | fn | sync + barrier; returns the MAX elapsed time over ranks (seconds)."""
    sync()
    if dist is not None and dist.get_world_size() > 1:
        dist.barrier()
        sync()
    t0 = time.perf_counter()
    fn()
    sync()
    if dist is not None and dist.get_world_size() > 1:
        dist.barrier()
        sync()
    el = time.perf_counter() - t0
    if dist is not None and dist.get_world_size() > 1:
        t = torch.tensor([el], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el
