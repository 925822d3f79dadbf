"""Multi-GPU plumbing for the image-sharded denoise path.

The hot path shards per image: every image owns its edited-token set, K/V cache and velocity
cache, so ranks never exchange anything during denoising.  One process per GPU (RCCL = backend
"nccl" on ROCm; "gloo" in the CPU tests), weights replicated.  The only collectives are the barrier
around the timed region, a MAX-reduce of the elapsed time and an all_gather of the final latents.

Optional second axis: CFG-branch sharding for the families whose reference
patch sets run the two CFG branches as two separate forwards with separate K/V caches (Qwen-Image-Edit,
Step1X-Edit v1p2: `k_cache_even/odd`).  Two ranks own ONE image; rank 2p runs the 'cond' forward, rank
2p+1 the 'uncond' forward, and per computed step they exchange `noise_pred` ([1, K_e or L, 64] bf16,
<= 512 KB: one 2-rank all_gather, latency-bound on a single xGMI link).  Everything after the exchange
(CFG combine, scheduler step, region partition, decay-cache decisions) is replicated and deterministic,
so both ranks hold bit-identical latents / ids and no further traffic is needed.
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


def gather_latents(local: Sequence[torch.Tensor], image_ids: Sequence[int], n_images: int, dist=None,
                   like: Optional[torch.Tensor] = None, force: bool = False) -> List[Optional[torch.Tensor]]:
    """Every rank ends up with the latents of all images, in image order.  Ranks may hold different
    numbers of images (ragged batch, down to NONE when n_images < world size): shorter ranks pad with a dummy that is
    dropped again.  All images must share one latent shape; a rank without images learns it from `like` (a template
    tensor) or, failing that, from rank 0 (one small broadcast) - it must never enter the collective empty-handed."""
    # `force`: run the collective even in a world of one (bench.py --force-collectives: RCCL init + all_gather exercised on
    # a one-GPU box)
    if dist is None or (dist.get_world_size() == 1 and not force):
        out: List[Optional[torch.Tensor]] = [None] * n_images
        for j, t in zip(image_ids, local):
            out[j] = t
        return out
    world = dist.get_world_size()
    per = (n_images + world - 1) // world
    out = [None] * n_images
    template = local[0] if len(local) else like
    if n_images < world:
        # some ranks own no image (image j lives on rank j % world): every rank learns the latent shape from rank 0, which
        # always owns image 0 - the same collective sequence on every rank, nobody enters all_gather empty-handed
        meta = [(tuple(local[0].shape), str(local[0].dtype).replace("torch.", ""))] if dist.get_rank() == 0 else [None]
        dist.broadcast_object_list(meta, src=0)
        if template is None:
            shape, dt = meta[0]
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            template = torch.zeros(shape, dtype=getattr(torch, dt), device=dev)
    assert template is not None, "a rank without images needs n_images < world size (round-robin placement) or `like`"
    for slot in range(per):
        have = slot < len(local)
        mine = local[slot] if have else torch.zeros_like(template)
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


def timed(fn: Callable[[], None], sync: Callable[[], None], dist=None, force: bool = False) -> float:
    """barrier + sync | fn | sync + barrier; returns the MAX elapsed time over ranks (seconds).  `force`: run the barrier
    and the MAX-reduce in a world of one too."""
    coll = dist is not None and (dist.get_world_size() > 1 or force)
    sync()
    if coll:
        dist.barrier()
        sync()
    t0 = time.perf_counter()
    fn()
    sync()
    if coll:
        dist.barrier()
        sync()
    el = time.perf_counter() - t0
    if coll:
        t = torch.tensor([el], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el


class CfgBranchPair:
    """One rank's view of a CFG pair: `role` is the branch this rank computes; `exchange(mine)` returns
    (cond, uncond) on both ranks."""

    ROLES = ("cond", "uncond")

    def __init__(self, dist, group, index_in_pair: int):
        self.dist, self.group, self.index = dist, group, int(index_in_pair)

    @property
    def role(self) -> str:
        return self.ROLES[self.index]

    def exchange(self, mine: torch.Tensor):
        dev = mine.device
        host = self.dist.get_backend(self.group) == "gloo" and mine.is_cuda     # debugging path (ranks sharing one GPU)
        send = mine.contiguous().cpu() if host else mine.contiguous()
        both = [torch.empty_like(send), torch.empty_like(send)]
        self.dist.all_gather(both, send, group=self.group)
        if host:
            both = [b.to(dev) for b in both]
        both[self.index] = mine           # keep the local tensor (no copy; identical bytes)
        return both[0], both[1]


def make_cfg_pair(dist) -> CfgBranchPair:
    """Ranks (2p, 2p+1) form pair p.  Collective over the WORLD (every rank must create every group)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if world % 2:
        raise ValueError(f"CFG-branch sharding needs an even number of ranks, got {world}")
    mine = None
    for p in range(world // 2):
        g = dist.new_group(ranks=[2 * p, 2 * p + 1])
        if rank // 2 == p:
            mine = g
    return CfgBranchPair(dist, mine, rank % 2)


_side_streams = {}


def side_stream(main):
    """The side stream paired with `main` (one per caller stream, created on first use)."""
    key = (main.device_index, main.cuda_stream)
    side = _side_streams.get(key)
    if side is None:
        side = _side_streams[key] = torch.cuda.Stream(device=main.device)
    return side


def _branch_streams_mode() -> int:
    """RGN_BRANCH_STREAMS: 0 = the two forwards of a step run one after the other on the caller's stream (reference order),
    1 (default) = the uncond forward of a step that can profit runs on a side stream, 2 = of every computed step."""
    import os
    return int(os.environ.get("RGN_BRANCH_STREAMS", "1"))


def branches_concurrent(owner, key, region_step: bool) -> bool:
    """May the two forwards of this computed step run on two streams?  Not the first time a (step kind, text lengths) key is
    seen by `owner` (the manager of the edit / the vanilla pipeline call): that forward pair builds the lazily cached tables
    (rotary rows of the edited ids, K/V slabs) the later ones share, and runs in order.  Owners clear `_branch_warm` whenever
    they drop those tables."""
    mode = _branch_streams_mode()
    if mode <= 0 or (mode == 1 and not region_step):
        return False
    warm = owner.__dict__.setdefault("_branch_warm", set())
    if key in warm:
        return True
    warm.add(key)
    return False


def branch_batching() -> bool:
    """RGN_BATCH_BRANCHES (default 1): the two CFG forwards of a computed step run as ONE batched pass through the trunk (the
    reference's B = 2 forward, Step1XEdit/inplace.py:381-399); 0 = two forwards (in sequence or on two streams)."""
    import os
    return os.environ.get("RGN_BATCH_BRANCHES", "1") != "0"


def run_cfg_branches(pair: Optional[CfgBranchPair], run_cond: Callable[[], torch.Tensor], run_uncond: Callable[[], torch.Tensor],
                     concurrent: bool = False, batch_on=None):
    """Both CFG forwards of one computed step: without a pair sequentially in the reference's order (cond, then uncond) - or,
    with `concurrent` (the caller says the two forwards touch disjoint state: per-branch K/V caches, tables already built),
    the uncond forward on a SIDE STREAM forked from the caller's stream and joined before the combine: the same launches
    with the same arguments, so the results are bit-identical, but a region step's launches (72 ... 288 workgroups on 256
    CUs, a drain and a fill at every kernel boundary) interleave with the other branch's instead of leaving CUs idle.
    The engine keeps one activation workspace and one split-K / KV-split scratch per stream.  With a pair: one branch per
    rank + exchange."""
    if pair is not None:
        return pair.exchange(run_cond() if pair.role == "cond" else run_uncond())
    if batch_on is not None and branch_batching() and hasattr(batch_on, "begin_batch"):
        # `batch_on` = the transformer both closures call: it records the two forwards and executes them as one batched pass
        # (per-branch K/V caches, rotary tables and AdaLN vectors; weights streamed once, one launch per Linear)
        batch_on.begin_batch()
        try:
            pos, neg = run_cond(), run_uncond()
        except BaseException:
            batch_on.abort_batch()
            raise
        outs = batch_on.end_batch()
        # a forward that did not record (a wrapped / replaced forward, or closures calling another object than `batch_on`)
        # hands back real tensors: they ARE the two sequential results (advisor finding, round 3)
        return tuple(o.resolve(outs) if hasattr(o, "resolve") else o for o in (pos, neg))
    if not (concurrent and torch.cuda.is_available()):
        pos = run_cond()
        return pos, run_uncond()
    main = torch.cuda.current_stream()
    side = side_stream(main)
    side.wait_stream(main)                       # fork: everything enqueued so far (latents, caches of earlier steps) is visible
    try:
        pos = run_cond()                         # host order stays cond -> uncond; the GPU runs them side by side
        with torch.cuda.stream(side):
            neg = run_uncond()
    finally:
        # join ALWAYS: if a branch raised, whatever the side stream already enqueued still reads / writes the shared K/V
        # caches and workspaces - later main-stream work must not overtake it
        main.wait_stream(side)
    neg.record_stream(main)
    return pos, neg
