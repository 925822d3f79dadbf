"""CPU, world_size 2, gloo: the N>1 plumbing of the image-sharded path (no GPU needed)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_images, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from regione_amd import dist as D
    dist = D.init("gloo")
    ids = D.shard_images(n_images, world, rank)
    # stand-in "edit": the latent of image j is a deterministic function of j only
    local = [torch.full((1, 8, 64), float(j)).to(torch.bfloat16) + torch.arange(64).to(torch.bfloat16) for j in ids]
    holder = {}

    def job():
        holder["all"] = D.gather_latents(local, ids, n_images, dist)
    el = D.timed(job, lambda: None, dist)
    ok = all(torch.equal(holder["all"][j], torch.full((1, 8, 64), float(j)).to(torch.bfloat16) + torch.arange(64).to(torch.bfloat16))
             for j in range(n_images))
    q.put((rank, ids, ok, el))
    dist.barrier()
    dist.destroy_process_group()


def _run(n_images):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_images, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_image_sharding_even_batch():
    res = _run(8)
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 7]
    assert all(r[2] for r in res)
    assert res[0][3] == res[1][3] > 0            # MAX-reduced time is identical on every rank


def test_image_sharding_ragged_batch():
    res = _run(5)                                # assets/data.jsonl has 5 prompts
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    assert all(r[2] for r in res)


def test_image_sharding_fewer_images_than_ranks():
    """n_images < world: rank 1 owns nothing - it must still take part in the gather (advisor finding, round 1: it used to
    index local[0] and die while rank 0 sat in all_gather)."""
    res = _run(1)
    assert res[0][1] == [0] and res[1][1] == []
    assert all(r[2] for r in res)


def test_shard_images_partition():
    from regione_amd import dist as D
    for n in (0, 1, 5, 8, 17):
        for w in (1, 2, 4, 8):
            parts = [D.shard_images(n, w, r) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_branch_stream_policy_first_pair_of_a_step_kind_runs_in_order(monkeypatch):
    """regione_amd.dist.branches_concurrent: the two forwards of a CFG step may share the GPU on two streams only after a
    forward pair of the same (step kind, text lengths) has run in order since the owner dropped its lazily built tables."""
    from regione_amd import dist as D

    class Owner:
        pass
    o = Owner()
    monkeypatch.setenv("RGN_BRANCH_STREAMS", "1")
    assert D.branches_concurrent(o, (False, 512, 384), True) is False          # first region step: builds the tables
    assert D.branches_concurrent(o, (False, 512, 384), True) is True
    assert D.branches_concurrent(o, (True, 512, 384), False) is False           # full steps stay sequential in mode 1
    o._branch_warm = set()                                                      # what a manager's refresh() does
    assert D.branches_concurrent(o, (False, 512, 384), True) is False
    monkeypatch.setenv("RGN_BRANCH_STREAMS", "2")
    assert D.branches_concurrent(o, (True, 512, 384), False) is False and D.branches_concurrent(o, (True, 512, 384), False) is True
    monkeypatch.setenv("RGN_BRANCH_STREAMS", "0")
    assert D.branches_concurrent(o, (False, 512, 384), True) is False
    # without a GPU (or without `concurrent`) the helper keeps the reference order
    order = []
    a, b = D.run_cfg_branches(None, lambda: order.append("cond") or 1, lambda: order.append("uncond") or 2, concurrent=False)
    assert (a, b) == (1, 2) and order == ["cond", "uncond"]



def test_cfg_branches_batched_protocol_records_then_executes_in_call_order(monkeypatch):
    """regione_amd.dist.run_cfg_branches(batch_on=transformer): between begin_batch() and end_batch() each branch's forward call
    only records and returns a handle that remembers the indexing applied to it; end_batch() executes once and the handles
    resolve in call order.  RGN_BATCH_BRANCHES=0 and transformers without the protocol keep two forwards."""
    import torch
    from regione_amd import dist as D
    from regione_amd.harness.flux import BranchHandle

    class Tr:
        def __init__(self):
            self.log, self._batch = [], None

        def begin_batch(self):
            self._batch = []
            self.log.append("begin")

        def abort_batch(self):
            self._batch = None
            self.log.append("abort")

        def end_batch(self):
            recs, self._batch = self._batch, None
            self.log.append(("end", len(recs)))
            return [torch.full((1, 6, 4), float(r)) for r in recs]

        def __call__(self, tag):
            if self._batch is None:
                self.log.append(("eager", tag))
                return (torch.full((1, 6, 4), float(tag)),)
            self._batch.append(tag)
            return (BranchHandle(len(self._batch) - 1),)
    tr = Tr()
    monkeypatch.setenv("RGN_BATCH_BRANCHES", "1")
    pos, neg = D.run_cfg_branches(None, lambda: tr(7)[0][:, :4], lambda: tr(9)[0][:, :2], batch_on=tr)
    assert tr.log == ["begin", ("end", 2)] and pos.shape == (1, 4, 4) and neg.shape == (1, 2, 4)
    assert float(pos[0, 0, 0]) == 7.0 and float(neg[0, 0, 0]) == 9.0
    tr.log.clear()
    monkeypatch.setenv("RGN_BATCH_BRANCHES", "0")
    pos, neg = D.run_cfg_branches(None, lambda: tr(7)[0][:, :4], lambda: tr(9)[0][:, :2], batch_on=tr)
    assert tr.log == [("eager", 7), ("eager", 9)] and float(neg[0, 0, 0]) == 9.0
    # a failing branch aborts the recording
    tr.log.clear()
    monkeypatch.setenv("RGN_BATCH_BRANCHES", "1")
    import pytest
    with pytest.raises(ZeroDivisionError):
        D.run_cfg_branches(None, lambda: tr(1)[0], lambda: 1 / 0, batch_on=tr)
    assert tr.log == ["begin", "abort"] and tr._batch is None
