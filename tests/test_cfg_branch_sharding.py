"""CFG-branch sharding (regione_amd/dist.py CfgBranchPair).

CPU part (gloo, world 2 and 4): pairing, exchange, and that a replicated loop around a one-branch-per-rank
forward reproduces the sequential loop bit for bit.  GPU part (-m gpu): two processes share cuda:0 and exchange
over gloo - the Qwen-Image-Edit and Step1X-Edit v1p2 engines, RegionE enabled and disabled, give latents / ids /
plans BIT-IDENTICAL to the unsharded run, each rank having run only its own branch.
"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(worker, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _env(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))


# ------------------------------------------------------------------------------------------------ CPU
def _fake_branch(x, tag, image):
    # deterministic stand-in for one CFG forward of image `image`
    s = 0.75 if tag == "cond" else -0.5
    return (x.float() * s + float(image) + torch.arange(x.shape[-1]) * 0.01).to(torch.bfloat16)


def _host_worker(rank, world, port, q):
    _env(rank, world, port)
    from regione_amd import dist as D
    dist = D.init("gloo")
    pair = D.make_cfg_pair(dist)
    image = rank // 2
    calls = []

    def loop(pair):
        x = torch.linspace(-1, 1, 8 * 64).reshape(1, 8, 64).to(torch.bfloat16) + image
        for _ in range(5):
            def run(tag):
                calls.append(tag)
                return _fake_branch(x, tag, image)
            pos, neg = D.run_cfg_branches(pair, lambda: run("cond"), lambda: run("uncond"))
            x = (x.float() + 0.1 * (neg.float() + 4.0 * (pos.float() - neg.float()))).to(torch.bfloat16)
        return x
    seq = loop(None)
    n_seq = len(calls)
    del calls[:]
    sharded = loop(pair)
    q.put((rank, pair.role, bool(torch.equal(seq, sharded)), n_seq, list(calls)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_cfg_pair_exchange_reproduces_sequential_loop(world):
    res = _spawn(_host_worker, world)
    for rank, role, same, n_seq, calls in res:
        assert role == ("cond", "uncond")[rank % 2]
        assert same
        assert n_seq == 10 and calls == [role] * 5           # one forward per step instead of two


def test_cfg_pair_needs_even_world_and_supported_family():
    from regione_amd import dist as D

    class OddWorld:
        def get_world_size(self): return 3
        def get_rank(self): return 0
    with pytest.raises(ValueError):
        D.make_cfg_pair(OddWorld())

    from regione_amd import RegionEHelper

    class FluxKontextPipeline:          # only the class NAME matters to the dispatch (tool/RegionE.py)
        pass
    with pytest.raises(NotImplementedError):
        RegionEHelper(FluxKontextPipeline()).shard_cfg_branches(object())


# ------------------------------------------------------------------------------------------------ GPU
def _gpu_worker(rank, world, port, q, family):
    _env(rank, world, port)
    from regione_amd import RegionEHelper, synth
    from regione_amd import dist as D
    dist = D.init("gloo")                       # two ranks on ONE GPU: RCCL refuses that, gloo carries the exchange
    pair = D.make_cfg_pair(dist)
    torch.cuda.set_device(0)
    h = w = 16
    Tp, Tn = 32, 24
    if family == "qwen":
        from regione_amd.harness import qwen as HQ
        cfg = synth.FluxConfig(**synth.QWEN_TOY)
        wts = synth.make_flux_weights(cfg, seed=6, dtype=torch.bfloat16, w_std=0.05)
        pipe = HQ.QwenImageEditPipeline(HQ.QwenImageTransformer2DModel(cfg, "cuda").load_state_dict(wts))
    else:
        from regione_amd.harness import step1x as HS
        cfg = synth.FluxConfig(guidance_embeds=False, **synth.TOY)
        wts = synth.make_flux_weights(cfg, seed=5, dtype=torch.bfloat16, w_std=0.05)
        pipe = HS.Step1XEditPipelineV1P2(HS.Step1XEditTransformer2DModel(cfg, "cuda").load_state_dict(wts))
    lat, img, prompt, y = synth.make_edit_inputs(h, w, Tp, cfg, seed=9, dtype=torch.bfloat16)
    _, _, nprompt, ny = synth.make_edit_inputs(h, w, Tn, cfg, seed=10, dtype=torch.bfloat16)
    # a region by construction: the condition image equals the start latents except inside a block
    img = lat.clone()
    blk = torch.zeros(h, w, dtype=torch.bool)
    blk[4:10, 5:12] = True
    img[0, blk.flatten()] = -lat[0, blk.flatten()]
    kw = dict(image=img.cuda(), prompt_embeds=prompt.cuda(), negative_prompt_embeds=nprompt.cuda(), height=h * 16,
              width=w * 16, latents=lat.cuda(), true_cfg_scale=4.0, return_dict=False)
    if family != "qwen":
        kw.update(pooled_prompt_embeds=y.cuda(), negative_pooled_prompt_embeds=ny.cuda())
    tags = []
    tr = pipe.transformer
    inner = tr._run

    def counting_run(*a, **k):
        tags.append((a[7] if len(a) > 7 else k.get("attention_kwargs") or {}).get("tag"))
        return inner(*a, **k)
    tr._run = counting_run
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.5)
    out = {}
    for mode in ("vanilla", "regione"):
        if mode == "regione":
            helper.enable()
        for sharded in (False, True):
            helper.shard_cfg_branches(pair if sharded else None)
            del tags[:]
            trace = {}
            lat_out = pipe(**(dict(kw, trace=trace) if mode == "regione" else kw))[0]
            torch.cuda.synchronize()
            ids = pipe._regione_manager.edited_ids.cpu() if mode == "regione" else None
            out[mode, sharded] = (lat_out.cpu(), ids, "".join(trace.get("kind", [])), list(tags))
    res = {}
    for mode in ("vanilla", "regione"):
        a, b = out[mode, False], out[mode, True]
        res[mode] = dict(latents_equal=bool(torch.equal(a[0], b[0])),
                         ids_equal=(a[1] is None or bool(torch.equal(a[1], b[1]))), plan=(a[2], b[2]),
                         forwards=(len(a[3]), len(b[3])), my_tags=sorted(set(b[3])), n_ids=(0 if a[1] is None else int(a[1].numel())),
                         checksum=float(b[0].float().sum()))
    q.put((rank, pair.role, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["qwen", "step1x_v1p2"])
def test_cfg_branch_sharding_bit_identical_on_gpu(family):
    res = _spawn(_gpu_worker, 2, family)
    for rank, role, r in res:
        for mode in ("vanilla", "regione"):
            m = r[mode]
            assert m["latents_equal"] and m["ids_equal"], (rank, mode, m)
            assert m["plan"][0] == m["plan"][1]
            assert m["forwards"][1] * 2 == m["forwards"][0] and m["my_tags"] == [role]
        assert 0 < r["regione"]["n_ids"] < 256 and "R" in r["regione"]["plan"][0]
    assert res[0][2]["regione"]["checksum"] == res[1][2]["regione"]["checksum"]      # both ranks hold the same latents
