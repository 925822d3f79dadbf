#!/usr/bin/env python3
"""CFG-branch sharding at full size: ranks (2p, 2p+1) share one image, one branch each.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
        tools/cfg_shard_run.py [--family qwen|step1x_v1p2] [--size 1024] [--out FILE]
    (single-GPU box, plumbing check only - no speed-up to be had:  ... --share-gpu --dist-backend gloo)

Per rank: one RegionE edit with both branches computed locally (the unsharded time of the same engine), then the same
edit with the branches sharded over the pair; checks that the two give BIT-IDENTICAL latents and ids, and that both ranks
of the pair agree; rank 0 prints one JSON line with both wall-clocks (MAX over ranks) and the exchange volume.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench as B  # noqa: E402
import run_configs as RC  # noqa: E402
from regione_amd import RegionEHelper, synth  # noqa: E402
from regione_amd import dist as D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", default="qwen", choices=["qwen", "step1x_v1p2"])
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--toy", action="store_true")
    ap.add_argument("--edit-frac", type=float, default=0.25)
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rank, local_rank, world = D.env_world()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = torch.device("cuda", 0 if args.share_gpu else local_rank)
    torch.cuda.set_device(device)
    dist = D.init(args.dist_backend, device)
    pair = D.make_cfg_pair(dist)

    from regione_amd.harness import qwen as HQ, step1x as HS
    h_tok = w_tok = args.size // 16
    T, Tn = (32, 24) if args.toy else (512, 384)
    if args.family == "qwen":
        cfg = synth.FluxConfig(**(synth.QWEN_TOY if args.toy else synth.QWEN))
        tr = HQ.QwenImageTransformer2DModel(cfg, device).load_state_dict_stream(RC.weights_stream(cfg, device, 42))
        pipe = HQ.QwenImageEditPipeline(tr)
        scale = 4.0
    else:
        cfg = synth.FluxConfig(guidance_embeds=False, **(synth.TOY if args.toy else {}))
        tr = HS.Step1XEditTransformer2DModel(cfg, device).load_state_dict_stream(RC.weights_stream(cfg, device, 42))
        pipe = HS.Step1XEditPipelineV1P2(tr)
        scale = 6.0
    image = rank // 2                                  # the pair's image
    lat, img, prompt, pooled = synth.make_edit_inputs(h_tok, w_tok, T, cfg, seed=110 + image, dtype=torch.bfloat16)
    _, _, nprompt, npooled = synth.make_edit_inputs(h_tok, w_tok, Tn, cfg, seed=210 + image, dtype=torch.bfloat16)
    to = lambda t: t.to(device) if t is not None else None
    lat, img, prompt, nprompt, pooled, npooled = map(to, (lat, img, prompt, nprompt, pooled, npooled))
    helper = RegionEHelper(pipe)
    with contextlib.redirect_stdout(sys.stderr):
        helper.set_params()
    helper.enable()
    box = RC.make_box(h_tok, w_tok, args.edit_frac)
    B.install_region_injection(pipe, h_tok, w_tok, box, img[0:1], seed=7)
    kw = dict(image=img, prompt_embeds=prompt, negative_prompt_embeds=nprompt, height=args.size, width=args.size, latents=lat,
              true_cfg_scale=scale, return_dict=False)
    if args.family != "qwen":
        kw.update(pooled_prompt_embeds=pooled, negative_pooled_prompt_embeds=npooled)

    def run(sharded):
        helper.shard_cfg_branches(pair if sharded else None)
        pipe(**kw)                                     # warm (allocator, tables, communicator)
        holder = {}

        def job():
            holder["out"] = pipe(**kw)[0]
        el = D.timed(job, torch.cuda.synchronize, dist)
        return el, holder["out"], pipe._regione_manager.edited_ids.clone()
    t_seq, o_seq, ids_seq = run(False)
    t_sh, o_sh, ids_sh = run(True)
    same = bool(torch.equal(o_seq, o_sh) and torch.equal(ids_seq, ids_sh))
    pos, neg = pair.exchange(o_sh)                     # the partner's final latents: must equal mine
    agree = bool(torch.equal(pos, neg))
    flags = torch.tensor([int(same), int(agree)], dtype=torch.int32, device=device if args.dist_backend == "nccl" else "cpu")
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    K_e, L = int(ids_sh.shape[1]), h_tok * w_tok
    res = dict(case=f"{args.family} {args.size}^2 CFG-branch sharding", world=world, backend=args.dist_backend,
               shared_gpu=bool(args.share_gpu), K_e=K_e, L=L, both_branches_per_rank_edit_s=t_seq, sharded_edit_s=t_sh,
               speedup_from_sharding=t_seq / t_sh, sharded_bit_identical_to_unsharded=bool(flags[0].item()),
               pair_ranks_agree=bool(flags[1].item()), exchange_bytes_full_step=L * 64 * 2, exchange_bytes_region_step=K_e * 64 * 2)
    if rank == 0:
        print(json.dumps(res), flush=True)
        if args.out:
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            json.dump(res, open(args.out, "w"), indent=1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
