#!/usr/bin/env python3
"""Where a region step's time goes, launch by launch (GPU box only).

    cd /tmp && rocprofv3 --kernel-trace -d out -o rs -- python $REPO/tools/probes/region_step_trace.py run [family] [blocks]
    python tools/probes/region_step_trace.py report out/**/rs_results.db

`run`: a family's engine at its public dimensions with `blocks` double (+ `blocks` single) blocks, one warm RegionE edit, then one
more; `report`: the dispatches of the LAST region step of the trace in order - kernel, duration, idle gap since the previous
dispatch ended - plus totals per kernel family and the share of the step that is gaps."""
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(family="qwen", blocks=8):
    import contextlib
    import torch
    import bench as B
    from regione_amd import RegionEHelper, synth
    from regione_amd.harness import flux as HF, qwen as HQ, step1x as HS
    from tools.run_configs import weights_stream, make_box
    dev = torch.device("cuda", 0)
    size, T, Tn = (512, 512, 512) if family == "step1x" else (1024, 512, 384 if family == "qwen" else 512)
    h = w = size // 16
    if family == "qwen":
        cfg = synth.FluxConfig(**dict(synth.QWEN, n_double=blocks))
        pipe = HQ.QwenImageEditPipeline(HQ.QwenImageTransformer2DModel(cfg, dev).load_state_dict_stream(weights_stream(cfg, dev, 42)))
    elif family == "step1x":
        cfg = synth.FluxConfig(guidance_embeds=False, n_double=blocks, n_single=blocks)
        pipe = HS.Step1XEditPipeline(HS.Step1XEditTransformer2DModel(cfg, dev).load_state_dict_stream(weights_stream(cfg, dev, 42)))
    else:
        cfg = synth.FluxConfig(n_double=blocks, n_single=blocks)
        pipe = HF.FluxKontextPipeline(HF.FluxTransformer2DModel(cfg, dev).load_state_dict_stream(weights_stream(cfg, dev, 42)))
    lat, img, prompt, pooled = [t.to(dev) if t is not None else None for t in synth.make_edit_inputs(h, w, T, cfg, seed=110)]
    _, _, nprompt, npooled = [t.to(dev) if t is not None else None for t in synth.make_edit_inputs(h, w, Tn, cfg, seed=111)]
    helper = RegionEHelper(pipe)
    with contextlib.redirect_stdout(sys.stderr):
        helper.set_params()
    helper.enable()
    B.install_region_injection(pipe, h, w, make_box(h, w, 0.25), img[0:1], seed=7)
    kw = dict(image=img, prompt_embeds=prompt, height=size, width=size, latents=lat, return_dict=False)
    if family == "qwen":
        kw.update(negative_prompt_embeds=nprompt, true_cfg_scale=4.0)
    elif family == "step1x":
        kw.update(pooled_prompt_embeds=pooled, negative_prompt_embeds=nprompt, negative_pooled_prompt_embeds=npooled, true_cfg_scale=6.0)
    else:
        kw.update(pooled_prompt_embeds=pooled, guidance_scale=2.5)
    for _ in range(2):
        tr = {}
        pipe(trace=tr, **kw)
        torch.cuda.synchronize()
    print("kinds", "".join(tr["kind"]), "K_e", int(pipe._regione_manager.edited_ids.shape[1]), file=sys.stderr)


def report(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    namecol = "display_name" if "display_name" in scols else "kernel_name"
    rows = list(c.execute(f"select s.{namecol}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    # region steps = the stretches between two euler_kernel dispatches that contain no gemv (full steps recompute nothing else,
    # but region steps are the SHORT ones): take the last stretch shorter than half of the longest
    cuts = [i for i, r in enumerate(rows) if "euler_kernel" in r[0]]
    spans = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b - a > 20]
    longest = max(rows[b][2] - rows[a][2] for a, b in spans)
    reg = [(a, b) for a, b in spans if rows[b][2] - rows[a][2] < 0.5 * longest]
    a, b = reg[-1]
    seq = rows[a + 1:b]
    t0, t1 = seq[0][1], seq[-1][2]
    busy = sum(e - s for _, s, e in seq)
    print(f"# last region step: {len(seq)} dispatches, {1e-6 * (t1 - t0):.3f} ms wall, {1e-6 * busy:.3f} ms in kernels, "
          f"{100.0 * (1 - busy / (t1 - t0)):.1f} % idle between dispatches")
    fam = {}
    prev = None
    for i, (n, s, e) in enumerate(seq):
        short = n.split("(")[0].replace("void rgn::", "").replace("rgn::", "")[:60]
        gap = 0 if prev is None else s - prev
        f = fam.setdefault(short, [0, 0, 0])
        f[0] += 1; f[1] += e - s; f[2] += max(gap, 0)
        if i < int(os.environ.get("TRACE_ROWS", "40")):
            print(f"{short:<62} {1e-3 * (e - s):8.1f} us   gap {1e-3 * gap:6.1f} us")
        prev = e
    print("# per kernel: calls, total ms, avg us, idle ms in front of it")
    for k, (n, tot, gap) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:<62} {n:>5} {1e-6 * tot:>9.3f} {1e-3 * tot / n:>9.1f} {1e-6 * gap:>9.3f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(*(sys.argv[2:3] or ["qwen"]), *(int(x) for x in sys.argv[3:4]))
    else:
        report(sys.argv[2])
