#!/usr/bin/env python3
"""Timing-protocol twin of the reference drivers (src/FluxKontext/main.py:41-129) for the denoise loop (GPU box only).

Same protocol: read a jsonl of {"key", "instruction"} items, 3 untimed warm-up edits, then per item
`torch.cuda.synchronize(); t0; pipe(...); torch.cuda.synchronize(); t1`, and a `time_consuming.json` with the
reference's keys (`num_item`, `ave_time_consuming`, `time_consuming_list`).  What differs, by necessity: this image
has no checkpoints, text encoders or VAE, so each item's prompt embeddings / latents are synthetic, seeded by a hash
of (key, instruction, --seed); the timed region is therefore the denoise loop alone (the reference's includes prompt
encoding and VAE decode).  With --compare the same items are also run full-token and the latent-space PSNR of the
two outputs is reported per item (the stand-in for evaluation/metric_all_task.py:85-100, which needs decoded images).

    python tools/edit_driver.py --image_path /path/to/data.jsonl --use_regione --compare --output_dir result/FluxKontext

Hosted mode (`--pipeline_factory pkg.module:function`): the END-TO-END protocol of the reference drivers on a stock pipeline
OBJECT - `pipe = factory(); RegionEHelper(pipe).set_params(...).enable(); pipe(image=, prompt=)` per item, the way
src/FluxKontext/main.py:41-129 does it - with the three stages of an edit reported per item and averaged:
`stages = {encode_s, loop_s, decode_s}` (image preprocessing + prompt encoders + VAE encode | the denoise loop on the HIP
engine | VAE decode + post-processing; `HostedOutput.timing`).  The factory returns any of the five pipeline classes the
reference patches (a diffusers `from_pretrained(...)` object on a box that has diffusers and the checkpoint); this image has
neither, so the default factory `tests.host_standins:make_<family>` builds the stand-in pipelines of the test-suite (toy
trunk, toy VAE and prompt encoders) - the plumbing and the report format are what it exercises here.
"""
import argparse
import contextlib
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from regione_amd import RegionEHelper, synth  # noqa: E402
from tools.run_configs import weights_stream  # noqa: E402


def item_seed(key: str, instruction: str, seed: int) -> int:
    return int.from_bytes(hashlib.sha256(f"{key}|{instruction}|{seed}".encode()).digest()[:4], "little")


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.float(), b.float()
    mse = float(((a - b) ** 2).mean())
    peak = float(b.abs().max())
    return float("inf") if mse == 0 else 10.0 * torch.log10(torch.tensor(peak * peak / mse)).item()


def hosted_main(a):
    """End-to-end protocol on a stock pipeline object (see the module docstring)."""
    import importlib
    mod, fn = a.pipeline_factory.split(":")
    pipe = getattr(importlib.import_module(mod), fn)()
    helper = RegionEHelper(pipe)
    with contextlib.redirect_stdout(sys.stderr):
        helper.set_params(num_inference_steps=a.num_inference_steps, warmup_step=a.warmup_step, post_step=a.post_step,
                          refresh_step=a.refresh_step, threshold=a.threshold, cache_threshold=a.cache_threshold,
                          erosion_dilation=a.erosion_dilation)
    items = [json.loads(line) for line in open(a.image_path) if line.strip()]

    def picture(key):
        """The item's input image: <key> if it is a readable image file (needs PIL on the box), else a seeded synthetic one."""
        try:
            from PIL import Image
            import numpy as np
            im = np.asarray(Image.open(key).convert("RGB"), dtype="float32") / 255.0
            return torch.from_numpy(im).permute(2, 0, 1).unsqueeze(0)
        except Exception:
            g = torch.Generator().manual_seed(item_seed(key, "", a.seed))
            p = torch.rand(1, 3, a.size, a.size, generator=g)
            p[:, :, a.size // 4: a.size // 2, a.size // 3: 2 * a.size // 3] = 0.0
            return p

    def edit(key, instruction):
        g = torch.Generator().manual_seed(item_seed(key, instruction, a.seed))
        return pipe(image=picture(key), prompt=instruction, generator=g, output_type="pt", num_inference_steps=a.num_inference_steps)

    def run_all(tag):
        for _ in range(3):
            edit("assets/demo_0", "just warmup!")
        times, stages = [], []
        for index, data in enumerate(items):
            torch.cuda.synchronize()
            t0 = time.time()
            out = edit(data["key"], data["instruction"])
            torch.cuda.synchronize()
            times.append(time.time() - t0)
            stages.append(dict(getattr(out, "timing", {})))
            print(f"[{tag} {index + 1} / {len(items)}] {data['key']}: {data['instruction']!r}  Time consuming: {times[-1]}s  {stages[-1]}",
                  file=sys.stderr)
        return times, stages

    def mean_stages(stages):
        keys = ("encode_s", "loop_s", "decode_s")
        return {k: sum(s.get(k, 0.0) for s in stages) / max(len(stages), 1) for k in keys}
    os.makedirs(a.output_dir, exist_ok=True)
    if a.use_regione:
        helper.enable()
    else:            # full-token loop on the HIP engine through the same hosted call (class swap without the RegionE patch set)
        from regione_amd import adapters
        adapters.attach(pipe)
        adapters.swap_host_class(pipe)
    times, stages = run_all("RegionE" if a.use_regione else "full-token")
    report = {"num_item": len(times), "ave_time_consuming": sum(times) / len(times), "time_consuming_list": times,
              "stages": mean_stages(stages), "stages_per_item": stages, "pipeline": type(pipe).__name__}
    if a.compare and a.use_regione:
        helper.disable()
        from regione_amd import adapters
        adapters.swap_host_class(pipe)
        rtimes, rstages = run_all("full-token")
        report["full_token"] = {"ave_time_consuming": sum(rtimes) / len(rtimes), "time_consuming_list": rtimes, "stages": mean_stages(rstages)}
        report["speedup"] = report["full_token"]["ave_time_consuming"] / report["ave_time_consuming"]
        report["loop_speedup"] = report["full_token"]["stages"]["loop_s"] / max(report["stages"]["loop_s"], 1e-9)
    json.dump(report, open(os.path.join(a.output_dir, "time_consuming.json"), "w"), indent=4)
    print(json.dumps(report))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=110)
    ap.add_argument("--num_inference_steps", type=int, default=28)
    ap.add_argument("--guidance_scale", type=float, default=2.5)
    ap.add_argument("--use_regione", action="store_true")
    ap.add_argument("--warmup_step", type=int, default=6)
    ap.add_argument("--post_step", type=int, default=2)
    ap.add_argument("--refresh_step", type=str, default="16")
    ap.add_argument("--threshold", type=float, default=0.93)
    ap.add_argument("--cache_threshold", type=float, default=0.04)
    ap.add_argument("--erosion_dilation", action="store_true")
    ap.add_argument("--image_path", type=str, required=True, help="jsonl with one {'key', 'instruction'} object per line")
    ap.add_argument("--output_dir", type=str, default="result/FluxKontext/Demo/RegionE")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--toy", action="store_true", help="toy-size engine (plumbing check)")
    ap.add_argument("--compare", action="store_true", help="also run full-token and report latent PSNR per item")
    ap.add_argument("--pipeline_factory", type=str, default=None,
                    help="hosted mode: 'pkg.module:function' returning a stock pipeline object (FluxKontext / Step1XEdit(V1P2) / "
                         "QwenImageEdit(Plus)Pipeline); e.g. tests.host_standins:make_flux")
    ap.add_argument("--overlay_dir", type=str, default=None,
                    help="with --use_regione: write <key>.mask.png per item - the edited-token partition painted on the pixel grid "
                         "(tools/overlay.py; reference src/Step1X-Edit-v1p2/inplace.py:456-497)")
    a = ap.parse_args()

    if a.pipeline_factory:
        return hosted_main(a)
    from regione_amd.harness import flux as HF
    dev = torch.device("cuda", 0)
    cfg = synth.FluxConfig(**synth.TOY) if a.toy else synth.FluxConfig()
    pipe = HF.FluxKontextPipeline(HF.FluxTransformer2DModel(cfg, dev).load_state_dict_stream(weights_stream(cfg, dev, 42)))
    helper = RegionEHelper(pipe)
    with contextlib.redirect_stdout(sys.stderr):
        helper.set_params(num_inference_steps=a.num_inference_steps, warmup_step=a.warmup_step, post_step=a.post_step,
                          refresh_step=a.refresh_step, threshold=a.threshold, cache_threshold=a.cache_threshold,
                          erosion_dilation=a.erosion_dilation)
    h = w = a.size // 16
    T = 32 if a.toy else 512
    items = [json.loads(line) for line in open(a.image_path) if line.strip()]

    def edit(key, instruction):
        s = item_seed(key, instruction, a.seed)
        lat, img, prompt, pooled = [t.to(dev) for t in synth.make_edit_inputs(h, w, T, cfg, seed=s)]
        return pipe(image=img, prompt_embeds=prompt, pooled_prompt_embeds=pooled, height=a.size, width=a.size, latents=lat,
                    num_inference_steps=a.num_inference_steps, guidance_scale=a.guidance_scale, return_dict=False)[0]

    edited = []

    def run_all(tag):
        print("Warmup...", file=sys.stderr)
        for _ in range(3):
            edit("assets/demo_0", "just warmup!")
        outs, times = {}, []
        for index, data in enumerate(items):
            torch.cuda.synchronize()
            t0 = time.time()
            outs[data["key"]] = edit(data["key"], data["instruction"])
            torch.cuda.synchronize()
            t1 = time.time()
            times.append(t1 - t0)
            M = getattr(pipe, "_regione_manager", None)
            if a.overlay_dir and M is not None and M.edited_ids is not None:
                from tools import overlay
                os.makedirs(a.overlay_dir, exist_ok=True)
                overlay.save_overlay(os.path.join(a.overlay_dir, os.path.basename(data["key"]) + ".mask.png"),
                                     M.edited_ids.cpu().numpy(), a.size, a.size)
                edited.append(int(M.edited_ids.shape[1]))
            print(f"[{tag} {index + 1} / {len(items)}] {data['key']}: {data['instruction']!r}  Time consuming: {t1 - t0}s", file=sys.stderr)
        return outs, times

    os.makedirs(a.output_dir, exist_ok=True)
    report = {}
    if a.use_regione:
        helper.enable()
    outs, times = run_all("RegionE" if a.use_regione else "full-token")
    report = {"num_item": len(times), "ave_time_consuming": sum(times) / len(times), "time_consuming_list": times,
              # end-to-end = encode + loop + decode (hosted pipelines report the three stages in `.timing`); this driver has no
              # encoders / VAE (synthetic embeddings and latents), so its timed region is the loop alone
              "stages": {"encode_s": 0.0, "loop_s": sum(times) / len(times), "decode_s": 0.0}}
    if edited:
        report["edited_tokens"] = edited
    if a.compare and a.use_regione:
        helper.disable()
        ref, rtimes = run_all("full-token")
        report["full_token"] = {"ave_time_consuming": sum(rtimes) / len(rtimes), "time_consuming_list": rtimes}
        report["speedup"] = report["full_token"]["ave_time_consuming"] / report["ave_time_consuming"]
        report["latent_psnr_vs_full_token_db"] = {k: psnr(outs[k], ref[k]) for k in outs}
    for k, v in outs.items():
        torch.save(v.cpu(), os.path.join(a.output_dir, os.path.basename(k) + ".latent.pt"))
    json.dump(report, open(os.path.join(a.output_dir, "time_consuming.json"), "w"), indent=4)
    print(json.dumps(report))


if __name__ == "__main__":
    main()
