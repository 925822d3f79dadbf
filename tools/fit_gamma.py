#!/usr/bin/env python3
"""Fit the Adaptive-Velocity-Decay table `gamma` for a pipeline and a step count (GPU box only).

The reference ships one fitted 27-entry table per model family (FluxKontext/inplace.py:47-50, …) and forbids any
other step count (utils.py:391: "Changing the inference step requires fitting a new gamma") - but not the fitting
code.  On a cache-served step the loop uses `noise_pred_i = cache * ratio_i`, `ratio_i = gamma[i-1] * (1 + (t_i -
t_{i-1}) / 1000)` (inplace.py:295-318), where `cache` is the last computed velocity.  The least-squares scalar that
maps the previous velocity onto the current one is r_i = <v_i, v_{i-1}> / <v_{i-1}, v_{i-1}>, so

    gamma[i-1] = mean_over_samples(r_i) / (1 + (t_i - t_{i-1}) / 1000),      i = 1 .. N-1

measured on FULL-token denoising runs (RegionE disabled) of the given engine.

    python tools/fit_gamma.py --steps 50 --size 1024 --samples 4 [--toy] [--out gamma_50.json]
    helper.set_params(num_inference_steps=50, gamma=json.load(open("gamma_50.json"))["gamma"])

With the synthetic random-weight engines of this repository the numbers only exercise the procedure; fit on the real
checkpoint for a usable table."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from regione_amd import synth  # noqa: E402
from tools.run_configs import weights_stream  # noqa: E402


def fit(pipe, call, n_steps: int, samples: int):
    """call(seed) runs one full-token edit; returns the fp16 table [n_steps - 1] and the per-step ratios."""
    sch = pipe.scheduler
    vs = []
    orig = sch.step

    def step(model_output, timestep, sample, **kw):
        vs.append(model_output.float().flatten())
        return orig(model_output, timestep, sample, **kw)
    sch.step = step
    num = torch.zeros(n_steps - 1, dtype=torch.float64)
    try:
        for s in range(samples):
            vs.clear()
            call(s)
            assert len(vs) == n_steps
            for i in range(1, n_steps):
                num[i - 1] += float(torch.dot(vs[i], vs[i - 1]) / torch.dot(vs[i - 1], vs[i - 1]))
    finally:
        sch.step = orig
    ratio = num / samples
    ts = sch.timesteps.double().cpu()
    gamma = ratio / (1 + (ts[1:] - ts[:-1]) / 1000)
    return gamma.to(torch.float16), ratio


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=28)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=2)
    ap.add_argument("--toy", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from regione_amd.harness import flux as HF
    dev = torch.device("cuda", 0)
    cfg = synth.FluxConfig(**synth.TOY) if a.toy else synth.FluxConfig()
    pipe = HF.FluxKontextPipeline(HF.FluxTransformer2DModel(cfg, dev).load_state_dict_stream(weights_stream(cfg, dev, 42)))
    h = w = a.size // 16
    T = 32 if a.toy else 512

    def call(seed):
        lat, img, prompt, pooled = [t.to(dev) for t in synth.make_edit_inputs(h, w, T, cfg, seed=1000 + seed)]
        pipe(image=img, prompt_embeds=prompt, pooled_prompt_embeds=pooled, height=a.size, width=a.size, latents=lat,
             num_inference_steps=a.steps, guidance_scale=2.5, return_dict=False)
    gamma, ratio = fit(pipe, call, a.steps, a.samples)
    res = dict(num_inference_steps=a.steps, size=a.size, samples=a.samples, gamma=[float(x) for x in gamma],
               ratio=[float(x) for x in ratio])
    print(json.dumps(res))
    if a.out:
        json.dump(res, open(a.out, "w"))


if __name__ == "__main__":
    main()
