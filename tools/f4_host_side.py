#!/usr/bin/env python3
"""SURVEY.md section 8(f) row 4 - the host side of a FLUX.1-Kontext edit, MEASURED (GPU box only; measurement tool, nothing
here is on the product path and nothing here is a HIP kernel of this repository).

DESIGN.md section 7 closed the row "VAE decode / text encoders" as "deliberately host-side" on an ESTIMATE (6.6 TFLOP against the
loop's 1 630 TFLOP).  No `diffusers` / `transformers` checkpoints exist in this image, so this tool builds the public
architectures of the three host modules as plain PyTorch-ROCm modules with random weights (bf16, eager - what a stock diffusers
host runs) and times them on the MI355X next to the HIP denoise loop of the same process:

  encode_s = VAE encode of the 1024 x 1024 input image  (AutoencoderKL of FLUX.1: 128-256-512-512 channels, 2 ResNet blocks
             per level, one mid attention, 16 latent channels)             [reference call site FluxKontext/inplace.py:113-166]
           + T5-XXL encoder over 512 prompt tokens (24 layers, d_model 4096, 64 heads x 64, gated-GELU d_ff 10240; 4.7 B params)
           + CLIP-L text encoder over 77 tokens (12 layers, d 768)                                        [inplace.py:168-226]
  loop_s   = the 28-step RegionE edit on the HIP engine (bench.py's workload, K_e 25 %)
  decode_s = VAE decode of the 128 x 128 x 16 latent to 1024 x 1024                                       [inplace.py:396-402]

Random weights change no timing (dense convolutions / GEMMs); the figures are the host's share of an end-to-end edit.

    python tools/f4_host_side.py > profiles/r06_f4_host_side.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- AutoencoderKL (FLUX.1 VAE) [EXT: public architecture] ---------------------------------------------------------------------
class ResnetBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.n1, self.c1 = nn.GroupNorm(32, cin, eps=1e-6), nn.Conv2d(cin, cout, 3, padding=1)
        self.n2, self.c2 = nn.GroupNorm(32, cout, eps=1e-6), nn.Conv2d(cout, cout, 3, padding=1)
        self.skip = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.c1(F.silu(self.n1(x)))
        h = self.c2(F.silu(self.n2(h)))
        return (x if self.skip is None else self.skip(x)) + h


class MidAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = nn.GroupNorm(32, c, eps=1e-6)
        self.q, self.k, self.v, self.o = (nn.Linear(c, c) for _ in range(4))

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.norm(x).view(b, c, h * w).transpose(1, 2)
        a = F.scaled_dot_product_attention(self.q(t)[:, None], self.k(t)[:, None], self.v(t)[:, None])[:, 0]
        return x + self.o(a).transpose(1, 2).view(b, c, h, w)


class Encoder(nn.Module):
    def __init__(self, ch=(128, 256, 512, 512), zc=16):
        super().__init__()
        self.conv_in = nn.Conv2d(3, ch[0], 3, padding=1)
        blocks, c = [], ch[0]
        for i, co in enumerate(ch):
            blocks += [ResnetBlock(c, co), ResnetBlock(co, co)]
            c = co
            if i < len(ch) - 1:
                blocks.append(nn.Conv2d(c, c, 3, stride=2, padding=0))
        self.down = nn.ModuleList(blocks)
        self.mid = nn.Sequential(ResnetBlock(c, c), MidAttention(c), ResnetBlock(c, c))
        self.norm_out, self.conv_out = nn.GroupNorm(32, c, eps=1e-6), nn.Conv2d(c, 2 * zc, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down:
            x = b(F.pad(x, (0, 1, 0, 1))) if isinstance(b, nn.Conv2d) else b(x)
        return self.conv_out(F.silu(self.norm_out(self.mid(x))))


class Decoder(nn.Module):
    def __init__(self, ch=(128, 256, 512, 512), zc=16):
        super().__init__()
        c = ch[-1]
        self.conv_in = nn.Conv2d(zc, c, 3, padding=1)
        self.mid = nn.Sequential(ResnetBlock(c, c), MidAttention(c), ResnetBlock(c, c))
        blocks = []
        for i, co in enumerate(reversed(ch)):
            blocks += [ResnetBlock(c, co), ResnetBlock(co, co), ResnetBlock(co, co)]
            c = co
            if i < len(ch) - 1:
                blocks.append(nn.Upsample(scale_factor=2.0, mode="nearest"))
                blocks.append(nn.Conv2d(c, c, 3, padding=1))
        self.up = nn.ModuleList(blocks)
        self.norm_out, self.conv_out = nn.GroupNorm(32, c, eps=1e-6), nn.Conv2d(c, 3, 3, padding=1)

    def forward(self, z):
        x = self.mid(self.conv_in(z))
        for b in self.up:
            x = b(x)
        return self.conv_out(F.silu(self.norm_out(x)))


# ---- T5 encoder / CLIP text encoder [EXT: public architectures] ---------------------------------------------------------------
class T5Block(nn.Module):
    def __init__(self, d=4096, heads=64, dkv=64, dff=10240):
        super().__init__()
        self.h, self.dkv = heads, dkv
        self.ln1, self.ln2 = nn.RMSNorm(d, eps=1e-6), nn.RMSNorm(d, eps=1e-6)
        self.q, self.k, self.v = (nn.Linear(d, heads * dkv, bias=False) for _ in range(3))
        self.o = nn.Linear(heads * dkv, d, bias=False)
        self.wi0, self.wi1, self.wo = nn.Linear(d, dff, bias=False), nn.Linear(d, dff, bias=False), nn.Linear(dff, d, bias=False)

    def forward(self, x, bias):
        b, s, _ = x.shape
        t = self.ln1(x)
        sp = lambda y: y.view(b, s, self.h, self.dkv).transpose(1, 2)
        a = F.scaled_dot_product_attention(sp(self.q(t)), sp(self.k(t)), sp(self.v(t)), attn_mask=bias, scale=1.0)
        x = x + self.o(a.transpose(1, 2).reshape(b, s, -1))
        t = self.ln2(x)
        return x + self.wo(F.gelu(self.wi0(t), approximate="tanh") * self.wi1(t))


class T5Encoder(nn.Module):
    def __init__(self, layers=24, d=4096, heads=64):
        super().__init__()
        self.emb = nn.Embedding(32128, d)
        self.rel = nn.Embedding(32, heads)
        self.blocks = nn.ModuleList(T5Block(d, heads) for _ in range(layers))
        self.final = nn.RMSNorm(d, eps=1e-6)

    def forward(self, ids):
        s = ids.shape[1]
        bucket = (torch.arange(s, device=ids.device)[None] - torch.arange(s, device=ids.device)[:, None]).abs().clamp(max=31)
        bias = self.rel(bucket).permute(2, 0, 1)[None].to(self.emb.weight.dtype)
        x = self.emb(ids)
        for blk in self.blocks:
            x = blk(x, bias)
        return self.final(x)


class ClipText(nn.Module):
    def __init__(self, layers=12, d=768, heads=12):
        super().__init__()
        self.emb, self.pos = nn.Embedding(49408, d), nn.Embedding(77, d)
        self.layers = nn.ModuleList(nn.TransformerEncoderLayer(d, heads, 4 * d, activation="gelu", batch_first=True, norm_first=True)
                                    for _ in range(layers))
        self.final = nn.LayerNorm(d)

    def forward(self, ids):
        x = self.emb(ids) + self.pos(torch.arange(ids.shape[1], device=ids.device))[None]
        for l in self.layers:
            x = l(x)
        return self.final(x)


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def nparams(m):
    return sum(p.numel() for p in m.parameters())


@torch.no_grad()
def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    bf = torch.bfloat16
    res = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "dtype": "bf16", "image": "1024 x 1024",
           "note": "public architectures, random weights, PyTorch-ROCm eager (what a stock diffusers host runs); medians of 5"}
    enc, dec = Encoder().to(dev, bf), Decoder().to(dev, bf)
    img = torch.randn(1, 3, 1024, 1024, device=dev, dtype=bf)
    z = torch.randn(1, 16, 128, 128, device=dev, dtype=bf)
    res["vae_encode_s"] = timed(lambda: enc(img))
    res["vae_decode_s"] = timed(lambda: dec(z))
    res["vae_params_m"] = round((nparams(enc) + nparams(dec)) / 1e6, 1)
    del enc, dec
    # round 6: the same two stages on libregione_hip.so (regione_amd/vae.py; synthetic weights of the diffusers layout)
    from regione_amd import vae as V
    hdec = V.HipVaeDecoder(V.synthetic_decoder_state_dict(3, device=dev), dev)
    henc = V.HipVaeEncoder(V.synthetic_decoder_state_dict(4, device=dev, shapes=V.encoder_param_shapes()), dev)
    res["hip_vae_encode_s"] = timed(lambda: henc.encode(img.clamp(-1, 1)))
    res["hip_vae_decode_s"] = timed(lambda: hdec.decode(z))
    del hdec, henc
    t5 = T5Encoder().to(dev, bf)
    ids = torch.randint(0, 32000, (1, 512), device=dev)
    res["t5_xxl_512_tokens_s"] = timed(lambda: t5(ids))
    res["t5_params_b"] = round(nparams(t5) / 1e9, 2)
    del t5
    clip = ClipText().to(dev, bf)
    cids = torch.randint(0, 49000, (1, 77), device=dev)
    res["clip_l_77_tokens_s"] = timed(lambda: clip(cids))
    del clip
    torch.cuda.empty_cache()
    # the HIP denoise loop of the same process (bench.py's workload)
    import contextlib
    import bench as B
    from regione_amd import RegionEHelper, synth
    cfg = synth.FluxConfig()
    pipe = B.build_pipeline(cfg, dev, seed=42)
    h = w = 64
    lat, im, prompt, pooled = [t.to(dev) for t in synth.make_edit_inputs(h, w, 512, cfg, seed=110, dtype=bf)]
    helper = RegionEHelper(pipe)
    with contextlib.redirect_stdout(sys.stderr):
        helper.set_params(threshold=0.88, cache_threshold=0.04)
    helper.enable()
    B.install_region_injection(pipe, h, w, (17, 47, 17, 47), im[0:1], seed=7)
    run = lambda: pipe(image=im, prompt_embeds=prompt, pooled_prompt_embeds=pooled, height=1024, width=1024, latents=lat,
                       guidance_scale=2.5, return_dict=False)
    res["loop_s"] = timed(run, reps=3)
    helper.disable()
    res["loop_full_token_s"] = timed(run, reps=1)
    res["encode_s"] = res["vae_encode_s"] + res["t5_xxl_512_tokens_s"] + res["clip_l_77_tokens_s"]
    res["decode_s"] = res["vae_decode_s"]
    tot = res["encode_s"] + res["loop_s"] + res["decode_s"]
    res["end_to_end_s"] = tot
    res["host_share_of_end_to_end"] = (res["encode_s"] + res["decode_s"]) / tot
    hip_tot = res["hip_vae_encode_s"] + res["t5_xxl_512_tokens_s"] + res["clip_l_77_tokens_s"] + res["loop_s"] + res["hip_vae_decode_s"]
    res["end_to_end_hip_vae_s"] = hip_tot
    res["host_share_of_end_to_end_hip_vae"] = (res["t5_xxl_512_tokens_s"] + res["clip_l_77_tokens_s"]) / hip_tot
    res["end_to_end_speedup_vs_full_token_loop_hip_vae"] = (hip_tot - res["loop_s"] + res["loop_full_token_s"]) / hip_tot
    res["end_to_end_speedup_vs_full_token_loop"] = (res["encode_s"] + res["loop_full_token_s"] + res["decode_s"]) / tot
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
