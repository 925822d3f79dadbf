"""A PyTorch restatement of the [EXT] `AutoencoderKL` decoder of the public FLUX.1 / Step1X-Edit checkpoints (diffusers layout and
parameter names: `decoder.conv_in`, `decoder.mid_block.{resnets,attentions}`, `decoder.up_blocks.N.{resnets,upsamplers}`,
`decoder.conv_norm_out`, `decoder.conv_out`) - the checker of regione_amd/vae.py (test infrastructure; `diffusers` is not installable
in this image, and nothing of the VAE lives in /root/reference: the reference only CALLS `self.vae.decode`, FluxKontext/inplace.py:396-402).

Semantics restated from the public diffusers sources: ResnetBlock2D (GroupNorm(32, eps 1e-6) -> SiLU -> 3 x 3 conv, twice, 1 x 1
`conv_shortcut` when the width changes, output_scale_factor 1), UNetMidBlock2D's `Attention` (one head of width C, GroupNorm without
activation, scaled dot product, `to_out.0`, residual connection), UpDecoderBlock2D (layers_per_block + 1 ResNets, nearest 2 x upsample +
3 x 3 conv), `conv_norm_out` -> SiLU -> `conv_out`."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, eps=1e-6):
        super().__init__()
        self.norm1, self.conv1 = nn.GroupNorm(32, cin, eps=eps), nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2, self.conv2 = nn.GroupNorm(32, cout, eps=eps), nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (self.conv_shortcut(x) if hasattr(self, "conv_shortcut") else x) + h


class Attention(nn.Module):
    def __init__(self, c, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, c, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Identity()])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(q @ k.transpose(1, 2) / (c ** 0.5), dim=-1)
        return x + self.to_out[0](p @ v).transpose(1, 2).reshape(b, c, h, w)


class MidBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c), ResnetBlock2D(c, c)])
        self.attentions = nn.ModuleList([Attention(c)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n, up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout) for j in range(n)])
        if up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.upsamplers[0](x) if hasattr(self, "upsamplers") else x


class Decoder(nn.Module):
    def __init__(self, block_out_channels=(128, 256, 512, 512), latent_channels=16, layers_per_block=2):
        super().__init__()
        ch = list(reversed(block_out_channels))
        self.conv_in = nn.Conv2d(latent_channels, ch[0], 3, padding=1)
        self.mid_block = MidBlock(ch[0])
        self.up_blocks = nn.ModuleList()
        cin = ch[0]
        for i, co in enumerate(ch):
            self.up_blocks.append(UpDecoderBlock2D(cin, co, layers_per_block + 1, i < len(ch) - 1))
            cin = co
        self.conv_norm_out, self.conv_out = nn.GroupNorm(32, cin, eps=1e-6), nn.Conv2d(cin, 3, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n, down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout) for j in range(n)])
        if down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.downsamplers[0](x) if hasattr(self, "downsamplers") else x


class Encoder(nn.Module):
    """diffusers `Encoder` of the AutoencoderKL: conv_in, DownEncoderBlock2D x 4 (layers_per_block ResNets, stride-2 conv after the first
    three), the mid block, conv_norm_out -> SiLU -> conv_out to 2 x latent channels (mean | logvar)."""

    def __init__(self, block_out_channels=(128, 256, 512, 512), latent_channels=16, layers_per_block=2):
        super().__init__()
        ch = list(block_out_channels)
        self.conv_in = nn.Conv2d(3, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        cin = ch[0]
        for i, co in enumerate(ch):
            self.down_blocks.append(DownEncoderBlock2D(cin, co, layers_per_block, i < len(ch) - 1))
            cin = co
        self.mid_block = MidBlock(cin)
        self.conv_norm_out, self.conv_out = nn.GroupNorm(32, cin, eps=1e-6), nn.Conv2d(cin, 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class AutoencoderKLStandIn(nn.Module):
    """`vae.decode(z, return_dict=False)[0]` / `vae.encode(x).latent_dist` of the host pipeline (no quant / post_quant conv: FLUX.1)."""

    def __init__(self, **kw):
        super().__init__()
        self.decoder = Decoder(**kw)
        self.encoder = Encoder(**kw)

    def encode(self, x, return_dict=True):
        moments = self.encoder(x)

        class Dist:
            mean = moments[:, : moments.shape[1] // 2]

            def mode(self_):
                return self_.mean

            def sample(self_, generator=None):
                lv = moments[:, moments.shape[1] // 2:].clamp(-30, 20)
                return self_.mean + torch.exp(0.5 * lv) * torch.randn(self_.mean.shape, generator=generator).to(moments)
        return type("AutoencoderKLOutput", (), {"latent_dist": Dist()})()

    def decode(self, z, return_dict=True):
        out = self.decoder(z)
        return (out,) if not return_dict else type("DecoderOutput", (), {"sample": out})()


def seeded(seed=0, **kw):
    """Checkpoint-like statistics without a checkpoint: default conv / linear initialisation, GroupNorm weights near 1 with spread, biases."""
    torch.manual_seed(seed)
    m = AutoencoderKLStandIn(**kw)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.GroupNorm):
                mod.weight.copy_(1.0 + 0.2 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
    return m.eval()
