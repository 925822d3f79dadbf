"""VAE decode on the HIP kernels.

The reference ends every edit with `image = self.vae.decode(latents, return_dict=False)[0]` (FluxKontext/inplace.py:396-402; the
Step1X-Edit pipelines make the same call) and its latency protocol times the whole `pipe(...)` call (src/FluxKontext/main.py:62-73),
so the decode is part of the reference's end-to-end edit time.  The module is the [EXT] `AutoencoderKL` decoder of the public
FLUX.1 / Step1X-Edit checkpoints (diffusers layout: conv_in -> mid block (ResNet, one attention head of width 512, ResNet) ->
four up levels of three ResNet blocks (512, 512, 256, 128 channels; nearest 2 x upsample + 3 x 3 conv after the first three) ->
GroupNorm + SiLU + conv_out); nothing of it lives in /root/reference: [EXT], unpinned, checked against an fp32 PyTorch module of the
same architecture (tests/host_vae.py) instead of the oracle.

`HipVaeDecoder(state_dict, device)` adopts a decoder's parameters (diffusers names, with or without the `decoder.` prefix) once:
3 x 3 kernels re-laid [Cout, Cin, 3, 3] -> [Cout, 3, 3, Cin] bf16.  `decode(z)` runs ~95 launches, all `rgn::` kernels:

  * every convolution = rgn_conv_bf16: an implicit GEMM on the hand-scheduled 256 x 256 MFMA loop over a zero-bordered, pixel-major
    activation image (csrc/vae.hip header), bias / ResNet skip / border zeroing in its epilogue;
  * GroupNorm(32) + SiLU = rgn_groupnorm_silu (statistics pass + apply pass, bit-reproducible);
  * the mid-block attention = three GEMMs + rgn_softmax_rows;
  * nearest upsample, NCHW <-> padded pixel-major conversions = row kernels.

No CPU / eager fallback: a missing library raises RegionEHipError like every other op of the package.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib, ops

_p, _stream = ops._p, ops._stream


class PaddedImage:
    """A zero-bordered pixel-major activation image [Hp * Wp, C] bf16 inside a storage tensor with guard rows on both sides."""

    def __init__(self, H: int, W: int, C: int, device):
        self.H, self.W, self.C = H, W, C
        self.Hp, self.Wp = H + 2, W + 2
        self.rows = self.Hp * self.Wp
        g = self.Wp + 72                     # guard rows: a window reads up to Wp + 1 + group rows outside the image, pixel groups write group - 1
        self.storage = torch.zeros((self.rows + 2 * g, C), dtype=torch.bfloat16, device=device)
        self.t = self.storage[g:g + self.rows]

    def ptr(self) -> int:
        return self.t.data_ptr()


class _Pool:
    """Per-decoder activation buffers, reused across layers and calls (everything runs stream-ordered on one stream)."""

    def __init__(self, device):
        self.device = device
        self.free: Dict[Tuple[int, int, int], List[PaddedImage]] = {}

    def get(self, H, W, C) -> PaddedImage:
        lst = self.free.setdefault((H, W, C), [])
        return lst.pop() if lst else PaddedImage(H, W, C, self.device)

    def put(self, img: PaddedImage):
        self.free.setdefault((img.H, img.W, img.C), []).append(img)


class ConvWeights:
    """One convolution's parameters as rgn_conv_bf16 wants them.  `w4`: [Cout, kh, kw, Cin] fp32 (kh = kw = 3 or 1).  group = 1: the matrix
    [Cout, taps * Cin].  group > 1 (narrow outputs): the block-Toeplitz matrix over a window of group + 2 pixels per kernel row - output row =
    `group` pixels x `ldy` columns, row p * ldy + c holds w4[c, ky, kx] at window pixel p + kx of kernel row ky (include/regione_hip.h)."""

    def __init__(self, w4: torch.Tensor, bias: torch.Tensor, group: int = 1, ldy: Optional[int] = None):
        co, kh, kw, ci = w4.shape
        assert kh == kw and kh in (1, 3)
        self.taps, self.cout, self.cin, self.group = kh * kw, co, ci, group
        self.ldy = co if ldy is None else ldy
        dev = w4.device
        if group == 1:
            self.w = w4.reshape(co, -1).to(torch.bfloat16).contiguous()
            self.b = bias.to(dev, torch.bfloat16).contiguous()
        else:
            win = group + 2 if kh == 3 else group
            t = torch.zeros((group, self.ldy, kh, win, ci), dtype=torch.float32, device=dev)
            for p in range(group):
                for kx in range(kw):
                    t[p, :co, :, p + kx, :] = w4[:, :, kx, :]
            self.w = t.reshape(group * self.ldy, -1).to(torch.bfloat16).contiguous()
            b = torch.zeros((group, self.ldy), dtype=torch.float32, device=dev)
            b[:, :co] = bias.to(dev, torch.float32)
            self.b = b.reshape(-1).to(torch.bfloat16).contiguous()


_gn_ws: Dict[Tuple[int, int], torch.Tensor] = {}
_gn_owner: Dict[Tuple[int, int], Tuple[Optional["PaddedImage"], int]] = {}      # whose statistics the workspace's partial sums hold, tile count


def _gn_key(device):
    return (device.index if device.index is not None else torch.cuda.current_device(), _stream())


def _gn_workspace(device) -> torch.Tensor:
    key = _gn_key(device)
    if key not in _gn_ws:
        _gn_ws[key] = torch.empty(_lib.lib().rgn_groupnorm_workspace_bytes() // 4, dtype=torch.float32, device=device)
    return _gn_ws[key]


def conv(x: PaddedImage, cw: ConvWeights, out: PaddedImage, resid: Optional[PaddedImage] = None, gn: bool = False):
    """out = conv(x) + bias (+ resid), border rows zero (rgn_conv_bf16).  gn = True: the epilogue also leaves the GroupNorm statistics of `out`
    in the stream's groupnorm workspace (the next `groupnorm_silu(out, ...)` then skips its statistics pass)."""
    if cw.cin != x.C or out.C != cw.ldy or (resid is not None and (resid.C != out.C or resid.rows != out.rows)) or out.rows != x.rows:
        raise _lib.RegionEHipError(f"conv: weights for {cw.cin} -> {cw.cout} (row stride {cw.ldy}) on images with {x.C} -> {out.C} channels")
    import ctypes
    dev = x.t.device
    nblk = ctypes.c_int(0)
    ws = _gn_workspace(dev) if gn else None
    _gn_owner[_gn_key(dev)] = (None, 0)                  # whatever the workspace held is about to be overwritten (or is left stale: same answer)
    rc = _lib.lib().rgn_conv_bf16(x.ptr(), x.C, _p(cw.w), _p(cw.b), None if resid is None else resid.ptr(), out.ptr(), out.C, x.Hp, x.Wp,
                                  x.C, cw.cout, cw.taps, cw.group, _p(ws), ctypes.byref(nblk) if gn else None, _stream())
    _lib.check(rc, "rgn_conv_bf16")
    if gn:
        _gn_owner[_gn_key(dev)] = (out, nblk.value)
    return out


def groupnorm_silu(x: PaddedImage, gamma: torch.Tensor, beta: torch.Tensor, out: PaddedImage, silu: bool = True, eps: float = 1e-6):
    dev = x.t.device
    ws = _gn_workspace(dev)
    owner, nblk = _gn_owner.get(_gn_key(dev), (None, 0))
    pre = nblk if owner is x else 0                      # the convolution that wrote x left its tiles' sums in the workspace
    _gn_owner[_gn_key(dev)] = (None, 0)                  # consumed
    rc = _lib.lib().rgn_groupnorm_silu(x.ptr(), out.ptr(), x.Hp, x.Wp, x.C, _p(gamma), _p(beta), float(eps), int(silu), _p(ws), pre, _stream())
    _lib.check(rc, "rgn_groupnorm_silu")
    return out


def upsample2x(x: PaddedImage, out: PaddedImage):
    assert out.H == 2 * x.H and out.W == 2 * x.W and out.C == x.C
    _lib.check(_lib.lib().rgn_upsample2x(x.ptr(), out.ptr(), x.Hp, x.Wp, x.C, _stream()), "rgn_upsample2x")
    return out


def decoder_param_shapes(block_out_channels=(128, 256, 512, 512), latent_channels: int = 16, layers_per_block: int = 2):
    """Parameter names and shapes of the [EXT] AutoencoderKL decoder (diffusers state-dict layout) - for synthetic weights (bench.py,
    tools): a real checkpoint's `vae.decoder.state_dict()` has exactly these entries."""
    ch = list(reversed(block_out_channels))
    s: Dict[str, Tuple[int, ...]] = {}

    def conv_(n, co, ci, k):
        s[n + ".weight"], s[n + ".bias"] = (co, ci, k, k), (co,)

    def norm_(n, c):
        s[n + ".weight"], s[n + ".bias"] = (c,), (c,)

    def res_(n, ci, co):
        norm_(n + ".norm1", ci); conv_(n + ".conv1", co, ci, 3); norm_(n + ".norm2", co); conv_(n + ".conv2", co, co, 3)
        if ci != co:
            conv_(n + ".conv_shortcut", co, ci, 1)
    top = ch[0]
    conv_("conv_in", top, latent_channels, 3)
    res_("mid_block.resnets.0", top, top); res_("mid_block.resnets.1", top, top)
    a = "mid_block.attentions.0."
    norm_(a + "group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[a + n + ".weight"], s[a + n + ".bias"] = (top, top), (top,)
    cin = top
    for i, co in enumerate(ch):
        for j in range(layers_per_block + 1):
            res_(f"up_blocks.{i}.resnets.{j}", cin if j == 0 else co, co)
        if i < len(ch) - 1:
            conv_(f"up_blocks.{i}.upsamplers.0.conv", co, co, 3)
        cin = co
    norm_("conv_norm_out", cin)
    conv_("conv_out", 3, cin, 3)
    return s


def encoder_param_shapes(block_out_channels=(128, 256, 512, 512), latent_channels: int = 16, layers_per_block: int = 2):
    """The same for the encoder (`vae.encoder.state_dict()`)."""
    ch = list(block_out_channels)
    s: Dict[str, Tuple[int, ...]] = {}

    def conv_(n, co, ci, k):
        s[n + ".weight"], s[n + ".bias"] = (co, ci, k, k), (co,)

    def norm_(n, c):
        s[n + ".weight"], s[n + ".bias"] = (c,), (c,)

    def res_(n, ci, co):
        norm_(n + ".norm1", ci); conv_(n + ".conv1", co, ci, 3); norm_(n + ".norm2", co); conv_(n + ".conv2", co, co, 3)
        if ci != co:
            conv_(n + ".conv_shortcut", co, ci, 1)
    conv_("conv_in", ch[0], 3, 3)
    cin = ch[0]
    for i, co in enumerate(ch):
        for j in range(layers_per_block):
            res_(f"down_blocks.{i}.resnets.{j}", cin if j == 0 else co, co)
        if i < len(ch) - 1:
            conv_(f"down_blocks.{i}.downsamplers.0.conv", co, co, 3)
        cin = co
    res_("mid_block.resnets.0", cin, cin); res_("mid_block.resnets.1", cin, cin)
    a = "mid_block.attentions.0."
    norm_(a + "group_norm", cin)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[a + n + ".weight"], s[a + n + ".bias"] = (cin, cin), (cin,)
    norm_("conv_norm_out", cin)
    conv_("conv_out", 2 * latent_channels, cin, 3)
    return s


def synthetic_decoder_state_dict(seed: int = 0, device="cpu", shapes=None, **kw):
    """Seeded weights of that layout: convolutions / linears U(-1, 1) / sqrt(fan_in) (PyTorch's default scale), norm weights 1 + 0.2 N(0, 1).
    `shapes=encoder_param_shapes()` draws an encoder."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in (decoder_param_shapes(**kw) if shapes is None else shapes).items():
        if name.endswith(".bias"):
            sd[name] = 0.1 * torch.randn(shape, generator=g, device=device)
        elif len(shape) == 1:
            sd[name] = 1.0 + 0.2 * torch.randn(shape, generator=g, device=device)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            sd[name] = (torch.rand(shape, generator=g, device=device) * 2 - 1) / math.sqrt(fan_in)
    return sd


def conv_s2(x: PaddedImage, cw: "ConvWeights", out: PaddedImage, rows_table: torch.Tensor):
    """The encoder's downsampler: 3 x 3, stride 2, F.pad (0, 1, 0, 1) (rgn_conv_s2_bf16); `rows_table` = downsample_rows(x, out)."""
    if cw.cin != x.C or out.C != cw.ldy or cw.group != 1 or cw.taps != 9 or out.H * 2 != x.H or out.W * 2 != x.W:
        raise _lib.RegionEHipError(f"conv_s2: weights for {cw.cin} -> {cw.cout} on {x.H} x {x.W} x {x.C} -> {out.H} x {out.W} x {out.C}")
    rc = _lib.lib().rgn_conv_s2_bf16(x.ptr(), _p(cw.w), _p(cw.b), out.ptr(), out.C, x.Hp, x.Wp, x.C, cw.cout, _p(rows_table), _stream())
    _lib.check(rc, "rgn_conv_s2_bf16")
    return out


class UpConvWeights:
    """`upsamplers.0.conv` for rgn_conv_up2_bf16: the four phase matrices [4][Cout, 2, 2, Cin] with the taps a low-resolution pixel is seen
    through summed (in fp32, then rounded to bf16 once).  `w4`: [Cout, 3, 3, Cin] fp32."""

    def __init__(self, w4: torch.Tensor, bias: torch.Tensor):
        co, kh, kw, ci = w4.shape
        assert kh == 3 and kw == 3
        self.cout, self.cin = co, ci
        fold = ([[0], [1, 2]], [[0, 1], [2]])                    # phase 0: window rows (y - 1, y) <- taps (0), (1, 2); phase 1: (y, y + 1) <- (0, 1), (2)
        mats = []
        for a in range(2):
            for b in range(2):
                t = torch.zeros((co, 2, 2, ci), dtype=torch.float32, device=w4.device)
                for r in range(2):
                    for c in range(2):
                        for ky in fold[a][r]:
                            for kx in fold[b][c]:
                                t[:, r, c, :] += w4[:, ky, kx, :]
                mats.append(t.reshape(co, -1))
        self.w = torch.stack(mats).to(torch.bfloat16).contiguous()
        self.b = bias.to(w4.device, torch.bfloat16).contiguous()


def upsample_rows(H: int, W: int, device) -> torch.Tensor:
    """rgn_conv_up2_bf16's row tables [4][(H + 2) * (W + 2)] for an H x W low-resolution image: low-resolution padded pixel (py, px), phase
    (a, b) -> padded row of pixel (2 (py - 1) + a, 2 (px - 1) + b) of the (2H + 2) x (2W + 2) image; border pixels -> the first guard row
    behind that image.  Built once per size (setup)."""
    Hp, Wp, Wq = H + 2, W + 2, 2 * W + 2
    m = torch.arange(Hp * Wp, dtype=torch.int64, device=device)
    py, px = m // Wp, m % Wp
    ok = (py >= 1) & (py <= H) & (px >= 1) & (px <= W)
    scratch = (2 * H + 2) * Wq
    out = []
    for a in range(2):
        for b in range(2):
            row = (2 * (py - 1) + a + 1) * Wq + 2 * (px - 1) + b + 1
            out.append(torch.where(ok, row, torch.full_like(row, scratch)))
    return torch.stack(out).contiguous()


def conv_up2(x: PaddedImage, uw: UpConvWeights, out: PaddedImage, rows_table: torch.Tensor, gn: bool = False):
    """out = conv3x3(upsample2x(x)) + bias without the upsampled image (rgn_conv_up2_bf16); `rows_table` = upsample_rows(x.H, x.W)."""
    if uw.cin != x.C or out.C != uw.cout or out.H != 2 * x.H or out.W != 2 * x.W:
        raise _lib.RegionEHipError(f"conv_up2: weights for {uw.cin} -> {uw.cout} on {x.H} x {x.W} x {x.C} -> {out.H} x {out.W} x {out.C}")
    import ctypes
    dev = x.t.device
    nblk = ctypes.c_int(0)
    ws = _gn_workspace(dev) if gn else None
    _gn_owner[_gn_key(dev)] = (None, 0)
    rc = _lib.lib().rgn_conv_up2_bf16(x.ptr(), _p(uw.w), _p(uw.b), out.ptr(), out.C, x.Hp, x.Wp, x.C, uw.cout, _p(rows_table), _p(ws),
                                      ctypes.byref(nblk) if gn else None, _stream())
    _lib.check(rc, "rgn_conv_up2_bf16")
    if gn:
        _gn_owner[_gn_key(dev)] = (out, nblk.value)
    return out


def downsample_rows(H: int, W: int, device) -> torch.Tensor:
    """rgn_conv_s2_bf16's row table for an H x W input (padded pitch W + 2): GEMM row m = yo * (W + 2) + xo -> padded row of output pixel
    (yo, xo) in the (H / 2 + 2) x (W / 2 + 2) image, or - for the unused columns xo >= W / 2 of the wide grid - the first guard row behind
    the output image (written, never read).  Built once per size (setup, not on the decode / encode path)."""
    Wp, Ho, Wo = W + 2, H // 2, W // 2
    m = torch.arange(Ho * Wp, dtype=torch.int64, device=device)
    yo, xo = m // Wp, m % Wp
    row = (yo + 1) * (Wo + 2) + xo + 1
    return torch.where(xo < Wo, row, torch.full_like(row, (Ho + 2) * (Wo + 2))).contiguous()


class _KLBase:
    """Parameter adoption shared by the decoder and the encoder."""

    def _init_params(self, state_dict, device, prefix, pixel_groups):
        self.device = torch.device(device)
        self._sd = {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in state_dict.items()}
        self.p: Dict[str, torch.Tensor] = {}
        self.c: Dict[str, ConvWeights] = {}
        self.pixel_groups = pixel_groups
        self.fuse_gn = True              # GroupNorm statistics in the producing convolution's epilogue (False: the standalone statistics pass)
        self.pool = _Pool(self.device)
        self._attn_buf = {}

    def _done(self, what, ignore=()):
        unused = [k for k in self._sd if not k.startswith(tuple(ignore))]
        if unused:
            raise _lib.RegionEHipError(f"{what}: state dict entries it does not know: {unused[:6]}")
        del self._sd


    def _attention(self, a, top):
        self._vec(a + "group_norm.weight"); self._vec(a + "group_norm.bias")
        for n in ("to_q", "to_k", "to_out.0"):
            self._conv(a + n)
        self.p[a + "to_v.weight"] = self._take(a + "to_v.weight").reshape(top, top).to(self.device, torch.bfloat16).contiguous()
        self._vec(a + "to_v.bias")

    # -- parameter adoption -------------------------------------------------------------------------------------------------------
    def _take(self, name):
        if name not in self._sd:
            raise _lib.RegionEHipError(f"{type(self).__name__}: parameter {name} missing from the state dict")
        return self._sd.pop(name)

    def _vec(self, name):
        self.p[name] = self._take(name).to(self.device, torch.bfloat16).contiguous()

    def _conv(self, prefix, pad_in: Optional[int] = None, ldy: Optional[int] = None):
        w = self._take(prefix + ".weight").to(self.device, torch.float32)          # [Cout, Cin, kh, kw] (or [Cout, Cin]: a Linear)
        if w.dim() == 2:
            w = w[:, :, None, None]
        co, ci, kh, kw = w.shape
        w = w.permute(0, 2, 3, 1)                                                  # [Cout, kh, kw, Cin]
        if pad_in is not None and ci < pad_in:
            w = torch.nn.functional.pad(w, (0, pad_in - ci))
        # narrow outputs fill the 256-wide MFMA tile through pixel groups: 128 channels -> 2 pixels per GEMM row, the RGB head -> 8
        group = 2 if co == 128 else (8 if co <= 8 else 1)
        self.c[prefix] = ConvWeights(w, self._take(prefix + ".bias"), group=group if self.pixel_groups else 1, ldy=ldy)

    def _resnet(self, prefix):
        for n in ("norm1", "norm2"):
            self._vec(f"{prefix}.{n}.weight"); self._vec(f"{prefix}.{n}.bias")
        self._conv(prefix + ".conv1"); self._conv(prefix + ".conv2")
        if prefix + ".conv_shortcut.weight" in self._sd:
            self._conv(prefix + ".conv_shortcut")

    # -- blocks -------------------------------------------------------------------------------------------------------------------
    def _run_resnet(self, x: PaddedImage, prefix: str, cout: int) -> PaddedImage:
        P, Cv, pool = self.p, self.c, self.pool
        n = groupnorm_silu(x, P[prefix + ".norm1.weight"], P[prefix + ".norm1.bias"], pool.get(x.H, x.W, x.C), eps=self.eps)
        h = conv(n, Cv[prefix + ".conv1"], pool.get(x.H, x.W, cout), gn=self.fuse_gn)
        pool.put(n)
        n2 = groupnorm_silu(h, P[prefix + ".norm2.weight"], P[prefix + ".norm2.bias"], pool.get(x.H, x.W, cout), eps=self.eps)
        skip = x
        if prefix + ".conv_shortcut" in Cv:
            skip = conv(x, Cv[prefix + ".conv_shortcut"], pool.get(x.H, x.W, cout))
        conv(n2, Cv[prefix + ".conv2"], h, resid=skip, gn=self.fuse_gn)      # h is not an input of this launch; every ResNet output feeds a GroupNorm
        pool.put(n2)
        if skip is not x:
            pool.put(skip)
        pool.put(x)
        return h

    def _run_attention(self, x: PaddedImage) -> PaddedImage:
        """One head of width C over every pixel (diffusers `Attention`, residual_connection=True): three GEMMs + a row softmax.  The border
        pixels of the padded image are masked out as keys by the softmax pass and zeroed as outputs by the last projection's epilogue."""
        P, Cv, pool, a = self.p, self.c, self.pool, "mid_block.attentions.0."
        C, rows = x.C, x.rows
        ldp = ops.padded(rows, 64)
        key = (rows, C)
        if key not in self._attn_buf:
            self._attn_buf[key] = (torch.zeros((rows, ldp), dtype=torch.bfloat16, device=self.device),       # S / P
                                   torch.zeros((C, ldp), dtype=torch.bfloat16, device=self.device))          # V^T (padding columns stay 0)
        S, vt = self._attn_buf[key]
        n = groupnorm_silu(x, P[a + "group_norm.weight"], P[a + "group_norm.bias"], pool.get(x.H, x.W, C), silu=False, eps=self.eps)
        q = conv(n, Cv[a + "to_q"], pool.get(x.H, x.W, C))
        k = conv(n, Cv[a + "to_k"], pool.get(x.H, x.W, C))
        ops.gemm(P[a + "to_v.weight"], n.t, None, vt[:, :rows])                  # V^T = W_v X^T; b_v is added behind P V (rows of P sum to 1)
        ops.gemm(q.t, k.t, None, S[:, :rows])                                     # S = Q K^T
        _lib.check(_lib.lib().rgn_softmax_rows(_p(S), ldp, x.Hp, x.Wp, 1.0 / math.sqrt(C), _stream()), "rgn_softmax_rows")
        o = q                                                                     # q is dead: O = P V + b_v
        ops.gemm(S, vt, P[a + "to_v.bias"], o.t)
        out = conv(o, Cv[a + "to_out.0"], k, resid=x, gn=self.fuse_gn)       # k is dead
        pool.put(n); pool.put(o); pool.put(x)
        return out


class HipVaeDecoder(_KLBase):
    """AutoencoderKL decoder ([EXT] diffusers layout) on libregione_hip.so.  `decode(z)`: z [1, Cz, h, w] -> image [1, 3, 8h, 8w] bf16."""

    def __init__(self, state_dict, device, block_out_channels=(128, 256, 512, 512), latent_channels: int = 16, layers_per_block: int = 2,
                 norm_eps: float = 1e-6, pixel_groups: bool = True):
        self.ch = tuple(block_out_channels)
        self.zc, self.nres, self.eps = latent_channels, layers_per_block + 1, norm_eps
        self._init_params(state_dict, device, "decoder.", pixel_groups)
        self.u: Dict[str, UpConvWeights] = {}
        self.fuse_upsample = True        # nearest 2 x upsample folded into its convolution (rgn_conv_up2_bf16); False: upsample pass + 3 x 3
        self._up_rows = {}
        top = self.ch[-1]
        for c in self.ch:
            if c not in (128, 256, 512):
                raise _lib.RegionEHipError(f"HipVaeDecoder: block width {c} (the kernels cover 128 / 256 / 512 channels)")
        self._conv("conv_in", pad_in=64)
        for r in (0, 1):
            self._resnet(f"mid_block.resnets.{r}")
        self._attention("mid_block.attentions.0.", top)
        cin = top
        self.levels = []
        for i, co in enumerate(reversed(self.ch)):
            for j in range(self.nres):
                self._resnet(f"up_blocks.{i}.resnets.{j}")
            up = i < len(self.ch) - 1
            if up:
                n = f"up_blocks.{i}.upsamplers.0.conv"
                w = self._sd[n + ".weight"].to(self.device, torch.float32).permute(0, 2, 3, 1)
                self.u[n] = UpConvWeights(w, self._sd[n + ".bias"])          # the fused upsample + convolution (four 2 x 2 phases)
                self._conv(n)                                               # and the plain form (fuse_upsample = False)
            self.levels.append((cin, co, up))
            cin = co
        self._vec("conv_norm_out.weight"); self._vec("conv_norm_out.bias")
        self._conv("conv_out", ldy=8)
        if any(k.startswith("post_quant_conv") for k in self._sd):
            raise _lib.RegionEHipError("HipVaeDecoder: post_quant_conv is not part of the FLUX.1 / Step1X-Edit VAE (use_post_quant_conv = False)")
        self._done("HipVaeDecoder", ignore=("encoder.", "quant_conv"))

    # -- decode -------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        if not z.is_cuda or z.dim() != 4 or z.shape[0] != 1 or z.shape[1] != self.zc:
            raise _lib.RegionEHipError(f"HipVaeDecoder.decode: one latent image [1, {self.zc}, h, w] on the GPU, got {tuple(z.shape)} on {z.device}")
        z = z.to(torch.bfloat16).contiguous()
        h, w = z.shape[2], z.shape[3]
        P, Cv, pool, L = self.p, self.c, self.pool, _lib.lib()
        zin = pool.get(h, w, 64)
        _lib.check(L.rgn_nchw_to_padded(_p(z), zin.ptr(), self.zc, h, w, 64, _stream()), "rgn_nchw_to_padded")
        top = self.ch[-1]
        x = conv(zin, Cv["conv_in"], pool.get(h, w, top), gn=self.fuse_gn)
        pool.put(zin)
        x = self._run_resnet(x, "mid_block.resnets.0", top)
        x = self._run_attention(x)
        x = self._run_resnet(x, "mid_block.resnets.1", top)
        for i, (cin, co, up) in enumerate(self.levels):
            for j in range(self.nres):
                x = self._run_resnet(x, f"up_blocks.{i}.resnets.{j}", co)
            if up:
                name = f"up_blocks.{i}.upsamplers.0.conv"
                if self.fuse_upsample:
                    key = (x.H, x.W)
                    if key not in self._up_rows:
                        self._up_rows[key] = upsample_rows(x.H, x.W, self.device)
                    y = conv_up2(x, self.u[name], pool.get(2 * x.H, 2 * x.W, x.C), self._up_rows[key], gn=self.fuse_gn)
                    pool.put(x)
                    x = y
                else:
                    u = upsample2x(x, pool.get(2 * x.H, 2 * x.W, x.C))
                    pool.put(x)
                    x = conv(u, Cv[name], pool.get(u.H, u.W, u.C), gn=self.fuse_gn)
                    pool.put(u)
        n = groupnorm_silu(x, P["conv_norm_out.weight"], P["conv_norm_out.bias"], pool.get(x.H, x.W, x.C), eps=self.eps)
        pool.put(x)
        y = conv(n, Cv["conv_out"], pool.get(n.H, n.W, 8))
        pool.put(n)
        img = torch.empty((1, 3, y.H, y.W), dtype=torch.bfloat16, device=self.device)
        _lib.check(L.rgn_padded_to_nchw(y.ptr(), 8, _p(img), 3, y.H, y.W, _stream()), "rgn_padded_to_nchw")
        pool.put(y)
        return img

    def flops(self, h: int, w: int) -> float:
        """Algorithmic FLOPs of one decode of an h x w latent (2 * valid pixels * Cout * taps * Cin per convolution + the attention GEMMs)."""
        top = self.ch[-1]
        px = h * w
        f = 2.0 * px * top * 9 * self.zc
        res = lambda p, ci, co: 2.0 * p * (9 * ci * co + 9 * co * co + (ci * co if ci != co else 0))
        f += 2 * res(px, top, top) + 2.0 * px * top * top * 4 + 4.0 * px * px * top
        for cin, co, up in self.levels:
            f += res(px, cin, co) + (self.nres - 1) * res(px, co, co)
            if up:
                px *= 4
                f += 2.0 * px * 9 * co * co
        return f + 2.0 * px * 9 * self.ch[0] * 3


class LatentDist:
    """`vae.encode(x).latent_dist` of diffusers' AutoencoderKL (DiagonalGaussianDistribution) over the HIP encoder's moments: what the host's
    `retrieve_latents(encoder_output, generator, sample_mode)` reads - `.mode()` (FLUX.1-Kontext: sample_mode = "argmax") and `.sample()`."""

    def __init__(self, moments: torch.Tensor):
        self.mean, logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(logvar.float(), -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar).to(moments.dtype)

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device if generator is None else generator.device,
                            dtype=torch.float32).to(self.mean.device, self.mean.dtype)
        return self.mean + self.std * noise


class EncoderOutput:
    def __init__(self, moments):
        self.latent_dist = LatentDist(moments)


class HipVaeEncoder(_KLBase):
    """AutoencoderKL encoder ([EXT] diffusers layout: conv_in, four down levels of two ResNet blocks (128, 256, 512, 512; a stride-2 3 x 3
    convolution with F.pad (0, 1, 0, 1) after the first three), the mid block, GroupNorm + SiLU + conv_out to 2 x latent channels; no
    quant_conv in the FLUX.1 / Step1X-Edit VAE) on libregione_hip.so - the VAE encode of the condition image the host's `prepare_latents`
    runs before the loop (reference call site FluxKontext/inplace.py:210-226).  `encode(x)`: image [1, 3, H, W] (H, W multiples of 8) ->
    moments [1, 2 Cz, H / 8, W / 8] bf16; `encode_dist(x)` wraps them as `vae.encode(x)` does (`.latent_dist.mode() / .sample()`)."""

    def __init__(self, state_dict, device, block_out_channels=(128, 256, 512, 512), latent_channels: int = 16, layers_per_block: int = 2,
                 norm_eps: float = 1e-6, pixel_groups: bool = True):
        self.ch = tuple(block_out_channels)
        self.zc, self.nres, self.eps = latent_channels, layers_per_block, norm_eps
        self._init_params(state_dict, device, "encoder.", pixel_groups)
        for c in self.ch:
            if c not in (128, 256, 512):
                raise _lib.RegionEHipError(f"HipVaeEncoder: block width {c} (the kernels cover 128 / 256 / 512 channels)")
        self._conv("conv_in", pad_in=64)
        cin = self.ch[0]
        self.levels = []
        for i, co in enumerate(self.ch):
            for j in range(self.nres):
                self._resnet(f"down_blocks.{i}.resnets.{j}")
            down = i < len(self.ch) - 1
            if down:
                pg, self.pixel_groups = self.pixel_groups, False         # the stride-2 launch stores through a row table: no pixel groups
                self._conv(f"down_blocks.{i}.downsamplers.0.conv")
                self.pixel_groups = pg
            self.levels.append((cin, co, down))
            cin = co
        top = self.ch[-1]
        for r in (0, 1):
            self._resnet(f"mid_block.resnets.{r}")
        self._attention("mid_block.attentions.0.", top)
        self._vec("conv_norm_out.weight"); self._vec("conv_norm_out.bias")
        self._conv("conv_out")
        if any(k.startswith("quant_conv") for k in self._sd):
            raise _lib.RegionEHipError("HipVaeEncoder: quant_conv is not part of the FLUX.1 / Step1X-Edit VAE (use_quant_conv = False)")
        self._done("HipVaeEncoder", ignore=("decoder.", "post_quant_conv"))
        self._rows = {}

    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda or x.dim() != 4 or x.shape[0] != 1 or x.shape[1] != 3 or x.shape[2] % 8 or x.shape[3] % 8:
            raise _lib.RegionEHipError(f"HipVaeEncoder.encode: one image [1, 3, H, W] on the GPU, H and W multiples of 8; got {tuple(x.shape)} on {x.device}")
        x = x.to(torch.bfloat16).contiguous()
        H, W = x.shape[2], x.shape[3]
        P, Cv, pool, L = self.p, self.c, self.pool, _lib.lib()
        xin = pool.get(H, W, 64)
        _lib.check(L.rgn_nchw_to_padded(_p(x), xin.ptr(), 3, H, W, 64, _stream()), "rgn_nchw_to_padded")
        h = conv(xin, Cv["conv_in"], pool.get(H, W, self.ch[0]), gn=self.fuse_gn)
        pool.put(xin)
        for i, (cin, co, down) in enumerate(self.levels):
            for j in range(self.nres):
                h = self._run_resnet(h, f"down_blocks.{i}.resnets.{j}", co)
            if down:
                key = (h.H, h.W)
                if key not in self._rows:
                    self._rows[key] = downsample_rows(h.H, h.W, self.device)
                d = conv_s2(h, Cv[f"down_blocks.{i}.downsamplers.0.conv"], pool.get(h.H // 2, h.W // 2, h.C), self._rows[key])
                pool.put(h)
                h = d
        top = self.ch[-1]
        h = self._run_resnet(h, "mid_block.resnets.0", top)
        h = self._run_attention(h)
        h = self._run_resnet(h, "mid_block.resnets.1", top)
        n = groupnorm_silu(h, P["conv_norm_out.weight"], P["conv_norm_out.bias"], pool.get(h.H, h.W, h.C), eps=self.eps)
        pool.put(h)
        y = conv(n, Cv["conv_out"], pool.get(n.H, n.W, 2 * self.zc))
        pool.put(n)
        out = torch.empty((1, 2 * self.zc, y.H, y.W), dtype=torch.bfloat16, device=self.device)
        _lib.check(L.rgn_padded_to_nchw(y.ptr(), y.C, _p(out), 2 * self.zc, y.H, y.W, _stream()), "rgn_padded_to_nchw")
        pool.put(y)
        return out

    def encode_dist(self, x: torch.Tensor) -> EncoderOutput:
        return EncoderOutput(self.encode(x))

    def flops(self, H: int, W: int) -> float:
        px = H * W
        f = 2.0 * px * self.ch[0] * 9 * 3
        res = lambda p, ci, co: 2.0 * p * (9 * ci * co + 9 * co * co + (ci * co if ci != co else 0))
        for cin, co, down in self.levels:
            f += res(px, cin, co) + (self.nres - 1) * res(px, co, co)
            if down:
                px //= 4
                f += 2.0 * px * 9 * co * co
        top = self.ch[-1]
        return f + 2 * res(px, top, top) + 2.0 * px * top * top * 4 + 4.0 * px * px * top + 2.0 * px * 9 * top * 2 * self.zc
