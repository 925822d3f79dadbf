"""-m gpu: the HIP VAE decode (regione_amd/vae.py, csrc/vae.hip, rgn_conv_bf16) against an fp32 PyTorch
module of the public AutoencoderKL decoder with seeded weights (tests/host_vae.py).  [EXT] / unpinned: nothing of the VAE lives in
/root/reference (the reference calls `self.vae.decode`, FluxKontext/inplace.py:396-402).  Tolerance: PSNR >= 40 dB on the decoded
image tensor (BASELINE.json north_star's figure for latents, applied to the image), peak = the reference tensor's value range."""
import math

import pytest
import torch

from regione_amd import _lib, ops, vae as V
from tests import host_vae

pytestmark = pytest.mark.gpu


def _psnr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    mse = float(((a - b) ** 2).mean())
    peak = float(b.max() - b.min())
    return 10 * math.log10(peak * peak / max(mse, 1e-30))


def _fp32_on_cpu(fn):
    """The fp32 PyTorch reference on the host cores (test infrastructure): at 1024 x 1024 the CPU is several times faster than the first call
    of the GPU library's fp32 convolutions; 64 threads (more is slower on the 128-core box)."""
    n = torch.get_num_threads()
    torch.set_num_threads(min(n, 64))
    try:
        with torch.no_grad():
            return fn()
    finally:
        torch.set_num_threads(n)


def _padded(x_nchw, cpad=None):
    """[1, C, H, W] fp32 -> PaddedImage (bf16)"""
    _, C, H, W = x_nchw.shape
    img = V.PaddedImage(H, W, cpad or C, "cuda")
    z = x_nchw.to("cuda", torch.bfloat16).contiguous()
    _lib.check(_lib.lib().rgn_nchw_to_padded(ops._p(z), img.ptr(), C, H, W, img.C, ops._stream()), "nchw_to_padded")
    return img


def _unpadded(img, C=None):
    C = C or img.C
    out = torch.empty((1, C, img.H, img.W), dtype=torch.bfloat16, device="cuda")
    _lib.check(_lib.lib().rgn_padded_to_nchw(img.ptr(), img.C, ops._p(out), C, img.H, img.W, ops._stream()), "padded_to_nchw")
    return out


def _border_is_zero(img):
    t = img.t.view(img.Hp, img.Wp, img.C)
    return bool((t[0] == 0).all() and (t[-1] == 0).all() and (t[:, 0] == 0).all() and (t[:, -1] == 0).all())


@pytest.mark.parametrize("H,W,cin,cout,group", [(16, 16, 128, 128, 1), (16, 16, 128, 128, 2), (24, 40, 256, 128, 2), (33, 17, 512, 256, 1),
                                                 (128, 128, 512, 512, 1), (72, 56, 64, 512, 1), (40, 72, 128, 3, 8), (25, 31, 128, 3, 8),
                                                 (127, 129, 128, 128, 2), (258, 258, 64, 256, 1), (514, 258, 128, 128, 2)])
@pytest.mark.parametrize("resid", [False, True])
@pytest.mark.parametrize("geometry", [128, 256])
def test_conv3x3_implicit_gemm_vs_torch(H, W, cin, cout, group, resid, geometry):
    """3 x 3 / stride 1 / zero padding 1 as an implicit GEMM over the zero-bordered image; both geometries (small images run on the
    128 x 128 tiles, large on the hand-scheduled 256 x 256 loop; 258 x 258 / 514 x 258: whole rounds + a remainder as quarter tiles);
    ragged sizes; the ResNet skip in the epilogue; zero border kept;
    pixel groups (block-Toeplitz weights: 2 pixels per GEMM row for 128 output channels, 8 for the RGB head with row stride 8)."""
    g = torch.Generator().manual_seed(H * 1000 + cin + cout)
    ldy = 8 if cout == 3 else cout
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / math.sqrt(9 * cin))
    b = torch.randn(cout, generator=g) * 0.1
    r = torch.randn(1, cout, H, W, generator=g) if resid else None
    xb, wb, bb = x.bfloat16().float(), w.bfloat16().float(), b.bfloat16().float()
    ref = torch.nn.functional.conv2d(xb.double(), wb.double(), bb.double(), padding=1).float()
    if resid:
        ref = ref + r.bfloat16().float()
    xi, out = _padded(x), V.PaddedImage(H, W, ldy, "cuda")
    cw = V.ConvWeights(w.permute(0, 2, 3, 1).cuda(), b.cuda(), group=group, ldy=ldy)
    with _lib.plan_override(gemm_geometry=geometry):           # both tile geometries on every shape
        V.conv(xi, cw, out, resid=_padded(r, ldy) if resid else None)
    torch.cuda.synchronize()
    assert _border_is_zero(out)
    got = _unpadded(out, cout)
    assert _psnr(got, ref) >= 45.0, _psnr(got, ref)
    assert float((got.float().cpu() - ref).abs().max()) <= 0.05 * float(ref.abs().max())
    if ldy > cout:                                             # padding channels of the output rows stay zero
        assert bool((out.t[:, cout:] == 0).all())


@pytest.mark.parametrize("H,W,cin,cout,group", [(16, 16, 512, 256, 1), (128, 128, 256, 128, 1), (128, 128, 256, 128, 2), (40, 24, 512, 512, 1),
                                                 (33, 47, 256, 128, 2)])
def test_conv1x1_vs_torch(H, W, cin, cout, group):
    g = torch.Generator().manual_seed(7 + cin)
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(cout, cin, generator=g) / math.sqrt(cin)
    b = torch.randn(cout, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(x.bfloat16().double(), w.bfloat16().double()[:, :, None, None], b.bfloat16().double()).float()
    xi, out = _padded(x), V.PaddedImage(H, W, cout, "cuda")
    V.conv(xi, V.ConvWeights(w[:, None, None, :].cuda(), b.cuda(), group=group), out)
    torch.cuda.synchronize()
    assert _border_is_zero(out)
    assert _psnr(_unpadded(out), ref) >= 45.0


@pytest.mark.parametrize("H,W,C", [(16, 16, 128), (31, 50, 256), (128, 128, 512), (256, 256, 128)])
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm_silu_vs_torch(H, W, C, silu):
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(1, C, H, W, generator=g) * 2.0 + 3.0 * torch.randn(1, C, 1, 1, generator=g)      # per-channel offsets: mean >> 0
    gamma, beta = 1.0 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    xb = x.bfloat16().double()
    ref = torch.nn.functional.group_norm(xb, 32, gamma.bfloat16().double(), beta.bfloat16().double(), eps=1e-6)
    ref = (torch.nn.functional.silu(ref) if silu else ref).float()
    xi, out = _padded(x), V.PaddedImage(H, W, C, "cuda")
    V.groupnorm_silu(xi, gamma.to("cuda", torch.bfloat16), beta.to("cuda", torch.bfloat16), out, silu=silu)
    torch.cuda.synchronize()
    assert _border_is_zero(out)
    got = _unpadded(out)
    assert _psnr(got, ref) >= 48.0, _psnr(got, ref)
    # bit-reproducible (no atomics in the statistics)
    out2 = V.PaddedImage(H, W, C, "cuda")
    V.groupnorm_silu(xi, gamma.to("cuda", torch.bfloat16), beta.to("cuda", torch.bfloat16), out2, silu=silu)
    assert torch.equal(out.t, out2.t)


@pytest.mark.parametrize("H,W,cin,cout,group", [(24, 40, 128, 128, 2), (64, 64, 256, 256, 1), (33, 47, 512, 512, 1), (130, 126, 256, 128, 2),
                                                 (128, 128, 64, 512, 1), (258, 258, 64, 256, 1), (514, 258, 128, 128, 2)])
@pytest.mark.parametrize("geometry", [128, 256])
def test_groupnorm_statistics_from_the_convolution_epilogue(H, W, cin, cout, group, geometry):
    """rgn_conv_bf16(gn_partial): the statistics of the image the convolution stores (after bias, ResNet skip and border zeroing), so that the
    GroupNorm that follows skips its own pass over it.  Same normalised image as the standalone pass up to the fp32 fold order of the sums."""
    g = torch.Generator().manual_seed(H + W + cin)
    x = torch.randn(1, cin, H, W, generator=g)
    r = torch.randn(1, cout, H, W, generator=g) * 2.0 + 3.0 * torch.randn(1, cout, 1, 1, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    b = torch.randn(cout, generator=g)
    gamma, beta = (1.0 + 0.2 * torch.randn(cout, generator=g)).to("cuda", torch.bfloat16), (0.1 * torch.randn(cout, generator=g)).to("cuda", torch.bfloat16)
    xi, ri = _padded(x), _padded(r)
    cw = V.ConvWeights(w.permute(0, 2, 3, 1).cuda(), b.cuda(), group=group)
    outs = []
    for fused in (True, False):
        y, n = V.PaddedImage(H, W, cout, "cuda"), V.PaddedImage(H, W, cout, "cuda")
        with _lib.plan_override(gemm_geometry=geometry):
            V.conv(xi, cw, y, resid=ri, gn=fused)
        owner, nblk = V._gn_owner[V._gn_key(y.t.device)]
        assert (owner is y and nblk > 0) if fused else owner is None
        V.groupnorm_silu(y, gamma, beta, n)
        assert V._gn_owner[V._gn_key(y.t.device)][0] is None           # consumed
        torch.cuda.synchronize()
        outs.append((_unpadded(y), _unpadded(n)))
    assert torch.equal(outs[0][0], outs[1][0])                           # the convolution's output does not depend on the statistics option
    assert _psnr(outs[0][1], outs[1][1].float()) >= 70.0
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(outs[0][0].double().cpu(), 32, gamma.double().cpu(), beta.double().cpu(), eps=1e-6)).float()
    assert _psnr(outs[0][1], ref) >= 48.0


@pytest.mark.parametrize("H,W,C", [(16, 16, 256), (19, 23, 512), (128, 128, 512), (256, 256, 256), (40, 24, 128)])
@pytest.mark.parametrize("geometry", [128, 256])
def test_upsample_folded_into_its_convolution_vs_torch(H, W, C, geometry):
    """rgn_conv_up2_bf16: conv3x3(nearest 2 x upsample(x)) as four 2 x 2 phase convolutions of the low-resolution image (one launch of four
    problems, tap-summed weights, row tables) against interpolate + conv2d; zero border; GroupNorm statistics of the whole output."""
    g = torch.Generator().manual_seed(H * 7 + C)
    x = torch.randn(1, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)
    b = torch.randn(C, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x.bfloat16().double(), scale_factor=2.0, mode="nearest"), w.bfloat16().double(),
                                     b.bfloat16().double(), padding=1).float()
    xi, out = _padded(x), V.PaddedImage(2 * H, 2 * W, C, "cuda")
    uw = V.UpConvWeights(w.permute(0, 2, 3, 1).cuda(), b.cuda())
    with _lib.plan_override(gemm_geometry=geometry):
        V.conv_up2(xi, uw, out, V.upsample_rows(H, W, "cuda"), gn=True)
    torch.cuda.synchronize()
    assert _border_is_zero(out)
    assert _psnr(_unpadded(out), ref) >= 44.0, _psnr(_unpadded(out), ref)        # tap sums are rounded to bf16 once: a hair below the plain form
    gamma, beta = torch.ones(C, device="cuda", dtype=torch.bfloat16), torch.zeros(C, device="cuda", dtype=torch.bfloat16)
    n1, n2 = V.PaddedImage(2 * H, 2 * W, C, "cuda"), V.PaddedImage(2 * H, 2 * W, C, "cuda")
    V.groupnorm_silu(out, gamma, beta, n1)                                       # statistics from the four problems' tiles
    V.groupnorm_silu(out, gamma, beta, n2)                                       # consumed: the standalone pass
    torch.cuda.synchronize()
    assert _psnr(_unpadded(n1), _unpadded(n2).float()) >= 70.0


def test_upsample2x_exact():
    x = torch.randn(1, 256, 19, 23)
    xi = _padded(x)
    out = V.upsample2x(xi, V.PaddedImage(38, 46, 256, "cuda"))
    torch.cuda.synchronize()
    assert _border_is_zero(out)
    ref = torch.nn.functional.interpolate(x.bfloat16().float(), scale_factor=2.0, mode="nearest")
    assert torch.equal(_unpadded(out).float().cpu(), ref)


@pytest.mark.parametrize("H,W,cin,cout", [(16, 16, 128, 128), (64, 48, 256, 256), (256, 256, 128, 128), (34, 18, 512, 512)])
@pytest.mark.parametrize("geometry", [128, 256])
def test_conv_stride2_vs_torch(H, W, cin, cout, geometry):
    """The encoder's downsampler: F.pad(x, (0, 1, 0, 1)) + Conv2d(3 x 3, stride 2, padding 0) through the wide-grid GEMM + row table."""
    g = torch.Generator().manual_seed(H + cin)
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    b = torch.randn(cout, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(x.bfloat16().double(), (0, 1, 0, 1)), w.bfloat16().double(), b.bfloat16().double(),
                                     stride=2).float()
    xi, out = _padded(x), V.PaddedImage(H // 2, W // 2, cout, "cuda")
    cw = V.ConvWeights(w.permute(0, 2, 3, 1).cuda(), b.cuda())
    with _lib.plan_override(gemm_geometry=geometry):
        V.conv_s2(xi, cw, out, V.downsample_rows(H, W, "cuda"))
    torch.cuda.synchronize()
    assert _border_is_zero(out)
    assert _psnr(_unpadded(out), ref) >= 45.0


@pytest.mark.parametrize("H,W", [(64, 64), (128, 192), (1024, 1024)])
def test_encoder_vs_fp32_module(H, W):
    """VAE encode of the condition image (the host's prepare_latents; FluxKontext/inplace.py:210-226): moments (mean | logvar) of the HIP
    encoder against the fp32 module, >= 40 dB; at 1024 x 1024 also the time."""
    import time
    m = host_vae.seeded(7)
    enc = V.HipVaeEncoder(m.state_dict(), "cuda")
    x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(H)).clamp(-1, 1)
    ref = _fp32_on_cpu(lambda: m.encoder(x.bfloat16().float()))
    got = enc.encode(x.cuda())
    torch.cuda.synchronize()
    assert got.shape == (1, 32, H // 8, W // 8) and torch.isfinite(got.float()).all()
    p = _psnr(got, ref)
    d = enc.encode_dist(x.cuda()).latent_dist
    assert torch.equal(d.mode(), got[:, :16]) and d.sample(torch.Generator().manual_seed(0)).shape == (1, 16, H // 8, W // 8)
    msg = f"[vae] encode {H} x {W}: HIP vs fp32 module {p:.1f} dB"
    if H == 1024:
        xc = x.cuda()
        for _ in range(2):
            enc.encode(xc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            enc.encode(xc)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        msg += f"; {ms:.1f} ms ({enc.flops(H, W) / ms / 1e9:.0f} TFLOP/s; eager bf16 module: 37-38 ms, tools/f4_host_side.py)"
        assert ms <= 15.0, ms
    print(msg)
    assert p >= 40.0, p


def _decoder_pair(seed, pixel_groups=True, **kw):
    m = host_vae.seeded(seed, **kw)
    dec = V.HipVaeDecoder(m.state_dict(), "cuda", pixel_groups=pixel_groups, **kw)
    return m, dec


@pytest.mark.parametrize("h,w,groups,fuse", [(16, 16, True, True), (24, 40, True, True), (24, 40, False, True), (24, 40, True, False),
                                             (24, 40, True, "plain_upsample")])
def test_decoder_small_latents_vs_fp32_module(h, w, groups, fuse):
    """The whole decoder (every block type incl. the mid-block attention and the three upsamples) on small latents; with / without pixel groups
    and with / without the GroupNorm statistics in the convolution epilogues."""
    m, dec = _decoder_pair(3, pixel_groups=groups)
    dec.fuse_gn = bool(fuse)
    dec.fuse_upsample = fuse != "plain_upsample"
    z = torch.randn(1, 16, h, w, generator=torch.Generator().manual_seed(h))
    with torch.no_grad():
        ref = m.decode(z.bfloat16().float(), return_dict=False)[0]
    img = dec.decode(z.cuda())
    torch.cuda.synchronize()
    assert img.shape == (1, 3, 8 * h, 8 * w) and img.dtype == torch.bfloat16
    assert torch.isfinite(img.float()).all()
    p = _psnr(img, ref)
    print(f"[vae] {8 * h} x {8 * w}: HIP decode vs fp32 module {p:.1f} dB")
    assert p >= 40.0, p


def test_decoder_1024_vs_fp32_module_and_timing():
    """The headline size: 128 x 128 x 16 latent -> 1024 x 1024 image (FluxKontext/inplace.py:396-402) against the fp32 module (run on the host
    cores: test infrastructure).  Time: VERDICT round 5 next #4 asked for <= 20 ms (the eager bf16 module of the same box takes 73-74 ms:
    tools/vae_decode_bench.py, profiles/r06_vae_decode_bench.json); asserted with slack for a slow box."""
    import time
    m, dec = _decoder_pair(5)
    z = torch.randn(1, 16, 128, 128, generator=torch.Generator().manual_seed(1))
    ref = _fp32_on_cpu(lambda: m.decode(z.bfloat16().float(), return_dict=False)[0])     # 12 s on 64 host threads (MIOpen's fp32 path: 85 s)
    img = dec.decode(z.cuda())
    torch.cuda.synchronize()
    p = _psnr(img, ref)
    zc = z.cuda()
    for _ in range(2):
        dec.decode(zc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dec.decode(zc)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"[vae] 1024 x 1024: HIP decode vs fp32 module {p:.1f} dB; {ms:.1f} ms ({dec.flops(128, 128) / ms / 1e9:.0f} TFLOP/s)")
    assert p >= 40.0, p
    assert ms <= 22.0, ms


def test_decoder_rejects_what_it_does_not_implement():
    m = host_vae.seeded(1)
    sd = dict(m.state_dict())
    sd["decoder.extra.weight"] = torch.zeros(1)
    with pytest.raises(_lib.RegionEHipError):
        V.HipVaeDecoder(sd, "cuda")
    sd = dict(m.state_dict())
    del sd["decoder.mid_block.attentions.0.to_q.weight"]
    with pytest.raises(_lib.RegionEHipError):
        V.HipVaeDecoder(sd, "cuda")
    dec = V.HipVaeDecoder(m.state_dict(), "cuda")
    with pytest.raises(_lib.RegionEHipError):
        dec.decode(torch.zeros(1, 16, 8, 8))                 # CPU tensor: no fallback


def test_every_gpu_kernel_of_a_decode_and_an_encode_is_a_libregione_hip_kernel():
    """SURVEY.md section 7 "no Python fallback on the GPU path", for the f4 kernels: a warm decode / encode (buffers of this size pooled, weights
    re-laid at adoption) dispatches `rgn::` kernels only - no at::native elementwise / fill / copy kernel."""
    from tests.test_gpu_no_eager_kernels import _foreign, _gpu_activity_names
    m = host_vae.seeded(2)
    dec, enc = V.HipVaeDecoder(m.state_dict(), "cuda"), V.HipVaeEncoder(m.state_dict(), "cuda")
    z = torch.randn(1, 16, 24, 32).to("cuda", torch.bfloat16)
    x = torch.randn(1, 3, 192, 256).clamp(-1, 1).to("cuda", torch.bfloat16)
    dec.decode(z), enc.encode(x)                              # warm
    torch.cuda.synchronize()
    img, names = _gpu_activity_names(lambda: dec.decode(z))
    assert len(names) > 90 and any("gemm_bf16_kernel" in n for n in names) and any("gn_apply_kernel" in n for n in names), names[:5]
    assert _foreign(names) == [], _foreign(names)
    mom, names = _gpu_activity_names(lambda: enc.encode(x))
    assert len(names) > 60 and _foreign(names) == [], _foreign(names)
    assert img.shape == (1, 3, 192, 256) and mom.shape == (1, 32, 24, 32)
