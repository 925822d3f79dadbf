"""CPU: the host logic of the f4 kernels (regione_amd/vae.py) - no GPU, no library calls.

The implicit-GEMM view of a convolution is checked by EMULATING the GEMM the kernel runs with plain torch on the CPU: rows of a zero-bordered
pixel-major image, the K walk of `rgn_conv_bf16` (kernel rows of `window` contiguous pixels, a jump to the next image row), the re-laid / block-
Toeplitz weight matrix of `ConvWeights` - against `torch.nn.functional.conv2d`.  Same for the stride-2 launch and its row table."""
import math

import torch

from regione_amd import vae as V
from tests import host_vae


def _padded_rows(x, guard):
    """[1, C, H, W] -> ([guard + Hp * Wp + guard, C] pixel-major zero-bordered image, offset of row 0)"""
    _, C, H, W = x.shape
    img = torch.zeros(H + 2, W + 2, C, dtype=x.dtype)
    img[1:-1, 1:-1] = x[0].permute(1, 2, 0)
    flat = torch.cat([torch.zeros(guard, C, dtype=x.dtype), img.reshape(-1, C), torch.zeros(guard, C, dtype=x.dtype)])
    return flat, guard


def _gemm_rows(flat, base, first_row, lda_pixels, M, window, Wp, taps):
    """The A matrix the kernel walks: row m starts at pixel first_row + m * lda_pixels; a kernel row = `window` contiguous pixels, then + Wp."""
    rows = []
    for m in range(M):
        p0 = base + first_row + m * lda_pixels
        ks = [flat[p0 + ky * Wp: p0 + ky * Wp + window].reshape(-1) for ky in range(3 if taps == 9 else 1)]
        rows.append(torch.cat(ks))
    return torch.stack(rows)


def test_param_shapes_are_the_modules_state_dicts():
    m = host_vae.seeded(0)
    sd = m.state_dict()
    assert {k[len("decoder."):]: tuple(v.shape) for k, v in sd.items() if k.startswith("decoder.")} == V.decoder_param_shapes()
    assert {k[len("encoder."):]: tuple(v.shape) for k, v in sd.items() if k.startswith("encoder.")} == V.encoder_param_shapes()
    syn = V.synthetic_decoder_state_dict(1)
    assert {k: tuple(v.shape) for k, v in syn.items()} == V.decoder_param_shapes()


def test_conv_weights_as_a_gemm_equals_conv2d_with_and_without_pixel_groups():
    g = torch.Generator().manual_seed(0)
    H, W, ci = 6, 9, 4
    for co, ldy, group in ((5, 8, 1), (5, 8, 2), (3, 8, 4), (5, 5, 1)):
        x = torch.randn(1, ci, H, W, generator=g, dtype=torch.float64)
        w = torch.randn(co, ci, 3, 3, generator=g, dtype=torch.float64)
        b = torch.randn(co, generator=g, dtype=torch.float64)
        ref = torch.nn.functional.conv2d(x, w, b, padding=1)[0].permute(1, 2, 0)                  # [H, W, co]
        cw = V.ConvWeights(w.permute(0, 2, 3, 1).float(), b.float(), group=group, ldy=ldy)
        Wm = cw.w.double() if False else None
        # the bf16 rounding of ConvWeights is not under test here: rebuild the same matrix in fp64
        co_, kh, kw, ci_ = co, 3, 3, ci
        if group == 1:
            Wm, bias = w.permute(0, 2, 3, 1).reshape(co, -1), b
        else:
            t = torch.zeros(group, ldy, 3, group + 2, ci, dtype=torch.float64)
            for p in range(group):
                for kx in range(3):
                    t[p, :co, :, p + kx, :] = w.permute(0, 2, 3, 1)[:, :, kx, :]
            Wm = t.reshape(group * ldy, -1)
            bias = torch.zeros(group, ldy, dtype=torch.float64)
            bias[:, :co] = b
            bias = bias.reshape(-1)
        assert tuple(cw.w.shape) == tuple(Wm.shape) and torch.allclose(cw.w.double(), Wm, atol=2e-2, rtol=2e-2)
        Wp, Hp = W + 2, H + 2
        flat, base = _padded_rows(x, guard=Wp + 8)
        rows = Hp * Wp
        M = (rows + group - 1) // group
        A = _gemm_rows(flat, base, -(Wp + 1), group, M, group + 2, Wp, 9)
        out = (A @ Wm.T + bias).reshape(M * group, -1)[:rows]                                     # [rows, ldy or co]
        got = out.reshape(Hp, Wp, -1)[1:-1, 1:-1, :co]
        assert torch.allclose(got, ref, atol=1e-9), (co, group)


def test_one_by_one_block_diagonal_groups():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 4, 5, 6, generator=g, dtype=torch.float64)
    w = torch.randn(3, 4, generator=g, dtype=torch.float64)
    cw = V.ConvWeights(w[:, None, None, :].float(), torch.zeros(3), group=2, ldy=4)
    assert tuple(cw.w.shape) == (8, 8) and cw.taps == 1
    Wm = cw.w.double()
    assert torch.count_nonzero(Wm[:4, 4:]) == 0 and torch.count_nonzero(Wm[4:, :4]) == 0 and torch.count_nonzero(Wm[3]) == 0      # block diagonal, padding channel zero
    assert torch.allclose(Wm[:3, :4], w, atol=2e-2) and torch.allclose(Wm[4:7, 4:], w, atol=2e-2)


def test_stride_two_row_table_and_gemm_equal_padded_strided_conv2d():
    g = torch.Generator().manual_seed(2)
    H, W, ci, co = 8, 12, 3, 4
    x = torch.randn(1, ci, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(co, ci, 3, 3, generator=g, dtype=torch.float64)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (0, 1, 0, 1)), w, stride=2)[0].permute(1, 2, 0)       # [H/2, W/2, co]
    Wp = W + 2
    flat, base = _padded_rows(x, guard=Wp + 8)
    M = (H // 2) * Wp
    A = _gemm_rows(flat, base, Wp + 1, 2, M, 3, Wp, 9)                       # row m starts at image pixel 2 m + Wp + 1
    out = A @ w.permute(0, 2, 3, 1).reshape(co, -1).T
    table = V.downsample_rows(H, W, "cpu")
    Ho, Wo = H // 2, W // 2
    assert table.shape == (M,) and int(table.max()) == (Ho + 2) * (Wo + 2)   # the scratch row: first guard row behind the output image
    dst = torch.zeros((Ho + 2) * (Wo + 2) + 1, co, dtype=torch.float64)
    dst[table] = out                                                         # the epilogue's row scatter
    got = dst[:-1].reshape(Ho + 2, Wo + 2, co)
    assert torch.allclose(got[1:-1, 1:-1], ref, atol=1e-9)
    border = torch.cat([got[0].reshape(-1), got[-1].reshape(-1), got[:, 0].reshape(-1), got[:, -1].reshape(-1)])
    assert torch.count_nonzero(border) == 0                                  # valid rows only ever land on interior pixels
    assert len(set(table[table < (Ho + 2) * (Wo + 2)].tolist())) == Ho * Wo


def test_latent_dist_is_the_diagonal_gaussian_of_the_moments():
    mom = torch.randn(1, 8, 3, 3)
    mom[:, 4:] *= 20
    d = V.EncoderOutput(mom).latent_dist
    assert torch.equal(d.mode(), mom[:, :4])
    s = d.sample(torch.Generator().manual_seed(5))
    noise = torch.randn(mom[:, :4].shape, generator=torch.Generator().manual_seed(5))
    assert torch.allclose(s, mom[:, :4] + torch.exp(0.5 * mom[:, 4:].clamp(-30, 20)) * noise)


def test_decoder_flops_formula():
    """10.47 TFLOP for the 1024 x 1024 decode (the figure bench.py's end_to_end.vae_decode_tflops divides by the time)."""
    f = V.HipVaeDecoder.flops.__get__(type("D", (), dict(ch=(128, 256, 512, 512), zc=16, nres=3, levels=[(512, 512, True), (512, 512, True), (512, 256, True), (256, 128, False)]))())(128, 128)
    assert abs(f / 1e12 - 10.47) < 0.05, f


def test_upsample_folded_into_its_convolution_four_phases_equal_interpolate_then_conv2d():
    """rgn_conv_up2_bf16's host side: tap-summed 2 x 2 phase weights + the row tables, emulated as four GEMMs over the LOW-resolution image."""
    g = torch.Generator().manual_seed(3)
    H, W, ci, co = 5, 7, 3, 4
    x = torch.randn(1, ci, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(co, ci, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(co, generator=g, dtype=torch.float64)
    ref = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)[0].permute(1, 2, 0)
    uw = V.UpConvWeights(w.permute(0, 2, 3, 1).float(), b.float())
    assert tuple(uw.w.shape) == (4, co, 4 * ci)
    fold = ([[0], [1, 2]], [[0, 1], [2]])
    Wp, Hp = W + 2, H + 2
    flat, base = _padded_rows(x, guard=Wp + 8)
    table = V.upsample_rows(H, W, "cpu")
    Wq = 2 * W + 2
    dst = torch.zeros((2 * H + 2) * Wq + 1, co, dtype=torch.float64)
    for ph in range(4):
        a, bb = ph >> 1, ph & 1
        Wm = torch.zeros(co, 2, 2, ci, dtype=torch.float64)
        for r in range(2):
            for c in range(2):
                for ky in fold[a][r]:
                    for kx in fold[bb][c]:
                        Wm[:, r, c, :] += w.permute(0, 2, 3, 1)[:, ky, kx, :]
        assert torch.allclose(uw.w[ph].double(), Wm.reshape(co, -1), atol=3e-2, rtol=3e-2)
        rows = []
        for m in range(Hp * Wp):
            p0 = base + m + (a - 1) * Wp + (bb - 1)
            rows.append(torch.cat([flat[p0 + kr * Wp: p0 + kr * Wp + 2].reshape(-1) for kr in range(2)]))
        out = torch.stack(rows) @ Wm.reshape(co, -1).T + b
        dst[table[ph]] = out
    got = dst[:-1].reshape(2 * H + 2, Wq, co)
    assert torch.allclose(got[1:-1, 1:-1], ref, atol=1e-9)
    assert torch.count_nonzero(got[0]) == 0 and torch.count_nonzero(got[:, 0]) == 0 and torch.count_nonzero(got[-1]) == 0 and torch.count_nonzero(got[:, -1]) == 0
    valid = table[table < (2 * H + 2) * Wq]
    assert valid.numel() == 4 * H * W and len(set(valid.tolist())) == 4 * H * W          # every high-resolution pixel exactly once
