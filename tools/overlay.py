#!/usr/bin/env python3
"""Edited-region overlay (reference: src/Step1X-Edit-v1p2/inplace.py:456-497).

The reference's debugging build paints the partition over the decoded image: edited token id -> cell
(id // W_tok, id % W_tok) of the token grid, nearest-upsampled by 2 * vae_scale_factor to pixels, white with alpha 160,
alpha-composited over the RGBA image.  Same arithmetic here as library functions + a small CLI:

    python tools/overlay.py --ids ids.npy --height 1024 --width 1024 [--image edit.png] --out overlay.png

`ids.npy`: the edited token ids of an edit (`pipeline._regione_manager.edited_ids.cpu().numpy()`, shape [1, K] or [K]).
Without --image the mask itself is written (white = edited).  tools/edit_driver.py --overlay-dir writes one per item.
"""
import argparse
import os
import sys

import numpy as np


def token_ids_to_mask(token_ids, height, width, vae_scale_factor=8, patch_size=2):
    """[K] (or [1, K]) int ids -> uint8 [height, width] pixel mask (1 = edited).  Row-major token grid of
    (height // (patch*vae), width // (patch*vae)) cells (utils.py:337-340), each cell patch*vae pixels square."""
    ids = np.asarray(token_ids).reshape(-1).astype(np.int64)
    cell = vae_scale_factor * patch_size
    h_tok, w_tok = int(height) // cell, int(width) // cell
    if ids.size and (ids.min() < 0 or ids.max() >= h_tok * w_tok):
        raise ValueError(f"token id outside the {h_tok} x {w_tok} grid")
    grid = np.zeros((h_tok, w_tok), np.uint8)
    grid[ids // w_tok, ids % w_tok] = 1
    return np.kron(grid, np.ones((cell, cell), np.uint8))                 # nearest-neighbour upsampling


def overlay_rgba(image_rgba, mask, alpha=160):
    """Alpha-composite a white layer of opacity `alpha` (reference: 160) where mask == 1 over an RGBA uint8 image
    [H, W, 4] (PIL.Image.alpha_composite arithmetic, integer rounding)."""
    img = np.asarray(image_rgba).astype(np.float64)
    assert img.ndim == 3 and img.shape[2] == 4 and img.shape[:2] == mask.shape, (img.shape, mask.shape)
    a_src = mask.astype(np.float64) * alpha / 255.0
    a_dst = img[..., 3] / 255.0
    a_out = a_src + a_dst * (1 - a_src)
    rgb = (255.0 * a_src[..., None] + img[..., :3] * (a_dst * (1 - a_src))[..., None]) / np.maximum(a_out, 1e-12)[..., None]
    out = np.concatenate([rgb, 255.0 * a_out[..., None]], axis=-1)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def save_overlay(path, token_ids, height, width, image=None, vae_scale_factor=8):
    from PIL import Image
    mask = token_ids_to_mask(token_ids, height, width, vae_scale_factor)
    if image is None:
        Image.fromarray(mask * 255, mode="L").save(path)
        return mask
    im = image if isinstance(image, Image.Image) else Image.open(image)
    im = im.convert("RGBA").resize((mask.shape[1], mask.shape[0]))
    Image.fromarray(overlay_rgba(np.asarray(im), mask), mode="RGBA").save(path)
    return mask


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ids", required=True)
    ap.add_argument("--height", type=int, required=True)
    ap.add_argument("--width", type=int, required=True)
    ap.add_argument("--image")
    ap.add_argument("--out", required=True)
    ap.add_argument("--vae-scale-factor", type=int, default=8)
    a = ap.parse_args()
    m = save_overlay(a.out, np.load(a.ids), a.height, a.width, a.image, a.vae_scale_factor)
    print(f"{a.out}: {int(m.sum())} of {m.size} pixels edited ({100.0 * m.mean():.1f} %)")


if __name__ == "__main__":
    sys.exit(main())
