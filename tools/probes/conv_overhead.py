#!/usr/bin/env python3
"""Where a VAE convolution's time goes: K sweep at fixed M, N (time per K tile per round vs the fixed cost per tile) and epilogue options."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regione_amd import vae as V


def t_us(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / reps


def main():
    for H, cout in ((512, 256), (1024, 256), (256, 512)):
        rows = (H + 2) ** 2
        ntile = -(-rows // 256) * -(-cout // 256)
        rounds = -(-ntile // 256)
        for cin in (64, 128, 256, 512):
            x = V.PaddedImage(H, H, cin, "cuda"); x.t.normal_()
            y = V.PaddedImage(H, H, cout, "cuda"); r = V.PaddedImage(H, H, cout, "cuda"); r.t.normal_()
            w = torch.randn(cout, 3, 3, cin, device="cuda") / (3 * cin ** 0.5)
            cw = V.ConvWeights(w, torch.zeros(cout, device="cuda"))
            a = t_us(lambda: V.conv(x, cw, y))
            b = t_us(lambda: V.conv(x, cw, y, resid=r))
            c = t_us(lambda: V.conv(x, cw, y, resid=r, gn=True))
            nk = 9 * cin // 64
            print(f"H={H} cout={cout} cin={cin}: tiles {ntile} rounds {rounds} K-tiles {nk}: plain {a:7.1f} us = {a / rounds:5.1f} per round; +resid {b:7.1f}; +resid+gn {c:7.1f}  "
                  f"({2.0 * H * H * 9 * cin * cout / a / 1e6:6.0f} TFLOP/s plain)", flush=True)
            del x, y, r


if __name__ == "__main__":
    main()
