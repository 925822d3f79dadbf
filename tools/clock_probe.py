#!/usr/bin/env python3
"""Shader clock / power under sustained load of one kernel (GPU box only): queues ~2 s of back-to-back
launches and polls rocm-smi while the GPU drains them.   python tools/clock_probe.py [attn|gemm|vendor|idle]"""
import sys, os, subprocess, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from regione_amd import ops


def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.search(r"sclk clock level.*?\((\d+)Mhz\)", out)
    mclk = re.search(r"mclk clock level.*?\((\d+)Mhz\)", out)
    pw = re.search(r"Power \(W\):\s*([\d.]+)", out)
    return (sclk and sclk.group(1), mclk and mclk.group(1), pw and pw.group(1))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "attn"
    rnd = lambda *s: (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)
    if what == "attn":
        H, S = 24, 8704
        q, k, vt = rnd(S, H * 128), rnd(S, H * 128), rnd(H * 128, S)
        out = torch.empty_like(q)
        fn, n = (lambda: ops.attention(q, k, vt, out, S, H)), 3000
    elif what == "gemm":
        A, W, b = rnd(8704, 3072), rnd(21504, 3072) * 0.05, rnd(21504)
        out = torch.empty(8704, 21504, dtype=torch.bfloat16, device="cuda")
        fn, n = (lambda: ops.gemm(A, W, b, out)), 2000
    elif what == "vendor":          # reference point: the vendor library's kernel on the same shape (hipBLASLt via torch.addmm)
        A, W, b = rnd(8704, 3072), rnd(21504, 3072) * 0.05, rnd(21504)
        out = torch.empty(8704, 21504, dtype=torch.bfloat16, device="cuda")
        fn, n = (lambda: torch.addmm(b, A, W.t(), out=out)), 2000
    else:
        fn, n = (lambda: None), 0
    print("idle     sclk/mclk/W:", smi())
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    t0 = time.time()
    while not e.query() and time.time() - t0 < 20:
        print(f"t={time.time()-t0:5.2f}s sclk/mclk/W:", smi(), flush=True)
    torch.cuda.synchronize()
    if n:
        print(f"{what}: {s.elapsed_time(e)/n*1e3:.1f} us per launch sustained over {n} launches")


if __name__ == "__main__":
    main()
