#!/usr/bin/env python3
"""Vendor measuring stick for the attention kernel (GPU box only): torch's scaled_dot_product_attention on ROCm, every fused
backend this build offers (flash = AOTriton / CK, memory-efficient), against regione_amd's attention on the SAME problems -
24 heads x 128, Skv 8704, bf16, non-causal - at the full-step (Sq 8704) and region-step (Sq 1536 / 708) query counts.
Back-to-back launches for ~1.5 s per row (sustained clocks), HIP events on the launch stream.

    python tools/probes/vendor_attn.py > gpurun_out/r05/vendor_attn.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from regione_amd import ops


def timed(fn, seconds=1.5):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    n = max(5, int(seconds * 1e3 / max(s.elapsed_time(e), 1e-3)))
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3, n


def main():
    from torch.nn.attention import SDPBackend, sdpa_kernel
    H, S = 24, 8704
    rnd = lambda *s: (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)
    rows = []
    for Sq in (8704, 1536, 708):
        flops = 4.0 * Sq * S * H * 128
        q2, k2, vt2 = rnd(Sq, H * 128), rnd(S, H * 128), rnd(H * 128, S)
        o2 = torch.empty_like(q2)
        us, n = timed(lambda: ops.attention(q2, k2, vt2, o2, S, H))
        rows.append(dict(kernel="regione_amd attention (dynamic max)", Sq=Sq, Skv=S, us=us, launches=n, tflops=flops / us / 1e6))
        us, n = timed(lambda: ops.attention(q2, k2, vt2, o2, S, H, score_bound=8.0))
        rows.append(dict(kernel="regione_amd attention (static shift, the pipeline's path)", Sq=Sq, Skv=S, us=us, launches=n,
                         tflops=flops / us / 1e6))
        q, k, v = rnd(1, H, Sq, 128), rnd(1, H, S, 128), rnd(1, H, S, 128)
        for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("mem_efficient", SDPBackend.EFFICIENT_ATTENTION)):
            try:
                with sdpa_kernel([be]):
                    us, n = timed(lambda: F.scaled_dot_product_attention(q, k, v))
                rows.append(dict(kernel=f"torch SDPA backend={name}", Sq=Sq, Skv=S, us=us, launches=n, tflops=flops / us / 1e6))
            except Exception as ex:                       # backend not built into this torch
                rows.append(dict(kernel=f"torch SDPA backend={name}", Sq=Sq, Skv=S, error=str(ex)[:200]))
        # [B, S, H, D] layout (what flash_attn_func / the reference's call site hands over, inplace.py:796-801)
        qb, kb, vb = (t.transpose(1, 2).contiguous().transpose(1, 2) for t in (q, k, v))
        try:
            with sdpa_kernel([SDPBackend.FLASH_ATTENTION]):
                us, n = timed(lambda: F.scaled_dot_product_attention(qb, kb, vb))
            rows.append(dict(kernel="torch SDPA backend=flash, [B,S,H,D] memory layout", Sq=Sq, Skv=S, us=us, launches=n,
                             tflops=flops / us / 1e6))
        except Exception as ex:
            rows.append(dict(kernel="torch SDPA backend=flash, [B,S,H,D] memory layout", Sq=Sq, Skv=S, error=str(ex)[:200]))
    print(json.dumps(dict(torch=torch.__version__, hip=torch.version.hip, device=torch.cuda.get_device_name(0),
                          peak_bf16_tflops=2500.0, rows=rows), indent=1))


if __name__ == "__main__":
    main()
