import sys; sys.path.insert(0,'/root/repo')
import torch
from regione_amd import ops, _lib
from tools.probes.conv_overhead import t_us
for M in (264196, 1052676):
    for K in (576, 1152, 2304, 4608):
        A=torch.randn(M,K,device='cuda',dtype=torch.bfloat16); W=torch.randn(256,K,device='cuda',dtype=torch.bfloat16)/K**0.5
        b=torch.zeros(256,device='cuda',dtype=torch.bfloat16); out=torch.empty(M,256,device='cuda',dtype=torch.bfloat16)
        with _lib.plan_override(gemm_pieces=1):
            t=t_us(lambda: ops.gemm(A,W,b,out))
        tiles=-(-M//256); rounds=-(-tiles//256)
        print(f"plain GEMM M={M} K-tiles {K//64}: {t:7.1f} us = {t/rounds:5.1f} per round", flush=True)
        del A,W,out
