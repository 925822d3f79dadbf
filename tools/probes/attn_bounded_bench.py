import os, sys
sys.path.insert(0, ".")
import torch, tools.bench_kernels as B
from regione_amd import ops
orig = ops.attention
ops.attention = lambda q,k,vt,out,skv,H,**kw: orig(q,k,vt,out,skv,H,score_bound=12.0)
B.ops.attention = ops.attention
B.bench_attn()
