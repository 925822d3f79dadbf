"""GPU cfg_combine vs the oracle (torch CPU) on random rows: element mismatches per mode / dtype."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from regione_amd import ops
from oracle import regione_oracle as O
torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float32):
    for fam, mode, scale in (("flux", 0, 6.0), ("step1x", 1, 6.0), ("qwen", 2, 4.0), ("step1x", 1, 3.7), ("qwen", 2, 2.3)):
        pos, neg = torch.randn(1, 8192, 64).to(dt), torch.randn(1, 8192, 64).to(dt)
        ref = O.cfg_combine(fam, pos, neg, scale, t=torch.tensor(1e9), power=0.4)
        got = ops.cfg_combine(pos.cuda(), neg.cuda(), scale, mode, 0.4).cpu()
        bad = (got != ref)
        print(dt, fam, scale, "mismatching elements", int(bad.sum()), "rows", int(bad.any(-1).sum()), "max rel", float(((got.float() - ref.float()).abs() / ref.float().abs().clamp_min(1e-6)).max()))
