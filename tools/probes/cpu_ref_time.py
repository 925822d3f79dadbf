import sys, time; sys.path.insert(0, '/root/repo')
import torch
from tests import host_vae
m = host_vae.seeded(5)
z = torch.randn(1, 16, 128, 128)
print("threads", torch.get_num_threads(), flush=True)
for n in (64, 128):
    torch.set_num_threads(n)
    t0 = time.time()
    with torch.no_grad():
        r = m.decode(z, return_dict=False)[0]
    print(n, "cpu fp32 decode s", round(time.time() - t0, 1), flush=True)
x = torch.randn(1, 3, 1024, 1024)
t0 = time.time()
with torch.no_grad():
    r = m.encoder(x)
print("cpu fp32 encode s", round(time.time() - t0, 1), flush=True)
t0 = time.time()
mg = m.cuda()
with torch.no_grad():
    r = mg.decode(z.cuda(), return_dict=False)[0]; torch.cuda.synchronize()
print("gpu fp32 decode (first call) s", round(time.time() - t0, 1), flush=True)
