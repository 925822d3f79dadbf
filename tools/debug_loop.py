import sys, torch, numpy as np
sys.path.insert(0, '.')
from tests.conftest import load_golden
from oracle import regione_oracle as O
from regione_amd import RegionEHelper, synth
from regione_amd.harness import flux as H
from tests.test_host_logic import FakeTransformer
name = sys.argv[1] if len(sys.argv) > 1 else "loop_bf16_32"
g = load_golden(name)
h, w = g["h"], g["w"]; dt = torch.bfloat16 if g["bf16"] else torch.float32; L = h*w
lat, img, _, _ = synth.make_edit_inputs(h, w, 8, synth.FluxConfig(), seed=g["seed"], dtype=dt)
tgt = synth.region_target(h, w, tuple(int(x) for x in g["box"]), img, seed=g["tseed"], ramp=g["ramp"])
full = torch.cat([tgt, img[0].float()], 0)
tr = FakeTransformer(full, w, L, device="cuda")
pipe = H.FluxKontextPipeline(tr)
helper = RegionEHelper(pipe); helper.set_params(threshold=g["threshold"], cache_threshold=g["cache_threshold"], refresh_step=str(g["refresh_step"])); helper.enable()
trace = {}
out = pipe(image=img, prompt_embeds=torch.zeros(1,8,4), pooled_prompt_embeds=torch.zeros(1,4), height=h*16, width=w*16, latents=lat, return_dict=False, trace=trace)[0]
# oracle trace
trc = FakeTransformer(full, w, L)
st = O.RegionState(); st.set_parameters(28, 6, 2, str(g["refresh_step"]), g["threshold"], g["cache_threshold"], True)
otr = {}
O.denoise(lambda x, t, ids: trc(hidden_states=x, timestep=(t.expand(1).to(x.dtype)/1000), img_ids=ids)[0], st, lat, img, synth.flux_latent_ids(h, w), 8, h, w, trace=otr)
for i in range(28):
    a, b = trace["noise_pred"][i].cpu(), otr["noise_pred"][i]
    c, d = trace["latents"][i].cpu(), otr["latents"][i]
    print(i, trace["kind"][i], "np_equal", torch.equal(a, b), "nbad", int((a != b).sum()), "lat_equal", torch.equal(c, d), "nbad", int((c != d).sum()), tuple(c.shape))
