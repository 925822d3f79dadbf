"""-m gpu: every kernel a RegionE edit dispatches is one of libregione_hip.so's (SURVEY.md section 7:
"no Python fallback on the GPU path").

The whole 28-step edit of each family - full steps, the partition step, region steps, cache-served steps, the compaction /
restoration steps, CFG combines, both CFG branches - runs under torch.profiler; every GPU activity it records must be a kernel
of namespace `rgn::` or a runtime copy / fill (`__amd_rocclr_*`, Memcpy / Memset records: host-built tables going to the device,
the 4-byte K_e read, device-to-device row copies).  An `at::native::*` elementwise / cat / fill kernel anywhere in the loop fails
the test with its name.  (Round 4's rocprofv3 listing showed ~15 such templates; the per-edit ones were `torch.cat([latents,
image_latents])`, the bf16 adds of the time-text embedding, `torch.cat((arange(T), ids + T))` and the cache slabs' zero fill.)
"""
import pytest
import torch

from regione_amd import RegionEHelper, synth
from regione_amd.harness import flux as H

pytestmark = pytest.mark.gpu


def _build(family, golden):
    from regione_amd.harness import qwen as HQ, step1x as HS
    cu = lambda t: t.cuda() if t is not None else None
    h = w = 16
    if family == "qwen":
        cfg = synth.FluxConfig(**synth.QWEN_TOY)
        wts = synth.make_flux_weights(cfg, seed=6, dtype=torch.bfloat16, w_std=0.05)
        pipe = HQ.QwenImageEditPipeline(HQ.QwenImageTransformer2DModel(cfg, "cuda").load_state_dict(wts))
        img = golden("qwen_toy_bf16")["image_latents"]
    elif family.startswith("flux"):
        cfg = synth.FluxConfig(**synth.TOY)
        wts = synth.make_flux_weights(cfg, seed=42, dtype=torch.bfloat16, w_std=0.05)
        pipe = H.FluxKontextPipeline(H.FluxTransformer2DModel(cfg, "cuda").load_state_dict(wts))
        img = golden("toy_bf16")["image_latents"]
    else:
        cfg = synth.FluxConfig(guidance_embeds=False, **synth.TOY)
        wts = synth.make_flux_weights(cfg, seed=5, dtype=torch.bfloat16, w_std=0.05)
        cls = HS.Step1XEditPipelineV1P2 if family.endswith("v1p2") else HS.Step1XEditPipeline
        pipe = cls(HS.Step1XEditTransformer2DModel(cfg, "cuda").load_state_dict(wts))
        img = golden("s1xv2_toy_bf16" if family.endswith("v1p2") else "s1x_toy_bf16")["image_latents"]
    Tn = 32 if family in ("step1x", "flux_true_cfg") else 24
    lat, _, prompt, y = [cu(t) for t in synth.make_edit_inputs(h, w, 32, cfg, seed=9 if not family.startswith("flux") else 42, dtype=torch.bfloat16)]
    _, _, nprompt, ny = [cu(t) for t in synth.make_edit_inputs(h, w, Tn, cfg, seed=10, dtype=torch.bfloat16)]
    kw = dict(image=cu(img), prompt_embeds=prompt, height=h * 16, width=w * 16, latents=lat, return_dict=False)
    if family != "flux":
        kw.update(negative_prompt_embeds=nprompt, true_cfg_scale=4.0)
    if family != "qwen":
        kw.update(pooled_prompt_embeds=y)
        if family != "flux":
            kw.update(negative_pooled_prompt_embeds=ny)
    if family.startswith("flux"):
        kw.update(guidance_scale=2.5)
    return pipe, kw


def _gpu_activity_names(fn):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        out = fn()
        torch.cuda.synchronize()
    names = []
    for e in prof.events():
        if str(getattr(e, "device_type", "")).endswith("CUDA"):
            names.append(e.name)
    return out, names


def _foreign(names):
    ok = lambda n: ("rgn::" in n) or n.startswith("__amd_rocclr_") or n.lower().startswith(("memcpy", "memset"))
    return sorted({n[:120] for n in names if not ok(n)})


@pytest.mark.parametrize("family", ["flux", "flux_true_cfg", "step1x", "step1x_v1p2", "qwen"])
def test_every_gpu_kernel_of_a_regione_edit_is_a_libregione_hip_kernel(family, golden):
    pipe, kw = _build(family, golden)
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.5)
    helper.enable()
    pipe(**kw)                                            # warm: workspaces, rotary tables, cache slabs of this size
    torch.cuda.synchronize()
    kinds = {}
    out, names = _gpu_activity_names(lambda: pipe(trace=None, **kw)[0])
    M = pipe._regione_manager
    assert 0 < int(M.edited_ids.shape[1]) < 256, "the edit must contain region steps"
    assert len(names) > 200 and any("gemm_bf16_kernel" in n for n in names) and any("attention" in n for n in names), names[:5]
    assert any("arp_sim_kernel" in n for n in names) and any("euler_kernel" in n for n in names)
    assert _foreign(names) == [], _foreign(names)
    assert torch.isfinite(out.float()).all()
    # and the full-token loop on the same engine
    helper.disable()
    pipe(**kw)
    _, names = _gpu_activity_names(lambda: pipe(**kw)[0])
    assert _foreign(names) == [], _foreign(names)


def test_the_profiler_sees_torch_eager_kernels_when_there_are_some():
    """The check above is only worth something if an at::native kernel WOULD be reported."""
    a = torch.randn(1 << 16, device="cuda")
    _, names = _gpu_activity_names(lambda: torch.cat([a, a]) * 2.0)
    assert _foreign(names), names
