"""-m gpu: the reference's CALL SHAPE on a stock pipeline object (RegionE/README.md:85-113, tool/RegionE.py:9-27):

    helper = RegionEHelper(pipe); helper.set_params(...); helper.enable()
    image = pipe(image=..., prompt=...)            # the user keeps calling the pipeline itself
    helper.disable()

for all five pipeline classes, against host stand-ins with the method surface the reference's patched `__call__`s use
(tests/host_standins.py; diffusers itself is not installable here).  Checked: the class swap and its undo, the host
methods called in the reference's order, the loop == the engine's latent-level call on the same inputs (bit for bit
where the text path is static), Step1X's connector driven once per computed forward, Qwen-Image-Edit-2509 with a LIST of
condition images, the per-stage wall-clock fields.
"""
import pytest
import torch

from regione_amd import RegionEHelper, adapters as A

import host_standins as HS

pytestmark = pytest.mark.gpu


def _picture(h=256, w=256, seed=5):
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(1, 3, h, w, generator=g)
    p[:, :, h // 4: h // 4 + h // 3, w // 3: w // 3 + w // 3] = 0.0
    return p


def _gen():
    return torch.Generator().manual_seed(1)


def test_flux_stock_pipeline_call_shape():
    pipe = HS.FluxKontextPipeline(HS.stub_trunk("flux"))
    cls = type(pipe)
    helper = RegionEHelper(pipe)                                 # like the reference's: only the class name is read here
    assert helper.name == "FluxKontextPipeline" and helper.pipeline is pipe and not A.is_engine_pipeline(pipe)
    assert getattr(pipe, "_regione_engine", None) is None
    helper.set_params(threshold=0.5)
    helper.enable()                                              # weights adopted here, once (pipe._regione_engine)
    assert A.is_engine_pipeline(pipe._regione_engine) and helper._engine() is pipe._regione_engine
    assert type(pipe).__name__ == "RegionEFluxKontextPipeline" and isinstance(pipe, cls)
    trace = {}
    out = pipe(image=_picture(), prompt="make the square red", generator=_gen(), output_type="pt", guidance_scale=2.5,
               preferred_resolutions=[(256, 256)], trace=trace)
    assert tuple(out.images.shape) == (1, 3, 256, 256) and torch.isfinite(out.images).all()
    assert [c[0] for c in pipe.calls] == ["encode_prompt", "prepare_latents", "unpack", "free"]
    assert len(trace["kind"]) == 28 and "".join(trace["kind"]).startswith("FFFFFF") and "R" in trace["kind"]
    assert set(out.timing) == {"encode_s", "loop_s", "decode_s"} and all(v >= 0 for v in out.timing.values())
    # default auto-resize snaps to a Kontext training resolution (advisor finding: the table is the DEFAULT)
    assert (1024, 1024) in A.PREFERRED_KONTEXT_RESOLUTIONS and len(A.PREFERRED_KONTEXT_RESOLUTIONS) == 17
    # the loop the patched pipeline ran == the patched engine's latent-level call on the same packed inputs
    lat = pipe(image=_picture(), prompt="make the square red", generator=_gen(), output_type="latent", guidance_scale=2.5,
               preferred_resolutions=[(256, 256)]).images
    pe, pp, _ = pipe.encode_prompt(prompt="make the square red")
    l0, il, _, _ = pipe.prepare_latents(_picture() * 2 - 1, 1, 16, 256, 256, torch.bfloat16, None, _gen())
    direct = helper._engine()(image=il.cuda(), prompt_embeds=pe.cuda(), pooled_prompt_embeds=pp.cuda(), height=256, width=256,
                               latents=l0.cuda(), guidance_scale=2.5, return_dict=False)[0]
    assert torch.equal(lat, direct)
    helper.disable()
    assert type(pipe) is cls and getattr(pipe, "_regione_host_class", None) is None
    # a second helper re-uses the adopted engine instead of copying 12 B parameters again
    assert RegionEHelper(pipe)._engine() is helper._engine()
    with pytest.raises(KeyError):
        RegionEHelper(type("StableDiffusionPipeline", (), {"transformer": None})())


@pytest.mark.parametrize("v1p2", [False, True])
def test_step1x_stock_pipeline_call_shape_with_host_connector(v1p2):
    cls = HS.Step1XEditPipelineV1P2 if v1p2 else HS.Step1XEditPipeline
    trunk = HS.stub_trunk("step1x")
    pipe = cls(trunk)
    conn = HS.ToyConnector().to(torch.bfloat16)
    object.__setattr__(trunk, "connector", conn)                 # the host's own text path (timestep dependent)
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.5)
    helper.enable()
    assert type(pipe).__name__ == "RegionE" + cls.__name__
    trace = {}
    kw = dict(image=_picture(), prompt="turn the sky green", generator=_gen(), output_type="latent", trace=trace)
    if not v1p2:
        kw["latents"] = None
    a = pipe(**kw).images
    kinds = "".join(trace["kind"])
    assert len(kinds) == 28 and torch.isfinite(a.float()).all() and a.shape == (1, 256, 64)
    names = [c[0] for c in pipe.calls]
    assert names[:4] == ["encode_image", "encode_prompt", "encode_prompt", "prepare_latents"]
    assert pipe.calls[0][1] == (3 if v1p2 else 2)                # v1p2's encode_image takes size_level
    assert pipe.calls[2][1] == ""                                # no negative prompt given + an image -> "" (inplace.py:231)
    # the connector ran once per branch per COMPUTED forward, never on cache-served steps
    computed = kinds.count("F") + kinds.count("R")
    assert conn.calls == 2 * computed and kinds.count("C") > 0
    assert "connector" not in helper._engine().transformer.__dict__          # hook removed after the call
    # deterministic; image-space output goes through the host's decode + _output_process_image
    conn.calls = 0
    kw["generator"] = _gen()
    b = pipe(**kw).images
    assert torch.equal(a, b)
    kw.update(output_type="pt", generator=_gen())
    img = pipe(**kw)
    assert tuple(img.images.shape) == (1, 3, 256, 256) and ("output_process", (256, 256)) in pipe.calls
    think = dict(enable_thinking_mode=False, enable_reflection_mode=False)
    if v1p2:
        # the reference driver's own call, verbatim (src/Step1X-Edit-v1p2/main.py:42-43, :69-77): both switches False
        c = pipe(image=_picture(), prompt="turn the sky green", num_inference_steps=28, true_cfg_scale=6.0, generator=_gen(),
                 output_type="latent", **think).images
        d = pipe(image=_picture(), prompt="turn the sky green", num_inference_steps=28, true_cfg_scale=6.0, generator=_gen(),
                 output_type="latent", max_try_cnt=3).images
        assert torch.equal(c, d) and torch.isfinite(c.float()).all()
        for k in think:                                          # the VLM retry loop itself is not hosted: asked for -> loud
            with pytest.raises(NotImplementedError, match=k):
                pipe(image=_picture(), prompt="x", generator=_gen(), output_type="latent", **{k: True})
    else:                                                        # v1p1's __call__ has no such arguments
        with pytest.raises(TypeError, match="enable_thinking_mode"):
            pipe(image=_picture(), prompt="x", generator=_gen(), output_type="latent", latents=None, **think)
    helper.disable()
    assert type(pipe) is cls


def test_qwen_stock_pipeline_call_shape():
    pipe = HS.QwenImageEditPipeline(HS.stub_trunk("qwen"))
    van = type(pipe)
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.5)
    helper.enable()
    trace = {}
    kw = dict(image=_picture(), prompt="add a hat", negative_prompt=" ", true_cfg_scale=4.0, output_type="latent")
    lat = pipe(trace=trace, generator=_gen(), **kw).images               # output = the 1024^2-area grid of the input image
    assert lat.shape == (1, 4096, 64) and torch.isfinite(lat.float()).all() and len(trace["kind"]) == 28
    assert [c[0] for c in pipe.calls[:3]] == ["encode_prompt", "encode_prompt", "prepare_latents"]
    # an output size that differs from the condition image's grid: the reference's own partition fails on the shape
    # mismatch (utils.py:310-312); here it is refused with a clear message
    with pytest.raises(AssertionError, match="output token grid"):
        pipe(generator=_gen(), height=256, width=256, **kw)
    helper.disable()
    assert type(pipe) is van
    # vanilla (full-token) loop through the wrapper object == the engine's latent-level call on the same inputs
    hosted = A.adopt(pipe)
    lat_v = hosted(generator=_gen(), height=256, width=256, **kw).images
    pe, _ = pipe.encode_prompt(prompt="add a hat")
    ne, _ = pipe.encode_prompt(prompt=" ")
    img = pipe.image_processor.preprocess(pipe.image_processor.resize(_picture(), 1024, 1024), 1024, 1024).unsqueeze(2)
    l0, il = pipe.prepare_latents(img, 1, 16, 256, 256, torch.bfloat16, None, _gen())
    direct = hosted.engine(image=il.cuda(), prompt_embeds=pe.cuda(), negative_prompt_embeds=ne.cuda(), height=256, width=256,
                           latents=l0.cuda(), true_cfg_scale=4.0, return_dict=False, cond_shapes=[(64, 64)])[0]
    assert lat_v.shape == (1, 256, 64) and torch.equal(lat_v, direct)


def test_qwen_plus_list_of_condition_images():
    """Qwen-Image-Edit-2509: two condition images of different aspect ratios; each is resized for the VLM (384^2 area) and
    for the VAE (1024^2 area); the LAST one fixes the output size and is the partition's reference image."""
    pipe = HS.QwenImageEditPlusPipeline(HS.stub_trunk("qwen"))
    helper = RegionEHelper(pipe)
    assert helper.name == "QwenImageEditPlusPipeline"
    helper.set_params(threshold=0.5)
    helper.enable()
    trace = {}
    images = [_picture(192, 384, seed=2), _picture(256, 256, seed=3)]
    out = pipe(image=images, prompt="put the object of image 1 into image 2", negative_prompt=" ", true_cfg_scale=4.0,
               generator=_gen(), output_type="latent", trace=trace).images
    L = 64 * 64                                                    # last image 1:1 -> 1024 x 1024 -> 64 x 64 tokens
    assert out.shape == (1, L, 64) and torch.isfinite(out.float()).all() and len(trace["kind"]) == 28
    assert pipe.calls[0] == ("encode_prompt", "put the object of image 1 into image 2", 2)
    eng = helper._engine()
    M = eng._regione_manager
    # first image 2:1 -> calculate_dimensions(1024^2, 2) = 1440 x 736 -> 46 x 90 tokens; K/V rows = T + L + both images
    n1 = (736 // 16) * (1440 // 16)
    assert M.latent_ids.shape[0] == L + n1 + L
    proc = eng.transformer.transformer_blocks[0].attn.processor
    T = pipe.encode_prompt(prompt="put the object of image 1 into image 2")[0].shape[1]
    assert proc.caches["cond"][2] == T + L + n1 + L
    assert M.condition_latent.shape[1] == L and 0 < M.edited_ids.shape[1] <= L     # (a random toy trunk edits everything)
    helper.disable()


def test_flux_hosted_decode_runs_on_the_hip_vae():
    """a host whose `vae` is an AutoencoderKL (diffusers layout) gets its decoder adopted onto the HIP kernels
    at `enable()` (`pipe._regione_hip_vae`); the image equals the host module's own decode of the same latents (fp32 on the
    CPU) to >= 40 dB; `pipe._regione_hip_vae = False` keeps the host module; a stand-in VAE without that layout is left alone."""
    import math
    import host_vae
    from regione_amd import vae as V

    class KL(host_vae.AutoencoderKLStandIn):
        dtype = torch.float32
        config = HS.Vae.config
        n_decodes = n_encodes = 0

        def decode(self, z, return_dict=True):
            KL.n_decodes += 1
            return super().decode(z, return_dict=return_dict)

        def encode(self, x, return_dict=True):
            KL.n_encodes += 1
            return super().encode(x.float().to(next(self.parameters()).device), return_dict=return_dict)

    class FluxKontextPipeline(HS.FluxKontextPipeline):          # RegionEHelper dispatches on the class NAME, like the reference
        def _latents(self, image, dtype, generator, latents):            # diffusers' _encode_vae_image: retrieve_latents(vae.encode(image), "argmax")
            z = self.vae.encode(image).latent_dist.mode()
            z = (z - self.vae.config.shift_factor) * self.vae.config.scaling_factor
            image_latents = self._pack_latents(z.float().cpu()).to(dtype)
            if latents is None:
                latents = torch.randn(image_latents.shape, generator=generator).to(dtype)
            return latents, image_latents

    torch.manual_seed(11)
    pipe = FluxKontextPipeline(HS.stub_trunk("flux"))
    pipe.vae = KL().eval()
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.5)
    helper.enable()
    kw = dict(image=_picture(), prompt="make the square red", guidance_scale=2.5, preferred_resolutions=[(256, 256)])
    out = pipe(generator=_gen(), output_type="pt", **kw)
    assert isinstance(pipe._regione_hip_vae, V.HipVaeDecoder) and KL.n_decodes == 0          # the host module did not decode
    assert isinstance(pipe._regione_hip_vae_encoder, V.HipVaeEncoder) and KL.n_encodes == 0  # ... nor encode (prepare_latents' vae.encode)
    assert "encode" not in pipe.vae.__dict__                                                # the binding is undone after prepare_latents
    # the condition latents the loop saw = the host module's own encode of the same image (fp32 on the CPU) to >= 40 dB
    with torch.no_grad():
        zr = host_vae.AutoencoderKLStandIn.encode(pipe.vae, _picture() * 2 - 1).latent_dist.mode()
    zh = pipe._regione_hip_vae_encoder.encode((_picture() * 2 - 1).cuda())[:, :16].float().cpu()
    assert 10 * math.log10(float(zr.max() - zr.min()) ** 2 / max(float(((zh - zr) ** 2).mean()), 1e-30)) >= 40.0
    assert tuple(out.images.shape) == (1, 3, 256, 256) and torch.isfinite(out.images.float()).all()
    lat = pipe(generator=_gen(), output_type="latent", **kw).images
    z = pipe._unpack_latents(lat.float().cpu(), 256, 256, 8) / KL.config.scaling_factor + KL.config.shift_factor
    with torch.no_grad():
        ref = pipe.image_processor.postprocess(super(KL, pipe.vae).decode(z.bfloat16().float(), return_dict=False)[0])
    mse = float(((out.images.float().cpu() - ref) ** 2).mean())
    assert 10 * math.log10(1.0 / max(mse, 1e-30)) >= 40.0
    # opting out keeps the host module
    pipe._regione_hip_vae = False
    pipe.vae.cuda()                                           # the loop's latents live on the GPU: the host module must too
    out2 = pipe(generator=_gen(), output_type="pt", **kw)
    assert KL.n_decodes == 1 and KL.n_encodes == 1 and tuple(out2.images.shape) == (1, 3, 256, 256)
    helper.disable()
    # a VAE without the AutoencoderKL layout (the stand-in of the other tests) is not touched
    pipe2 = HS.FluxKontextPipeline(HS.stub_trunk("flux"))
    assert A.hip_vae_for(pipe2, torch.device("cuda", 0)) is None and pipe2._regione_hip_vae is None


def test_step1x_hosted_encode_and_decode_run_on_the_hip_vae():
    """The Step1X-Edit pipelines make the same two VAE calls (Step1XEdit/inplace.py `prepare_latents` / `self.vae.decode`): a hosted Step1X
    pipeline whose `vae` is an AutoencoderKL gets both on the HIP kernels too; the host module's own methods are never entered."""
    import host_vae
    from regione_amd import vae as V

    class KL(host_vae.AutoencoderKLStandIn):
        dtype = torch.float32
        config = HS.Vae.config
        n = 0

        def decode(self, z, return_dict=True):
            KL.n += 1
            return super().decode(z, return_dict=return_dict)

        def encode(self, x, return_dict=True):
            KL.n += 1
            return super().encode(x, return_dict=return_dict)

    class Step1XEditPipeline(HS.Step1XEditPipeline):             # RegionEHelper dispatches on the class NAME
        def _latents(self, image, dtype, generator, latents):
            z = self.vae.encode(image).latent_dist.mode()
            image_latents = self._pack_latents(z.float().cpu()).to(dtype)
            if latents is None:
                latents = torch.randn(image_latents.shape, generator=generator).to(dtype)
            return latents, image_latents

    torch.manual_seed(12)
    trunk = HS.stub_trunk("step1x")
    pipe = Step1XEditPipeline(trunk)
    object.__setattr__(trunk, "connector", HS.ToyConnector().to(torch.bfloat16))
    pipe.vae = KL().eval()
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.5)
    helper.enable()
    out = pipe(image=_picture(), prompt="turn the sky green", generator=_gen(), output_type="pt", latents=None)
    assert tuple(out.images.shape) == (1, 3, 256, 256) and torch.isfinite(out.images.float()).all()
    assert isinstance(pipe._regione_hip_vae, V.HipVaeDecoder) and isinstance(pipe._regione_hip_vae_encoder, V.HipVaeEncoder) and KL.n == 0
    assert "encode" not in pipe.vae.__dict__
    helper.disable()
