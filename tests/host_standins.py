"""Stand-ins for the five diffusers pipeline classes the reference patches (RegionE/tool/RegionE.py:1-13), with the METHOD
SURFACE the reference's patched `__call__`s use outside the denoise loop (FluxKontext/inplace.py:112-240,:396-410;
Step1XEdit/inplace.py:185-330,:437-455; Step1XEditV1P2/inplace.py:214-300; QwenImageEdit/inplace.py:180-330,:434-455;
QwenImageEditPlus/inplace.py:189-300).  diffusers is not installable in this image: the transformer
trunks are the torch.nn module trees of tests/host_trunks.py (host parameter naming, with their own vanilla forwards and
stock attention processors), everything else is a small
deterministic toy (16-channel 8x 'VAE', hash-seeded 'prompt encoders', a timestep-dependent Step1X 'connector').
Test infrastructure only.
"""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def stub_trunk(family):
    """A host transformer as `from_pretrained` would hand it over: host parameter names (Qwen: img_mod / txt_mod / img_mlp /
    txt_mlp / img_in / txt_in), stock attention processors installed, its own `forward` runnable on the CPU."""
    import host_trunks as HT
    torch.manual_seed(3)
    mod = {"flux": HT.FluxTransformer2DModel, "step1x": HT.Step1XEditTransformer2DModel, "qwen": HT.QwenImageHostTransformer2DModel}[family]()
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if p.dim() == 1 and not n.endswith("bias"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif p.dim() == 1:
                p.copy_(0.01 * torch.randn_like(p))
            else:
                p.copy_(0.05 * torch.randn_like(p))
    return HT.install_vanilla_processors(mod.to(torch.bfloat16))


class ToyConnector(nn.Module):
    """Step1X 'connector' stand-in: a token refiner that DEPENDS ON THE TIMESTEP (like the real Qwen2 connector) and a pooled
    vector from the masked mean.  Lives under `connector.` in the host state dict, which the adapter leaves on the host."""

    def __init__(self, joint_dim=256, pooled_dim=64):
        super().__init__()
        g = torch.Generator().manual_seed(11)
        self.refine = nn.Linear(joint_dim, joint_dim)
        self.pool = nn.Linear(joint_dim, pooled_dim)
        with torch.no_grad():
            self.refine.weight.copy_(0.03 * torch.randn(joint_dim, joint_dim, generator=g))
            self.pool.weight.copy_(0.05 * torch.randn(pooled_dim, joint_dim, generator=g))
        self.calls = 0

    def forward(self, x, t, mask):
        self.calls += 1
        m = mask.unsqueeze(-1).to(x.dtype)
        y = self.pool((x * m).sum(1) / m.sum(1))
        return x + self.refine(x) * (1.0 + t.to(x.dtype).view(-1, 1, 1)), y


class ImageProcessor:
    """torch-tensor images [B, 3, H, W] in [0, 1] (the real one also takes PIL)."""

    def get_default_height_width(self, img):
        return img.shape[-2], img.shape[-1]

    def resize(self, image, h, w):
        return torch.nn.functional.interpolate(image, size=(h, w), mode="nearest")

    def preprocess(self, image, h, w):
        return self.resize(image, h, w) * 2 - 1

    def postprocess(self, image, output_type="pt"):
        return (image / 2 + 0.5).clamp(0, 1)


class Vae:
    dtype = torch.float32

    class config:
        scaling_factor, shift_factor = 0.36, 0.12
        z_dim = 16
        latents_mean, latents_std = [0.01 * i for i in range(16)], [1.0 + 0.02 * i for i in range(16)]

    def encode_pixels(self, image):
        x = torch.nn.functional.avg_pool2d(image, 8)
        return torch.cat([x * (0.5 + 0.1 * i) for i in range(6)], 1)[:, :16]

    def decode(self, lat, return_dict=False):
        if lat.dim() == 5:                                   # Qwen: [B, C, 1, H, W]
            return (torch.nn.functional.interpolate(lat[:, :3, 0], scale_factor=8, mode="nearest").unsqueeze(2),)
        return (torch.nn.functional.interpolate(lat[:, :3], scale_factor=8, mode="nearest"),)


def _pseudo(prompt, *shape):
    g = torch.Generator().manual_seed(sum(map(ord, prompt or "")) + 17)
    return torch.randn(*shape, generator=g).to(torch.bfloat16)


class _Base:
    vae_scale_factor, latent_channels, default_sample_size = 8, 16, 128
    _execution_device = torch.device("cpu")

    def __init__(self, trunk):
        self.transformer, self.scheduler = trunk, None
        self.image_processor, self.vae = ImageProcessor(), Vae()
        self.calls = []

    @staticmethod
    def _pack_latents(x):
        b, c, h, w = x.shape
        return x.view(b, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(b, (h // 2) * (w // 2), c * 4)

    def _unpack_latents(self, latents, height, width, vae_scale_factor):
        self.calls.append(("unpack", tuple(latents.shape)))
        b, n, c = latents.shape
        h, w = 2 * (height // (vae_scale_factor * 2)), 2 * (width // (vae_scale_factor * 2))
        return latents.view(b, h // 2, w // 2, c // 4, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(b, c // 4, h, w)

    def maybe_free_model_hooks(self):
        self.calls.append(("free",))

    def check_inputs(self, *a, **k):
        pass

    def _latents(self, image, dtype, generator, latents):
        image_latents = self._pack_latents(self.vae.encode_pixels(image)).to(dtype)
        if latents is None:
            latents = torch.randn(image_latents.shape, generator=generator).to(dtype)
        return latents, image_latents


class FluxKontextPipeline(_Base):
    def encode_prompt(self, prompt=None, prompt_2=None, prompt_embeds=None, pooled_prompt_embeds=None, device=None,
                      num_images_per_prompt=1, max_sequence_length=512, lora_scale=None):
        self.calls.append(("encode_prompt", prompt))
        if prompt_embeds is None:
            prompt_embeds, pooled_prompt_embeds = _pseudo(prompt, 1, 32, 256), _pseudo(prompt + "#", 1, 64)
        return prompt_embeds, pooled_prompt_embeds, torch.zeros(prompt_embeds.shape[1], 3)

    def prepare_latents(self, image, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        self.calls.append(("prepare_latents", height, width, num_channels_latents))
        return (*self._latents(image, dtype, generator, latents), None, None)


class Step1XEditPipeline(_Base):
    """v1p1: encode_image(image, width, height, device, n); encode_prompt(...) -> (embeds, mask, text_ids)."""
    V1P2 = False

    def __init__(self, trunk):
        super().__init__(trunk)
        trunk.connector = ToyConnector()              # shadows the stub's pass-through method; parameters = `connector.*`

    def encode_image(self, image, width, height, *rest):
        self.calls.append(("encode_image", len(rest)))
        h, w = image.shape[-2] // 16 * 16, image.shape[-1] // 16 * 16
        return self.image_processor.preprocess(image, h, w), image, dict(orig=tuple(image.shape[-2:])), w, h

    def _embeds(self, prompt):
        e = _pseudo(prompt, 1, 24, 256)
        mask = torch.ones(1, 24)
        mask[:, 20 - len(prompt or "") % 5:] = 0
        return e, mask

    def encode_prompt(self, ref_image=None, prompt=None, prompt_embeds=None, prompt_embeds_mask=None, device=None,
                      num_images_per_prompt=1):
        self.calls.append(("encode_prompt", prompt))
        if prompt_embeds is None:
            prompt_embeds, prompt_embeds_mask = self._embeds(prompt)
        return prompt_embeds, prompt_embeds_mask, torch.zeros(prompt_embeds.shape[1], 3)

    def prepare_latents(self, image, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        self.calls.append(("prepare_latents", height, width))
        return (*self._latents(image, dtype, generator, latents), None, None)

    def _output_process_image(self, out, img_info):
        self.calls.append(("output_process", img_info["orig"]))
        return out


class Step1XEditPipelineV1P2(Step1XEditPipeline):
    """v1p2: encode_image takes size_level; encode_prompt returns a record; prepare_latents takes no `latents`; the trunk has
    a `text_token_mapping` in front of the context embedder (Step1XEditV1P2/inplace.py:606-609)."""
    V1P2 = True

    class Record:
        pass

    def __init__(self, trunk):
        super().__init__(trunk)
        ttm = nn.Linear(32, 256, bias=False)
        with torch.no_grad():
            ttm.weight.copy_(0.02 * torch.randn(256, 32, generator=torch.Generator().manual_seed(5)))
        trunk.__dict__["text_token_mapping"] = ttm.to(torch.bfloat16)     # kept out of state_dict(): runs on the host

    def encode_prompt(self, ref_image=None, prompt=None, device=None, num_images_per_prompt=1):
        self.calls.append(("encode_prompt", prompt))
        r = self.Record()
        r.embedding, r.mask = self._embeds(prompt)
        r.txt_ids = torch.zeros(r.embedding.shape[1], 3)
        r.text_embeds, r.text_masks = _pseudo((prompt or "") + "t", 1, 24, 32), r.mask.to(torch.bfloat16)
        return r

    def prepare_latents(self, image, batch_size, num_channels_latents, height, width, dtype, device, generator):
        self.calls.append(("prepare_latents", height, width))
        return (*self._latents(image, dtype, generator, None), None, None)


class QwenImageEditPipeline(_Base):
    def encode_prompt(self, image=None, prompt=None, prompt_embeds=None, prompt_embeds_mask=None, device=None,
                      num_images_per_prompt=1, max_sequence_length=512):
        self.calls.append(("encode_prompt", prompt, len(image) if isinstance(image, list) else 1))
        if prompt_embeds is None:
            n = 20 + len(prompt or "") % 7                 # cond / uncond text lengths differ, like real prompts
            prompt_embeds, prompt_embeds_mask = _pseudo(prompt, 1, n, 256), torch.ones(1, n)
        return prompt_embeds, prompt_embeds_mask

    def prepare_latents(self, image, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        self.calls.append(("prepare_latents", height, width))
        images = image if isinstance(image, list) else [image]
        cond = torch.cat([self._pack_latents(self.vae.encode_pixels(im[:, :, 0])).to(dtype) for im in images], dim=1)
        if latents is None:
            h, w = height // 16, width // 16
            latents = torch.randn(1, h * w, 64, generator=generator).to(dtype)
        return latents, cond


class QwenImageEditPlusPipeline(QwenImageEditPipeline):
    pass



# factories for tools/edit_driver.py --pipeline_factory tests.host_standins:make_<family> (this image has no diffusers /
# checkpoints: the stand-ins exercise the hosted end-to-end protocol and its report)
def make_flux():
    return FluxKontextPipeline(stub_trunk("flux"))


def _with_connector(pipe):
    # the trunk class defines `connector` as a METHOD (the fixtures' pass-through stub), which wins over a registered
    # sub-module in attribute lookup: shadow it on the instance, like tests/test_hosted_pipelines.py does
    object.__setattr__(pipe.transformer, "connector", ToyConnector().to(torch.bfloat16))
    return pipe


def make_step1x():
    return _with_connector(Step1XEditPipeline(stub_trunk("step1x")))


def make_step1x_v1p2():
    return _with_connector(Step1XEditPipelineV1P2(stub_trunk("step1x")))


def make_qwen():
    return QwenImageEditPipeline(stub_trunk("qwen"))
