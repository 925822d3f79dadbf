"""Plugin facade: same class, method names, kwargs, defaults, asserts and pipeline-class dispatch as
/root/reference/RegionE/tool/RegionE.py:1-51, on top of the HIP patch sets."""
import copy
import importlib

config = {
    "FluxKontextPipeline": {"num_inference_steps": 28, "warmup_step": 6, "post_step": 2, "refresh_step": "16",
                            "threshold": 0.93, "cache_threshold": 0.04, "erosion_dilation": True},
    "Step1XEditPipeline": {"num_inference_steps": 28, "warmup_step": 6, "post_step": 2, "refresh_step": "16",
                           "threshold": 0.88, "cache_threshold": 0.02, "erosion_dilation": True},
    "Step1XEditPipelineV1P2": {"num_inference_steps": 28, "warmup_step": 6, "post_step": 2, "refresh_step": "16",
                               "threshold": 0.88, "cache_threshold": 0.02, "erosion_dilation": True},
    "QwenImageEditPipeline": {"num_inference_steps": 28, "warmup_step": 6, "post_step": 2, "refresh_step": "16",
                              "threshold": 0.80, "cache_threshold": 0.03, "erosion_dilation": True},
    "QwenImageEditPlusPipeline": {"num_inference_steps": 28, "warmup_step": 6, "post_step": 2, "refresh_step": "16",
                                  "threshold": 0.80, "cache_threshold": 0.03, "erosion_dilation": True},
}

# pipeline class name -> patch-set module (RegionE/tool/RegionE.py:15-41 dispatch)
_FAMILY = {
    "FluxKontextPipeline": "FluxKontext",
    "Step1XEditPipeline": "Step1XEdit",
    "Step1XEditPipelineV1P2": "Step1XEditV1P2",
    "QwenImageEditPipeline": "QwenImageEdit",
    "QwenImageEditPlusPipeline": "QwenImageEditPlus",
}


def resample_gamma(table, num_inference_steps: int):
    """Linear re-sampling of a fitted 27-entry decay table to `num_inference_steps - 1` entries over the same
    normalised step axis, in fp16 like the shipped tables (extension; a table fitted at the target step count with
    tools/fit_gamma.py is the principled choice).  gamma_i scales the velocity over ONE step, so the per-step decay is
    re-expressed per unit of normalised time: g' = g ** (27 / (N - 1))."""
    import torch
    g = torch.as_tensor(table, dtype=torch.float64)
    n_old, n_new = g.numel(), num_inference_steps - 1
    x_new = torch.linspace(0, n_old - 1, n_new, dtype=torch.float64)
    lo = x_new.floor().long().clamp(max=n_old - 2)
    w = x_new - lo
    interp = g[lo] * (1 - w) + g[lo + 1] * w
    return (interp ** (n_old / n_new)).to(torch.float16)


class RegionEHelper(object):
    def __init__(self, pipeline=None):
        # Three kinds of `pipeline`:
        #   * a regione_amd.harness pipeline (latent-level, HIP transformer): patched directly;
        #   * a STOCK diffusers pipeline object (the reference's case, RegionE/README.md:85-113): enable() adopts its transformer
        #     weights onto the HIP engine once (regione_amd.adapters.attach, kept at pipeline._regione_engine), puts the patch
        #     set on that engine and swaps pipeline.__class__, so the user keeps calling pipeline(image=, prompt=);
        #     like the reference's, this constructor only looks at the class name;
        #   * a regione_amd.adapters.HostedPipeline wrapper (host + engine): the patch set goes on its engine.
        self.host = None
        if pipeline is not None:
            from .. import adapters
            if not adapters.is_engine_pipeline(pipeline) and not isinstance(pipeline, adapters.HostedPipeline):
                self.host = pipeline
                self.pipeline = pipeline
                self.name = adapters._host_name(pipeline)
            else:
                self.pipeline = getattr(pipeline, "_regione_engine", pipeline)
                self.name = self.pipeline.__class__.__name__
                if self.name.startswith("RegionE") and getattr(self.pipeline, "_regione_vanilla_class", None) is not None:
                    self.name = self.pipeline._regione_vanilla_class.__name__        # a second helper on an enabled pipeline
        # per-helper copy: the reference mutates the module-level dict in set_params (RegionE.py:43-51),
        # which leaks settings between helpers; same defaults, no leak.
        self.config = copy.deepcopy(config[self.name])

    def _family(self):
        try:
            return importlib.import_module(f"regione_amd.{_FAMILY[self.name]}.inplace")
        except ModuleNotFoundError as e:
            raise NotImplementedError(f"RegionE patch set for {self.name} is not built yet") from e

    def _engine(self):
        """The HIP pipeline the patch set goes on: `self.pipeline`, or - stock host pipeline - its adopted engine."""
        if self.host is None:
            return self.pipeline
        from .. import adapters
        return adapters.attach(self.host)

    def enable(self):
        assert self.pipeline is not None
        eng = self._family().warp_modules(self._engine(), **self.config)
        if self.host is not None:
            from .. import adapters
            adapters.swap_host_class(self.host)          # hook (1) on the user's own pipeline object
        else:
            self.pipeline = eng

    def disable(self):
        assert self.pipeline is not None
        eng = self._family().unwarp_modules(self._engine())
        if self.host is not None:
            from .. import adapters
            adapters.restore_host_class(self.host)
        else:
            self.pipeline = eng

    def shard_cfg_branches(self, pair):
        """Extension: run the 'cond' and 'uncond' forwards of one image on the two ranks of
        `pair` (regione_amd.dist.make_cfg_pair) and exchange noise_pred per computed step.  Only for the families whose
        reference patch sets keep one K/V cache per branch; `pair=None` switches it off.  Works enabled or disabled
        (the full-token loop shards the same way)."""
        if pair is not None and _FAMILY[self.name] not in ("QwenImageEdit", "QwenImageEditPlus", "Step1XEditV1P2"):
            raise NotImplementedError(f"{self.name} does not run CFG as two independent forwards (FLUX is guidance-distilled, "
                                      "Step1X-Edit v1p1 batches the branches): shard by image instead")
        self._engine()._cfg_pair = pair

    def set_params(self, num_inference_steps=28, warmup_step=None, post_step=None, refresh_step=None, threshold=None,
                   cache_threshold=None, erosion_dilation=None, strict_reference=None, gamma=None, gpu_eager_scalars=None):
        # reference: 28 steps only (tool/RegionE.py:44).  Extension: `gamma` = N-1 fitted decay factors (list / tensor,
        # e.g. from tools/fit_gamma.py) or "resample" (the family's 27-entry table linearly re-sampled to N-1 entries,
        # SURVEY.md section 8d config 5) lifts the restriction.
        assert num_inference_steps == 28 or gamma is not None, "num_inference_steps must be 28"
        if gamma is not None:
            if isinstance(gamma, str):
                assert gamma == "resample"
                gamma = resample_gamma(self._family().gamma, num_inference_steps)
            self.config['num_inference_steps'] = num_inference_steps
            self.config['gamma'] = gamma
        if warmup_step is not None: self.config['warmup_step'] = warmup_step
        if post_step is not None: self.config['post_step'] = post_step
        if refresh_step is not None: self.config['refresh_step'] = refresh_step
        if threshold is not None: self.config['threshold'] = threshold
        if cache_threshold is not None: self.config['cache_threshold'] = cache_threshold
        if erosion_dilation is not None: self.config['erosion_dilation'] = erosion_dilation
        if strict_reference is not None: self.config['strict_reference'] = strict_reference   # extension, see FluxKontext/inplace.py
        if gpu_eager_scalars is not None: self.config['gpu_eager_scalars'] = gpu_eager_scalars   # extension, see FluxKontext/utils.py:set_parameters
        print(f"RegionEHelper: set_params {self.config}")
