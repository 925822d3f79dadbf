"""Step1X-Edit v1p2 region utilities (RegionE/Step1XEditV1P2/utils.py): the FLUX ones plus the
negative-prompt text length (`neg_txt_length`, utils.py:445)."""
from ..FluxKontext.utils import (FluxKontextManager, ids_gather, ids_scatter, remove_scattered_points,  # noqa: F401
                                 token_selector)


class Step1XEditV1P2Manager(FluxKontextManager):
    """RegionE/Step1XEditV1P2/utils.py:347-457."""

    neg_txt_length = None

    def refresh(self, latents, image_latents, latent_ids, text_ids, neg_text_ids, patch_size=2, vae_scale_factor=8,
                height=None, width=None) -> None:
        super().refresh(latents, image_latents, latent_ids, text_ids, patch_size, vae_scale_factor, height, width)
        self.neg_txt_length = neg_text_ids.size(0)
