"""Qwen-Image-Edit region utilities (RegionE/QwenImageEdit/utils.py): the FLUX ones with 1-D latent ids
(`rope=True` branch of ids_gather, utils.py:259-261) and a manager whose refresh takes no text ids
(utils.py:422-448)."""
import math

import torch

from ..FluxKontext.utils import (FluxKontextManager, ids_gather, ids_scatter, remove_scattered_points,  # noqa: F401
                                 token_selector)


def calculate_dimensions(target_area, ratio):
    """utils.py:96-103."""
    width = math.sqrt(target_area * ratio)
    height = width / ratio
    width = round(width / 32) * 32
    height = round(height / 32) * 32
    return width, height, None


class QwenImageEditManager(FluxKontextManager):
    def refresh(self, latents, image_latents, latent_ids, patch_size=2, vae_scale_factor=8, height=None, width=None) -> None:
        super().refresh(latents, image_latents, latent_ids, torch.zeros(0, 3), patch_size, vae_scale_factor, height, width)
        self.txt_length = None          # per-branch text lengths arrive with each forward
