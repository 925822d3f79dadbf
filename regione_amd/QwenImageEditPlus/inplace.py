"""RegionE patch set for Qwen-Image-Edit-2509 ("Plus") - RegionE/QwenImageEditPlus/inplace.py.
Same kernels and protocol as QwenImageEdit; the deltas are its own gamma table (:47-50) and multi-image
conditioning (:229-244, :296-299), which lives in the shared code: `cond_shapes` (one rotary frame per condition image,
K/V rows = text + latent + every image) in harness/qwen.py + QwenImageEdit/inplace.py, the list handling in front of the
loop (per-image `calculate_dimensions`, VAE sizes, `encode_prompt(image=[...])`) in adapters._hosted_qwen."""
import torch

from ..QwenImageEdit import inplace as q
from ..QwenImageEdit.inplace import unwarp_modules  # noqa: F401

gamma = torch.tensor([1.0186, 1.0241, 1.0236, 1.0205, 1.0298, 1.0221, 1.0248, 1.0246, 1.0269,
                      1.0275, 1.0323, 1.0311, 1.0298, 1.0353, 1.0343, 1.0397, 1.0387, 1.0393,
                      1.0404, 1.0458, 1.0507, 1.0418, 1.0518, 1.0426, 1.0311, 1.0068, 0.7628], dtype=torch.float16)


class RegionEQwenImageEditPlusPipeline(q.RegionEQwenImageEditPipeline):
    gamma = gamma


def warp_modules(pipeline, **args):
    return q.warp_modules(pipeline, pipeline_cls=RegionEQwenImageEditPlusPipeline, **args)
