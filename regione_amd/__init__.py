"""regione_amd - RegionE's region-aware denoising hot path, native on AMD MI355X (gfx950).

    from regione_amd import RegionEHelper          # drop-in for `from RegionE import RegionEHelper`
"""
from .tool.RegionE import RegionEHelper  # noqa: F401

__all__ = ["RegionEHelper"]
