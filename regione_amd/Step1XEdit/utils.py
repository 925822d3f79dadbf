"""Step1X-Edit region utilities: identical to the FLUX ones (the reference's two utils.py differ only in
the manager's class name, `diff RegionE/FluxKontext/utils.py RegionE/Step1XEdit/utils.py`)."""
from ..FluxKontext.utils import (FluxKontextManager, ids_gather, ids_scatter, remove_scattered_points,  # noqa: F401
                                 token_selector)


class Step1XEditManager(FluxKontextManager):
    """RegionE/Step1XEdit/utils.py:337-445."""
