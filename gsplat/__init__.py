"""Drop-in `gsplat` package for MTGS on MI355X.

MTGS imports `from gsplat.rendering import rasterization`
(/root/reference/mtgs/scene_model/mtgs_scene_graph.py:20-23) and
`from gsplat.cuda._wrapper import spherical_harmonics`
(/root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:15-18).  Putting the
repository root on PYTHONPATH makes those imports resolve to mtgs_amd's HIP implementation.
The version string is the one MTGS pins (/root/reference/requirements.txt:12).
"""
from mtgs_amd.rendering import rasterization
from mtgs_amd.wrapper import (fully_fused_projection, isect_offset_encode, isect_tiles,
                              rasterize_to_pixels, spherical_harmonics)

__version__ = "1.4.0"
__all__ = ["rasterization", "spherical_harmonics", "fully_fused_projection", "isect_tiles",
           "isect_offset_encode", "rasterize_to_pixels"]
