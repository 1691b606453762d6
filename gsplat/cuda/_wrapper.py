"""gsplat.cuda._wrapper -> mtgs_amd.wrapper (see gsplat/__init__.py)."""
from mtgs_amd.wrapper import (fully_fused_projection, isect_offset_encode, isect_tiles,
                              rasterize_to_pixels, spherical_harmonics)

__all__ = ["spherical_harmonics", "fully_fused_projection", "isect_tiles", "isect_offset_encode",
           "rasterize_to_pixels"]
