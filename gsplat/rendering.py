"""gsplat.rendering -> mtgs_amd.rendering (see gsplat/__init__.py)."""
from mtgs_amd.rendering import rasterization

__all__ = ["rasterization"]
