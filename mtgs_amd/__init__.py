"""mtgs_amd -- MI355X (gfx950) Gaussian-splatting rasterizer behind gsplat 1.4.0's Python API,
the hot path of OpenDriveLab/MTGS (rasterization() + spherical_harmonics()).

    from mtgs_amd import rasterization, spherical_harmonics      # or: `import gsplat` (shim package)

Importing this package does not load the HIP library or touch the GPU; the first operator call
does, and raises if libmtgs_rast.so has not been built (python -m mtgs_amd.build).

Beyond gsplat's surface: `mtgs_amd.graph_mode` / `mtgs_amd.graphs.GraphedIteration` (an iteration as one HIP graph launch),
`mtgs_amd.tight_lists` (opt-in shorter tile lists), `mtgs_amd.dist` (view-parallel data parallelism), `mtgs_amd.nodes` / `.loss`
/ `.densify` / `.optim` (the fused neighbours of the path).
"""
from .rendering import rasterization
from .wrapper import (exact_lists, fully_fused_projection, graph_mode, isect_offset_encode, isect_tiles, lists_are_tight,
                      rasterize_to_pixels, sh_lazy, sh_prefill, spherical_harmonics, tight_lists)

__version__ = "0.1.0"
__all__ = ["rasterization", "spherical_harmonics", "fully_fused_projection", "isect_tiles",
           "isect_offset_encode", "rasterize_to_pixels", "graph_mode", "exact_lists", "tight_lists", "lists_are_tight", "sh_prefill", "sh_lazy"]
