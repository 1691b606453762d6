"""gsplat.rendering.rasterization (v1.4.0) on MI355X: same signature, argument meaning, return
values and `meta` keys, for the option set MTGS drives.

Reference call site: /root/reference/mtgs/scene_model/mtgs_scene_graph.py:641-662 (kwargs built at
:641-659; `info["means2d"].retain_grad()`, `.absgrad`, `info["radii"]` consumed at :663-669 and
:1170-1178).  Options gsplat accepts that are not implemented here raise NotImplementedError naming
the option -- they never silently differ.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor
from typing_extensions import Literal

from .wrapper import (MAX_CHANNELS, SUPPORTED_CHANNELS, _LazySH, fused_rasterization, isect_offset_encode, isect_tiles,
                      projection_with_opacities, rasterize_to_pixels, rasterize_to_pixels_with_depth,
                      spherical_harmonics)


class _CameraPosition(torch.autograd.Function):
    @staticmethod
    def forward(ctx, viewmat):
        from ._lib import call, ptr, require_gpu, stream_of
        require_gpu(viewmat)
        V = viewmat.detach().to(torch.float32).contiguous()
        out = torch.empty(3, dtype=torch.float32, device=V.device)
        call("mtgs_campos_fwd", ptr(V), ptr(out), stream_of(V))
        ctx.save_for_backward(V)
        ctx.dtype = viewmat.dtype
        return out

    @staticmethod
    def backward(ctx, v_c):
        from ._lib import call, ptr, stream_of
        (V,) = ctx.saved_tensors
        v = v_c.to(torch.float32).contiguous()
        g = torch.empty((4, 4), dtype=torch.float32, device=V.device)
        call("mtgs_campos_bwd", ptr(V), ptr(v), ptr(g), stream_of(V))
        return g.to(ctx.dtype)


def camera_position(viewmat: Tensor) -> Tensor:
    """torch.inverse(viewmat)[:3, 3] of one [4,4] view matrix [A t; 0 1] -- the camera position gsplat's sh_degree path takes
    its view directions from (gsplat/rendering.py) -- as -A^-1 t in one launch per direction (the LU route and its backward
    are ~25 one-element launches).  Differentiable with respect to the view matrix."""
    assert viewmat.shape == (4, 4), viewmat.shape
    return _CameraPosition.apply(viewmat)


def rasterization(
    means: Tensor,  # [N, 3]
    quats: Tensor,  # [N, 4]
    scales: Tensor,  # [N, 3]
    opacities: Tensor,  # [N]
    colors: Tensor,  # [(C,) N, D] or [(C,) N, K, 3]
    viewmats: Tensor,  # [C, 4, 4]
    Ks: Tensor,  # [C, 3, 3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = True,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: Literal["RGB", "D", "ED", "RGB+D", "RGB+ED"] = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: Literal["classic", "antialiased"] = "classic",
    channel_chunk: int = 32,
    distributed: bool = False,
    camera_model: Literal["pinhole", "ortho", "fisheye"] = "pinhole",
    covars: Optional[Tensor] = None,
    color_source=None,
) -> Tuple[Tensor, Tensor, Dict]:
    """Rasterize a set of 3D Gaussians (N) to a batch of image planes (C).

    Returns (render_colors [C, height, width, X], render_alphas [C, height, width, 1], meta).

    color_source (extension, not in gsplat's signature; mtgs_amd.nodes.ColorSource from collect_gaussians(...,
    deferred_colors=True)): the RGB channels are evaluated from the nodes' SH coefficients for the VISIBLE Gaussians only;
    `colors` then holds the other colour channels ([N, DX]) or is None; with `color_source.camera_normals = camera_to_world[3,4]`
    MTGS's three camera-space normal channels follow the colours, likewise computed for the visible Gaussians only.
    """
    meta: Dict = {}
    N = means.shape[0]
    C = viewmats.shape[0]
    assert means.shape == (N, 3), means.shape
    assert opacities.shape == (N,), opacities.shape
    assert viewmats.shape == (C, 4, 4), viewmats.shape
    assert Ks.shape == (C, 3, 3), Ks.shape
    assert render_mode in ["RGB", "D", "ED", "RGB+D", "RGB+ED"], render_mode
    assert rasterize_mode in ["classic", "antialiased"], rasterize_mode

    # options of gsplat 1.4.0 that MTGS never enables (mtgs_scene_graph.py:641-659)
    if packed:
        raise NotImplementedError("rasterization: packed=True is not implemented (MTGS passes packed=False)")
    if sparse_grad:
        raise NotImplementedError("rasterization: sparse_grad=True is not implemented (requires packed=True)")
    if distributed:
        raise NotImplementedError("rasterization: distributed=True (Gaussian-sharded) is not implemented; "
                                  "use view-parallel data parallelism (mtgs_amd.dist)")
    if camera_model != "pinhole":
        raise NotImplementedError(f"rasterization: camera_model={camera_model!r} is not implemented")
    if covars is not None:
        raise NotImplementedError("rasterization: covars is not implemented (pass quats and scales)")
    if tile_size != 16:
        raise NotImplementedError(f"rasterization: tile_size={tile_size} is not implemented (only 16)")
    assert quats.shape == (N, 4), quats.shape
    assert scales.shape == (N, 3), scales.shape

    # MTGS's colours -- clamp(spherical_harmonics(...) + 0.5, 0, 1) -- still deferred (wrapper._LazySH): evaluated for the Gaussians the
    # projection finds visible, by the rasterization itself; whatever this call cannot take over is evaluated now, for all of them
    if type(colors) is _LazySH:
        src = None
        if (color_source is None and sh_degree is None and C == 1 and backgrounds is None and colors.dim() == 2
                and render_mode in ["RGB", "RGB+D", "RGB+ED"] and (colors.shape[1] + int(render_mode != "RGB")) in SUPPORTED_CHANNELS
                and colors.shape[1] + int(render_mode != "RGB") <= 8):
            src = colors.raster_source(N, width, height)
        if src is None:
            colors = colors._materialize()
        else:
            with_depth = render_mode != "RGB"
            extra = src[2].unsqueeze(0) if len(src) > 2 else None      # (channels behind the colours -- MTGS's normals --, as they are)
            render_colors, render_alphas, m = fused_rasterization(
                means, quats, scales, opacities, extra, viewmats, Ks, None, width, height, eps2d, near_plane, far_plane, radius_clip,
                rasterize_mode == "antialiased", with_depth, render_mode == "RGB+ED", absgrad, color_source=src[0],
                sh_source=(src[1], None))
            meta.update({"camera_ids": None, "gaussian_ids": None, "radii": m["radii"], "means2d": m["means2d"], "depths": m["depths"],
                         "conics": m["conics"], "opacities": m["opacities"], "tile_width": math.ceil(width / 16.0),
                         "tile_height": math.ceil(height / 16.0), "tiles_per_gauss": m["tiles_per_gauss"], "isect_ids": m["isect_ids"],
                         "flatten_ids": m["flatten_ids"], "isect_offsets": m["isect_offsets"], "width": width, "height": height,
                         "tile_size": tile_size, "n_cameras": C})
            for k in ("n_visible", "n_intersections", "overflow", "n_listed"):
                if k in m:
                    meta[k] = m[k]
            return render_colors, render_alphas, meta
    if color_source is not None:
        if sh_degree is not None or C != 1 or render_mode in ["D", "ED"] or backgrounds is not None:
            raise NotImplementedError("rasterization(color_source=...): one camera, an RGB render mode, no sh_degree / backgrounds")
        cols = None if colors is None else (colors if colors.dim() == 3 else colors.unsqueeze(0))
        render_colors, render_alphas, m = fused_rasterization(
            means, quats, scales, opacities, cols, viewmats, Ks, None, width, height, eps2d, near_plane, far_plane, radius_clip,
            rasterize_mode == "antialiased", render_mode in ["RGB+D", "RGB+ED"], render_mode == "RGB+ED", absgrad,
            color_source=color_source)
        meta.update({"camera_ids": None, "gaussian_ids": None, "radii": m["radii"], "means2d": m["means2d"], "depths": m["depths"],
                     "conics": m["conics"], "opacities": m["opacities"], "tile_width": math.ceil(width / 16.0),
                     "tile_height": math.ceil(height / 16.0), "tiles_per_gauss": m["tiles_per_gauss"], "isect_ids": m["isect_ids"],
                     "flatten_ids": m["flatten_ids"], "isect_offsets": m["isect_offsets"], "width": width, "height": height,
                     "tile_size": tile_size, "n_cameras": C})
        for k in ("n_visible", "n_intersections", "overflow", "n_listed"):
            if k in m:
                meta[k] = m[k]
        return render_colors, render_alphas, meta
    if sh_degree is None:
        # colors are post-activation values [N, D] or [C, N, D]
        assert (colors.dim() == 2 and colors.shape[0] == N) or (
            colors.dim() == 3 and colors.shape[:2] == (C, N)), colors.shape
    else:
        # colors are SH coefficients [N, K, 3] or [C, N, K, 3]
        assert (colors.dim() == 3 and colors.shape[0] == N and colors.shape[2] == 3) or (
            colors.dim() == 4 and colors.shape[:2] == (C, N) and colors.shape[3] == 3), colors.shape
        assert (sh_degree + 1) ** 2 <= colors.shape[-2], colors.shape

    tile_width = math.ceil(width / float(tile_size))
    tile_height = math.ceil(height / float(tile_size))
    with_depth = render_mode in ["RGB+D", "RGB+ED", "D", "ED"]
    expected = render_mode in ["ED", "RGB+ED"]
    camera_ids, gaussian_ids = None, None

    # Fast path -- everything MTGS drives: colours given per Gaussian, a channel count the compositing kernels
    # take in one launch.  Projection, binning and compositing run as ONE autograd node (same kernels and
    # results as the operator-by-operator composition below; see wrapper._FusedRasterization).
    if sh_degree is None:
        depth_only = render_mode in ["D", "ED"]
        n_blend = (0 if depth_only else colors.shape[-1]) + int(with_depth)
        if n_blend in SUPPORTED_CHANNELS and n_blend <= min(channel_chunk, MAX_CHANNELS):
            cols = None
            if not depth_only:
                cols = colors if colors.dim() == 3 else (colors.unsqueeze(0) if C == 1 else colors.expand(C, -1, -1))
            render_colors, render_alphas, m = fused_rasterization(
                means, quats, scales, opacities, cols, viewmats, Ks, None if depth_only else backgrounds, width,
                height, eps2d, near_plane, far_plane, radius_clip, rasterize_mode == "antialiased", with_depth,
                expected, absgrad)
            meta.update({"camera_ids": camera_ids, "gaussian_ids": gaussian_ids, "radii": m["radii"],
                         "means2d": m["means2d"], "depths": m["depths"], "conics": m["conics"],
                         "opacities": m["opacities"], "tile_width": tile_width, "tile_height": tile_height,
                         "tiles_per_gauss": m["tiles_per_gauss"], "isect_ids": m["isect_ids"],
                         "flatten_ids": m["flatten_ids"], "isect_offsets": m["isect_offsets"], "width": width,
                         "height": height, "tile_size": tile_size, "n_cameras": C})
            for k in ("n_visible", "n_intersections", "overflow", "n_listed"):   # only inside mtgs_amd.graph_mode
                if k in m:
                    meta[k] = m[k]
            return render_colors, render_alphas, meta

    # gsplat's own call style (SH coefficients + sh_degree) through the same one-node path: SH + clamp_min evaluated for the
    # Gaussians the projection found visible (gsplat masks SH with radii > 0 too), straight into their records; the
    # coefficient gradient comes back through compact rows and is expanded once, the view directions stay differentiable.
    if (sh_degree is not None and C == 1 and colors.dim() == 3 and colors.shape[1] <= 16 and sh_degree <= 3
            and backgrounds is None and render_mode in ["RGB", "RGB+D", "RGB+ED"] and colors.dtype == torch.float32
            and (3 + int(with_depth)) in SUPPORTED_CHANNELS and N > 0):
        from .nodes import sh_coefficient_source
        campos = camera_position(viewmats[0])                      # = torch.inverse(viewmats)[0, :3, 3], differentiable, one launch
        cs = sh_coefficient_source(colors, sh_degree, campos)
        render_colors, render_alphas, m = fused_rasterization(
            means, quats, scales, opacities, None, viewmats, Ks, None, width, height, eps2d, near_plane, far_plane, radius_clip,
            rasterize_mode == "antialiased", with_depth, expected, absgrad, color_source=cs, sh_source=(colors, campos))
        meta.update({"camera_ids": camera_ids, "gaussian_ids": gaussian_ids, "radii": m["radii"], "means2d": m["means2d"],
                     "depths": m["depths"], "conics": m["conics"], "opacities": m["opacities"], "tile_width": tile_width,
                     "tile_height": tile_height, "tiles_per_gauss": m["tiles_per_gauss"], "isect_ids": m["isect_ids"],
                     "flatten_ids": m["flatten_ids"], "isect_offsets": m["isect_offsets"], "width": width, "height": height,
                     "tile_size": tile_size, "n_cameras": C})
        for k in ("n_visible", "n_intersections", "overflow", "n_listed"):
            if k in m:
                meta[k] = m[k]
        return render_colors, render_alphas, meta

    # (1) projection, fused with `opacities.repeat(C, 1) [* compensations]`
    radii, means2d, depths, conics, compensations, opacities = projection_with_opacities(
        means, quats, scales, viewmats, Ks, opacities, width, height, eps2d=eps2d, near_plane=near_plane,
        far_plane=far_plane, radius_clip=radius_clip, calc_compensations=(rasterize_mode == "antialiased"))

    meta.update({"camera_ids": camera_ids, "gaussian_ids": gaussian_ids, "radii": radii,
                 "means2d": means2d, "depths": depths, "conics": conics, "opacities": opacities})

    # (2) colours [C, N, D]
    if sh_degree is None:
        if colors.dim() == 2:
            # (C == 1, MTGS's case: a plain view -- expand()'s backward would launch a reduction over C)
            colors = colors.unsqueeze(0) if C == 1 else colors.expand(C, -1, -1)
    else:
        camtoworlds = torch.inverse(viewmats)  # [C, 4, 4]
        dirs = means[None, :, :] - camtoworlds[:, None, :3, 3]  # [C, N, 3]
        masks = radii > 0
        shs = colors.expand(C, -1, -1, -1) if colors.dim() == 3 else colors
        colors = spherical_harmonics(sh_degree, dirs, shs, masks=masks)  # [C, N, 3]
        colors = torch.clamp_min(colors + 0.5, 0.0)

    # (3) tile binning
    tiles_per_gauss, isect_ids, flatten_ids = isect_tiles(
        means2d, radii, depths, tile_size, tile_width, tile_height, packed=False, n_cameras=C,
        camera_ids=camera_ids, gaussian_ids=gaussian_ids)
    isect_offsets = isect_offset_encode(isect_ids, C, tile_width, tile_height)

    meta.update({"tile_width": tile_width, "tile_height": tile_height,
                 "tiles_per_gauss": tiles_per_gauss, "isect_ids": isect_ids,
                 "flatten_ids": flatten_ids, "isect_offsets": isect_offsets, "width": width,
                 "height": height, "tile_size": tile_size, "n_cameras": C})

    # (4) compositing.  The depth channel of the "+D"/"+ED"/"D"/"ED" modes and the expected-depth
    # normalisation are blended in the same pass (gsplat: torch.cat + rasterize_to_pixels + division).
    if render_mode in ["D", "ED"]:
        colors, backgrounds = None, None  # gsplat: colors = depths[..., None], backgrounds = zeros
    n_channels = (0 if colors is None else colors.shape[-1]) + int(with_depth)
    chunk = min(channel_chunk, MAX_CHANNELS)
    if n_channels <= chunk:
        if with_depth:
            render_colors, render_alphas = rasterize_to_pixels_with_depth(
                means2d, conics, colors, opacities, depths, expected, width, height, tile_size, isect_offsets,
                flatten_ids, backgrounds=backgrounds, absgrad=absgrad)
        else:
            render_colors, render_alphas = rasterize_to_pixels(
                means2d, conics, colors, opacities, width, height, tile_size, isect_offsets, flatten_ids,
                backgrounds=backgrounds, packed=False, absgrad=absgrad)
    else:
        # many channels: gsplat's own composition, in channel chunks
        if with_depth:
            colors = torch.cat((colors, depths[..., None]), dim=-1)
            if backgrounds is not None:
                backgrounds = torch.cat([backgrounds, torch.zeros(C, 1, device=backgrounds.device)], dim=-1)
        n_chunks = (colors.shape[-1] + chunk - 1) // chunk
        render_colors, render_alphas = [], []
        for i in range(n_chunks):
            colors_chunk = colors[..., i * chunk:(i + 1) * chunk]
            bg_chunk = backgrounds[..., i * chunk:(i + 1) * chunk] if backgrounds is not None else None
            rc, ra = rasterize_to_pixels(means2d, conics, colors_chunk, opacities, width, height,
                                         tile_size, isect_offsets, flatten_ids, backgrounds=bg_chunk,
                                         packed=False, absgrad=absgrad)
            render_colors.append(rc)
            render_alphas.append(ra)
        render_colors = torch.cat(render_colors, dim=-1)
        render_alphas = render_alphas[0]
        if expected:
            render_colors = torch.cat(
                [render_colors[..., :-1], render_colors[..., -1:] / render_alphas.clamp(min=1e-10)], dim=-1)

    return render_colors, render_alphas, meta
