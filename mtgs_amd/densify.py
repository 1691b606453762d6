"""Densification statistics of MTGS Gaussian nodes in one launch per node.

`update_statistics` does what MTGSSceneModel.update_submodel_statistics followed by
VanillaGaussianSplattingModel.after_train do per step and node
(/root/reference/mtgs/scene_model/mtgs_scene_graph.py:1157-1183,
 /root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:448-474).  Under view-parallel data
parallelism every rank must take identical refine decisions: `mtgs_amd.dist.all_reduce_stats` sum-/max-reduces the
three arrays before `refinement_after` reads them.
"""
from __future__ import annotations

import torch
from torch import Tensor

from ._lib import call, ptr, require_gpu, stream_of


@torch.no_grad()
def update_statistics(xys_grad_norm: Tensor, vis_counts: Tensor, max_2Dsize: Tensor, radii: Tensor, xys_grad: Tensor,
                      width: int, height: int, start: int = 0) -> None:
    """In place, for the node whose n = xys_grad_norm.numel() Gaussians are rows [start, start + n) of the collected
    arrays: radii[(1,) N] int32 (info["radii"]), xys_grad[(1,) N, 2] (info["means2d"].absgrad or .grad).
    Visible (radii > 0):  xys_grad_norm += |xys_grad * (width, height) * 0.5|,  vis_counts += 1,
    max_2Dsize = max(max_2Dsize, radii)."""
    require_gpu(xys_grad_norm, vis_counts, max_2Dsize, radii, xys_grad)
    n = xys_grad_norm.numel()
    assert vis_counts.numel() == n and max_2Dsize.numel() == n
    for t in (xys_grad_norm, vis_counts, max_2Dsize):
        assert t.dtype == torch.float32 and t.is_contiguous(), "statistics must be contiguous float32"
    radii = radii.reshape(-1)
    grads = xys_grad.reshape(-1, 2)
    assert radii.numel() == grads.shape[0] and start >= 0 and start + n <= radii.numel(), (radii.shape, grads.shape, start, n)
    r = radii[start:start + n].contiguous()
    if r.dtype != torch.int32:
        r = r.to(torch.int32)
    g = grads[start:start + n].to(torch.float32).contiguous()
    call("mtgs_densify_stats", n, ptr(r), ptr(g), int(width), int(height), ptr(xys_grad_norm), ptr(vis_counts),
         ptr(max_2Dsize), stream_of(xys_grad_norm))
