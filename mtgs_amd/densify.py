"""Densification statistics of MTGS Gaussian nodes in one launch per node.

`update_statistics` does what MTGSSceneModel.update_submodel_statistics followed by
VanillaGaussianSplattingModel.after_train do per step and node
(/root/reference/mtgs/scene_model/mtgs_scene_graph.py:1157-1183,
 /root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:448-474).  Under view-parallel data
parallelism every rank must take identical refine decisions: `mtgs_amd.dist.all_reduce_stats` sum-/max-reduces the
three arrays before `refinement_after` reads them.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
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


_STATS_DESC = np.dtype([("n", "<i8"), ("first_block", "<i8"), ("start", "<i8"), ("xys_grad_norm", "<u8"), ("vis_counts", "<u8"),
                        ("max_2dsize", "<u8")], align=True)


@torch.no_grad()
def update_statistics_all(stats: Sequence[Tuple[Tensor, Tensor, Tensor]], radii: Tensor, xys_grad: Tensor, width: int,
                          height: int, starts: Optional[Sequence[int]] = None) -> None:
    """`update_statistics` for every node of the scene graph in ONE launch: stats[i] = (xys_grad_norm, vis_counts,
    max_2Dsize) of node i, whose Gaussians are rows [starts[i], starts[i] + n_i) of the collected arrays (default: the nodes
    follow each other in order, as get_gaussians concatenates them)."""
    from ._lib import load
    if not stats:
        return
    flat = [t for s in stats for t in s]
    require_gpu(radii, xys_grad, *flat)
    for t in flat:
        assert t.dtype == torch.float32 and t.is_contiguous(), "statistics must be contiguous float32"
    n = np.asarray([s[0].numel() for s in stats], dtype=np.int64)
    for s_, k in zip(stats, n):
        assert s_[1].numel() == k and s_[2].numel() == k
    st = np.cumsum(n) - n if starts is None else np.asarray(starts, dtype=np.int64)
    r = radii.reshape(-1)
    r = r if r.dtype == torch.int32 else r.to(torch.int32)
    r = r.contiguous()
    g = xys_grad.reshape(-1, 2).to(torch.float32).contiguous()
    assert r.numel() == g.shape[0] and (st >= 0).all() and int((st + n).max()) <= r.numel(), (r.shape, g.shape)
    if load().mtgs_stats_desc_bytes() != _STATS_DESC.itemsize:
        raise RuntimeError("mtgs_stats_desc layout mismatch between libmtgs_rast.so and mtgs_amd.densify")
    tab = np.zeros(len(stats), dtype=_STATS_DESC)
    nblk = (n + 255) // 256
    tab["n"], tab["start"], tab["first_block"] = n, st, np.cumsum(nblk) - nblk
    for j, k in enumerate(("xys_grad_norm", "vis_counts", "max_2dsize")):
        tab[k] = [s_[j].data_ptr() for s_ in stats]
    from .nodes import upload_table
    tab_dev = upload_table(tab, r.device)
    call("mtgs_densify_stats_batch", len(stats), ptr(tab_dev), int(nblk.sum()), ptr(r), ptr(g), int(width), int(height),
         stream_of(r))
