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


@torch.no_grad()
def update_statistics_rows(stats: Sequence[Tuple[Tensor, Tensor, Tensor]], radii: Tensor, grad_rows: Tensor, vis_ids: Tensor,
                           width: int, height: int, starts: Optional[Sequence[int]] = None, absgrad: bool = True,
                           n_vis: Optional[int] = None, n_vis_dev: Optional[Tensor] = None) -> None:
    """`update_statistics_all` from the COMPACT gradient rows of the one-node rasterization instead of a dense
    means2d gradient: grad_rows[n_vis, 16] (columns 0-1 the 2-D gradient, 2-3 its absolute-value sum) and vis_ids[n_vis]
    (flat index of each row's Gaussian), as the data-parallel exchange keeps them after the backward
    (mtgs_amd.dist.SparseGradExchange.grad_rows / .vis_ids) -- in that mode no dense gradient exists.  Same update,
    visible Gaussians only (mtgs_scene_graph.py:1157-1183, vanilla_gaussian_splatting.py:448-474).  n_vis_dev (device int64,
    mtgs_front_fwd's packed totals): the row count on the device (graph mode: the buffers are capacity-sized)."""
    from ._lib import load
    if not stats:
        return
    flat = [t for s_ in stats for t in s_]
    require_gpu(radii, grad_rows, vis_ids, *flat)
    for t in flat:
        assert t.dtype == torch.float32 and t.is_contiguous(), "statistics must be contiguous float32"
    n = np.asarray([s_[0].numel() for s_ in stats], dtype=np.int64)
    st = np.cumsum(n) - n if starts is None else np.asarray(starts, dtype=np.int64)
    assert (np.diff(st) >= 0).all(), "nodes in the order of the collected arrays"
    r = radii.reshape(-1)
    r = (r if r.dtype == torch.int32 else r.to(torch.int32)).contiguous()
    assert grad_rows.dim() == 2 and grad_rows.shape[1] >= 4 and grad_rows.dtype == torch.float32 and grad_rows.is_contiguous()
    assert vis_ids.dtype == torch.int32 and vis_ids.is_contiguous()
    nv = int(vis_ids.numel() if n_vis is None else n_vis)
    assert nv <= vis_ids.numel() and nv <= grad_rows.shape[0] and int((st + n).max()) <= r.numel()
    if load().mtgs_stats_desc_bytes() != _STATS_DESC.itemsize:
        raise RuntimeError("mtgs_stats_desc layout mismatch between libmtgs_rast.so and mtgs_amd.densify")
    tab = np.zeros(len(stats), dtype=_STATS_DESC)
    tab["n"], tab["start"] = n, st
    for j, k in enumerate(("xys_grad_norm", "vis_counts", "max_2dsize")):
        tab[k] = [s_[j].data_ptr() for s_ in stats]
    from .nodes import upload_table
    tab_dev = upload_table(tab, r.device)
    call("mtgs_densify_stats_rows", nv, ptr(vis_ids), ptr(grad_rows), int(grad_rows.shape[1]), 2 if absgrad else 0, ptr(r),
         len(stats), ptr(tab_dev), int(width), int(height), ptr(n_vis_dev), stream_of(r))


# ------------------------------------------------------------------------------------------------ refinement on device
import ctypes as _C
from dataclasses import dataclass
from typing import Dict


@dataclass
class RefineConfig:
    """The control fields of GaussianSplattingControlConfig that refinement_after reads
    (vanilla_gaussian_splatting.py:29-66; values of the shipped config/MTGS.py:59-71 with iteration_factor 1)."""
    refine_every: int = 100
    stop_split_at: int = 15000
    reset_alpha_every: int = 30
    continue_cull_post_densification: bool = False
    cull_alpha_thresh: float = 0.005
    cull_scale_thresh: float = 0.5
    densify_size_thresh: float = 0.2
    densify_grad_thresh: float = 0.001
    n_split_samples: int = 2
    clone_sample_means: bool = True
    stop_screen_size_at: int = 15000
    cull_screen_size: float = 150.0
    split_screen_size: float = 100.0


@torch.no_grad()
def refine_gaussians(params: Dict[str, Tensor], stats: Tuple[Tensor, Tensor, Tensor], cfg: RefineConfig, step: int, seed: int,
                     moments: Optional[Dict[str, Tuple[Tensor, Tensor]]] = None, extras: Optional[Dict[str, Tensor]] = None,
                     before_rows=None):
    """VanillaGaussianSplattingModel.refinement_after (densification branch, step < stop_split_at) for one node, on the
    device: split / duplicate / cull, every per-Gaussian tensor of `params` (means[N,3], scales[N,3] log, quats[N,4] wxyz,
    opacities[N,1] logit, + any other [N, ...] tensors: features_dc / features_rest / features_adapters ...) compacted and
    appended in the reference's order, the Adam moments `moments[name] = (exp_avg, exp_avg_sq)` following their rows
    (zeros for new Gaussians, dup_in_optim / remove_from_optim).  stats = (xys_grad_norm, vis_counts, max_2Dsize).

    Rank-deterministic: the split / clone samples come from a counter-based generator keyed by (seed, step, Gaussian index,
    sample) -- with all-reduced statistics (mtgs_amd.dist.all_reduce_stats) every rank of a data-parallel job produces
    bit-identical tensors, so N stays identical without a broadcast (the reference draws torch.randn per rank, :642, :687).
    ONE host synchronisation: the new N (tensor sizes).  Returns (new_params, new_moments | None, info) with
    info = {n_before, n_after, n_old_kept (old Gaussians kept), n_children (split children), n_dups, n_split (split parents; a device scalar),
    src_index, kind}.
    extras {name: [N, ...] tensor of a 4-byte dtype}: moved like a parameter (a child / duplicate takes its parent's row) and
    returned in info["extras"] -- the row-lazy optimizer's `last` stamps.  before_rows(parents bool [N]): called after the
    decisions are known and before any row is copied, with the Gaussians that get children or a duplicate -- a caller whose rows are
    lazy brings THOSE rows up to date there (FusedAdam.catch_up_rows) instead of flushing every row.
    The four GEOMETRY tensors (means, scales, quats, opacities) must be CURRENT when this function is called: the split / cull
    decisions and the children's positions are computed from them before the hook runs, so the hook is only good for tensors that
    are copied afterwards (colours, extras).  A caller whose geometry rows are lazy flushes them first."""
    from ._lib import call, ptr, stream_of
    means, scales, quats, opac = (params[k] for k in ("means", "scales", "quats", "opacities"))
    require_gpu(means, scales, quats, opac, *stats)
    N, dev = means.shape[0], means.device
    assert means.shape == (N, 3) and scales.shape == (N, 3) and quats.shape == (N, 4) and opac.numel() == N
    for t in params.values():
        assert t.dtype == torch.float32 and t.shape[0] == N, "every parameter is float32 with one row per Gaussian"
    gn, vc, m2 = (t.reshape(-1).to(torch.float32).contiguous() for t in stats)
    assert gn.numel() == N and vc.numel() == N and m2.numel() == N
    S = int(cfg.n_split_samples)
    th = (_C.c_float * 6)(cfg.densify_grad_thresh, cfg.densify_size_thresh, cfg.split_screen_size, cfg.cull_alpha_thresh,
                          cfg.cull_scale_thresh, cfg.cull_screen_size)
    cull_big = step > cfg.refine_every * cfg.reset_alpha_every
    opt = (_C.c_int * 5)(S, int(step < cfg.stop_screen_size_at), int(cull_big), int(cull_big and step < cfg.stop_screen_size_at),
                         int(cfg.clone_sample_means))
    st = stream_of(means)
    c = {k: params[k].detach().contiguous() for k in params}
    counts = torch.empty((2 + S, max(N, 1)), dtype=torch.int32, device=dev)
    flags = torch.empty(max(N, 1), dtype=torch.uint8, device=dev)
    call("mtgs_refine_classify", N, ptr(c["means"]), ptr(c["scales"]), ptr(c["quats"]), ptr(c["opacities"]), ptr(gn), ptr(vc), ptr(m2),
         th, opt, _C.c_uint64(seed & (2 ** 64 - 1)), int(step), ptr(counts), ptr(flags), st)
    counts = counts[:, :N]
    incl = torch.cumsum(counts, dim=1, dtype=torch.int64)
    pos = (incl - counts).contiguous()
    totals = incl[:, -1] if N > 0 else torch.zeros(2 + S, dtype=torch.int64, device=dev)
    bases = (torch.cumsum(totals, 0) - totals).contiguous()
    tot_h = totals.tolist()                      # the one host synchronisation: the new tensor sizes
    n_out = int(sum(tot_h))
    src_index = torch.empty(max(n_out, 1), dtype=torch.int32, device=dev)
    kind = torch.empty(max(n_out, 1), dtype=torch.uint8, device=dev)
    new = {"means": torch.empty((n_out, 3), dtype=torch.float32, device=dev),
           "scales": torch.empty((n_out, 3), dtype=torch.float32, device=dev)}
    call("mtgs_refine_apply", N, n_out, ptr(flags), ptr(pos), ptr(bases), ptr(c["means"]), ptr(c["scales"]), ptr(c["quats"]), th, opt,
         _C.c_uint64(seed & (2 ** 64 - 1)), int(step), ptr(src_index), ptr(kind), ptr(new["means"]), ptr(new["scales"]), st)

    if before_rows is not None and N > 0:
        assert 1 <= S <= 5, S                      # (flag byte: bit 0 old row kept, bits 1..S children, bit 1+S duplicate, bit 7 split parent)
        before_rows((flags[:N] & (((1 << (S + 1)) - 1) << 1)) != 0)      # (bits 1 .. 1 + S: a child or the duplicate is kept)

    def rows(src, zero_new):
        w = src.numel() // max(N, 1)
        dst = torch.empty((n_out,) + tuple(src.shape[1:]), dtype=torch.float32, device=dev)
        call("mtgs_refine_rows", n_out, w, ptr(src.contiguous()), ptr(src_index), ptr(kind), int(zero_new), ptr(dst), st)
        return dst

    for k, t in c.items():
        if k not in new:
            new[k] = rows(t, False)
    new_moments = None
    if moments is not None:
        new_moments = {k: (rows(a.detach(), True), rows(b.detach(), True)) for k, (a, b) in moments.items()}
    moved = {}
    for k, t in (extras or {}).items():
        assert t.shape[0] == N and t.element_size() == 4, "extras: [N, ...] tensors of a 4-byte dtype"
        moved[k] = rows(t.contiguous().view(torch.float32), False).view(t.dtype)      # (a bit-for-bit row copy)
    # (n_split stays a DEVICE scalar: children can be culled one by one, so the number of split parents does not follow from
    #  the totals, and a second host synchronisation is not worth a log line -- int(info["n_split"]) reads it when wanted)
    info = {"n_before": N, "n_after": n_out, "n_old_kept": tot_h[0], "n_children": sum(tot_h[1:1 + S]), "n_dups": tot_h[1 + S],
            "n_split": (flags[:N] >> 7).sum() if N else torch.zeros((), dtype=torch.int64, device=dev),
            "src_index": src_index[:n_out], "kind": kind[:n_out], "extras": moved}
    return new, new_moments, info


@torch.no_grad()
def reset_opacities(opacities: Tensor, cfg: RefineConfig, moments: Optional[Tuple[Tensor, Tensor]] = None) -> None:
    """The opacity reset of refinement_after (:555-572): clamp the logits at logit(2 cull_alpha_thresh), zero their moments."""
    v = 2.0 * cfg.cull_alpha_thresh
    opacities.clamp_(max=float(np.log(v / (1.0 - v))))
    if moments is not None:
        moments[0].zero_()
        moments[1].zero_()
