"""Fused per-node activations for MTGS Gaussian nodes (the caller side of the
rasterization path).

`node_gaussians(...)` returns what `VanillaGaussianSplattingModel.get_gaussians` returns
(/root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:299-341: means, exp(scales),
normalised quats, sigmoid(opacities), rgbs = clamp(SH(n, dirs, cat(features_dc, features_rest)) + 0.5, 0, 1)),
computed by ONE HIP kernel per direction (csrc/node.hip) instead of ~12 PyTorch launches per direction: the
coefficients are read where they are (no `torch.cat` copy of [N,16,3], no dirs / clamp temporaries) and the backward
writes the gradients of features_dc / features_rest directly.  For MultiColorGaussianSplattingModel
(multi_color_gaussian_splatting.py:77-101) pass `features_dc_add=features_adapters[:, t]` and
`features_rest=features_rest[:, t]`: strided views are read in place.

No CPU / PyTorch fallback (mtgs_amd._lib raises without the HIP library).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
from torch import Tensor

from ._lib import call, host_i64, ptr, require_gpu, stream_of


def _rows(t: Optional[Tensor], width: int):
    """(tensor, row stride in floats) of a [N, ...] tensor whose rows are `width` contiguous floats."""
    if t is None:
        return None, width
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32 tensor, got {t.dtype}")
    if t.is_contiguous():
        return t, width
    N = t.shape[0]
    inner_ok = t[0].is_contiguous() if N > 0 else True
    if N > 1 and inner_ok and t.stride(0) >= width:
        return t, t.stride(0)
    return t.contiguous(), width


def _launch_fwd(means, scales_raw, quats_raw, opacities_raw, features_dc, features_dc_add, features_rest, cam, degree, use_sh,
                trav, scales, quats, opacities, rgbs, mask, pose=None, means_out=None):
    """One mtgs_node_fwd launch into the given (contiguous) output tensors.  Returns what the backward needs.
    pose: float32[7] (instance quaternion wxyz | translation) of a rigid node, or None; means_out: [N,3] or None."""
    N = means.shape[0]
    Kr = features_rest.shape[-2]
    T = features_rest.shape[1] if trav >= 0 else 0
    if trav >= 0:   # FULL per-traversal parameters: slice `trav` is read in place
        features_rest = features_rest[:, trav]
        features_dc_add = None if features_dc_add is None else features_dc_add[:, trav]
    means_c, scales_c, quats_c = means.detach().contiguous(), scales_raw.contiguous(), quats_raw.contiguous()
    opac_c = opacities_raw.reshape(N).contiguous()
    dc, s_dc = _rows(features_dc, 3)
    dca, s_dca = _rows(features_dc_add, 3)
    rest, s_rest = _rows(features_rest, Kr * 3)
    call("mtgs_node_fwd", N, Kr, int(degree), int(use_sh), ptr(means_c), ptr(scales_c), ptr(quats_c), ptr(opac_c), ptr(dc),
         ptr(dca), ptr(rest), host_i64([s_dc, s_dca, s_rest]), ptr(cam), ptr(scales), ptr(quats), ptr(opacities),
         ptr(rgbs), ptr(mask), ptr(pose), ptr(means_out), stream_of(means))
    return means_c, quats_c, (N, Kr, int(degree), int(use_sh), opacities_raw.shape, features_dc_add is not None, T, int(trav))


def _launch_bwd(means_c, quats_c, cam, scales, opacities, rgbs, mask, dims, v_scales, v_quats, v_opacities, v_rgbs, pose=None,
                v_means=None, want_means=False):
    """One mtgs_node_bwd launch.  Returns (g_scales, g_quats, g_opacities, g_dc, g_dc_add, g_rest, g_means, g_pose);
    g_means / g_pose are None unless a rigid node's pose is given (g_means then = R^T v_means, g_pose float32[7])."""
    N, Kr, degree, use_sh, opac_shape, has_add, T, trav = dims
    dev = means_c.device
    z = lambda g, shape: (torch.zeros(shape, dtype=torch.float32, device=dev) if g is None else g.to(torch.float32).contiguous())
    v_scales, v_quats = z(v_scales, (N, 3)), z(v_quats, (N, 4))
    v_opacities, v_rgbs = z(v_opacities, (N,)), z(v_rgbs, (N, 3))
    g_scales = torch.empty((N, 3), dtype=torch.float32, device=dev)
    g_quats = torch.empty((N, 4), dtype=torch.float32, device=dev)
    g_opac = torch.empty((N,), dtype=torch.float32, device=dev)
    g_dc = torch.empty((N, 3), dtype=torch.float32, device=dev)
    if T:
        g_rest = torch.empty((N, T, Kr, 3), dtype=torch.float32, device=dev)
        g_add = torch.empty((N, T, 3), dtype=torch.float32, device=dev) if has_add else None
    else:
        g_rest = torch.empty((N, Kr, 3), dtype=torch.float32, device=dev)
        g_add = None
    g_means = g_pose = v_means_c = None
    if pose is not None:
        v_means_c = None if v_means is None else v_means.to(torch.float32).contiguous()
        g_means = torch.empty((N, 3), dtype=torch.float32, device=dev) if want_means else None
        g_pose = torch.zeros(7, dtype=torch.float32, device=dev)   # accumulated with atomics
    call("mtgs_node_bwd", N, Kr, degree, use_sh, ptr(means_c), ptr(quats_c), ptr(cam), ptr(scales), ptr(opacities),
         ptr(rgbs), ptr(mask), ptr(v_scales), ptr(v_quats), ptr(v_opacities), ptr(v_rgbs), ptr(g_scales), ptr(g_quats),
         ptr(g_opac), ptr(g_dc), ptr(g_rest), ptr(g_add), T, max(trav, 0), ptr(pose), ptr(v_means_c), ptr(g_means), ptr(g_pose),
         stream_of(means_c))
    if not T:
        g_add = g_dc if has_add else None
    return g_scales, g_quats, g_opac.reshape(opac_shape), g_dc, g_add, g_rest, g_means, g_pose


class _NodeActivations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, scales_raw, quats_raw, opacities_raw, features_dc, features_dc_add, features_rest, cam_pos,
                degree, use_sh, trav, inst_quat, inst_trans):
        require_gpu(means, scales_raw, quats_raw, opacities_raw, features_dc, features_dc_add, features_rest, cam_pos, inst_quat,
                    inst_trans)
        N, dev = means.shape[0], means.device
        cam = cam_pos.detach().reshape(3).to(torch.float32).contiguous()
        pose = None if inst_quat is None else torch.cat([inst_quat.detach().reshape(4), inst_trans.detach().reshape(3)]).to(torch.float32)
        scales = torch.empty((N, 3), dtype=torch.float32, device=dev)
        quats = torch.empty((N, 4), dtype=torch.float32, device=dev)
        opacities = torch.empty((N,), dtype=torch.float32, device=dev)
        rgbs = torch.empty((N, 3), dtype=torch.float32, device=dev)
        mask = torch.empty((N,), dtype=torch.uint8, device=dev)
        means_g = torch.empty((N, 3), dtype=torch.float32, device=dev) if pose is not None else None
        means_c, quats_c, ctx.dims = _launch_fwd(means, scales_raw, quats_raw, opacities_raw, features_dc, features_dc_add,
                                                 features_rest, cam, degree, use_sh, trav, scales, quats, opacities, rgbs, mask,
                                                 pose, means_g)
        ctx.save_for_backward(means_c, quats_c, cam, scales, opacities, rgbs, mask, pose)
        if means_g is None:   # static node: the means pass through unchanged (returned by the caller)
            means_g = torch.empty(0, device=dev)
            ctx.mark_non_differentiable(means_g)
        return scales, quats, opacities, rgbs, means_g

    @staticmethod
    def backward(ctx, v_scales, v_quats, v_opacities, v_rgbs, v_means_g):
        means_c, quats_c, cam, scales, opacities, rgbs, mask, pose = ctx.saved_tensors
        g_scales, g_quats, g_opac, g_dc, g_add, g_rest, g_means, g_pose = _launch_bwd(
            means_c, quats_c, cam, scales, opacities, rgbs, mask, ctx.dims, v_scales, v_quats, v_opacities, v_rgbs, pose, v_means_g,
            want_means=ctx.needs_input_grad[0])
        g_q = g_t = None
        if g_pose is not None:
            g_q, g_t = g_pose[:4], g_pose[4:]
        return (g_means, g_scales, g_quats, g_opac, g_dc, g_add, g_rest, None, None, None, None, g_q, g_t)


_NODE_KEYS = ("means", "scales", "quats", "opacities", "features_dc", "features_dc_add", "features_rest", "instance_quat",
              "instance_trans")
_NK = len(_NODE_KEYS)


# mtgs_node_desc of include/mtgs_rast.h (checked against mtgs_node_desc_bytes() on first use)
_DESC = np.dtype([(k, "<i8") for k in ("n", "first_block", "start")]
                 + [(k, "<u8") for k in ("means", "scales_raw", "quats_raw", "opacities_raw", "features_dc", "features_dc_add",
                                         "features_rest")]
                 + [(k, "<i8") for k in ("dc_stride", "dc_add_stride", "rest_stride")] + [("pose", "<u8"), ("pose_trans", "<u8")]
                 + [(k, "<i4") for k in ("k_rest", "use_sh", "n_traversals", "traversal", "pose_normalize", "skip_colors")]
                 + [(k, "<u8") for k in ("scales", "quats", "opacities", "rgbs", "clamp_mask", "means_out", "v_scales", "v_quats",
                                         "v_opacities", "v_rgbs", "v_means", "g_scales_raw", "g_quats_raw", "g_opacities_raw",
                                         "g_features_dc", "g_features_rest", "g_features_dc_add", "g_means", "g_pose", "g_pose_quat_row",
                                         "g_pose_trans_row", "frame_dev")], align=True)
_desc_checked = False


def _upload(tab: np.ndarray, dev) -> Tensor:
    global _desc_checked
    if not _desc_checked:
        from ._lib import load
        want = load().mtgs_node_desc_bytes()
        if want != _DESC.itemsize:
            raise RuntimeError(f"mtgs_node_desc is {want} bytes in libmtgs_rast.so, {_DESC.itemsize} in mtgs_amd.nodes")
        _desc_checked = True
    return upload_table(tab, dev)


def upload_table(tab: np.ndarray, dev) -> Tensor:
    """Host table -> device, WITHOUT blocking the host: a copy from pageable memory waits for everything already enqueued
    on the stream (the host could no longer run ahead of the GPU: +0.2 ms per iteration at MTGS's training size).  The
    staging buffer comes from PyTorch's pinned-memory cache, which does not hand it out again before the copy has run."""
    from .wrapper import staging_buffer
    raw = tab.view(np.uint8).reshape(-1)
    staged = staging_buffer(raw.size)
    staged.numpy()[:] = raw
    return staged.to(dev, non_blocking=True)


class ColorSource:
    """Visibility-first colours (collect_gaussians(..., deferred_colors=True) -> rasterization(..., color_source=...)).

    Holds the node table (csrc/viscolor.hip reads the SH coefficients of the VISIBLE Gaussians through it), and after the
    backward of the rasterization the coefficient gradient in ROW form: `rows[n_vis, 48]` (d L / d coefficient k, channel c at
    3 k + c) and `row_of[N]` (int32: a Gaussian's row, or -1 when the frame did not see it).  The colour parameters get no
    autograd gradient -- no dense [N, (T,) K, 3] tensor with zeros for ~85 % of the rows and for every other traversal is
    written; `apply_to(optimizer)` hands the rows to mtgs_amd.optim.FusedAdam, which gives Gaussians without a row the exact
    zero-gradient update.  `dense_gradients()` expands them (slow path: tests, other optimizers)."""

    def __init__(self, table, n_nodes, degree, cam, keep, node_params):
        self.table, self.n_nodes, self.degree, self.cam, self._keep = table, int(n_nodes), int(degree), cam, keep
        self.node_params = node_params     # per node: (start, n, features_dc, features_adapters | None, features_rest, traversal | None)
        self.rows = self.row_of = None
        self.autograd, self.width = False, 48   # (sh_coefficient_source: dense coefficient gradient + differentiable directions)
        self.dirs = None             # (sh_direction_source: the caller's own view directions [N, 3], no gradient, instead of means - cam)
        self.camera_normals = None   # camera_to_world [3,4] (device): the rasterization adds MTGS's three camera-space normal channels
        #                              (mtgs_scene_graph.py:526-545, 636-638) after the colours, computed for the VISIBLE Gaussians only
        self.node_geometry = None    # per node: (means, scales, quats, opacities leaves, is a rigid node)
        self.geometry_rows = False   # True (static nodes only): the rasterization's backward returns NO gradient for means / quats /
        #                              scales / opacities; the projection backward's per-visible rows go through mtgs_node_bwd_rows and
        #                              apply_to() hands them to the optimizer as row gradients of the nodes' raw geometry parameters
        self.geo_ws = None
        self.want_grad_rows = False  # True: the backward leaves the compact gradient rows in .grad_rows / .grad_row_ids / .grad_row_count
        #                              (columns 0-1 the 2-D gradient, 2-3 absgrad: densify.update_statistics_rows) and writes NO dense absgrad
        self.grad_rows = self.grad_row_ids = self.grad_row_count = None
        self.skip_zero_rows = True   # row-lazy parameters: visible rows whose colour gradient is exactly zero (occluded Gaussians: most
        #                              of the frustum-visible ones) are left lazy by the step instead of being stepped with zeros
        self.optimizer = None   # a FusedAdam with row-lazy colour parameters: prepare() peeks the visible rows for the colour kernel
        self.caught = None
        self.touch_first = False  # True (opt-in): the rasterization bins first and flags the Gaussians the frame composites FROM (one
        #                           pass of the compositing DECISIONS, mtgs_blend_touch_packed: costs about one forward); the
        #                           optimizer's peek, the SH evaluation and the normals then leave the others -- 90 % of the visible
        #                           ones in an opaque scene -- alone.  Exact (tests/test_gpu_nodes.py).  Pays when the peek of every
        #                           visible row costs more than a forward pass (a wash at 960x540 / 2M)
        self.row_flags = None     # the flags of the last frame (uint8 [cap_vis]) or None

    COEF_STRIDE = 52            # floats of a compact coefficient row: dc 3 | dc_add (adapter) 3 | rest 45 | pad

    def prepare(self, vis_rank: Tensor, cap_rows: int, vis_ids: Optional[Tensor] = None, totals: Optional[Tensor] = None,
                row_flags: Optional[Tensor] = None) -> Optional[Tensor]:
        """Called by the rasterization between its front end and the colour kernel.  With a row-lazy optimizer attached
        (`self.optimizer`, FusedAdam.set_row_lazy): the UP-TO-DATE coefficient rows of the Gaussians the frame sees, as one
        compact buffer [cap_rows, COEF_STRIDE] that the colour kernel reads instead of the parameters (FusedAdam.peek_rows:
        nothing in the optimizer changes) and that apply_to() hands back to the step.  vis_rank int32 [N]: row or -1.
        Returns the buffer, or None (no optimizer: the colour kernel reads the parameters in place)."""
        self.caught = self.row_ids = None
        self.row_flags = row_flags
        if self.optimizer is None:
            if any(isinstance(np_[5], Tensor) for np_ in self.node_params):
                raise RuntimeError("ColorSource: a device traversal_index needs the row-lazy optimizer's peek (ColorSource.optimizer)")
            return None
        out = torch.empty((max(int(cap_rows), 1), self.COEF_STRIDE), dtype=torch.float32, device=vis_rank.device)
        items = []
        for start, n, dc, adapters, rest, trav in self.node_params:
            ro = vis_rank[start:start + n]
            rid = None if vis_ids is None else (vis_ids, start, totals)     # (the frame's id list: rows straight from it)
            items.append((dc, ro, None, 0, rid))
            if adapters is not None:
                items.append((adapters, ro, trav if adapters.dim() == 3 else None, 3, rid))
            if rest.shape[-2] > 0:
                items.append((rest, ro, trav if rest.dim() == 4 else None, 6, rid))
        self.optimizer.peek_rows(items, out, row_flags=row_flags if vis_ids is not None else None)
        self.row_ids = None if vis_ids is None else (vis_ids, totals)
        self.caught = out
        return out

    def apply_to(self, optimizer) -> None:
        """optimizer.set_row_gradient(...) for every colour parameter of every node (call between backward() and step())."""
        assert self.rows is not None, "backward() of the rasterization first"
        if self.geometry_rows and self.geo_ws is not None:
            # the geometry half: rows of raw-parameter gradients of the visible Gaussians (mtgs_node_bwd_rows), one launch
            ws, ids, totals = self.geo_ws
            prow = torch.empty_like(ws)
            call("mtgs_node_bwd_rows", self.n_nodes, ptr(self.table), ptr(ids), ptr(totals), ws.shape[0], ptr(ws), 12, ptr(prow), stream_of(ws))
            for (start, n, *_), (mn, sc, qt, op, _rigid) in zip(self.node_params, self.node_geometry):
                ro = self.row_of[start:start + n]
                for p_, col in ((mn, 0), (sc, 3), (qt, 6), (op, 10)):
                    if p_.requires_grad:
                        optimizer.set_row_gradient(p_, prow, ro, col)
        c = getattr(self, "caught", None)
        if c is not None and (optimizer is not self.optimizer or c.shape[0] < self.rows.shape[0]):
            c = None     # (the peeked rows belong to the optimizer that made them, numbered like this frame's gradient rows)
        is_lazy = getattr(optimizer, "is_row_lazy", lambda p: False)
        ids = getattr(self, "row_ids", None)

        def ck(p, col, start):
            kw = {}
            if is_lazy(p) and self.skip_zero_rows:
                kw["zero_probe"] = 0      # rows[:, 0:3] = C0 * v_rgb (x clamp mask): zero iff the whole coefficient gradient is
            if is_lazy(p):
                if c is not None:
                    kw["caught"] = (c, col)
                if ids is not None:
                    kw["row_ids"] = (ids[0], start, ids[1])
                    if self.row_flags is not None and optimizer is self.optimizer:
                        kw["row_flags"] = self.row_flags
            return kw
        for start, n, dc, adapters, rest, trav in self.node_params:
            ro = self.row_of[start:start + n]
            if dc.requires_grad:
                optimizer.set_row_gradient(dc, self.rows, ro, 0, **ck(dc, 0, start))
            if adapters is not None and adapters.requires_grad:
                optimizer.set_row_gradient(adapters, self.rows, ro, 0, slice_index=trav if adapters.dim() == 3 else None,
                                           **ck(adapters, 3, start))
            if rest.requires_grad and rest.shape[-2] > 0:
                optimizer.set_row_gradient(rest, self.rows, ro, 3, slice_index=trav if rest.dim() == 4 else None, **ck(rest, 6, start))

    def dense_gradients(self):
        """[(features_dc grad, features_adapters grad | None, features_rest grad)] per node, dense (zeros where no row)."""
        assert self.rows is not None, "backward() of the rasterization first"
        out = []
        for start, n, dc, adapters, rest, trav in self.node_params:
            ro = self.row_of[start:start + n].long()
            vis = ro >= 0
            r = self.rows[ro[vis]]
            Kr = rest.shape[-2]
            g_dc = torch.zeros(n, 3, device=r.device)
            g_dc[vis] = r[:, 0:3]
            g_rest = torch.zeros_like(rest)
            g_ad = None
            if rest.dim() == 4:
                g_rest[vis, trav] = r[:, 3:3 + 3 * Kr].view(-1, Kr, 3)
            else:
                g_rest[vis] = r[:, 3:3 + 3 * Kr].view(-1, Kr, 3)
            if adapters is not None:
                g_ad = torch.zeros_like(adapters)
                if adapters.dim() == 3:
                    g_ad[vis, trav] = r[:, 0:3]
                else:
                    g_ad[vis] = r[:, 0:3]
            out.append((g_dc, g_ad, g_rest))
        return out


def sh_coefficient_source(coeffs: Tensor, sh_degree: int, campos: Tensor) -> ColorSource:
    """ColorSource for gsplat's own call style `rasterization(colors=coeffs[N,K,3], sh_degree=n)` (gsplat/rendering.py: SH masked
    with radii > 0, clamp_min(. + 0.5, 0), differentiable view directions): ONE descriptor over the coefficient tensor."""
    N, K = coeffs.shape[0], coeffs.shape[1]
    assert coeffs.dim() == 3 and coeffs.shape[2] == 3 and coeffs.dtype == torch.float32 and K <= 16 and sh_degree <= 3
    c = coeffs.detach().contiguous()
    tab = np.zeros(1, dtype=_DESC)
    tab["n"], tab["start"], tab["first_block"] = N, 0, 0
    tab["features_dc"], tab["features_rest"] = c.data_ptr(), c.data_ptr() + 12
    tab["dc_stride"] = tab["rest_stride"] = K * 3
    tab["dc_add_stride"] = 3
    tab["k_rest"], tab["use_sh"] = K - 1, 4
    cam = campos.detach().reshape(3).to(torch.float32).contiguous()
    cs = ColorSource(_upload(tab, c.device), 1, int(sh_degree), cam, [c], [(0, N, coeffs, None, coeffs, None)])
    cs.autograd, cs.width = True, K * 3
    return cs


_dir_tables: dict = {}      # (coefficient pointer, N, K, activation, device, stream) -> uploaded descriptor (its content is a function of the key)


def sh_direction_source(coeffs, sh_degree: int, dirs, use_sh: int) -> ColorSource:
    """ColorSource for MTGS's own call style -- rgbs = spherical_harmonics(n, viewdirs, colors); torch.clamp(rgbs + 0.5, 0, 1)
    (vanilla_gaussian_splatting.py:313-318; use_sh = 1) or clamp_min (use_sh = 4) -- whose deferred result reached rasterization()
    (wrapper._LazySH.raster_source): one descriptor per [n_i, 16, 3] coefficient tensor (`coeffs` / `dirs`: a tensor each, or equally long
    lists of them -- the nodes of a scene graph whose colours were concatenated, mtgs_scene_graph.py:451-452 -- in that order), the
    caller's directions as they are."""
    from . import wrapper
    cl = list(coeffs) if isinstance(coeffs, (list, tuple)) else [coeffs]
    dl = list(dirs) if isinstance(dirs, (list, tuple)) else [dirs]
    assert len(cl) == len(dl) >= 1 and sh_degree <= 3
    for c_, d_ in zip(cl, dl):
        assert c_.dim() == 3 and c_.shape[1:] == (16, 3) and c_.dtype == torch.float32 and d_.shape == (c_.shape[0], 3)
    K = 16
    cc = [c_.detach().contiguous() for c_ in cl]
    d = dl[0].detach().contiguous() if len(dl) == 1 else torch.cat([d_.detach() for d_ in dl], dim=0)      # (collected order, like `means`)
    dev = cc[0].device
    in_graph = wrapper._graph.caps is not None or torch.cuda.is_current_stream_capturing()
    key = (tuple((c_.data_ptr(), c_.shape[0]) for c_ in cc), int(use_sh), dev.index, torch.cuda.current_stream(dev).cuda_stream)
    table = None if in_graph else _dir_tables.get(key)
    starts = np.concatenate([[0], np.cumsum([c_.shape[0] for c_ in cc])]).astype(np.int64)
    if table is None:
        tab = np.zeros(len(cc), dtype=_DESC)
        for i, c_ in enumerate(cc):
            tab["n"][i], tab["start"][i] = c_.shape[0], starts[i]
            tab["features_dc"][i], tab["features_rest"][i] = c_.data_ptr(), c_.data_ptr() + 12
            tab["dc_stride"][i] = tab["rest_stride"][i] = K * 3
            tab["dc_add_stride"][i] = 3
            tab["k_rest"][i], tab["use_sh"][i] = K - 1, int(use_sh)
        tab["first_block"] = np.concatenate([[0], np.cumsum([-(-c_.shape[0] // 256) for c_ in cc])[:-1]])
        table = _upload(tab, dev)
        if not in_graph:      # (a table uploaded inside a capture has no content before the first replay)
            if len(_dir_tables) >= 16:
                _dir_tables.clear()
            _dir_tables[key] = table
    cs = ColorSource(table, len(cc), int(sh_degree), None, cc + [d],
                     [(int(starts[i]), c_.shape[0], cl[i], None, cl[i], None) for i, c_ in enumerate(cc)])
    cs.autograd, cs.width, cs.dirs = True, K * 3, d
    return cs


class _CollectNodes(torch.autograd.Function):
    """All nodes of a scene as ONE autograd node and ONE launch per direction (mtgs_node_fwd_batch / mtgs_node_bwd_batch: a
    table of node descriptors in device memory): every node's workgroups write straight into its slice of the collected
    tensors (no torch.cat of the per-node outputs), the backward reads the slices of the incoming gradients and writes
    every node's parameter gradients into slices of a few flat buffers."""

    @staticmethod
    def forward(ctx, cam_pos, specs, *flat):
        # specs[i] = (degree, use_sh, trav, frame[, deferred colours]); flat = _NK tensors (or None) per node in _NODE_KEYS order
        n_nodes = len(specs)
        deferred = len(specs[0]) > 4 and bool(specs[0][4])
        require_gpu(cam_pos, *[t for t in flat if t is not None])
        sizes = [flat[_NK * i].shape[0] for i in range(n_nodes)]
        total, dev = sum(sizes), flat[0].device
        degree = specs[0][0]
        assert all(sp[0] == degree for sp in specs), "one sh_degree_to_use per step"
        cam = cam_pos.detach().reshape(3).to(torch.float32).contiguous()
        means = torch.empty((total, 3), dtype=torch.float32, device=dev)
        scales = torch.empty((total, 3), dtype=torch.float32, device=dev)
        quats = torch.empty((total, 4), dtype=torch.float32, device=dev)
        opacities = torch.empty((total,), dtype=torch.float32, device=dev)
        rgbs = torch.empty((total, 3), dtype=torch.float32, device=dev)
        mask = torch.empty((total,), dtype=torch.uint8, device=dev)
        model_id = torch.empty((total,), dtype=torch.int64, device=dev)
        # rigid nodes: either the pose of the frame is given (instance_quat[4], instance_trans[3]: gathered with ONE cat into
        # [R', 7]) or the per-frame parameter tables + frame index (specs[i][3] >= 0: the kernel reads and normalises row `frame`)
        rigid = [i for i in range(n_nodes) if flat[_NK * i + 7] is not None]
        given = [i for i in rigid if specs[i][3] < 0]
        framed = [i for i in rigid if specs[i][3] >= 0]
        pose_all = None
        if given:
            pose_all = torch.cat([flat[_NK * i + j].detach().reshape(-1) for i in given for j in (7, 8)]).to(torch.float32)
            assert pose_all.numel() == 7 * len(given)
        pose_tabs = []
        for i in framed:
            pose_tabs += [flat[_NK * i + 7].detach().to(torch.float32).contiguous(), flat[_NK * i + 8].detach().to(torch.float32).contiguous()]
        tab = np.zeros(n_nodes, dtype=_DESC)
        saved, keep, dims = [], [], []
        col = {k: [] for k in ("means", "scales_raw", "quats_raw", "opacities_raw", "features_dc", "features_dc_add", "features_rest",
                               "dc_stride", "dc_add_stride", "rest_stride", "k_rest", "use_sh", "n_traversals", "traversal")}
        start = 0
        for i in range(n_nodes):
            m, sr, qr, orw, dc, add, rest, _, _ = flat[_NK * i:_NK * i + _NK]
            _, use_sh, trav = specs[i][:3]
            n = sizes[i]
            Kr = rest.shape[-2]
            T = rest.shape[1] if trav >= 0 else 0
            if trav >= 0:   # FULL per-traversal parameters: slice `trav` is read in place
                rest = rest[:, trav]
                add = None if add is None else add[:, trav]
            m_c, sr_c, qr_c = m.detach().contiguous(), sr.contiguous(), qr.contiguous()
            o_c = orw.reshape(n).contiguous()
            dc_c, s_dc = _rows(dc, 3)
            add_c, s_add = _rows(add, 3)
            rest_c, s_rest = _rows(rest, Kr * 3)
            keep += [sr_c, qr_c, o_c, dc_c, add_c, rest_c]   # (ColorSource: what the table's raw-parameter pointers refer to)
            saved += [m_c, qr_c]
            dims.append((n, Kr, orw.shape, add is not None, T, int(trav), start))
            for k, v in zip(col, (m_c.data_ptr(), sr_c.data_ptr(), qr_c.data_ptr(), o_c.data_ptr(), dc_c.data_ptr(),
                                  0 if add_c is None else add_c.data_ptr(), rest_c.data_ptr(), s_dc, s_add, s_rest, Kr, int(use_sh),
                                  T, max(int(trav), 0))):
                col[k].append(v)
            start += n
        for k, v in col.items():
            tab[k] = v
        n_arr = np.asarray(sizes, dtype=np.int64)
        starts = np.cumsum(n_arr) - n_arr
        nblk = (n_arr + 255) // 256
        tab["n"], tab["start"], tab["first_block"] = n_arr, starts, np.cumsum(nblk) - nblk
        tab["skip_colors"] = int(deferred)
        blk = int(nblk.sum())
        ustarts = starts.astype(np.uint64)
        for k, t, w in (("scales", scales, 12), ("quats", quats, 16), ("opacities", opacities, 4), ("rgbs", rgbs, 12),
                        ("clamp_mask", mask, 1), ("means_out", means, 12)):
            tab[k] = np.uint64(t.data_ptr()) + np.uint64(w) * ustarts
        if given:
            base = np.uint64(pose_all.data_ptr()) + np.uint64(28) * np.arange(len(given), dtype=np.uint64)
            tab["pose"][given], tab["pose_trans"][given] = base, base + np.uint64(16)
        if framed:
            fr = [specs[i][3] for i in framed]
            tab["pose"][framed] = [q.data_ptr() + 16 * f for q, f in zip(pose_tabs[0::2], fr)]
            tab["pose_trans"][framed] = [t.data_ptr() + 12 * f for t, f in zip(pose_tabs[1::2], fr)]
            tab["pose_normalize"][framed] = 1
            for i in framed:      # (a device frame index: spec frame 0 = row 0 of the tables, the kernels add the word's value)
                if len(specs[i]) > 5 and specs[i][5] is not None:
                    tab["frame_dev"][i] = specs[i][5].data_ptr()
                    keep.append(specs[i][5])
        tab_dev = _upload(tab, dev)
        call("mtgs_node_fwd_batch", n_nodes, ptr(tab_dev), blk, -1 if deferred else int(degree), ptr(cam), ptr(model_id),
             stream_of(means))
        ctx.color_source = None
        # (the table and what its raw-parameter pointers refer to: mtgs_node_bwd_rows of a caller that keeps geometry gradients
        #  as rows -- ColorSource.geometry_rows, the row form of the data-parallel exchange)
        _CollectNodes.last_table = (tab_dev, n_nodes, list(keep))
        if deferred:   # the front end of the rasterizer reads the coefficients of the VISIBLE Gaussians through this table
            node_params = [(int(starts[i]), sizes[i]) for i in range(n_nodes)]   # (+ the leaf parameters: collect_gaussians)
            _CollectNodes.last_color_source = ColorSource(tab_dev, n_nodes, int(degree), cam, keep, node_params)
        del keep   # (stream-ordered allocator: the launch above is already enqueued)
        ctx.tab, ctx.dims, ctx.blocks, ctx.degree, ctx.rigid, ctx.deferred = tab, dims, blk, int(degree), rigid, deferred
        ctx.framed = [(i, specs[i][3], flat[_NK * i + 7].shape[0]) for i in framed]
        ctx.n_flat = len(flat)
        ctx.set_materialize_grads(False)     # (no zero-filled cotangents: see backward)
        ctx.save_for_backward(cam, scales, opacities, rgbs, mask, pose_all, *saved, *pose_tabs)
        ctx.mark_non_differentiable(model_id)
        return means, scales, quats, opacities, rgbs, model_id

    @staticmethod
    def backward(ctx, v_means, v_scales, v_quats, v_opacities, v_rgbs, _v_id):
        if all(g is None for g in (v_means, v_scales, v_quats, v_opacities, v_rgbs)):
            # nothing arrived (ColorSource.geometry_rows: the rasterization keeps its geometry gradients as rows for the
            # optimizer): no launch, and the parameters' .grad stay None
            return (None, None) + (None,) * ctx.n_flat
        cam, scales, opacities, rgbs, mask, pose_all, *saved = ctx.saved_tensors
        dims, rigid, framed = ctx.dims, ctx.rigid, ctx.framed
        saved = saved[:2 * len(dims)]   # (the pose tables behind them are only kept alive: the table holds their addresses)
        n_nodes, total, dev = len(dims), scales.shape[0], scales.device
        need = ctx.needs_input_grad[2:]
        z = lambda g, shape: (torch.zeros(shape, dtype=torch.float32, device=dev) if g is None else g.to(torch.float32).contiguous())
        v_scales, v_quats = z(v_scales, (total, 3)), z(v_quats, (total, 4))
        v_opacities, v_rgbs = z(v_opacities, (total,)), (None if ctx.deferred else z(v_rgbs, (total, 3)))
        v_means_c = None if v_means is None else v_means.to(torch.float32).contiguous()
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        g_scales, g_quats, g_opac, g_dc = new(total, 3), new(total, 4), new(total), new(total, 3)
        # no node wants a colour gradient (features passed detached: the data-parallel exchange rebuilds the coefficient
        # gradient on the receivers): the kernel then skips the coefficient rows, the largest stream of the backward
        colours = (not ctx.deferred) and any(need[_NK * i + j] for i in range(n_nodes) for j in (4, 5, 6))
        rest_sizes = [n * Kr * 3 * max(T, 1) if colours else 0 for (n, Kr, _, _, T, _, _) in dims]
        add_sizes = [n * T * 3 if (T and has_add and colours) else 0 for (n, _, _, has_add, T, _, _) in dims]
        g_rest_flat, g_add_flat = new(sum(rest_sizes)), new(sum(add_sizes))
        if not colours:
            g_dc = None
        want_gm = bool(rigid) and any(need[_NK * i] for i in rigid)
        g_means_all = new(total, 3) if want_gm else None
        g_pose_all = torch.zeros((len(rigid), 7), dtype=torch.float32, device=dev) if rigid else None   # atomics
        rigid_row = {i: r for r, i in enumerate(rigid)}
        tab = ctx.tab.copy()
        ustarts = tab["start"].astype(np.uint64)
        at = lambda t, w: np.uint64(t.data_ptr()) + np.uint64(w) * ustarts
        tab["means"] = [t.data_ptr() for t in saved[0::2]]
        tab["quats_raw"] = [t.data_ptr() for t in saved[1::2]]
        tab["v_scales"], tab["v_quats"], tab["v_opacities"] = at(v_scales, 12), at(v_quats, 16), at(v_opacities, 4)
        tab["v_rgbs"] = 0 if ctx.deferred else at(v_rgbs, 12)
        tab["v_means"] = at(v_means_c, 12) if v_means_c is not None else 0
        tab["g_scales_raw"], tab["g_quats_raw"], tab["g_opacities_raw"] = at(g_scales, 12), at(g_quats, 16), at(g_opac, 4)
        tab["g_features_dc"] = at(g_dc, 12) if colours else 0
        rs, ads = np.asarray(rest_sizes, dtype=np.uint64), np.asarray(add_sizes, dtype=np.uint64)
        tab["g_features_rest"] = (np.uint64(g_rest_flat.data_ptr()) + np.uint64(4) * (np.cumsum(rs) - rs)) if colours else 0
        tab["g_features_dc_add"] = np.where(ads > 0, np.uint64(g_add_flat.data_ptr()) + np.uint64(4) * (np.cumsum(ads) - ads), np.uint64(0))
        if rigid:
            if g_means_all is not None:
                want = np.asarray([bool(need[_NK * i]) for i in rigid])
                tab["g_means"][rigid] = np.where(want, at(g_means_all, 12)[rigid], np.uint64(0))
            tab["g_pose"][rigid] = np.uint64(g_pose_all.data_ptr()) + np.uint64(28) * np.arange(len(rigid), dtype=np.uint64)
        gq_tab = gt_tab = None
        if framed:   # gradients of the per-frame pose parameters: zero but for row `frame` (written after the launch)
            frames = sum(F for _, _, F in framed)
            gq_tab = torch.zeros((frames, 4), dtype=torch.float32, device=dev)
            gt_tab = torch.zeros((frames, 3), dtype=torch.float32, device=dev)
            F_arr = np.asarray([F for _, _, F in framed], dtype=np.uint64)
            rows = (np.cumsum(F_arr) - F_arr) + np.asarray([f for _, f, _ in framed], dtype=np.uint64)
            idx = [i for i, _, _ in framed]
            tab["g_pose_quat_row"][idx] = np.uint64(gq_tab.data_ptr()) + np.uint64(16) * rows
            tab["g_pose_trans_row"][idx] = np.uint64(gt_tab.data_ptr()) + np.uint64(12) * rows
        tab_dev = _upload(tab, dev)
        call("mtgs_node_bwd_batch", n_nodes, ptr(tab_dev), ctx.blocks, ctx.degree, ptr(cam), stream_of(scales))
        # per-node views of the flat buffers, in _NODE_KEYS order
        sizes = [d[0] for d in dims]
        sp = lambda t: t.split(sizes) if total else [t] * n_nodes
        s_scales, s_quats, s_opac = sp(g_scales), sp(g_quats), sp(g_opac)
        s_dc = sp(g_dc) if colours else [None] * n_nodes
        s_rest, s_add = g_rest_flat.split(rest_sizes), g_add_flat.split(add_sizes)
        s_vm = sp(v_means_c) if v_means_c is not None else [None] * n_nodes
        s_gm = sp(g_means_all) if g_means_all is not None else [None] * n_nodes
        framed_grads = {}
        if framed:
            Fs = [F for _, _, F in framed]
            for (i, _, _), gq, gt in zip(framed, gq_tab.split(Fs), gt_tab.split(Fs)):
                framed_grads[i] = (gq, gt)
        grads = []
        for i, (n, Kr, opac_shape, has_add, T, trav, start) in enumerate(dims):
            if not colours:
                g_r = g_a = None
            elif T:
                g_r = s_rest[i].view(n, T, Kr, 3)
                g_a = s_add[i].view(n, T, 3) if has_add else None
            else:
                g_r = s_rest[i].view(n, Kr, 3)
                g_a = s_dc[i] if has_add else None
            if i in framed_grads:
                g_m, (g_q, g_t) = s_gm[i], framed_grads[i]
            elif i in rigid_row:
                gp = g_pose_all[rigid_row[i]]
                g_m, g_q, g_t = s_gm[i], gp[:4], gp[4:]
            else:
                g_m, g_q, g_t = s_vm[i], None, None
            grads += [g_m, s_scales[i], s_quats[i], s_opac[i].reshape(opac_shape), s_dc[i], g_a, g_r, g_q, g_t]
        return (None, None) + tuple(g if need[j] else None for j, g in enumerate(grads))


def collect_gaussians(nodes, camera_to_worlds: Tensor, sh_degree_to_use: int, model_sh_degree: int = 3,
                      raw_colors: bool = False, deferred_colors: bool = False) -> Dict[str, Tensor]:
    """MTGSSceneModel.get_gaussians for static nodes (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:408-461): the
    activated Gaussians of every node, concatenated in order, plus `model_id`.  `nodes` is a sequence of dicts of RAW
    parameters {"means", "scales", "quats", "opacities", "features_dc", "features_rest"} with, for multi-colour nodes,
    "features_adapters" [N,T,3], a 4-D "features_rest" [N,T,K-1,3] and "traversal_index"; for rigid nodes either the pose of the
    current frame "instance_quat" [4] (wxyz) and "instance_trans" [3] (see node_gaussians), or the per-frame pose PARAMETERS
    "instance_quats" [F,4], "instance_trans" [F,3] and "frame_idx": the kernel then reads row frame_idx and normalises the
    quaternion as RigidSubModel.get_object_pose does (rigid_node.py:139-144), and the backward returns the gradients of the
    full tables (zero but for that row).  One autograd node and ONE kernel
    launch per direction for the whole scene, however many nodes it has (a scene graph holds one rigid node per object
    instance in view): each node's workgroups write into its slice of the collected tensors (no torch.cat of per-node
    outputs); "model_id" is written by the same launch.
    raw_colors: "rgbs" is the SH value itself, without clamp(. + 0.5, 0, 1) -- what the data-parallel exchange takes
    (mtgs_amd.dist.SparseGradExchange.rasterization applies the activation and differentiates it).
    deferred_colors: VISIBILITY FIRST -- the node kernels run geometry-only (no coefficient reads, no "rgbs"), and the returned
    "color_source" (ColorSource) makes `rasterization(..., color_source=...)` evaluate SH + clamp for the Gaussians its
    projection found visible, straight into their records; the coefficient gradient comes back as compact rows
    (ColorSource.apply_to(FusedAdam)).  MTGS computes the colours of all Gaussians of all nodes every step
    (vanilla_gaussian_splatting.py:309-322); a camera sees ~15 % of them."""
    specs, flat, sizes = [], [], []
    use_sh = model_sh_degree > 0
    assert use_sh or not raw_colors, "raw_colors needs an SH colour model"
    if len(nodes) == 0:
        raise ValueError("collect_gaussians: no nodes (MTGS returns its empty outputs before reaching the rasterizer, "
                         "mtgs_scene_graph.py:595-598)")
    for nd in nodes:
        N = nd["means"].shape[0]
        for k in ("means", "scales", "quats", "opacities", "features_dc", "features_rest"):
            if nd[k].dtype != torch.float32:
                raise TypeError(f"collect_gaussians: {k} must be float32, got {nd[k].dtype}")
        rest, add, trav = nd["features_rest"], nd.get("features_adapters"), nd.get("traversal_index")
        trav_dev = isinstance(trav, Tensor)
        if trav_dev:
            # the traversal as an int32 DEVICE scalar (one captured iteration for every traversal): only the optimizer's peek /
            # step read it (ColorSource.optimizer with row-lazy colour parameters) -- the node table carries slice 0 unread
            if not deferred_colors or trav.dtype != torch.int32 or trav.numel() != 1 or not trav.is_cuda:
                raise ValueError("collect_gaussians: a device 'traversal_index' is one int32 on the GPU and needs deferred_colors")
            trav = 0
        if rest.dim() == 4:
            if trav is None:
                raise ValueError("collect_gaussians: per-traversal features_rest [N,T,K-1,3] needs 'traversal_index'")
            assert 0 <= trav < rest.shape[1] and (add is None or add.shape == (N, rest.shape[1], 3)), (trav, rest.shape)
            rest, add = rest.contiguous(), None if add is None else add.contiguous()
        else:
            assert rest.dim() == 3, rest.shape
            if add is not None and add.dim() == 3:   # [N,T,3] adapters with a shared features_rest: slice the adapters
                add = add[:, trav]
            trav = None
        Kr = rest.shape[-2]
        if Kr > 15 or sh_degree_to_use > 3:
            raise NotImplementedError("collect_gaussians: SH degree > 3 (MTGS configs use <= 3)")
        if use_sh:
            assert (sh_degree_to_use + 1) ** 2 <= Kr + 1, (sh_degree_to_use, rest.shape)
        assert nd["scales"].shape == (N, 3) and nd["quats"].shape == (N, 4) and nd["opacities"].numel() == N
        iq, it, frame, frame_dev = nd.get("instance_quat"), nd.get("instance_trans"), -1, None
        if nd.get("instance_quats") is not None:   # per-frame pose parameters [F,4] / [F,3] + the frame of this step
            assert iq is None, "pass either instance_quat (the pose) or instance_quats + frame_idx (the parameters)"
            iq, frame = nd["instance_quats"], nd["frame_idx"]
            assert iq.dim() == 2 and iq.shape[1] == 4 and it is not None and it.shape == (iq.shape[0], 3), (iq.shape, None if it is None else it.shape)
            if isinstance(frame, Tensor):    # the frame as an int32 DEVICE scalar (one captured iteration for every frame; its value --
                #                              0 <= frame < F -- is the caller's promise): the kernels add it to the tables' row 0
                if frame.dtype != torch.int32 or frame.numel() != 1 or not frame.is_cuda:
                    raise ValueError("collect_gaussians: a device 'frame_idx' is one int32 on the GPU")
                frame_dev, frame = frame, 0
            else:
                frame = int(frame)
                assert 0 <= frame < iq.shape[0], (frame, iq.shape)
        else:
            assert (iq is None) == (it is None) and (iq is None or (iq.numel() == 4 and it.numel() == 3)), "rigid pose: quat[4] + trans[3]"
        specs.append((int(sh_degree_to_use), 2 if raw_colors else int(use_sh), -1 if trav is None else int(trav), frame,
                      bool(deferred_colors), frame_dev))
        flat += [nd["means"], nd["scales"], nd["quats"], nd["opacities"], nd["features_dc"], add, rest, iq, it]
        sizes.append(N)
    cam_pos = camera_to_worlds[..., :3, 3].reshape(-1)[:3]
    assert not (deferred_colors and raw_colors)
    means, scales, quats, opacities, rgbs, model_id = _CollectNodes.apply(cam_pos, tuple(specs), *flat)
    out = {"means": means, "scales": scales, "quats": quats, "opacities": opacities, "rgbs": rgbs, "model_id": model_id}
    out["node_table"], _CollectNodes.last_table = getattr(_CollectNodes, "last_table", None), None
    if deferred_colors:
        out["rgbs"] = None
        cs, _CollectNodes.last_color_source = _CollectNodes.last_color_source, None
        cs.node_geometry = [(nd["means"], nd["scales"], nd["quats"], nd["opacities"],
                             nd.get("instance_quat") is not None or nd.get("instance_quats") is not None) for nd in nodes]
        cs.node_params = [(st, n, nd["features_dc"], nd.get("features_adapters"), nd["features_rest"], nd.get("traversal_index"))
                          for (st, n), nd in zip(cs.node_params, nodes)]
        out["color_source"] = cs
    return out


def node_gaussians(means: Tensor, scales: Tensor, quats: Tensor, opacities: Tensor, features_dc: Tensor,
                   features_rest: Tensor, camera_to_worlds: Tensor, sh_degree_to_use: int, model_sh_degree: int,
                   features_dc_add: Optional[Tensor] = None, traversal_index: Optional[int] = None,
                   instance_quat: Optional[Tensor] = None, instance_trans: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """The dict VanillaGaussianSplattingModel.get_gaussians(camera_to_worlds) returns, from the RAW parameters:
    means[N,3], scales[N,3] (log), quats[N,4], opacities[N,1] (logits), features_dc[N,3], features_rest[N,K-1,3];
    `sh_degree_to_use` = min(step // sh_degree_interval, sh_degree), `model_sh_degree` = the model's sh_degree
    (0 selects rgbs = sigmoid(features_dc), vanilla_gaussian_splatting.py:319-320).  camera_to_worlds[...,3,4]:
    only its translation is used, as in the reference (:314).
    Multi-colour nodes (multi_color_gaussian_splatting.py:77-86), either
      * pass the slices: features_rest[:, t], features_dc_add=features_adapters[:, t]  (autograd then zero-fills the
        full-size gradients and copies the slice in), or
      * pass the FULL per-traversal parameters features_rest[N,T,K-1,3], features_dc_add=features_adapters[N,T,3] and
        traversal_index=t: slice t is read in place and the backward writes the full-size gradients in its own pass.
    Rigid nodes (rigid_node.py:205-216, 283-291; fourier_features_dim = None as in the shipped configs): pass the pose
    get_object_pose returned -- instance_quat[4] (wxyz), instance_trans[3]; "means" is then the GLOBAL mean
    quat_to_rotmat(q) m + t, "quats" = quat_mult(q, q_local / |q_local|), the view directions use the global means, and
    gradients flow back to the local means, the local quaternions and the pose."""
    N = means.shape[0]
    assert (instance_quat is None) == (instance_trans is None), "instance_quat and instance_trans go together"
    if instance_quat is not None:
        assert instance_quat.numel() == 4 and instance_trans.numel() == 3, (instance_quat.shape, instance_trans.shape)
    if traversal_index is not None:
        assert features_rest.dim() == 4 and features_rest.shape[0] == N and features_rest.shape[3] == 3, features_rest.shape
        T = features_rest.shape[1]
        assert 0 <= traversal_index < T, (traversal_index, T)
        assert features_dc_add is None or features_dc_add.shape == (N, T, 3), features_dc_add.shape
        assert scales.shape == (N, 3) and quats.shape == (N, 4) and opacities.numel() == N and features_dc.shape == (N, 3)
        use_sh = model_sh_degree > 0
        if use_sh:
            assert (sh_degree_to_use + 1) ** 2 <= features_rest.shape[2] + 1, (sh_degree_to_use, features_rest.shape)
        if features_rest.shape[2] > 15 or sh_degree_to_use > 3:
            raise NotImplementedError("node_gaussians: SH degree > 3 (MTGS configs use <= 3)")
        cam_pos = camera_to_worlds[..., :3, 3].reshape(-1)[:3]
        s, q, o, rgb, mg = _NodeActivations.apply(means, scales, quats, opacities, features_dc,
                                                  None if features_dc_add is None else features_dc_add.contiguous(),
                                                  features_rest.contiguous(), cam_pos, int(sh_degree_to_use), bool(use_sh),
                                                  int(traversal_index), instance_quat, instance_trans)
        return {"means": means if instance_quat is None else mg, "scales": s, "quats": q, "opacities": o, "rgbs": rgb}
    assert scales.shape == (N, 3) and quats.shape == (N, 4), (scales.shape, quats.shape)
    assert opacities.numel() == N, opacities.shape
    assert features_dc.shape == (N, 3), features_dc.shape
    assert features_rest.dim() == 3 and features_rest.shape[0] == N and features_rest.shape[2] == 3, features_rest.shape
    if features_dc_add is not None:
        assert features_dc_add.shape == (N, 3), features_dc_add.shape
    use_sh = model_sh_degree > 0
    if use_sh:
        assert (sh_degree_to_use + 1) ** 2 <= features_rest.shape[1] + 1, (sh_degree_to_use, features_rest.shape)
    if features_rest.shape[1] > 15 or sh_degree_to_use > 3:
        raise NotImplementedError("node_gaussians: SH degree > 3 (MTGS configs use <= 3)")
    cam_pos = camera_to_worlds[..., :3, 3].reshape(-1)[:3]
    s, q, o, rgb, mg = _NodeActivations.apply(means, scales, quats, opacities, features_dc, features_dc_add, features_rest,
                                              cam_pos, int(sh_degree_to_use), bool(use_sh), -1, instance_quat, instance_trans)
    return {"means": means if instance_quat is None else mg, "scales": s, "quats": q, "opacities": o, "rgbs": rgb}


class _CameraSpaceNormals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, quats, scales, means, c2w, rgbs):
        require_gpu(quats, scales, means, c2w, rgbs)
        N, dev = quats.shape[0], quats.device
        q_c = quats.detach().to(torch.float32).contiguous()
        s_c = scales.detach().to(torch.float32).contiguous()
        m_c = means.detach().to(torch.float32).contiguous()
        cam = c2w.detach().to(torch.float32).reshape(-1, 3, 4)[0].contiguous()
        r_c = None if rgbs is None else rgbs.detach().to(torch.float32).contiguous()
        width = 3 if rgbs is None else 6
        out = torch.empty((N, width), dtype=torch.float32, device=dev)
        call("mtgs_normals_fwd", N, ptr(q_c), ptr(s_c), ptr(m_c), ptr(cam), ptr(r_c), ptr(out), width, stream_of(out))
        ctx.save_for_backward(q_c, s_c, m_c, cam)
        ctx.with_rgbs = rgbs is not None
        return out

    @staticmethod
    def backward(ctx, v_out):
        q_c, s_c, m_c, cam = ctx.saved_tensors
        N = q_c.shape[0]
        v = v_out.to(torch.float32)
        if v.stride(-1) != 1:
            v = v.contiguous()
        col0 = 3 if ctx.with_rgbs else 0
        g_quats = torch.empty_like(q_c)
        if N:
            call("mtgs_normals_bwd", N, ptr(q_c), ptr(s_c), ptr(m_c), ptr(cam), v.data_ptr() + 4 * col0, v.stride(0), ptr(g_quats),
                 stream_of(q_c))
        return g_quats, None, None, None, (v[:, :3] if ctx.with_rgbs else None)


def camera_space_normals(quats: Tensor, scales: Tensor, means: Tensor, camera_to_worlds: Tensor,
                         rgbs: Optional[Tensor] = None) -> Tensor:
    """MTGSSceneModel._get_gaussian_camera_space_normals (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:526-545):
    the shortest axis of every Gaussian -- column argmin(scales) of quat_to_rotmat(quats), normalised, flipped to face the
    camera -- rotated into camera space, [N,3].  quats / scales / means are the COLLECTED (activated) Gaussians,
    camera_to_worlds[(1,)3,4] the camera of the step (`camera.camera_to_worlds`; read on the device, no host copy).
    With `rgbs` [N,3] the result is `torch.cat([rgbs, normals], -1)` [N,6] -- what `predict_normals` hands to
    `rasterization(colors=...)` (:636-638) -- written by the same kernel.  Gradients: quats (and rgbs, passed through)."""
    N = quats.shape[0]
    assert quats.shape == (N, 4) and scales.shape == (N, 3) and means.shape == (N, 3), (quats.shape, scales.shape, means.shape)
    assert camera_to_worlds.shape[-2:] == (3, 4) and camera_to_worlds.numel() == 12, camera_to_worlds.shape
    assert rgbs is None or rgbs.shape == (N, 3), rgbs.shape
    return _CameraSpaceNormals.apply(quats, scales, means, camera_to_worlds, rgbs)


# ------------------------------------------------------------------------------------ rigid nodes: pose between frames, Fourier colour
def interpolate_quats(q1: Tensor, q2: Tensor, fraction) -> Tensor:
    """mtgs utils.interpolate_quats (utils.py:201-233): slerp with the nlerp branch above dot 0.9995, result normalised.
    Device-side torch (8 floats), differentiable like the reference."""
    q1 = q1[None] if q1.dim() == 1 else q1
    q2 = q2[None] if q2.dim() == 1 else q2
    q1 = q1 / torch.norm(q1, dim=-1, keepdim=True)
    q2 = q2 / torch.norm(q2, dim=-1, keepdim=True)
    dot = torch.clamp((q1 * q2).sum(dim=-1, keepdim=True), -1, 1)
    q2 = torch.where(dot < 0, -q2, q2)
    dot = torch.abs(dot)
    similar = dot > 0.9995
    lin = q1 + fraction * (q2 - q1)
    theta_0 = torch.acos(dot)
    theta = theta_0 * fraction
    s2 = torch.sin(theta) / torch.sin(theta_0)
    s1 = torch.cos(theta) - dot * s2
    out = torch.where(similar, lin, s1 * q1 + s2 * q2)
    return out / torch.norm(out, dim=-1, keepdim=True)


def object_pose(instance_quats: Tensor, instance_trans: Tensor, frame_idx: Optional[int] = None, timestamp=None,
                frame_timestamps: Optional[Tensor] = None, in_frame_mask: Optional[Tensor] = None, num_frames: Optional[int] = None):
    """RigidSubModel.get_object_pose (rigid_node.py:127-166) for a non-static object: (quat[4], trans[3]) of the frame, or
    (None, None) when the object is not in it.  With `frame_idx` the normalised row; otherwise the pose BETWEEN the two
    frames adjacent to `timestamp` (slerp of the quaternions, lerp of the translations); a timestamp that matches a frame
    returns that row as stored (the reference does not normalise it there, :155-157).
    The frame search reads three scalars back (as the reference's `if not in_frame_mask[...]` does)."""
    F = instance_quats.shape[0] if num_frames is None else num_frames
    if frame_idx is not None:
        if frame_idx >= F or (in_frame_mask is not None and not bool(in_frame_mask[frame_idx])):
            return None, None
        q = instance_quats[frame_idx]
        return q / q.norm(dim=-1, keepdim=True), instance_trans[frame_idx]
    assert timestamp is not None and frame_timestamps is not None, "frame_idx, or timestamp + frame_timestamps"
    ts = frame_timestamps.to(instance_quats.device)
    diffs = timestamp - ts
    inf = torch.full_like(diffs, float("inf"))
    prev_f = int(torch.argmin(torch.where(diffs >= 0, diffs, inf)))
    next_f = int(torch.argmin(torch.where(diffs <= 0, -diffs, inf)))
    if in_frame_mask is not None and not (bool(in_frame_mask[next_f]) and bool(in_frame_mask[prev_f])):
        return None, None
    if next_f == prev_f:
        return instance_quats[next_f], instance_trans[next_f]
    t = (timestamp - ts[prev_f]) / (ts[next_f] - ts[prev_f])
    return (interpolate_quats(instance_quats[prev_f], instance_quats[next_f], t).squeeze(0),
            torch.lerp(instance_trans[prev_f], instance_trans[next_f], t))


def idft_weights(x, dim: int, input_normalized: bool = True, device=None) -> Tensor:
    """mtgs utils.IDFT (utils.py:335-352) for ONE x: w[dim], cos on the even indices, sin(index + 1) on the odd ones."""
    x = torch.as_tensor(x, dtype=torch.float32, device=device).reshape(())
    idx = torch.arange(dim, dtype=torch.float32, device=x.device)
    odd = (torch.arange(dim, device=x.device) % 2) == 1
    k = torch.where(odd, idx + 1, idx)
    ang = x * k * (2 * math.pi / dim) if input_normalized else x * k
    return torch.where(odd, torch.sin(ang), torch.cos(ang))


class _FourierDC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features_dc, w):
        require_gpu(features_dc, w)
        ctx.w_shape = w.shape
        f, w = features_dc.contiguous(), w.to(torch.float32).reshape(-1).contiguous()
        N, F = f.shape[0], f.shape[1]
        assert w.numel() == F, (w.shape, f.shape)
        dc = torch.empty((N, 3), dtype=torch.float32, device=f.device)
        call("mtgs_fourier_dc_fwd", N, F, ptr(f), ptr(w), ptr(dc), stream_of(f))
        ctx.save_for_backward(f, w)
        return dc

    @staticmethod
    def backward(ctx, v_dc):
        f, w = ctx.saved_tensors
        N, F = f.shape[0], f.shape[1]
        v_f = torch.empty_like(f)
        nblk = (N * 3 + 255) // 256
        part = torch.empty((max(nblk, 1), F), dtype=torch.float32, device=f.device) if ctx.needs_input_grad[1] else None
        call("mtgs_fourier_dc_bwd", N, F, ptr(f), ptr(w), ptr(v_dc.contiguous()), ptr(v_f), ptr(part), stream_of(f))
        v_w = None
        if part is not None:
            v_w = (part.sum(0) if N > 0 else torch.zeros_like(w)).reshape(ctx.w_shape)
        return v_f, v_w


def fourier_features_dc(features_dc: Tensor, x, scale: float = 1.0, space: str = "temporal") -> Tensor:
    """RigidSubModel.get_true_features_dc / get_fourier_features (rigid_node.py:217-229): features_dc[N,F,3] ->
    true_features_dc[N,3] for the frame's normalised timestamp (space 'temporal') or the camera-object yaw ('spatial')."""
    assert features_dc.dim() == 3 and features_dc.shape[2] == 3, features_dc.shape
    w = idft_weights(torch.as_tensor(x, device=features_dc.device) * scale, features_dc.shape[1], space == "temporal",
                     device=features_dc.device)
    return _FourierDC.apply(features_dc, w)


def cam_obj_yaw(camera_to_world: Tensor, quat_cur_frame: Tensor) -> Tensor:
    """RigidSubModel.get_cam_obj_yaw (rigid_node.py:231-237)."""
    w, x, y, z = quat_cur_frame.unbind(-1)
    r00, r02 = 1 - 2 * (y * y + z * z), 2 * (x * z + w * y)        # quat_to_rotmat rows (utils.py:14-40)
    return torch.atan2(camera_to_world[..., 0, 0], camera_to_world[..., 0, 2]) - torch.atan2(r00, r02)
