"""Loss-head pieces on the consumer side of the rasterization path.

`masked_ssim(gt, pred, mask)` == mtgs.utils.ssim.MaskedSSIM(data_range=1.0, size_average=True, channel=3)(
gt.permute(2,0,1)[None], pred.permute(2,0,1)[None], mask=mask)  as MTGS calls it
(/root/reference/mtgs/scene_model/mtgs_scene_graph.py:322, :831-841; /root/reference/mtgs/utils/ssim.py), computed by
two HIP kernels (csrc/loss.hip) on the [H,W,3] images the rasterizer produces, with the gradient with respect to
`pred`.  Pinned by golden vectors generated from the reference module itself (tests/golden/make_ssim_golden.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from ._lib import call, ptr, require_gpu, stream_of


def _mask_u8(mask: Optional[Tensor], H: int, W: int) -> Optional[Tensor]:
    """[H,W] uint8 view of a pixel mask for the kernels.  A bool mask is REINTERPRETED (True is the byte 1): `.to(torch.uint8)`
    is a copy kernel, and the loss head takes the same mask four or five times per iteration."""
    if mask is None:
        return None
    m = mask.reshape(H, W).contiguous()
    return m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)


class _MaskedSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gt, pred, mask, win_sigma, data_range, K1, K2):
        require_gpu(gt, pred, mask)
        H, W = pred.shape[:2]
        gt_c = gt.detach().to(torch.float32).contiguous()
        pred_c = pred.detach().to(torch.float32).contiguous()
        mask_c = _mask_u8(mask, H, W)
        dev = pred.device
        n = C.c_size_t(0)
        call("mtgs_ssim_workspace_floats", W, H, C.byref(n))
        partials = torch.empty(n.value, dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad[1]
        gmaps = torch.empty(((H - 10), (W - 10), 9), dtype=torch.float32, device=dev) if need else None
        call("mtgs_ssim_fwd", W, H, ptr(gt_c), ptr(pred_c), ptr(mask_c), float(win_sigma), float(data_range), float(K1),
             float(K2), ptr(gmaps), ptr(partials), ptr(out), stream_of(pred))
        ctx.save_for_backward(gt_c, pred_c, gmaps, out)
        ctx.dims = (H, W, float(win_sigma), pred.dtype)
        return out[0]

    @staticmethod
    def backward(ctx, v_out):
        gt_c, pred_c, gmaps, out = ctx.saved_tensors
        H, W, win_sigma, dtype = ctx.dims
        v = v_out.to(torch.float32).reshape(1).contiguous()
        v_pred = torch.empty_like(pred_c)
        call("mtgs_ssim_bwd", W, H, ptr(gt_c), ptr(pred_c), ptr(gmaps), win_sigma, ptr(v), ptr(out), ptr(v_pred),
             stream_of(pred_c))
        return None, v_pred.to(dtype), None, None, None, None, None


def masked_ssim(gt: Tensor, pred: Tensor, mask: Optional[Tensor] = None, win_sigma: float = 1.5,
                data_range: float = 1.0, K: Tuple[float, float] = (0.01, 0.03)) -> Tensor:
    """gt, pred: [H, W, 3]; mask: [H, W, 1] / [H, W] bool or None.  Returns the scalar the reference's MaskedSSIM
    returns (mean of the SSIM map over the masked elements; the mask is cropped by the 5-pixel window margin).
    Differentiable with respect to `pred` (the reference's second argument); `gt` gets no gradient."""
    assert pred.dim() == 3 and pred.shape[2] == 3 and gt.shape == pred.shape, (gt.shape, pred.shape)
    H, W = pred.shape[:2]
    if H <= 10 or W <= 10:
        raise ValueError(f"masked_ssim: image {H}x{W} is smaller than the 11x11 window")
    if mask is not None:
        assert mask.numel() == H * W, mask.shape
    if gt.requires_grad:
        raise NotImplementedError("masked_ssim: gradient with respect to gt (the first argument) is not implemented")
    return _MaskedSSIM.apply(gt, pred, mask, win_sigma, data_range, K[0], K[1])


class _MaskedL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gt, pred, mask):
        require_gpu(gt, pred, mask)
        H, W = pred.shape[:2]
        gt_c = gt.detach().to(torch.float32).contiguous()
        pred_c = pred.detach().to(torch.float32).contiguous()
        mask_c = _mask_u8(mask, H, W)
        n = C.c_size_t(0)
        call("mtgs_l1_workspace_floats", W, H, C.byref(n))
        partials = torch.empty(n.value, dtype=torch.float32, device=pred.device)
        out = torch.empty(2, dtype=torch.float32, device=pred.device)
        ch = pred.shape[2]
        call("mtgs_l1_fwd", W, H, ch, ptr(gt_c), ptr(pred_c), ptr(mask_c), ptr(partials), ptr(out), stream_of(pred))
        ctx.save_for_backward(gt_c, pred_c, mask_c, out)
        ctx.dims = (H, W, ch, pred.dtype)
        return out[0]

    @staticmethod
    def backward(ctx, v_out):
        gt_c, pred_c, mask_c, out = ctx.saved_tensors
        H, W, ch, dtype = ctx.dims
        v = v_out.to(torch.float32).reshape(1).contiguous()
        v_pred = torch.empty_like(pred_c)
        call("mtgs_l1_bwd", W, H, ch, ptr(gt_c), ptr(pred_c), ptr(mask_c), ptr(v), ptr(out), ptr(v_pred), stream_of(pred_c))
        return None, v_pred.to(dtype), None


def masked_l1(gt: Tensor, pred: Tensor, mask: Optional[Tensor] = None) -> Tensor:
    """torch.abs(gt - pred)[mask.squeeze(-1)].mean() as MTGS forms its L1 loss
    (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:823; the same form carries the depth terms, :881-883 with
    [H,W,1] images, and the normal term, :934): gt, pred [H,W,C] with 1 <= C <= 8, mask [H,W,1] / [H,W] bool or None.
    One launch per direction instead of boolean-mask indexing (nonzero + gather, sorted index_put backward).
    Differentiable with respect to `pred`."""
    assert pred.dim() == 3 and 1 <= pred.shape[2] <= 8 and gt.shape == pred.shape, (gt.shape, pred.shape)
    if mask is not None:
        assert mask.numel() == pred.shape[0] * pred.shape[1], mask.shape
    if gt.requires_grad:
        raise NotImplementedError("masked_l1: gradient with respect to gt is not implemented")
    return _MaskedL1.apply(gt, pred, mask)


class _InverseDepthL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gt_depth, depth, mask, lo, hi, eps):
        require_gpu(gt_depth, depth, mask)
        H, W = depth.shape[:2]
        gt_c = gt_depth.detach().to(torch.float32).reshape(H, W).contiguous()
        d_c = depth.detach().to(torch.float32).reshape(H, W).contiguous()
        mask_c = _mask_u8(mask, H, W)
        n = C.c_size_t(0)
        call("mtgs_l1_workspace_floats", W, H, C.byref(n))
        partials = torch.empty(n.value, dtype=torch.float32, device=depth.device)
        out = torch.empty(2, dtype=torch.float32, device=depth.device)
        used = torch.empty((H, W), dtype=torch.uint8, device=depth.device)
        call("mtgs_inv_depth_l1_fwd", W, H, ptr(gt_c), ptr(d_c), ptr(mask_c), float(lo), float(hi), float(eps), ptr(used),
             ptr(partials), ptr(out), stream_of(depth))
        ctx.save_for_backward(gt_c, d_c, mask_c, out)
        ctx.cfg = (H, W, float(lo), float(hi), float(eps), depth.shape, depth.dtype)
        used = used.view(torch.bool).reshape(H, W, 1)
        ctx.mark_non_differentiable(used)
        ctx.set_materialize_grads(False)      # (no zero-filled "gradient" of the mask output)
        return out[0], used

    @staticmethod
    def backward(ctx, v_out, _v_mask):
        gt_c, d_c, mask_c, out = ctx.saved_tensors
        H, W, lo, hi, eps, shape, dtype = ctx.cfg
        if v_out is None:
            return None, None, None, None, None, None
        v = v_out.to(torch.float32).reshape(1).contiguous()
        v_d = torch.empty_like(d_c)
        call("mtgs_inv_depth_l1_bwd", W, H, ptr(gt_c), ptr(d_c), ptr(mask_c), lo, hi, eps, ptr(v), ptr(out), ptr(v_d), stream_of(d_c))
        return None, v_d.reshape(shape).to(dtype), None, None, None, None


def inverse_depth_l1(depth: Tensor, gt_depth: Tensor, mask: Optional[Tensor] = None, lo: float = 0.1, hi: float = 80.0,
                     eps: float = 1e-5):
    """MTGS's lidar depth term (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:849-858, 875-879, InverseL1):
        m = (gt_depth > lo) & (gt_depth < hi) & mask
        loss = torch.abs(1 / (gt_depth + eps) - 1 / (depth + eps))[m].mean()          (0 when m is empty)
    in one launch per direction.  depth, gt_depth [H,W,1] / [H,W]; mask [H,W,1] / [H,W] bool or None.
    Returns (loss, m [H,W,1] bool) -- the depth NCC term takes the same mask (:891).  Differentiable with respect to `depth`."""
    assert depth.numel() == gt_depth.numel() and depth.dim() in (2, 3), (depth.shape, gt_depth.shape)
    if mask is not None:
        assert mask.numel() == depth.numel(), mask.shape
    if gt_depth.requires_grad:
        raise NotImplementedError("inverse_depth_l1: gradient with respect to gt_depth is not implemented")
    return _InverseDepthL1.apply(gt_depth, depth, mask, lo, hi, eps)


class _Combine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stacked, weights, guard_mask, constant):
        require_gpu(stacked)
        n = stacked.numel()
        w = (C.c_float * n)(*weights)
        out = torch.empty(1, dtype=torch.float32, device=stacked.device)
        kept = torch.empty(1, dtype=torch.int32, device=stacked.device)
        call("mtgs_loss_combine_fwd", n, ptr(stacked), w, int(guard_mask), float(constant), ptr(out), ptr(kept), stream_of(stacked))
        ctx.save_for_backward(kept)
        ctx.cfg = (n, tuple(float(x) for x in weights))
        return out[0]

    @staticmethod
    def backward(ctx, v_out):
        (kept,) = ctx.saved_tensors
        n, weights = ctx.cfg
        v = v_out.to(torch.float32).reshape(1).contiguous()
        v_terms = torch.empty(n, dtype=torch.float32, device=v.device)
        call("mtgs_loss_combine_bwd", n, ptr(v), ptr(kept), (C.c_float * n)(*weights), ptr(v_terms), stream_of(v))
        return v_terms, None, None, None


def combine_losses(terms, weights, constant: float = 0.0, drop_if_not_finite=()) -> Tensor:
    """constant + sum_i weights[i] * terms[i] over scalar (0-dim) device tensors -- the sum of MTGS's loss dictionary
    (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:823-945: every term times its lambda; the trainer adds the values up).
    drop_if_not_finite: indices of terms that are left out when they are NaN / inf (the normal term: `if torch.isfinite(...)`,
    :939).  `0.2 * (1 - ssim)` is weight -0.2 and constant 0.2.  Two launches forward (the stack, the sum), one backward."""
    assert 1 <= len(terms) <= 16 and len(weights) == len(terms)
    stacked = torch.stack([t.reshape(()).to(torch.float32) for t in terms])
    guard = 0
    for i in drop_if_not_finite:
        guard |= 1 << int(i)
    return _Combine.apply(stacked, [float(w) for w in weights], guard, float(constant))


class _OutputHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, render, alpha, background, exposure, want_depth, normal_channel):
        require_gpu(render, alpha, background, exposure)
        D = render.shape[-1]
        H, W = render.shape[-3], render.shape[-2]
        dev = render.device
        r_c = render.detach().to(torch.float32).reshape(H, W, D).contiguous()
        a_c = alpha.detach().to(torch.float32).reshape(H, W).contiguous()
        bg_c = background.detach().to(torch.float32).reshape(3).contiguous()
        e_c = None if exposure is None else exposure.detach().to(torch.float32).reshape(12).contiguous()
        depth_ch = D - 1 if want_depth else -1
        dmax = r_c[..., depth_ch].max().reshape(1) if want_depth else None      # depth_im.detach().max() (:680)
        rgb = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
        app = torch.empty((H, W, 3), dtype=torch.float32, device=dev) if e_c is not None else None
        depth = torch.empty((H, W, 1), dtype=torch.float32, device=dev) if want_depth else None
        normal = torch.empty((H, W, 3), dtype=torch.float32, device=dev) if normal_channel >= 0 else None
        call("mtgs_head_fwd", W, H, D, depth_ch, int(normal_channel), ptr(r_c), ptr(a_c), ptr(bg_c), ptr(e_c), ptr(dmax), ptr(rgb),
             ptr(app), ptr(depth), ptr(normal), stream_of(render))
        ctx.save_for_backward(r_c, a_c, bg_c, e_c)
        ctx.cfg = (H, W, D, depth_ch, int(normal_channel), render.shape, alpha.shape, background.shape,
                   None if exposure is None else exposure.shape)
        ctx.set_materialize_grads(False)
        return rgb, app, depth, normal

    @staticmethod
    def backward(ctx, v_rgb, v_app, v_depth, v_normal):
        r_c, a_c, bg_c, e_c = ctx.saved_tensors
        H, W, D, depth_ch, normal_ch, r_shape, a_shape, bg_shape, e_shape = ctx.cfg
        dev = r_c.device
        c = lambda g: None if g is None else g.to(torch.float32).contiguous()
        v_rgb, v_app, v_depth, v_normal = c(v_rgb), c(v_app), c(v_depth), c(v_normal)
        v_render = torch.empty((H, W, D), dtype=torch.float32, device=dev)
        v_alpha = torch.empty((H, W), dtype=torch.float32, device=dev)
        need_bg, need_e = ctx.needs_input_grad[2], e_c is not None and ctx.needs_input_grad[3]
        v_bg = torch.empty(3, dtype=torch.float32, device=dev) if need_bg else None
        v_e = torch.empty(12, dtype=torch.float32, device=dev) if need_e else None
        n = C.c_size_t(0)
        call("mtgs_head_workspace_floats", W, H, C.byref(n))
        partials = torch.empty(n.value, dtype=torch.float32, device=dev)
        call("mtgs_head_bwd", W, H, D, depth_ch, normal_ch, ptr(r_c), ptr(a_c), ptr(bg_c), ptr(e_c), ptr(v_rgb), ptr(v_app),
             ptr(v_depth), ptr(v_normal), ptr(v_render), ptr(v_alpha), ptr(v_bg), ptr(v_e), ptr(partials), stream_of(r_c))
        return (v_render.reshape(r_shape), v_alpha.reshape(a_shape), None if v_bg is None else v_bg.reshape(bg_shape),
                None if v_e is None else v_e.reshape(e_shape), None, None)


def output_head(render: Tensor, alpha: Tensor, background: Tensor, exposure: Optional[Tensor] = None, depth: bool = True,
                normal_channel: int = -1):
    """What MTGSSceneModel.get_outputs derives from the rasterizer's result (one camera) before the losses
    (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:672-690), with the shipped appearance model
    (LearnableExposureRGBModel.forward, module/appearance.py:73-87), in ONE launch per direction:

        rgb            = clamp(render[..., :3] + (1 - alpha) * background, 0, 1)                      [H,W,3]
        rgb_appearance = clamp(rgb @ exposure[:3,:3] + exposure[:3,3], 0, 1)      (exposure [3,4]; None: not computed)
        depth          = where(alpha > 0, render[..., -1:], render[..., -1:].detach().max())          [H,W,1]  (depth=True)
        normal         = (n / |n| + 1) / 2, n = render[..., c:c+3]                [H,W,3]  (normal_channel = c >= 0)

    render [1,H,W,D] or [H,W,D], alpha [1,H,W,1] or [H,W,1], background [3].  Returns (rgb, rgb_appearance, depth, normal)
    with None for what was not requested.  Differentiable with respect to render, alpha, background and exposure."""
    D = render.shape[-1]
    assert render.dim() in (3, 4) and (render.dim() == 3 or render.shape[0] == 1) and D >= 3, render.shape
    assert alpha.numel() == render.numel() // D and background.numel() == 3, (alpha.shape, background.shape)
    assert exposure is None or exposure.shape == (3, 4), exposure.shape
    assert not depth or D >= 4, f"output_head: depth=True needs the depth channel behind the colours (render has {D} channels)"
    assert normal_channel < 0 or normal_channel + 3 <= D - int(bool(depth)), (normal_channel, D)
    return _OutputHead.apply(render, alpha, background, exposure, bool(depth), int(normal_channel))


_OOB_DESC = np.dtype([("n", "<i8"), ("first_block", "<i8"), ("start", "<i8"), ("means", "<u8"), ("opacities", "<u8"),
                      ("g_opacities", "<u8"), ("limit", "<f4", (3,)), ("reserved", "<f4")], align=True)


class _OobLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, radii, starts, limits, *flat):
        from ._lib import load
        from .nodes import upload_table
        require_gpu(radii, *flat)
        if load().mtgs_oob_desc_bytes() != _OOB_DESC.itemsize:
            raise RuntimeError("mtgs_oob_desc layout mismatch between libmtgs_rast.so and mtgs_amd.loss")
        n_nodes, dev = len(flat) // 2, radii.device
        means = [m.detach().to(torch.float32).contiguous() for m in flat[0::2]]
        opacs = [o.detach().to(torch.float32).reshape(-1).contiguous() for o in flat[1::2]]
        n = np.asarray([m.shape[0] for m in means], dtype=np.int64)
        r = radii.reshape(-1)
        r = (r if r.dtype == torch.int32 else r.to(torch.int32)).contiguous()
        st = np.asarray(starts, dtype=np.int64)
        assert st.shape == n.shape and (n_nodes == 0 or ((st >= 0).all() and int((st + n).max()) <= r.numel())), (st, n, r.shape)
        tab = np.zeros(n_nodes, dtype=_OOB_DESC)
        nblk = (n + 255) // 256
        tab["n"], tab["start"], tab["first_block"] = n, st, np.cumsum(nblk) - nblk
        tab["means"] = [m.data_ptr() for m in means]
        tab["opacities"] = [o.data_ptr() for o in opacs]
        tab["limit"] = np.asarray(limits, dtype=np.float32).reshape(n_nodes, 3)
        blocks = int(nblk.sum())
        flags = torch.empty(max(n_nodes, 1), dtype=torch.int32, device=dev)
        partials = torch.empty(max(2 * blocks, 1), dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        tab_dev = upload_table(tab, dev) if n_nodes else None
        call("mtgs_oob_fwd", n_nodes, ptr(tab_dev), blocks, ptr(r), ptr(flags), ptr(partials), ptr(out), stream_of(radii))
        ctx.tab, ctx.blocks, ctx.shapes = tab, blocks, [o.shape for o in flat[1::2]]
        ctx.save_for_backward(flags, out, *means, *opacs)
        return out[0]

    @staticmethod
    def backward(ctx, v_out):
        from .nodes import upload_table
        flags, out, *rest = ctx.saved_tensors
        n_nodes = len(ctx.shapes)
        if n_nodes == 0:
            return (None, None, None)
        dev = out.device
        sizes = [int(k) for k in ctx.tab["n"]]
        g_flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        tab = ctx.tab.copy()
        off = np.cumsum(ctx.tab["n"]) - ctx.tab["n"]
        tab["g_opacities"] = np.uint64(g_flat.data_ptr()) + np.uint64(4) * off.astype(np.uint64)
        v = v_out.to(torch.float32).reshape(1).contiguous()
        tab_dev = upload_table(tab, dev)
        call("mtgs_oob_bwd", n_nodes, ptr(tab_dev), ctx.blocks, ptr(flags), ptr(v), ptr(out), stream_of(out))
        grads = []
        for g, shape in zip(g_flat.split(sizes), ctx.shapes):
            grads += [None, g.reshape(shape)]
        return (None, None, None) + tuple(grads)


def oob_loss(nodes, radii: Tensor, starts, tolerance: float = 1.5) -> Tensor:
    """The out-of-box regulariser of MTGS's rigid object nodes (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:949-967,
    `oob_lambda` = 1.0 in config/MTGS.py) for every rigid node of the frame in one pass, without the per-node host
    synchronisations of the reference loop.  nodes: sequence of (means [n,3] -- the node's LOCAL means --, opacities [n,1]
    logits, instance_size (3 floats)); radii: info["radii"] of the frame; starts[i]: offset of node i in the collected
    arrays.  Returns  sum over { Gaussians of nodes with a visible Gaussian whose |mean| exceeds instance_size / 2 +
    tolerance on some axis } of -log(1 - sigmoid(opacity) + 1e-6), divided by their number (0 when there is none).
    Differentiable with respect to the opacities."""
    flat, limits = [], []
    for means, opacities, size in nodes:
        assert means.dim() == 2 and means.shape[1] == 3 and opacities.numel() == means.shape[0], (means.shape, opacities.shape)
        flat += [means, opacities]
        limits.append([float(s) / 2 + float(tolerance) for s in size])
    return _OobLoss.apply(radii, tuple(int(s) for s in starts), limits, *flat)


class _DepthNcc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, mask, patch_size, stride):
        require_gpu(pred, gt, mask)
        H, W = pred.shape[0], pred.shape[1]
        dev = pred.device
        p_c = pred.detach().to(torch.float32).reshape(H, W).contiguous()
        g_c = gt.detach().to(torch.float32).reshape(H, W).contiguous()
        m_c = _mask_u8(mask, H, W)
        n = C.c_int64(0)
        call("mtgs_ncc_patches", W, H, patch_size, stride, C.byref(n))
        stats = torch.empty(n.value * 6, dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        call("mtgs_ncc_fwd", W, H, patch_size, stride, ptr(p_c), ptr(g_c), ptr(m_c), ptr(stats), ptr(out), stream_of(pred))
        ctx.save_for_backward(p_c, g_c, stats, out)
        ctx.cfg = (H, W, patch_size, stride, pred.shape, pred.dtype)
        return out[0]

    @staticmethod
    def backward(ctx, v_out):
        p_c, g_c, stats, out = ctx.saved_tensors
        H, W, patch_size, stride, shape, dtype = ctx.cfg
        v = v_out.to(torch.float32).reshape(1).contiguous()
        v_pred = torch.empty_like(p_c)
        call("mtgs_ncc_bwd", W, H, patch_size, stride, ptr(p_c), ptr(g_c), ptr(stats), ptr(v), ptr(out), ptr(v_pred), stream_of(p_c))
        return v_pred.reshape(shape).to(dtype), None, None, None, None


def depth_ncc_loss(pred_depth: Tensor, gt_depth: Tensor, patch_size: int = 32, stride: int = 16,
                   mask: Optional[Tensor] = None) -> Tensor:
    """calculate_depth_ncc_loss(pred_depth, gt_depth, patch_size, stride, mask=mask)
    (/root/reference/mtgs/utils/geometric_loss.py:322-348; called at mtgs_scene_graph.py:886-894 with the config's
    ncc_patch_size = 32, ncc_stride = 16): one minus the mean, over the patches whose mask is entirely set, of the normalised
    cross-correlation between the two depth images.  pred_depth, gt_depth [H,W,1] (or [H,W]), mask [H,W,1] bool.  Two
    launches forward, one backward, no host synchronisation; differentiable with respect to pred_depth."""
    assert pred_depth.dim() in (2, 3) and pred_depth.numel() == pred_depth.shape[0] * pred_depth.shape[1], pred_depth.shape
    assert gt_depth.numel() == pred_depth.numel() and (mask is None or mask.numel() == pred_depth.numel()), (gt_depth.shape,)
    assert patch_size > 0 and stride > 0
    if gt_depth.requires_grad:
        raise NotImplementedError("depth_ncc_loss: gradient with respect to gt_depth is not implemented")
    return _DepthNcc.apply(pred_depth, gt_depth, mask, int(patch_size), int(stride))


class _TvLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image):
        require_gpu(image)
        H, W, Cc = image.shape[-3], image.shape[-2], image.shape[-1]
        x = image.detach().to(torch.float32).reshape(H, W, Cc).contiguous()
        n = C.c_size_t(0)
        call("mtgs_tv_workspace_floats", W, H, Cc, C.byref(n))
        partials = torch.empty(n.value, dtype=torch.float32, device=x.device)
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        call("mtgs_tv_fwd", W, H, Cc, ptr(x), ptr(partials), ptr(out), stream_of(x))
        ctx.save_for_backward(x)
        ctx.cfg = (H, W, Cc, image.shape, image.dtype)
        return out[0]

    @staticmethod
    def backward(ctx, v_out):
        (x,) = ctx.saved_tensors
        H, W, Cc, shape, dtype = ctx.cfg
        v = v_out.to(torch.float32).reshape(1).contiguous()
        v_x = torch.empty_like(x)
        call("mtgs_tv_bwd", W, H, Cc, ptr(x), ptr(v), ptr(v_x), stream_of(x))
        return v_x.reshape(shape).to(dtype)


def tv_loss(image: Tensor) -> Tensor:
    """TVLoss()(image) of MTGS's normal term (/root/reference/mtgs/utils/geometric_loss.py:293-303, used at
    mtgs_scene_graph.py:931-932): mean |image[:, :-1] - image[:, 1:]| + mean |image[:-1] - image[1:]| for one image
    [H,W,C] (or [1,H,W,C]).  Two launches forward, one backward."""
    assert image.dim() in (3, 4) and (image.dim() == 3 or image.shape[0] == 1), image.shape
    return _TvLoss.apply(image)
