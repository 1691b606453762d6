"""Differentiable torch restatement of the gsplat-1.4.0 path (any dtype, normally fp64).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see gsplat_oracle.c).  This is a SECOND, independent
restatement: it is written with whole-image tensor ops and gets every gradient from torch
autograd, so it shares no backward formula with oracle/gsplat_oracle.c or with the HIP kernels.
tests/ use it to check the hand-derived VJPs (projection, compensation, compositing, SH).

Follows gsplat/cuda/_torch_impl.py::{_quat_scale_to_covar_preci, _world_to_cam, _persp_proj,
_fully_fused_projection, _isect_tiles, accumulate, _eval_sh_bases_fast} (v1.4.0) and, for the
thresholds the torch implementation does not reproduce, the CUDA kernels' semantics (alpha
clamp 0.999, 1/255 skip, T <= 1e-4 stop-before-composite, sigma < 0 skip).
"""
from __future__ import annotations

import math

import torch

ALPHA_MAX = 0.999
ALPHA_MIN = 1.0 / 255.0
T_MIN = 1e-4


def quat_to_rotmat(q):
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(q.shape[:-1] + (3, 3))


def project(means, quats, scales, viewmats, Ks, W, H, eps2d=0.3, near=0.01, far=1e10,
            radius_clip=0.0):
    """Returns radii[C,N] (int64, no grad), means2d, depths, conics, compensations (all [C,N,..])."""
    Rq = quat_to_rotmat(quats)
    M = Rq * scales[:, None, :]
    covar = M @ M.transpose(-1, -2)                         # [N,3,3]
    R = viewmats[:, :3, :3]
    t = viewmats[:, :3, 3]
    mean_c = torch.einsum("cij,nj->cni", R, means) + t[:, None, :]
    covar_c = torch.einsum("cij,njk,clk->cnil", R, covar, R)
    fx, fy, cx, cy = Ks[:, 0, 0], Ks[:, 1, 1], Ks[:, 0, 2], Ks[:, 1, 2]
    x, y, z = mean_c.unbind(-1)
    tan_x, tan_y = 0.5 * W / fx, 0.5 * H / fy
    lim_xp = ((W - cx) / fx + 0.3 * tan_x)[:, None]
    lim_xn = (cx / fx + 0.3 * tan_x)[:, None]
    lim_yp = ((H - cy) / fy + 0.3 * tan_y)[:, None]
    lim_yn = (cy / fy + 0.3 * tan_y)[:, None]
    rz = 1.0 / z
    tx = z * torch.minimum(lim_xp, torch.maximum(-lim_xn, x * rz))
    ty = z * torch.minimum(lim_yp, torch.maximum(-lim_yn, y * rz))
    O = torch.zeros_like(z)
    J = torch.stack([fx[:, None] * rz, O, -fx[:, None] * tx * rz * rz,
                     O, fy[:, None] * rz, -fy[:, None] * ty * rz * rz], dim=-1).reshape(z.shape + (2, 3))
    cov2d = J @ covar_c @ J.transpose(-1, -2)
    means2d = torch.stack([fx[:, None] * x * rz + cx[:, None], fy[:, None] * y * rz + cy[:, None]], dim=-1)
    det_orig = cov2d[..., 0, 0] * cov2d[..., 1, 1] - cov2d[..., 0, 1] * cov2d[..., 1, 0]
    cov2d = cov2d + eps2d * torch.eye(2, dtype=cov2d.dtype)
    det = cov2d[..., 0, 0] * cov2d[..., 1, 1] - cov2d[..., 0, 1] * cov2d[..., 1, 0]
    comp = torch.sqrt(torch.clamp(det_orig / det, min=0.0))
    conics = torch.stack([cov2d[..., 1, 1] / det, -(cov2d[..., 0, 1] + cov2d[..., 1, 0]) / 2 / det,
                          cov2d[..., 0, 0] / det], dim=-1)
    with torch.no_grad():
        b = 0.5 * (cov2d[..., 0, 0] + cov2d[..., 1, 1])
        v1 = b + torch.sqrt(torch.clamp(b * b - det, min=0.01))
        radius = torch.ceil(3.0 * torch.sqrt(v1))
        ok = (z >= near) & (z <= far) & (det > 0) & (radius > radius_clip)
        ok &= ~((means2d[..., 0] + radius <= 0) | (means2d[..., 0] - radius >= W)
                | (means2d[..., 1] + radius <= 0) | (means2d[..., 1] - radius >= H))
        radii = torch.where(ok, radius, torch.zeros_like(radius)).long()
    return radii, means2d, z, conics, comp


def tile_rects(means2d, radii, tile_size, tw, th):
    """[C,N,4] = x0,y0,x1,y1 (half-open), following _torch_impl._isect_tiles."""
    with torch.no_grad():
        tm = means2d / tile_size
        tr = (radii.to(means2d.dtype) / tile_size)[..., None]
        lo = torch.floor(tm - tr)
        hi = torch.ceil(tm + tr)
        x0 = lo[..., 0].clamp(0, tw); y0 = lo[..., 1].clamp(0, th)
        x1 = hi[..., 0].clamp(0, tw); y1 = hi[..., 1].clamp(0, th)
        r = torch.stack([x0, y0, x1, y1], dim=-1).long()
        r[radii <= 0] = 0
    return r


def composite(means2d, conics, colors, opacities, radii, depths, W, H, tile_size=16, backgrounds=None,
              rects=None):
    """Front-to-back compositing for all pixels at once, one Gaussian per step.
    means2d[C,N,2] conics[C,N,3] colors[C,N,D] opacities[C,N]. Returns render[C,H,W,D], alpha[C,H,W,1]."""
    Cc, N, D = colors.shape
    tw, th = math.ceil(W / tile_size), math.ceil(H / tile_size)
    if rects is None:
        rects = tile_rects(means2d, radii, tile_size, tw, th)
    dt = means2d.dtype
    py, px = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    tyy, txx = py // tile_size, px // tile_size
    fx, fy = px.to(dt) + 0.5, py.to(dt) + 0.5
    renders, alphas = [], []
    for c in range(Cc):
        vis = torch.nonzero(radii[c] > 0).flatten()
        # canonical order = (depth as fp32 bits, gaussian index); fp32 cast mirrors the sort key
        d32 = depths[c, vis].detach().to(torch.float32)
        order = vis[torch.sort(d32, stable=True).indices]
        T = torch.ones(H, W, dtype=dt)
        done = torch.zeros(H, W, dtype=torch.bool)
        out = torch.zeros(H, W, D, dtype=dt)
        for g in order.tolist():
            x0, y0, x1, y1 = rects[c, g].tolist()
            if x1 <= x0 or y1 <= y0:
                continue
            in_rect = (txx >= x0) & (txx < x1) & (tyy >= y0) & (tyy < y1)
            dx = means2d[c, g, 0] - fx
            dy = means2d[c, g, 1] - fy
            a, b, cc = conics[c, g]
            sigma = 0.5 * (a * dx * dx + cc * dy * dy) + b * dx * dy
            alpha = torch.clamp(opacities[c, g] * torch.exp(-sigma), max=ALPHA_MAX)
            ok = in_rect & ~done & (sigma >= 0) & (alpha >= ALPHA_MIN)
            next_T = T * (1 - alpha)
            stop = ok & (next_T <= T_MIN)
            done = done | stop
            use = ok & ~stop
            w = torch.where(use, alpha * T, torch.zeros_like(T))
            out = out + w[..., None] * colors[c, g]
            T = torch.where(use, next_T, T)
        if backgrounds is not None:
            out = out + T[..., None] * backgrounds[c]
        renders.append(out)
        alphas.append((1 - T)[..., None])
    return torch.stack(renders), torch.stack(alphas)


def sh_bases(degree, dirs):
    """_eval_sh_bases_fast restated (degree <= 4); dirs are normalised inside."""
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    x, y, z = d.unbind(-1)
    b = [torch.full_like(x, 0.2820947917738781)]
    if degree >= 1:
        b += [-0.48860251190292 * y, 0.48860251190292 * z, -0.48860251190292 * x]
    if degree >= 2:
        z2 = z * z
        fTmp0B = -1.092548430592079 * z
        fC1 = x * x - y * y
        fS1 = 2 * x * y
        pSH6 = 0.9461746957575601 * z2 - 0.3153915652525201
        b += [0.5462742152960395 * fS1, fTmp0B * y, pSH6, fTmp0B * x, 0.5462742152960395 * fC1]
    if degree >= 3:
        fTmp0C = -2.285228997322329 * z2 + 0.4570457994644658
        fTmp1B = 1.445305721320277 * z
        fC2 = x * fC1 - y * fS1
        fS2 = x * fS1 + y * fC1
        pSH12 = z * (1.865881662950577 * z2 - 1.119528997770346)
        b += [-0.5900435899266435 * fS2, fTmp1B * fS1, fTmp0C * y, pSH12, fTmp0C * x, fTmp1B * fC1,
              -0.5900435899266435 * fC2]
    if degree >= 4:
        fTmp0D = z * (-4.683325804901025 * z2 + 2.007139630671868)
        fTmp1C = 3.31161143515146 * z2 - 0.47308734787878
        fTmp2B = -1.770130769779931 * z
        fC3 = x * fC2 - y * fS2
        fS3 = x * fS2 + y * fC2
        pSH20 = 1.984313483298443 * z * pSH12 - 1.006230589874905 * pSH6
        b += [0.6258357354491763 * fS3, fTmp2B * fS2, fTmp1C * fS1, fTmp0D * y, pSH20, fTmp0D * x,
              fTmp1C * fC1, fTmp2B * fC2, 0.6258357354491763 * fC3]
    return torch.stack(b, dim=-1)


def spherical_harmonics(degree, dirs, coeffs, masks=None):
    nb = (degree + 1) ** 2
    out = (sh_bases(degree, dirs)[..., :, None] * coeffs[..., :nb, :]).sum(dim=-2)
    if masks is not None:
        out = torch.where(masks[..., None], out, torch.zeros_like(out))
    return out


def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height,
                  near_plane=0.01, far_plane=1e10, radius_clip=0.0, eps2d=0.3, sh_degree=None,
                  tile_size=16, backgrounds=None, render_mode="RGB", rasterize_mode="classic"):
    """gsplat/rendering.py::rasterization restated (packed=False) on the functions above."""
    Cc = viewmats.shape[0]
    radii, means2d, depths, conics, comp = project(
        means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip)
    opac = opacities[None, :].expand(Cc, -1)
    if rasterize_mode == "antialiased":
        opac = opac * comp
    if sh_degree is None:
        cols = colors.expand(Cc, -1, -1) if colors.dim() == 2 else colors
    else:
        c2w = torch.inverse(viewmats)
        dirs = means[None] - c2w[:, None, :3, 3]
        shs = colors.expand(Cc, -1, -1, -1) if colors.dim() == 3 else colors
        cols = torch.clamp_min(spherical_harmonics(sh_degree, dirs, shs, masks=radii > 0) + 0.5, 0.0)
    bg = backgrounds
    if render_mode in ("RGB+D", "RGB+ED"):
        cols = torch.cat([cols, depths[..., None]], dim=-1)
        if bg is not None:
            bg = torch.cat([bg, torch.zeros(Cc, 1, dtype=bg.dtype)], dim=-1)
    elif render_mode in ("D", "ED"):
        cols = depths[..., None]
        if bg is not None:
            bg = torch.zeros(Cc, 1, dtype=bg.dtype)
    render, alpha = composite(means2d, conics, cols, opac, radii, depths, width, height, tile_size, bg)
    if render_mode in ("ED", "RGB+ED"):
        render = torch.cat([render[..., :-1], render[..., -1:] / alpha.clamp(min=1e-10)], dim=-1)
    meta = dict(radii=radii, means2d=means2d, depths=depths, conics=conics, opacities=opac,
                compensations=comp, colors=cols)
    return render, alpha, meta
