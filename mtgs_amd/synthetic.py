"""WB-v1 synthetic road-block generator.

The reference has no synthetic scene generator (its inputs are nuPlan road blocks, 20 TB, not
reachable); WB-v1 produces Gaussians at road-block scale (parser box is +-1.2x a 70-100 m block,
/root/reference/mtgs/dataset/nuplan_dataparser.py:61,388-390) and a pinhole camera like the nuPlan
rig (1920x1080, /root/reference/mtgs/dataset/nuplan_dataparser.py:68-69,413-423).

Everything is drawn on the CPU with a seeded torch.Generator in a FIXED order so that the same
seed gives bit-identical inputs on every machine.
"""
from __future__ import annotations

import math

import torch

SH_C0 = 0.2820947917738781


def make_scene(N: int, seed: int = 0, sh_degree: int | None = None, extent=(50.0, 7.5, 50.0)):
    """Returns dict(means[N,3], quats[N,4] wxyz unit, scales[N,3] (post-exp), opacities[N]
    (post-sigmoid), colors[N,3] or coeffs[N,K,3])."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    means = (torch.rand(N, 3, generator=g) * 2 - 1) * torch.tensor(extent)
    lo, hi = math.log(0.02), math.log(0.2)
    scales = torch.exp(torch.rand(N, 3, generator=g) * (hi - lo) + lo)
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=-1)
    opacities = torch.sigmoid(torch.randn(N, generator=g))
    out = dict(means=means, quats=quats, scales=scales, opacities=opacities)
    if sh_degree is None:
        out["colors"] = torch.rand(N, 3, generator=g)
    else:
        K = (sh_degree + 1) ** 2
        coeffs = torch.empty(N, K, 3)
        coeffs[:, 0, :] = (torch.rand(N, 3, generator=g) - 0.5) / SH_C0
        if K > 1:
            coeffs[:, 1:, :] = 0.1 * torch.randn(N, K - 1, 3, generator=g)
        out["coeffs"] = coeffs
    return out


def make_camera(W: int, H: int, yaw_deg: float = 0.0):
    """viewmat[1,4,4] (world->camera, OpenCV: +z forward) at the origin rotated `yaw_deg` about +y;
    Ks[1,3,3] with fx = fy = 0.8 W, principal point at the image centre."""
    a = math.radians(yaw_deg)
    c, s = math.cos(a), math.sin(a)
    # camera-to-world rotation about +y by yaw; viewmat is its inverse (transpose)
    R_c2w = torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])
    vm = torch.eye(4)
    vm[:3, :3] = R_c2w.T
    K = torch.tensor([[0.8 * W, 0.0, W / 2.0], [0.0, 0.8 * W, H / 2.0], [0.0, 0.0, 1.0]])
    return vm[None].contiguous(), K[None].contiguous()


def mtgs_c2w_to_viewmat(c2w: torch.Tensor) -> torch.Tensor:
    """camera_to_world (nerfstudio/OpenGL) -> gsplat viewmat, as MTGS does it
    (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:601-613): flip y,z, then analytic inverse."""
    R = c2w[:3, :3] @ torch.diag(torch.tensor([1.0, -1.0, -1.0], dtype=c2w.dtype, device=c2w.device))
    T = c2w[:3, 3:4]
    vm = torch.eye(4, dtype=c2w.dtype, device=c2w.device)
    vm[:3, :3] = R.T
    vm[:3, 3:4] = -R.T @ T
    return vm[None]
