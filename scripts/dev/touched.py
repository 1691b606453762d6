#!/usr/bin/env python3
"""Development: how many frustum-visible Gaussians of the headline frame receive any gradient from the compositing backward
(the others are occluded: their tile lists terminate before they are reached)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mtgs_amd import rasterization, spherical_harmonics, wrapper  # noqa: E402
from mtgs_amd.synthetic import make_camera, make_scene  # noqa: E402

dev = torch.device("cuda")
for N, W, H in ((2_000_000, 1920, 1080), (2_000_000, 960, 540), (500_000, 1920, 1080)):
    sc = make_scene(N, seed=0, sh_degree=3)
    vm, K = make_camera(W, H)
    P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    cam = torch.inverse(vm)[0, :3, 3].to(dev)
    rgb = torch.clamp(spherical_harmonics(3, P["means"].detach() - cam, P["coeffs"]) + 0.5, 0, 1)
    r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm.to(dev), K.to(dev), W, H, packed=False,
                               render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
    dbg = wrapper._debug_rows = {}
    g = torch.Generator().manual_seed(1)
    torch.autograd.backward([r, a], [torch.randn(r.shape, generator=g).to(dev), torch.randn(a.shape, generator=g).to(dev)])
    wrapper._debug_rows = None
    G = dbg["G"][: dbg["vis_ids"].numel()]
    touched = int((G != 0).any(1).sum())
    print(f"N {N} {W}x{H}: visible {G.shape[0]}, with a gradient {touched} ({100.0 * touched / G.shape[0]:.1f} %), "
          f"intersections {info['flatten_ids'].numel()}")
