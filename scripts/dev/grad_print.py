#!/usr/bin/env python3
"""Development: gradient checksums of one headline-like step with a given library build (A/B builds must print the same numbers up
to the order of the compositing atomics)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mtgs_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.use_library(sys.argv[1])
from mtgs_amd import rasterization, spherical_harmonics
from mtgs_amd.synthetic import make_camera, make_scene
dev = torch.device("cuda")
N, W, H = 500_000, 1920, 1080
sc = make_scene(N, seed=0, sh_degree=3)
vm, K = make_camera(W, H)
P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
vm, K = vm.to(dev), K.to(dev)
cam = torch.inverse(vm)[0, :3, 3]
g = torch.Generator().manual_seed(1)
Gc, Ga = torch.randn(1, H, W, 4, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)
rgb = torch.clamp(spherical_harmonics(3, P["means"].detach() - cam, P["coeffs"]) + 0.5, 0.0, 1.0)
r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, W, H, packed=False, render_mode="RGB+ED",
                           rasterize_mode="antialiased", absgrad=True)
info["means2d"].retain_grad()
torch.autograd.backward([r, a], [Gc, Ga])
print({k: (float(v.grad.double().sum()), float(v.grad.double().abs().sum())) for k, v in P.items()},
      float(info["means2d"].absgrad.double().sum()))
