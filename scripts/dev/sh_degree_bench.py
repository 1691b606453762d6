#!/usr/bin/env python3
"""Development: gsplat's own call style rasterization(colors=coeffs, sh_degree=3) at the headline size, forward + backward."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mtgs_amd import rasterization
from mtgs_amd.synthetic import make_camera, make_scene
dev = torch.device("cuda")
W, H, N = 1920, 1080, 2_000_000
sc = make_scene(N, seed=0, sh_degree=3)
vm, K = make_camera(W, H)
P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
vmd, Kd = vm.to(dev).requires_grad_(True), K.to(dev)
g = torch.Generator().manual_seed(1)
Gc, Ga = torch.randn(1, H, W, 4, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)
def step():
    for p in list(P.values()) + [vmd]: p.grad = None
    r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["coeffs"], vmd, Kd, W, H, sh_degree=3, packed=False,
                               render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
    torch.autograd.backward([r, a], [Gc, Ga])
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print(f"rasterization(sh_degree=3) fwd+bwd, 2M Gaussians 1920x1080: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step")
