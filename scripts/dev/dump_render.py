#!/usr/bin/env python3
"""Development: render the headline scene with a given build of the library and save render / alpha / gradients, to compare
two builds bit for bit:  dump_render.py [--lib X] --out a.pt ; dump_render.py --cmp a.pt b.pt"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--out", default=None)
ap.add_argument("--cmp", nargs=2, default=None)
ap.add_argument("--n", type=int, default=2_000_000)
args = ap.parse_args()
if args.cmp:
    a, b = torch.load(args.cmp[0]), torch.load(args.cmp[1])
    for k in a:
        d = (a[k].double() - b[k].double()).abs()
        print(f"{k:12s} differing entries {int((d > 0).sum()):9d} of {d.numel():10d}   max abs diff {float(d.max()):.3e}   max |a| {float(a[k].abs().max()):.3e}")
    sys.exit(0)
from mtgs_amd import _lib
if args.lib:
    _lib.use_library(args.lib)
from mtgs_amd import rasterization
from mtgs_amd.synthetic import make_camera, make_scene
dev = torch.device("cuda")
W, H = 1920, 1080
sc = make_scene(args.n, seed=0)
vm, K = make_camera(W, H)
P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
g = torch.Generator().manual_seed(1)
Gc, Ga = torch.randn(1, H, W, 4, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)
r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm.to(dev), K.to(dev), W, H, packed=False,
                           render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
info["means2d"].retain_grad()
torch.autograd.backward([r, a], [Gc, Ga])
torch.save({"render": r.detach().cpu(), "alpha": a.detach().cpu(), "means2d.grad": info["means2d"].grad.cpu(),
            "v_opacities": P["opacities"].grad.cpu(), "v_colors": P["colors"].grad.cpu()}, args.out)
