#!/usr/bin/env python3
"""Development: stage times of the fused rasterization path (HIP events around the C-ABI calls it makes), headline
workload by default.  `--lib` selects an A/B build (scripts/build_variant.py)."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mtgs_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--extra-channels", type=int, default=0)
ap.add_argument("--lib", default=None)
ap.add_argument("--sh", action="store_true", help="MTGS's step: spherical_harmonics -> clamp(+0.5) -> rasterization (deferred colours)")
args = ap.parse_args()
if args.lib:
    _lib.use_library(args.lib)
from mtgs_amd import rasterization, spherical_harmonics  # noqa: E402
from mtgs_amd.synthetic import make_camera, make_scene  # noqa: E402

dev = torch.device("cuda")
sc = make_scene(args.n, seed=0, sh_degree=3 if args.sh else None)
vm, K = make_camera(args.width, args.height)
P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
if args.extra_channels:
    P["colors"] = torch.cat([P["colors"].detach(), torch.rand(args.n, args.extra_channels, device=dev)], -1).requires_grad_(True)
vm, K = vm.to(dev).requires_grad_(True), K.to(dev)
g = torch.Generator().manual_seed(1)
D = (3 if args.sh else P["colors"].shape[-1]) + 1
Gc, Ga = torch.randn(1, args.height, args.width, D, generator=g).to(dev), torch.randn(1, args.height, args.width, 1, generator=g).to(dev)
names = ["mtgs_front_fwd", "mtgs_bin3_build", "mtgs_blend_fwd_packed", "mtgs_blend_bwd_packed", "mtgs_project_bwd", "mtgs_project_bwd_zeroed"]
if args.sh:
    names[1:1] = ["mtgs_vis_color_fwd_dirs"]
    names[-2:-2] = ["mtgs_vis_color_bwd_dirs"]
    cam = torch.inverse(vm.detach())[0, :3, 3]
    dirs = (P["means"].detach().to(dev) - cam).contiguous()


def step():
    cols = torch.clamp(spherical_harmonics(3, dirs, P["coeffs"]) + 0.5, 0.0, 1.0) if args.sh else P["colors"]
    r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], cols, vm, K, args.width, args.height,
                               packed=False, render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
    torch.autograd.backward([r, a], [Gc, Ga])
    return info


for _ in range(3):
    info = step()
torch.cuda.synchronize()
_lib.time_calls(names)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(args.reps):
    step()
t1.record()
torch.cuda.synchronize()
ms = _lib.timed_ms()
print(f"lib={args.lib or 'in-tree'} N={args.n} {args.width}x{args.height} D={D} n_vis={int((info['radii'] > 0).sum())} M={info['flatten_ids'].numel()}")
for n in names:
    v = sorted(ms.get(n) or [0.0])
    print(f"  {n:26s} median {v[len(v) // 2] * 1e3:8.1f} us   min {v[0] * 1e3:8.1f} us")
print(f"  whole step (raster only)   {t0.elapsed_time(t1) / args.reps * 1e3:8.1f} us")
