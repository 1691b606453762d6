#!/usr/bin/env python3
"""Per-node activations at the headline size (2M Gaussians, K = 16): the operator chain of
VanillaGaussianSplattingModel.get_gaussians as MTGS runs it on the drop-in (PyTorch ops + the HIP
spherical_harmonics) against mtgs_amd.nodes.node_gaussians (one HIP kernel per direction).  Forward + backward."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mtgs_amd import spherical_harmonics  # noqa: E402
from mtgs_amd.nodes import node_gaussians  # noqa: E402

dev = torch.device("cuda")
N, K = 2_000_000, 16
g = torch.Generator().manual_seed(0)
P = {"means": torch.randn(N, 3, generator=g) * 20, "scales": torch.randn(N, 3, generator=g) - 2,
     "quats": torch.randn(N, 4, generator=g), "opacities": torch.randn(N, 1, generator=g),
     "features_dc": torch.randn(N, 3, generator=g), "features_rest": torch.randn(N, K - 1, 3, generator=g) * 0.1}
P = {k: v.to(dev).requires_grad_(True) for k, v in P.items()}
c2w = torch.eye(4, device=dev)[None, :3]
cot = {"scales": torch.randn(N, 3, device=dev), "quats": torch.randn(N, 4, device=dev),
       "opacities": torch.randn(N, device=dev), "rgbs": torch.randn(N, 3, device=dev)}


def chain():
    out = {"scales": torch.exp(P["scales"]), "quats": P["quats"] / P["quats"].norm(dim=-1, keepdim=True),
           "opacities": torch.sigmoid(P["opacities"]).squeeze(-1)}
    colors = torch.cat((P["features_dc"][:, None, :], P["features_rest"]), dim=1)
    viewdirs = P["means"].detach() - c2w[..., :3, 3]
    viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
    out["rgbs"] = torch.clamp(spherical_harmonics(3, viewdirs, colors) + 0.5, 0.0, 1.0)
    return out


def fused():
    return node_gaussians(P["means"], P["scales"], P["quats"], P["opacities"], P["features_dc"], P["features_rest"], c2w, 3, 3)


def step(fn):
    for p in P.values():
        p.grad = None
    out = fn()
    torch.autograd.backward([out[k] for k in cot], [cot[k] for k in cot])


def t(fn, reps=10):
    for _ in range(3):
        step(fn)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        step(fn)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


a, b = t(chain), t(fused)
alg = N * (40 + 12 * K + 44) + N * (44 + 12 + 44 + 12 * K)
print(f"N={N} K={K}: operator chain {a:.0f} us, fused node kernels {b:.0f} us ({a / b:.1f}x); "
      f"fused: {alg / 1e6:.0f} MB algorithmic -> {alg / b / 1e3:.0f} GB/s")
