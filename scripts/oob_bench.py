#!/usr/bin/env python3
"""The out-of-box regulariser (mtgs_scene_graph.py:949-967, oob_lambda = 1.0 in config/MTGS.py) on a frame with `--objects`
rigid nodes among 2M collected Gaussians: the reference's per-node Python loop (model_id comparison over all Gaussians,
boolean-mask gathers, two host synchronisations per node) against mtgs_amd.loss.oob_loss.  Forward + backward, wall-clock."""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mtgs_amd.loss import oob_loss  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--objects", type=int, default=100)
ap.add_argument("--object-size", type=int, default=3000)
ap.add_argument("--static", type=int, default=1_700_000)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
sizes = [max(1, int(args.object_size * (0.3 + 1.4 * torch.rand(1, generator=g).item()))) for _ in range(args.objects)]
starts, s = [], args.static
for k in sizes:
    starts.append(s); s += k
total = s
radii = (torch.randint(0, 30, (1, total), generator=g) * (torch.rand(1, total, generator=g) < 0.15)).int().to(dev)
model_id = torch.zeros(total, dtype=torch.long)
for i, (st, k) in enumerate(zip(starts, sizes)):
    model_id[st:st + k] = i + 1
model_id = model_id.to(dev)
nodes = [((torch.randn(k, 3, generator=g) * 1.5).to(dev), (torch.randn(k, 1, generator=g)).to(dev).requires_grad_(True), [4.5, 2.0, 1.8])
         for k in sizes]


def chain():
    oob, overall = 0.0, 0
    visible_mask = (radii > 0).flatten()
    for i, (means, opac, size) in enumerate(nodes):
        model_mask = model_id == i + 1
        if visible_mask[model_mask].sum() == 0:
            continue
        instance_size = means.new_tensor(size)
        oob_mask = (means.abs() > (instance_size / 2 + 1.5)[None]).any(-1).detach()
        if oob_mask.sum() != 0:
            oob = oob + (-torch.log(1 - opac[oob_mask].sigmoid() + 1e-6)).sum()
            overall = overall + oob_mask.sum()
    return oob / overall


def fused():
    return oob_loss(nodes, radii, starts, tolerance=1.5)


def wall(fn):
    for _ in range(2):
        for _, o, _ in nodes:
            o.grad = None
        fn().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        for _, o, _ in nodes:
            o.grad = None
        fn().backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.reps * 1e3


a, b = float(chain()), float(fused())
print(f"{args.objects} rigid nodes in {total} Gaussians: loss chain {a:.6f} fused {b:.6f}")
tc, tf = wall(chain), wall(fused)
print(f"out-of-box regulariser, fwd+bwd: per-node loop {tc:.2f} ms -> fused {tf:.3f} ms ({tc / tf:.0f}x)")
