#!/usr/bin/env python3
"""MTGSSceneModel.get_gaussians on a scene graph with many object nodes (mtgs_scene_graph.py:408-461): a background node, a
road node and `--objects` rigid nodes of a few thousand Gaussians each (one per object instance in view), forward +
backward of the per-node activations, wall-clock per step (host + GPU: with hundreds of nodes the HOST is the limit):

  chain   : the reference's PyTorch operator chain per node (exp / normalise / sigmoid / cat / SH / clamp; rigid nodes:
            get_object_pose -- index + normalise the per-frame pose parameters --, quat_to_rotmat, matmul, quat_mult) + torch.cat of the per-node outputs, SH through the HIP spherical_harmonics
  pernode : mtgs_amd.nodes.node_gaussians per node (one fused kernel per node and direction) + torch.cat
  batched : mtgs_amd.nodes.collect_gaussians (ONE launch per direction for the whole scene graph)
"""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mtgs_amd import spherical_harmonics  # noqa: E402
from mtgs_amd.nodes import collect_gaussians, node_gaussians  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--objects", type=int, default=150)
ap.add_argument("--object-size", type=int, default=3000)
ap.add_argument("--background", type=int, default=1_200_000)
ap.add_argument("--road", type=int, default=350_000)
ap.add_argument("--frames", type=int, default=200, help="frames of every object's pose parameters instance_quats[F,4] / instance_trans[F,3]")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--only", default="chain,pernode,batched")
args = ap.parse_args()
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
K = 16


def params(n, rigid):
    p = {"means": torch.randn(n, 3, generator=g) * 20, "scales": torch.randn(n, 3, generator=g) - 2,
         "quats": torch.randn(n, 4, generator=g), "opacities": torch.randn(n, 1, generator=g),
         "features_dc": torch.randn(n, 3, generator=g), "features_rest": torch.randn(n, K - 1, 3, generator=g) * 0.1}
    if rigid:   # RigidSubModel: one pose per frame of the traversal (rigid_node.py:107-108)
        p["instance_quats"], p["instance_trans"] = torch.randn(args.frames, 4, generator=g), torch.randn(args.frames, 3, generator=g) * 10
    return {k: v.to(dev).requires_grad_(True) for k, v in p.items()}


sizes = [args.background, args.road] + [max(1, int(args.object_size * (0.3 + 1.4 * torch.rand(1, generator=g).item())))
                                        for _ in range(args.objects)]
nodes = [params(n, i >= 2) for i, n in enumerate(sizes)]
frame_of = [None, None] + [int(torch.randint(0, args.frames, (1,), generator=g)) for _ in range(args.objects)]
total = sum(sizes)
c2w = torch.eye(4, device=dev)[None, :3]
cot = {"means": torch.randn(total, 3, device=dev), "scales": torch.randn(total, 3, device=dev),
       "quats": torch.randn(total, 4, device=dev), "opacities": torch.randn(total, device=dev), "rgbs": torch.randn(total, 3, device=dev)}


def quat_to_rotmat(q):   # wxyz, as mtgs utils.quat_to_rotmat (no normalisation)
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                        2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(3, 3)


def quat_mult(a, b):
    w1, x1, y1, z1 = a.unbind(-1)
    w2, x2, y2, z2 = b.unbind(-1)
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], -1)


def pose_of(p, f):   # get_object_pose (rigid_node.py:139-144)
    return p["instance_quats"][f] / p["instance_quats"][f].norm(dim=-1, keepdim=True), p["instance_trans"][f]


def chain():
    parts = []
    for p, f in zip(nodes, frame_of):
        means, quats = p["means"], p["quats"] / p["quats"].norm(dim=-1, keepdim=True)
        if f is not None:
            iq, it = pose_of(p, f)
            means = means @ quat_to_rotmat(iq).T + it
            quats = quat_mult(iq[None], quats)
        colors = torch.cat((p["features_dc"][:, None, :], p["features_rest"]), dim=1)
        viewdirs = means.detach() - c2w[..., :3, 3]
        viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
        parts.append({"means": means, "scales": torch.exp(p["scales"]), "quats": quats, "opacities": torch.sigmoid(p["opacities"]).squeeze(-1),
                      "rgbs": torch.clamp(spherical_harmonics(3, viewdirs, colors) + 0.5, 0.0, 1.0)})
    return {k: torch.cat([q[k] for q in parts], 0) for k in cot}


def pernode():
    parts = []
    for p, f in zip(nodes, frame_of):
        iq, it = pose_of(p, f) if f is not None else (None, None)
        parts.append(node_gaussians(p["means"], p["scales"], p["quats"], p["opacities"], p["features_dc"], p["features_rest"], c2w, 3, 3,
                                    instance_quat=iq, instance_trans=it))
    return {k: torch.cat([q[k] for q in parts], 0) for k in cot}


def batched():
    return collect_gaussians([p if f is None else dict(p, frame_idx=f) for p, f in zip(nodes, frame_of)], c2w, 3, 3)


def step(fn):
    for p in nodes:
        for v in p.values():
            v.grad = None
    out = fn()
    torch.autograd.backward([out[k] for k in cot], [cot[k] for k in cot])


def wall(fn):
    for _ in range(2):
        step(fn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        step(fn)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.reps * 1e3


print(f"{len(nodes)} nodes ({args.objects} rigid objects), {total} Gaussians, forward + backward, wall-clock ms per step")
for name in args.only.split(","):
    print(f"  {name:8s} {wall({'chain': chain, 'pernode': pernode, 'batched': batched}[name]):8.2f} ms")
