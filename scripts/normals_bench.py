#!/usr/bin/env python3
"""Camera-space normals + the colour concatenation of `predict_normals` (the shipped MTGS.py config) at 2M Gaussians:
MTGSSceneModel._get_gaussian_camera_space_normals + torch.cat([rgbs, normals]) (mtgs_scene_graph.py:526-545, :636-638) as the
reference writes them in PyTorch, against mtgs_amd.nodes.camera_space_normals(..., rgbs=rgbs).  Forward + backward,
wall-clock (the reference's boolean-mask write synchronises the host)."""
import sys
import time
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mtgs_amd.nodes import camera_space_normals  # noqa: E402

dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
g = torch.Generator().manual_seed(0)
quats = torch.randn(N, 4, generator=g)
quats = (quats / quats.norm(dim=-1, keepdim=True)).to(dev).requires_grad_(True)
scales = torch.exp(torch.randn(N, 3, generator=g)).to(dev)
means = (torch.randn(N, 3, generator=g) * 20).to(dev)
rgbs = torch.rand(N, 3, generator=g).to(dev).requires_grad_(True)
A = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
c2w = torch.cat([A, torch.randn(3, 1, generator=g)], 1)[None].to(dev)
G = torch.randn(N, 6, generator=g).to(dev)


def quat_to_rotmat(quat):   # mtgs utils.quat_to_rotmat
    w, x, y, z = torch.unbind(quat, dim=-1)
    xx, yy, zz, xy, xz, yz, wx, wy, wz = x * x, y * y, z * z, x * y, x * z, y * z, w * x, w * y, w * z
    mat = torch.stack([1.0 - 2.0 * (yy + zz), 2.0 * (xy - wz), 2.0 * (xz + wy), 2.0 * (xy + wz), 1.0 - 2.0 * (xx + zz), 2.0 * (yz - wx),
                       2.0 * (xz - wy), 2.0 * (yz + wx), 1.0 - 2.0 * (xx + yy)], dim=-1)
    return mat.reshape(quat.shape[:-1] + (3, 3))


def chain():
    normals = F.one_hot(torch.argmin(scales, dim=-1), num_classes=3).float()
    rots = quat_to_rotmat(quats)
    normals = torch.bmm(rots, normals[:, :, None]).squeeze(-1)
    normals = F.normalize(normals, dim=1)
    viewdirs = -means.detach() + c2w.detach()[..., :3, 3]
    viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
    dots = (normals * viewdirs).sum(-1)
    negative_dot_indices = dots < 0
    normals[negative_dot_indices] = -normals[negative_dot_indices]
    normals = normals @ c2w.squeeze(0)[:3, :3]
    return torch.cat([rgbs, normals], dim=-1)


def fused():
    return camera_space_normals(quats, scales, means, c2w, rgbs=rgbs)


def wall(fn, reps=10):
    for _ in range(3):
        quats.grad = rgbs.grad = None
        (fn() * G).sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        quats.grad = rgbs.grad = None
        torch.autograd.backward(fn(), G)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


a = chain(); b = fused()
print(f"max |chain - fused| = {float((a - b).abs().max()):.2e}")
tc, tf = wall(chain), wall(fused)
print(f"N={N}: normals + cat, fwd+bwd: PyTorch chain {tc:.3f} ms -> fused {tf:.3f} ms ({tc / tf:.1f}x)")
