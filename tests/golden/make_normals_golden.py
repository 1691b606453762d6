#!/usr/bin/env python3
"""Golden vectors for the camera-space normals MTGS blends as three extra colour channels when `predict_normals` is on
(the shipped MTGS.py config), produced in the build container with the REFERENCE's own `quat_to_rotmat`
(/root/reference/mtgs/scene_model/gaussian_model/utils.py, imported by path) composed exactly as
MTGSSceneModel._get_gaussian_camera_space_normals does (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:526-545):

    normals  = one_hot(argmin(scales))                         # the shortest axis of the Gaussian
    normals  = normalize(bmm(quat_to_rotmat(quats), normals))
    viewdirs = normalize(cam_pos - means.detach())
    normals[dot(normals, viewdirs) < 0] *= -1                  # face the camera
    normals  = normals @ camera_to_worlds[:3, :3]              # world -> camera space

and the gradient of sum(normals * G) with respect to quats from autograd through those functions.
Writes tests/golden/normals_ref.npz (inputs + expected outputs only)."""
import importlib.util
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

spec = importlib.util.spec_from_file_location("ref_utils", "/root/reference/mtgs/scene_model/gaussian_model/utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def reference_normals(quats, scales, means, c2w):
    normals = F.one_hot(torch.argmin(scales, dim=-1), num_classes=3).to(quats.dtype)      # :527-529 (.float() there)
    rots = ref.quat_to_rotmat(quats)                                                      # :530
    normals = torch.bmm(rots, normals[:, :, None]).squeeze(-1)                            # :531
    normals = F.normalize(normals, dim=1)                                                 # :532
    viewdirs = -means.detach() + c2w.detach()[..., :3, 3]                                 # :534-536
    viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)                             # :537
    dots = (normals * viewdirs).sum(-1)                                                   # :538
    neg = dots < 0                                                                        # :539
    normals = torch.where(neg[:, None], -normals, normals)                                # :540 (in-place masked write there)
    return normals @ c2w.squeeze(0)[:3, :3]                                               # :543


out = {}
g = torch.Generator().manual_seed(99)
for name, N in (("a", 300), ("b", 64), ("c", 1)):
    quats = torch.randn(N, 4, generator=g, dtype=torch.float64)
    quats = quats / quats.norm(dim=-1, keepdim=True) * (1.0 if name != "b" else 1.3)     # get_gaussians hands over unit quats; b: not
    scales = torch.exp(torch.randn(N, 3, generator=g, dtype=torch.float64))
    if N > 10:   # ties: argmin takes the first of equal minima
        scales[3, :] = scales[3, 0]
        scales[4, 1] = scales[4, 2] = scales[4].min() * 0.5
        scales[5, 0] = scales[5, 2] = scales[5].min() * 0.5
    means = torch.randn(N, 3, generator=g, dtype=torch.float64) * 10
    A = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))[0]
    c2w = torch.cat([A, torch.randn(3, 1, generator=g, dtype=torch.float64) * 3], dim=1)[None]
    G = torch.randn(N, 3, generator=g, dtype=torch.float64)
    q = quats.clone().requires_grad_(True)
    n = reference_normals(q, scales, means, c2w)
    (n * G).sum().backward()
    for k, v in (("quats", quats), ("scales", scales), ("means", means), ("c2w", c2w), ("G", G), ("normals", n.detach()), ("g_quats", q.grad)):
        out[f"{name}_{k}"] = v.numpy()
np.savez_compressed(Path(__file__).parent / "normals_ref.npz", **out)
print({k: v.shape for k, v in out.items() if k.startswith("a_")})
