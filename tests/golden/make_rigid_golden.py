#!/usr/bin/env python3
"""Golden vectors for the rigid-node pose transform of mtgs_amd.nodes, produced by the REFERENCE's own helpers in the
build container: /root/reference/mtgs/scene_model/gaussian_model/utils.py (quat_to_rotmat, quat_mult) is imported by path
and composed exactly as RigidSubModel does (rigid_node.py:205-216):
    global_means = local_means @ quat_to_rotmat(q).T + t ;  global_quats = quat_mult(q, local_quats / |local_quats|)
with the gradients of  sum(global_means * Gm) + sum(global_quats * Gq)  from autograd through the reference functions.
Writes tests/golden/rigid_ref.npz (inputs + expected outputs only)."""
import importlib.util
from pathlib import Path

import numpy as np
import torch

spec = importlib.util.spec_from_file_location("ref_utils", "/root/reference/mtgs/scene_model/gaussian_model/utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
g = torch.Generator().manual_seed(77)
for name, N, unit in (("a", 257, True), ("b", 64, False), ("c", 1, True)):
    means = torch.randn(N, 3, generator=g, dtype=torch.float64) * 2
    quats = torch.randn(N, 4, generator=g, dtype=torch.float64)
    q = torch.randn(4, generator=g, dtype=torch.float64)
    if unit:   # get_object_pose normalises the per-frame quaternion (rigid_node.py:142); static nodes pass it raw (:136)
        q = q / q.norm()
    t = torch.randn(3, generator=g, dtype=torch.float64) * 5
    Gm, Gq = torch.randn(N, 3, generator=g, dtype=torch.float64), torch.randn(N, 4, generator=g, dtype=torch.float64)
    P = [x.clone().requires_grad_(True) for x in (means, quats, q, t)]
    gm = P[0] @ ref.quat_to_rotmat(P[2]).T + P[3]                               # rigid_node.py:205-209
    gq = ref.quat_mult(P[2], P[1] / P[1].norm(dim=-1, keepdim=True))            # rigid_node.py:211-214
    ((gm * Gm).sum() + (gq * Gq).sum()).backward()
    for k, v in (("means", means), ("quats", quats), ("q", q), ("t", t), ("Gm", Gm), ("Gq", Gq), ("global_means", gm.detach()),
                 ("global_quats", gq.detach()), ("g_means", P[0].grad), ("g_quats", P[1].grad), ("g_q", P[2].grad), ("g_t", P[3].grad)):
        out[f"{name}_{k}"] = v.numpy()
np.savez_compressed(Path(__file__).parent / "rigid_ref.npz", **out)
print({k: v.shape for k, v in out.items() if k.startswith("a_")})
