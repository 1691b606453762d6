#!/usr/bin/env python3
"""Golden vectors for the rigid-node helpers restated in mtgs_amd.nodes (object_pose / interpolate_quats, idft_weights /
fourier_features_dc), produced by the REFERENCE's own functions in the build container:
/root/reference/mtgs/scene_model/gaussian_model/utils.py (interpolate_quats :201-233, IDFT :335-352) imported by path and
composed as RigidSubModel does (rigid_node.py:145-166 pose between frames, :217-221 Fourier features), with autograd
gradients through the reference functions.  Writes tests/golden/pose_fourier_ref.npz (inputs + expected outputs only)."""
import importlib.util
from pathlib import Path

import numpy as np
import torch

spec = importlib.util.spec_from_file_location("ref_utils", "/root/reference/mtgs/scene_model/gaussian_model/utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
g = torch.Generator().manual_seed(2024)
# ---- slerp: generic pairs, nearly parallel pairs (nlerp branch), opposite hemisphere, un-normalised inputs
q1 = torch.randn(40, 4, generator=g, dtype=torch.float64)
q2 = torch.randn(40, 4, generator=g, dtype=torch.float64)
q2[:8] = q1[:8] + 1e-3 * torch.randn(8, 4, generator=g, dtype=torch.float64)      # dot > 0.9995
q2[8:12] = -q1[8:12] + 0.3 * torch.randn(4, 4, generator=g, dtype=torch.float64)   # dot < 0: flipped
frac = torch.rand(40, 1, generator=g, dtype=torch.float64)
out["slerp_q1"], out["slerp_q2"], out["slerp_t"] = q1.numpy(), q2.numpy(), frac.numpy()
out["slerp_out"] = torch.stack([ref.interpolate_quats(q1[i], q2[i], frac[i]).squeeze(0) for i in range(40)]).numpy()
# ---- pose between frames (rigid_node.py:145-166) with gradients to the pose parameters
F = 12
iq = torch.randn(F, 4, generator=g, dtype=torch.float64)
it = torch.randn(F, 3, generator=g, dtype=torch.float64) * 5
ts = torch.cumsum(torch.rand(F, generator=g, dtype=torch.float64) * 0.1 + 0.05, 0)
stamps = torch.cat([ts[3:4], (ts[:-1] + (ts[1:] - ts[:-1]) * torch.rand(F - 1, generator=g, dtype=torch.float64))[[0, 4, 9]]])
Gq, Gt = torch.randn(4, generator=g, dtype=torch.float64), torch.randn(3, generator=g, dtype=torch.float64)
out.update(pose_iq=iq.numpy(), pose_it=it.numpy(), pose_ts=ts.numpy(), pose_stamps=stamps.numpy(), pose_Gq=Gq.numpy(), pose_Gt=Gt.numpy())
for k, stamp in enumerate(stamps):
    A, B = iq.clone().requires_grad_(True), it.clone().requires_grad_(True)
    diffs = stamp - ts
    prev_f = torch.argmin(torch.where(diffs >= 0, diffs, float("inf")))
    next_f = torch.argmin(torch.where(diffs <= 0, -diffs, float("inf")))
    if next_f == prev_f:
        q, t = A[next_f], B[next_f]
    else:
        tt = (stamp - ts[prev_f]) / (ts[next_f] - ts[prev_f])
        q = ref.interpolate_quats(A[prev_f], A[next_f], tt).squeeze()
        t = torch.lerp(B[prev_f], B[next_f], tt)
    ((q * Gq).sum() + (t * Gt).sum()).backward()
    out[f"pose{k}_q"], out[f"pose{k}_t"], out[f"pose{k}_g_iq"], out[f"pose{k}_g_it"] = q.detach().numpy(), t.detach().numpy(), A.grad.numpy(), B.grad.numpy()
# ---- IDFT + Fourier features (rigid_node.py:217-221)
for name, dim, x, normalized in (("t5", 5, 0.37, True), ("t8", 8, 0.91, True), ("s6", 6, -1.234, False), ("s1", 1, 0.5, False)):
    w = ref.IDFT(torch.tensor(x, dtype=torch.float64), dim, normalized)
    N = 257
    fdc = torch.randn(N, dim, 3, generator=g, dtype=torch.float64).requires_grad_(True)
    Gd = torch.randn(N, 3, generator=g, dtype=torch.float64)
    dc = torch.sum(fdc * w[..., None], dim=1, keepdim=False)
    (dc * Gd).sum().backward()
    out.update({f"four_{name}_x": np.float64(x), f"four_{name}_dim": np.int64(dim), f"four_{name}_norm": np.bool_(normalized),
                f"four_{name}_w": w.numpy(), f"four_{name}_fdc": fdc.detach().numpy(), f"four_{name}_G": Gd.numpy(),
                f"four_{name}_dc": dc.detach().numpy(), f"four_{name}_g_fdc": fdc.grad.numpy(),
                f"four_{name}_g_w": (fdc.detach() * Gd[:, None, :]).sum((0, 2)).numpy()})
np.savez_compressed(Path(__file__).parent / "pose_fourier_ref.npz", **out)
print(sorted(k for k in out if k.startswith("pose0") or k.startswith("four_t5")))
