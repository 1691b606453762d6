#!/usr/bin/env python3
"""Golden vectors for mtgs_amd.loss.masked_ssim, produced by the REFERENCE's own module in the build container:
/root/reference/mtgs/utils/ssim.py is imported by path (it needs torch only) and run exactly as MTGS constructs and
calls it (mtgs_scene_graph.py:322, :831-841): MaskedSSIM(data_range=1.0, size_average=True, channel=3)(
gt.permute(2,0,1)[None], pred.permute(2,0,1)[None], mask=mask[H,W,1]); the gradient with respect to pred comes from
autograd through the reference.  Writes tests/golden/ssim_ref.npz (inputs + expected outputs only)."""
import importlib.util
from pathlib import Path

import numpy as np
import torch

spec = importlib.util.spec_from_file_location("ref_ssim", "/root/reference/mtgs/utils/ssim.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
g = torch.Generator().manual_seed(2024)
cases = {"a": (37, 53, "rand"), "b": (64, 48, "none"), "c": (30, 41, "block"), "d": (21, 70, "rand")}
for name, (H, W, mk) in cases.items():
    gt = torch.rand(H, W, 3, generator=g)
    # a prediction correlated with the ground truth, as during training
    pred = (gt + 0.25 * torch.randn(H, W, 3, generator=g)).clamp(0, 1)
    if name == "d":
        pred = torch.rand(H, W, 3, generator=g)
    if mk == "rand":
        mask = torch.rand(H, W, 1, generator=g) > 0.3
    elif mk == "block":
        mask = torch.ones(H, W, 1, dtype=torch.bool)
        mask[8:20, 10:30] = False
    else:
        mask = None
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        p = pred.to(dt).clone().requires_grad_(True)
        m = ref.MaskedSSIM(data_range=1.0, size_average=True, channel=3)
        val = m(gt.to(dt).permute(2, 0, 1)[None], p.permute(2, 0, 1)[None], mask=mask)
        val.backward()
        out[f"{name}_ssim_{tag}"] = val.detach().numpy()
        out[f"{name}_grad_{tag}"] = p.grad.numpy()
    out[f"{name}_gt"], out[f"{name}_pred"] = gt.numpy(), pred.detach().numpy()
    out[f"{name}_mask"] = np.zeros(0, dtype=bool) if mask is None else mask.numpy()
np.savez_compressed(Path(__file__).parent / "ssim_ref.npz", **out)
print({k: (v.shape, float(np.asarray(v).ravel()[0]) if v.size else None) for k, v in out.items() if "ssim" in k})
