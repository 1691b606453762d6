#!/usr/bin/env python3
"""Masked SSIM of the loss head, forward + backward: the operator chain of mtgs/utils/ssim.py as PyTorch runs it
on the GPU (5 grouped 'valid' convolutions x 2 passes + elementwise + masked_select, NCHW) against
mtgs_amd.loss.masked_ssim (two HIP kernels on the [H,W,3] images)."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mtgs_amd.loss import masked_ssim  # noqa: E402

dev = torch.device("cuda")


def window():
    coords = torch.arange(11, dtype=torch.float) - 5
    g = torch.exp(-(coords ** 2) / (2 * 1.5 ** 2))
    return (g / g.sum())[None, None].repeat(3, 1, 1, 1).to(dev)  # [3,1,1,11]


WIN = window()


def gfilter(x):
    x = F.conv2d(x, WIN.transpose(2, 3), groups=3)
    return F.conv2d(x, WIN, groups=3)


def chain(gt, pred, mask):
    X, Y = gt.permute(2, 0, 1)[None], pred.permute(2, 0, 1)[None]
    m = mask.permute(2, 0, 1).unsqueeze(0).expand_as(X)[..., 5:-5, 5:-5]
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2 = gfilter(X), gfilter(Y)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1, s2, s12 = gfilter(X * X) - mu1_sq, gfilter(Y * Y) - mu2_sq, gfilter(X * Y) - mu1_mu2
    cs = (2 * s12 + C2) / (s1 + s2 + C2)
    smap = ((2 * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs
    return torch.masked_select(smap, m).mean()


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for H, W in ((540, 960), (1080, 1920)):
    g = torch.Generator().manual_seed(0)
    gt = torch.rand(H, W, 3, generator=g).to(dev)
    pred = (gt + 0.2 * torch.randn(H, W, 3, generator=g).to(dev)).clamp(0, 1).requires_grad_(True)
    mask = (torch.rand(H, W, 1, generator=g) > 0.2).to(dev)

    def run(fn):
        pred.grad = None
        (1 - fn(gt, pred, mask)).backward()

    a, b = t(lambda: run(chain)), t(lambda: run(masked_ssim))
    d = abs(float(chain(gt, pred, mask)) - float(masked_ssim(gt, pred, mask)))
    print(f"{W}x{H}: operator chain {a:.0f} us, fused {b:.0f} us ({a / b:.1f}x), |difference| {d:.1e}")
