#!/usr/bin/env python3
"""Local (non-communication) cost of the sparse gradient exchange at the headline size, on one GPU:
pack, zero-fill of the dense outputs, and one mtgs_dp_accumulate launch per sender."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mtgs_amd import _lib, dist as mdist  # noqa: E402
from mtgs_amd._lib import call, ptr  # noqa: E402

dev = torch.device("cuda")
N, K = 2_000_000, 16
g = torch.Generator().manual_seed(0)
radii = (torch.rand(N, generator=g) < 0.15).int().to(dev)
vis = (radii > 0)
mk = lambda *s: (torch.randn(*s, generator=g).to(dev) * vis.view(-1, *([1] * (len(s) - 1)))).contiguous()
v_means, v_quats, v_scales, v_opac, v_rgb = mk(N, 3), mk(N, 4), mk(N, 3), mk(N), mk(N, 3)
means = torch.randn(N, 3, generator=g).to(dev)
cam = torch.tensor([0.3, -0.2, 0.1], device=dev)
ex = mdist.SparseGradExchange(N, K, dev)


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


st = torch.cuda.current_stream().cuda_stream
print("n_vis", int(vis.sum()))
print("exchange (world=1: pack + zero 472 MB + 1 accumulate + item sync): %.1f us" % t(lambda: ex.exchange(radii, means, cam, v_means, v_quats, v_scales, v_opac, v_rgb, 3)))
print("pack: %.1f us" % t(lambda: call("mtgs_dp_pack", N, ptr(radii), ptr(v_means), ptr(v_quats), ptr(v_scales), ptr(v_opac), ptr(v_rgb), ptr(ex.rows), N, ptr(ex.count), st)))
n_rows = int(ex.count.item())
flat = torch.zeros(N * 59, device=dev)
o = torch.split(flat, [3 * N, 4 * N, 3 * N, N, 48 * N])
print("zero-fill 472 MB: %.1f us" % t(lambda: flat.zero_()))
print("accumulate, one sender (%d rows): %.1f us" % (n_rows, t(lambda: call("mtgs_dp_accumulate", n_rows, ptr(ex.rows), N, K, 3, ptr(means), ptr(cam), ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(o[3]), ptr(o[4]), st))))
