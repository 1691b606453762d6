#!/usr/bin/env python3
"""Local (non-communication) cost of the sparse gradient exchange at the headline size, on one GPU: the
sender side (visibility map + ordered pack) and the receiver's one-pass reduction over W senders whose
visible sets are those of W cameras at yaw r * 45 deg of the benchmark scene (the rows themselves are random)."""
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mtgs_amd import _lib, dist as mdist, wrapper  # noqa: E402
from mtgs_amd._lib import call, ptr  # noqa: E402
from mtgs_amd.synthetic import make_camera, make_scene  # noqa: E402

dev = torch.device("cuda")
N, K, W, H = 2_000_000, 16, 1920, 1080
sc = {k: v.to(dev) for k, v in make_scene(N, seed=0, sh_degree=None).items()}
means = sc["means"]
g = torch.Generator().manual_seed(0)
mk = lambda *s: torch.randn(*s, generator=g).to(dev)
st = torch.cuda.current_stream().cuda_stream


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for world in (2, 4, 8):
    ex = mdist.SparseGradExchange(N, K, dev)
    nw, L = ex.n_words, ex.meta_len
    metas = torch.zeros(world, L, dtype=torch.int32, device=dev)
    rows, counts = [], []
    for r in range(world):
        vm, Kmat = make_camera(W, H, yaw_deg=45.0 * r)
        radii = wrapper.fully_fused_projection(means, None, sc["quats"], sc["scales"], vm.to(dev), Kmat.to(dev), W, H)[0][0]
        vis = (radii > 0)
        vg = [mk(N, 3), mk(N, 4), mk(N, 3), mk(N), mk(N, 3)]
        m = metas[r]
        pack = lambda: call("mtgs_dp_pack_ordered", N, ptr(radii), ptr(vg[0]), ptr(vg[1]), ptr(vg[2]), ptr(vg[3]), ptr(vg[4]),
                            ptr(m[4:]), ptr(m[4 + 2 * nw:]), ptr(m), ptr(ex.block_counts), ptr(ex.rows), N, st)
        if r == 0 and world == 2:
            print("sender: visibility map + ordered pack: %.1f us (n_vis %d)" % (t(pack), int(vis.sum())))
        pack()
        cam = torch.inverse(vm)[0, :3, 3].to(dev)
        m[1:4].copy_(cam.view(torch.int32))
        counts.append(int(m[0].item()))
        rows.append(ex.rows[:counts[-1]].clone())
    cap = max(counts)
    recv = torch.zeros(world, cap, 16, device=dev)
    for r in range(world):
        recv[r, :counts[r]] = rows[r]
    cams = metas[:, 1:4].contiguous().view(torch.float32)
    out = [torch.empty(N, 3, device=dev), torch.empty(N, 4, device=dev), torch.empty(N, 3, device=dev),
           torch.empty(N, device=dev), torch.empty(N, K, 3, device=dev)]
    red = lambda: call("mtgs_dp_reduce", world, N, K, 3, ptr(means), ptr(metas[:, 4:]), ptr(metas[:, 4 + 2 * nw:]), L * 4,
                       ptr(recv), cap * 16, ptr(cams), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), ptr(out[4]), st)
    covered = sum(counts)
    print("world %d: one-pass reduce over %d rows (%d MB received): %.1f us" % (world, covered, world * cap * 64 >> 20, t(red)))
