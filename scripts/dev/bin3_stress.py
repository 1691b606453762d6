"""Development: randomised stress of the sort-free binning (csrc/bin3.hip) against the operator path (global radix sort):
random image sizes, camera counts, Gaussian counts and scale distributions; ids and offsets must be bit-identical."""
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mtgs_amd import rasterization, wrapper as w  # noqa: E402
from mtgs_amd.synthetic import make_camera  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda")
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
rf = lambda: float(torch.rand(1, generator=g))
worst = 0
for case in range(cases):
    C = 1 if rf() < 0.7 else ri(2, 5)
    W, H = ri(17, 700), ri(17, 500)
    if rf() < 0.1:
        W, H = ri(1000, 2600), ri(600, 1500)
    N = ri(1, 200_000) if rf() < 0.8 else ri(1, 300)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    vms, Ks = zip(*[make_camera(W, H, yaw_deg=360.0 * rf()) for _ in range(C)])
    vm, K = torch.cat(vms).to(dev), torch.cat(Ks).to(dev)
    ext = 2.0 + 20.0 * rf()
    means = ((torch.rand(N, 3, generator=g) - 0.5) * 2 * ext).to(dev)
    if rf() < 0.3:      # equal depths along the view axis of camera 0: ties decided by the index
        means[:, 2] = torch.round(means[:, 2] * 2) / 2
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=1).to(dev)
    s_lo = 10 ** (-2.5 + 2.5 * rf())
    scales = (s_lo * (1 + 30 * torch.rand(N, 3, generator=g) ** (1 + 6 * rf()))).to(dev)
    opac = (0.02 + 0.9 * torch.rand(N, generator=g)).to(dev)
    cols = torch.rand(N, 3, generator=g).to(dev)
    mode = "antialiased" if rf() < 0.5 else "classic"
    try:
        r, a, info = rasterization(means=means, quats=quats, scales=scales, opacities=opac, colors=cols, viewmats=vm, Ks=K, width=W,
                                   height=H, packed=False, render_mode="RGB+ED", rasterize_mode=mode)
    except (NotImplementedError, RuntimeError) as e:      # more than 2^30 (2^31) intersections in one call: refused by name
        assert "intersections" in str(e), e
        continue
    radii, means2d, depths, conics, comps, oe = w.projection_with_opacities(means, quats, scales, vm, K, opac, W, H,
                                                                            calc_compensations=(mode == "antialiased"))
    _, isect_ids, flat = w.isect_tiles(means2d, radii, depths, 16, tw, th)
    off = w.isect_offset_encode(isect_ids, C, tw, th)
    ok = torch.equal(info["isect_offsets"], off) and torch.equal(info["flatten_ids"], flat) and torch.equal(info["isect_ids"], isect_ids)
    M = isect_ids.numel()
    if M:
        lens = torch.diff(torch.cat([off.reshape(-1), torch.tensor([M], device=dev, dtype=off.dtype)]))
        worst = max(worst, int(lens.max()))
    assert ok, f"case {case}: C={C} {W}x{H} N={N} M={M} mode={mode}"
    assert torch.isfinite(r).all() and torch.isfinite(a).all()
print(f"bin3 stress ok: {cases} cases, longest tile list {worst}")
