#!/usr/bin/env python3
"""scratch: tight tile lists vs gsplat's lists -- same image bits, same gradients, sublists; timing of the pieces."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import mtgs_amd
from mtgs_amd import rasterization, wrapper
from mtgs_amd.synthetic import make_camera, make_scene

dev = torch.device("cuda")
for (N, W, H) in ((200_000, 640, 480), (2_000_000, 1920, 1080)):
    sc = {k: v.to(dev) for k, v in make_scene(N, seed=0, sh_degree=None).items()}
    vm, K = make_camera(W, H)
    vm, K = vm.to(dev), K.to(dev)
    g = torch.Generator().manual_seed(1)
    Gc, Ga = torch.randn(1, H, W, 4, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)
    res = {}
    for mode in ("exact", "tight"):
        P = {k: sc[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
        with mtgs_amd.exact_lists(mode == "exact"):
            r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm, K, W, H, packed=False,
                                       render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
            info["means2d"].retain_grad()
            torch.autograd.backward([r, a], [Gc, Ga])
        n = int(info["n_listed"])
        res[mode] = (r.detach(), a.detach(), {k: p.grad.clone() for k, p in P.items()}, info["isect_offsets"].clone(),
                     info["flatten_ids"][:n].clone(), info["isect_ids"][:n].clone(), info["means2d"].absgrad.clone(), n, info["flatten_ids"].numel())
    e, t = res["exact"], res["tight"]
    print(f"N {N} {W}x{H}: listed {t[7]} of {e[7]} (M {e[8]}); render equal {torch.equal(e[0], t[0])} alpha equal {torch.equal(e[1], t[1])}")
    for k in e[2]:
        d = (e[2][k] - t[2][k]).abs().max().item()
        print(f"   grad {k}: max |diff| {d:.3e} of max {e[2][k].abs().max().item():.3e}")
    print("   absgrad diff", (e[6] - t[6]).abs().max().item())
    # sublist property per tile (check a sample of tiles)
    oe, ot = e[3].reshape(-1).tolist() + [e[7]], t[3].reshape(-1).tolist() + [t[7]]
    import random
    random.seed(0)
    bad = 0
    fe, ft = e[4].cpu(), t[4].cpu()
    for tile in random.sample(range(len(oe) - 1), 300):
        le, lt = fe[oe[tile]:oe[tile + 1]].tolist(), ft[ot[tile]:ot[tile + 1]].tolist()
        it = iter(le)
        if not all(x in it for x in lt):
            bad += 1
    print("   tiles whose tight list is not an ordered sublist:", bad)
