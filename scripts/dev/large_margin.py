#!/usr/bin/env python3
"""Development: the margins of tests/test_gpu_large.py::test_large_scene_culling_invariance[24M] over repeated runs
(fp32 atomic order is the only run-to-run difference)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import mtgs_amd as gs  # noqa: E402
from mtgs_amd.synthetic import make_camera  # noqa: E402
from tests.test_gpu_large import _scene, _step  # noqa: E402

N, W, H, scale_mul = 24_000_000, 1920, 1080, 3.0
worst = {}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    gen = torch.Generator(device="cuda").manual_seed(7)
    sc = _scene(N, gen, scale_mul)
    vm, K = make_camera(W, H)
    vm, K = vm.cuda(), K.cuda()
    Gc = torch.randn(1, H, W, 4, device="cuda", generator=gen)
    Ga = torch.randn(1, H, W, 1, device="cuda", generator=gen)
    P = {k: v.requires_grad_(True) for k, v in sc.items()}
    vmf = vm.clone().requires_grad_(True)
    r, a, info = _step(gs, P, vmf, K, W, H, Gc, Ga)
    idx = (info["radii"][0] > 0).nonzero()[:, 0]
    Q = {k: v.detach()[idx].clone().requires_grad_(True) for k, v in sc.items()}
    vms = vm.clone().requires_grad_(True)
    rs, as_, infos = _step(gs, Q, vms, K, W, H, Gc, Ga)
    out = {"img": float((r - rs).abs().max())}
    for k in P:
        out[k] = float((P[k].grad[idx] - Q[k].grad).abs().max()) / (float(Q[k].grad.abs().max()) + 1e-20)
    out["vm_rel"] = float(((vmf.grad - vms.grad).abs() / (2e-3 * vms.grad.abs() + 1e-3 * vms.grad.abs().max())).max())
    af, asub = info["means2d"].absgrad[0][idx], infos["means2d"].absgrad[0]
    out["absgrad"] = float((af - asub).abs().max()) / float(asub.abs().max())
    print(rep, {k: f"{v:.2e}" for k, v in out.items()}, flush=True)
    for k, v in out.items():
        worst[k] = max(worst.get(k, 0.0), v)
    del P, Q, sc, r, a, info, rs, as_, infos
print("worst", {k: f"{v:.2e}" for k, v in worst.items()}, "(bounds: gradients 2e-4, vm_rel 1, absgrad 2e-4)")
