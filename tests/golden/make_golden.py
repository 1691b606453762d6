#!/usr/bin/env python3
"""Generates the committed golden fixtures (run in the build container, where /root/reference is
mounted; the fixtures are data -- inputs and expected outputs -- and travel to the GPU box).

  ref_helpers.npz   outputs of the REFERENCE's own importable helpers
                    (/root/reference/mtgs/scene_model/gaussian_model/utils.py: quat_to_rotmat :14-40,
                    num_sh_bases :72-81, RGB2SH :83-88, SH2RGB :90-95) on seeded inputs.  They pin the
                    conventions this path shares with its callers: wxyz quaternion -> rotation matrix,
                    K = (deg+1)^2, and the SH DC constant (rgb = C0 * sh + 0.5).
  scene_*.npz       small seeded scenes: inputs, the CPU oracle's outputs for every stage, and the
                    gradients of L = sum(render*Gc) + sum(alpha*Ga) from fp64 autograd of the independent
                    torch restatement (oracle/torch_ref.py).  gsplat itself cannot be run here
                    (PARITY UNPINNED, see oracle/gsplat_oracle.c), so these are regression pins of the
                    oracle plus an autograd cross-check, not outputs of gsplat.
"""
import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
OUT = Path(__file__).resolve().parent

from oracle import oracle as orc, torch_ref as tr  # noqa: E402
from tests.util import small_scene, to_np  # noqa: E402


def ref_helpers():
    spec = importlib.util.spec_from_file_location(
        "ref_utils", "/root/reference/mtgs/scene_model/gaussian_model/utils.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    g = torch.Generator().manual_seed(123)
    quats = torch.nn.functional.normalize(torch.randn(64, 4, generator=g), dim=-1)
    rgb = torch.rand(32, 3, generator=g)
    sh = torch.randn(32, 3, generator=g)
    np.savez(OUT / "ref_helpers.npz",
             quats=quats.numpy(), rotmats=m.quat_to_rotmat(quats).numpy(),
             rgb=rgb.numpy(), rgb2sh=m.RGB2SH(rgb).numpy(), sh=sh.numpy(), sh2rgb=m.SH2RGB(sh).numpy(),
             degrees=np.arange(5), num_sh_bases=np.array([m.num_sh_bases(d) for d in range(5)]))


SCENES = {
    # name: (N, W, H, seed, D, render_mode, rasterize_mode, use_bg)
    "scene_classic_rgb": (150, 64, 48, 21, 3, "RGB", "classic", False),
    "scene_mtgs_like": (150, 70, 45, 22, 3, "RGB+ED", "antialiased", True),
}


def scene(name, N, W, H, seed, D, render_mode, rmode, use_bg):
    sc, vm, K = small_scene(N=N, W=W, H=H, seed=seed, D=D)
    a = to_np(sc)
    g = torch.Generator().manual_seed(seed + 1)
    n_out = D + (1 if render_mode != "RGB" else 0)
    bg = torch.rand(1, D, generator=g) if use_bg else None
    Gc = torch.randn(1, H, W, n_out, generator=g)
    Ga = torch.randn(1, H, W, 1, generator=g)
    render, alpha, m = orc.rasterization(a["means"], a["quats"], a["scales"], a["opacities"], a["colors"], vm.numpy(),
                                         K.numpy(), W, H, render_mode=render_mode, rasterize_mode=rmode,
                                         backgrounds=None if bg is None else bg.numpy())
    d = lambda t: t.double().clone().requires_grad_(True)
    P = [d(sc["means"]), d(sc["quats"]), d(sc["scales"]), d(sc["opacities"]), d(sc["colors"]), d(vm)]
    r2, a2, m2 = tr.rasterization(*P, K.double(), W, H, render_mode=render_mode, rasterize_mode=rmode,
                                  backgrounds=None if bg is None else bg.double())
    assert np.abs(render - r2.detach().numpy()).max() < 1e-4
    grads = torch.autograd.grad((r2 * Gc.double()).sum() + (a2 * Ga.double()).sum(), P)
    np.savez_compressed(
        OUT / f"{name}.npz", W=W, H=H, render_mode=render_mode, rasterize_mode=rmode,
        means=a["means"], quats=a["quats"], scales=a["scales"], opacities=a["opacities"], colors=a["colors"],
        viewmat=vm.numpy(), K=K.numpy(), backgrounds=np.zeros((0,), np.float32) if bg is None else bg.numpy(),
        Gc=Gc.numpy(), Ga=Ga.numpy(),
        render=render, alpha=alpha, radii=m["radii"], means2d=m["means2d"], depths=m["depths"], conics=m["conics"],
        compensations=np.zeros((0,), np.float32) if m["compensations"] is None else m["compensations"],
        tiles_per_gauss=m["tiles_per_gauss"], isect_ids=m["isect_ids"], flatten_ids=m["flatten_ids"],
        isect_offsets=m["isect_offsets"], last_ids=m["last_ids"],
        v_means=grads[0].numpy(), v_quats=grads[1].numpy(), v_scales=grads[2].numpy(),
        v_opacities=grads[3].numpy(), v_colors=grads[4].numpy(), v_viewmat=grads[5].numpy())


if __name__ == "__main__":
    ref_helpers()
    for name, cfg in SCENES.items():
        scene(name, *cfg)
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)
