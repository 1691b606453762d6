#!/usr/bin/env python3
"""THE PIN PATH for the hot-path oracle.  Run ONCE on any machine that has the real gsplat 1.4.0 (the version MTGS pins:
/root/reference/requirements.txt:12, `pip install git+https://github.com/nerfstudio-project/gsplat.git@v1.4.0`, CUDA):

    python tests/golden/make_gsplat_golden.py            # writes tests/golden/gsplat_1_4_0_{classic,mtgs}.npz

and commit the two files.  They use the schema of the fixtures make_golden.py generates from the CPU oracle (same
seeded scenes, same keys), but every OUTPUT in them comes from gsplat itself: `gsplat.rendering.rasterization` for the
image / alpha / meta tensors, torch autograd through gsplat's own backward kernels for the gradients.  Once present,
  tests/test_oracle_vs_torch_ref.py::test_oracle_reproduces_gsplat_fixture   (CPU: pins oracle/gsplat_oracle.c)
  tests/test_gpu_parity.py::test_hip_reproduces_gsplat_fixture               (GPU: pins the HIP path directly)
stop skipping, and the "PARITY UNPINNED" notes (oracle/gsplat_oracle.c) can go.

Tile lists: the real gsplat has ONE list form, and since round 5 it is what this package's default call returns -- the fixture's
isect_ids / flatten_ids / isect_offsets are compared bit for bit with the default `rasterization()`, and the opt-in tight lists
(`with mtgs_amd.tight_lists():`) are then checked as ordered sublists of the fixture's with a sentinel tail
(tests/util.py::assert_tile_lists(..., rerun=...)); `lists` / `n_intersections` in the file say so.

Neither gsplat nor CUDA exists in the build container of this repository (no network), so the files are NOT in the
tree yet; this script is the committed recipe.  It imports nothing of the HIP library (mtgs_amd.synthetic / tests.util
are plain torch)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
OUT = Path(__file__).resolve().parent

from tests.util import small_scene  # noqa: E402

# name: (N, W, H, seed, D, render_mode, rasterize_mode, use_bg) -- the scenes of make_golden.py
SCENES = {
    "gsplat_1_4_0_classic": (150, 64, 48, 21, 3, "RGB", "classic", False),
    "gsplat_1_4_0_mtgs": (150, 70, 45, 22, 3, "RGB+ED", "antialiased", True),
}


def main():
    import gsplat
    from gsplat.rendering import rasterization
    assert gsplat.__version__.startswith("1.4.0"), f"gsplat {gsplat.__version__}: the reference pins v1.4.0"
    assert "mtgs_amd" not in getattr(gsplat, "__file__", ""), "this is the drop-in shim, not the real gsplat"
    dev = torch.device("cuda")
    for name, (N, W, H, seed, D, render_mode, rmode, use_bg) in SCENES.items():
        sc, vm, K = small_scene(N=N, W=W, H=H, seed=seed, D=D)
        g = torch.Generator().manual_seed(seed + 1)
        n_out = D + (1 if render_mode != "RGB" else 0)
        bg = torch.rand(1, D, generator=g) if use_bg else None
        Gc = torch.randn(1, H, W, n_out, generator=g)
        Ga = torch.randn(1, H, W, 1, generator=g)
        P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
        vmd = vm.to(dev).requires_grad_(True)
        render, alpha, info = rasterization(
            means=P["means"], quats=P["quats"], scales=P["scales"], opacities=P["opacities"], colors=P["colors"],
            viewmats=vmd, Ks=K.to(dev), width=W, height=H, tile_size=16, packed=False, near_plane=0.01, far_plane=1e10,
            render_mode=render_mode, sparse_grad=False, absgrad=True, rasterize_mode=rmode,
            backgrounds=None if bg is None else bg.to(dev))          # the kwargs of mtgs_scene_graph.py:641-659
        info["means2d"].retain_grad()
        torch.autograd.backward([render, alpha], [Gc.to(dev), Ga.to(dev)])
        radii = info["radii"]
        if radii.dim() == 3:      # later gsplat versions: [C, N, 2]
            radii = radii.max(dim=-1).values
        n = lambda t: t.detach().cpu().numpy()
        np.savez_compressed(
            OUT / f"{name}.npz", W=W, H=H, render_mode=render_mode, rasterize_mode=rmode, source=f"gsplat {gsplat.__version__}",
            lists="gsplat", n_intersections=int(info["flatten_ids"].numel()),
            means=n(sc["means"]), quats=n(sc["quats"]), scales=n(sc["scales"]), opacities=n(sc["opacities"]),
            colors=n(sc["colors"]), viewmat=n(vm), K=n(K), backgrounds=np.zeros((0,), np.float32) if bg is None else n(bg),
            Gc=n(Gc), Ga=n(Ga), render=n(render), alpha=n(alpha), radii=n(radii).astype(np.int32), means2d=n(info["means2d"]),
            depths=n(info["depths"]), conics=n(info["conics"]), opacities_eff=n(info["opacities"]),
            tiles_per_gauss=n(info["tiles_per_gauss"]).astype(np.int32), isect_ids=n(info["isect_ids"]).astype(np.int64),
            flatten_ids=n(info["flatten_ids"]).astype(np.int32), isect_offsets=n(info["isect_offsets"]).astype(np.int32),
            v_means=n(P["means"].grad), v_quats=n(P["quats"].grad), v_scales=n(P["scales"].grad),
            v_opacities=n(P["opacities"].grad), v_colors=n(P["colors"].grad), v_viewmat=n(vmd.grad),
            v_means2d=n(info["means2d"].grad), v_means2d_abs=n(info["means2d"].absgrad))
        print("wrote", OUT / f"{name}.npz")


if __name__ == "__main__":
    main()
