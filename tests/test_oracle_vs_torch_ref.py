"""The C oracle against the independent torch restatement: forward values and every hand-derived
VJP (compositing, projection incl. viewmat and the antialiasing compensation, expected depth, SH)
against fp64 autograd; plus fp64 finite differences of the restatement itself and the golden
fixtures."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import torch_ref as tr
from tests.util import small_scene, to_np

GOLD = Path(__file__).resolve().parent / "golden"


def oracle_backward(oracle, a, vm, K, W, H, render_mode, rmode, bg, Gc, Ga):
    r, al, m = oracle.rasterization(a["means"], a["quats"], a["scales"], a["opacities"], a["colors"], vm, K, W, H,
                                    render_mode=render_mode, rasterize_mode=rmode, backgrounds=bg)
    Gc_raw, Ga_tot = Gc.copy(), Ga.copy()
    if render_mode in ("ED", "RGB+ED"):
        alc = np.maximum(al, 1e-10)
        Gc_raw[..., -1:] = Gc[..., -1:] / alc
        Ga_tot = Ga_tot + (-(m["render_raw"][..., -1:] / alc ** 2) * Gc[..., -1:]) * (al > 1e-10)
    v2d, vabs, vcon, vcol, vop = oracle.blend_bwd(m["means2d"], m["conics"], m["colors"], m["opacities"], m["backgrounds"],
                                                  W, H, 16, m["isect_offsets"], m["flatten_ids"], al, m["last_ids"], Gc_raw, Ga_tot)
    aa = rmode == "antialiased"
    D = a["colors"].shape[-1]
    v_depth = vcol[..., -1].copy() if render_mode != "RGB" else np.zeros_like(vop)
    v_comp = vop * a["opacities"][None] if aa else None
    vm_, vq, vs, vvm = oracle.project_bwd(a["means"], a["quats"], a["scales"], vm, K, W, H, 0.3, m["radii"], m["conics"],
                                          m["compensations"], v2d, v_depth, vcon, v_comp)
    v_opac = (vop * (m["compensations"] if aa else 1.0)).sum(0)
    return r, al, m, dict(means=vm_, quats=vq, scales=vs, opacities=v_opac, colors=vcol[..., :D].sum(0), viewmat=vvm)


@pytest.mark.parametrize("render_mode,rmode,D,use_bg,ncam", [
    ("RGB", "classic", 3, False, 1), ("RGB+ED", "antialiased", 3, True, 1), ("RGB+D", "classic", 5, False, 2)])
def test_oracle_forward_and_gradients_match_fp64_autograd(oracle, render_mode, rmode, D, use_bg, ncam):
    W, H = 70, 50
    sc, vm, K = small_scene(N=200, W=W, H=H, seed=5, D=D)
    if ncam == 2:
        vm = torch.cat([vm, vm.clone()]); vm[1, 0, 3] += 0.7
        K = torch.cat([K, K])
    a = to_np(sc)
    g = torch.Generator().manual_seed(1)
    n_out = D + (1 if render_mode != "RGB" else 0)
    Gc = torch.randn(ncam, H, W, n_out, generator=g)
    Ga = torch.randn(ncam, H, W, 1, generator=g)
    bg = torch.rand(ncam, D, generator=g) if use_bg else None
    r, al, m, grads = oracle_backward(oracle, a, vm.numpy(), K.numpy(), W, H, render_mode, rmode,
                                      None if bg is None else bg.numpy(), Gc.numpy(), Ga.numpy())
    d = lambda t: t.double().clone().requires_grad_(True)
    P = dict(means=d(sc["means"]), quats=d(sc["quats"]), scales=d(sc["scales"]), opacities=d(sc["opacities"]),
             colors=d(sc["colors"]), viewmat=d(vm))
    r2, a2, m2 = tr.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], P["viewmat"],
                                  K.double(), W, H, render_mode=render_mode, rasterize_mode=rmode,
                                  backgrounds=None if bg is None else bg.double())
    assert np.array_equal(m["radii"], m2["radii"].numpy())
    assert np.abs(r - r2.detach().numpy()).max() < 2e-5 * max(1, np.abs(r).max())
    assert np.abs(al - a2.detach().numpy()).max() < 5e-6
    ref = torch.autograd.grad((r2 * Gc.double()).sum() + (a2 * Ga.double()).sum(), list(P.values()))
    for name, rg in zip(P, ref):
        rg = rg.numpy()
        err, scale = np.abs(grads[name] - rg).max(), np.abs(rg).max()
        assert err <= 3e-5 * scale + 1e-6, f"{name}: {err:.3e} vs {scale:.3e}"


@pytest.mark.parametrize("degree", [0, 1, 2, 3, 4])
def test_oracle_sh_matches_autograd(oracle, degree):
    g = torch.Generator().manual_seed(degree)
    n, K = 64, 25
    dirs = torch.randn(n, 3, generator=g, dtype=torch.float64) * 2
    coeffs = torch.randn(n, K, 3, generator=g, dtype=torch.float64)
    vcol = torch.randn(n, 3, generator=g, dtype=torch.float64)
    d, c = dirs.clone().requires_grad_(True), coeffs.clone().requires_grad_(True)
    out = tr.spherical_harmonics(degree, d, c)
    gd, gc = torch.autograd.grad((out * vcol).sum(), [d, c], allow_unused=True)
    gd = torch.zeros_like(d) if gd is None else gd
    o = oracle.sh_fwd(degree, dirs.numpy(), coeffs.numpy())
    vc, vd = oracle.sh_bwd(degree, dirs.numpy(), coeffs.numpy(), vcol.numpy(), need_v_dirs=True)
    np.testing.assert_allclose(o, out.detach().numpy(), atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(vc, gc.numpy(), atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(vd, gd.numpy(), atol=1e-5, rtol=1e-4)
    assert np.all(vc[:, (degree + 1) ** 2:] == 0)


def test_torch_ref_finite_differences_fp64():
    """fp64 central differences of the restatement itself (incl. un-normalised quats, viewmat and
    the compensation path): autograd of the restatement is what pins the oracle's VJPs."""
    W, H = 40, 30
    sc, vm, K = small_scene(N=40, W=W, H=H, seed=9)
    g = torch.Generator().manual_seed(2)
    Gc = torch.randn(1, H, W, 4, generator=g, dtype=torch.float64)
    P = {k: v.double() for k, v in sc.items()}
    P["viewmat"] = vm.double()

    def loss(p):
        r, a, _ = tr.rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], p["viewmat"],
                                   K.double(), W, H, render_mode="RGB+ED", rasterize_mode="antialiased")
        return (r * Gc).sum() + 0.3 * a.sum()

    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    grads = dict(zip(Pg, torch.autograd.grad(loss(Pg), list(Pg.values()))))
    rng = np.random.default_rng(0)
    for name in ["means", "quats", "scales", "opacities", "colors", "viewmat"]:
        flat = P[name].reshape(-1)
        n_ok = 0
        for idx in rng.choice(flat.numel(), size=min(6, flat.numel()), replace=False):
            if name == "viewmat" and idx >= 12:
                continue
            eps = 1e-6
            vals = []
            for sgn in (+1, -1):
                q = {k: v.clone() for k, v in P.items()}
                q[name].reshape(-1)[idx] += sgn * eps
                vals.append(loss(q).item())
            fd = (vals[0] - vals[1]) / (2 * eps)
            an = grads[name].reshape(-1)[idx].item()
            # discrete decisions (alpha threshold, T stop) make the loss piecewise smooth: allow rare misses
            if abs(fd - an) <= 1e-4 * max(1.0, abs(an)):
                n_ok += 1
        assert n_ok >= 4, name


def test_reference_helper_conventions():
    """Fixtures produced by the REFERENCE's own helpers (tests/golden/make_golden.py)."""
    z = np.load(GOLD / "ref_helpers.npz")
    R = tr.quat_to_rotmat(torch.from_numpy(z["quats"]).double()).numpy()
    np.testing.assert_allclose(R, z["rotmats"], atol=1e-6)              # wxyz convention
    assert list(z["num_sh_bases"]) == [(d + 1) ** 2 for d in z["degrees"]]
    C0 = 0.28209479177387814
    np.testing.assert_allclose(z["rgb2sh"] * C0 + 0.5, z["rgb"], atol=1e-6)
    np.testing.assert_allclose(z["sh2rgb"], z["sh"] * C0 + 0.5, atol=1e-6)


def test_oracle_projection_uses_reference_quaternion_convention(oracle):
    """Sigma = R S^2 R^T with R from the reference's quat_to_rotmat: checked through the projected
    conic of an on-axis anisotropic Gaussian."""
    z = np.load(GOLD / "ref_helpers.npz")
    q, R = z["quats"][:8].astype(np.float32), z["rotmats"][:8].astype(np.float64)
    f, zc, W, H = 100.0, 4.0, 64, 64
    s = np.array([0.3, 0.1, 0.2])
    means = np.tile(np.array([[0, 0, zc]], np.float32), (8, 1))
    vm = np.eye(4, dtype=np.float32)[None]
    K = np.array([[[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]], np.float32)
    conics = oracle.project_fwd(means, q, np.tile(s.astype(np.float32), (8, 1)), vm, K, W, H)[3][0]
    for i in range(8):
        Sig = R[i] @ np.diag(s ** 2) @ R[i].T
        J = np.array([[f / zc, 0, 0], [0, f / zc, 0]])
        inv = np.linalg.inv(J @ Sig @ J.T + 0.3 * np.eye(2))
        np.testing.assert_allclose(conics[i], [inv[0, 0], inv[0, 1], inv[1, 1]], rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("name", ["scene_classic_rgb", "scene_mtgs_like"])
def test_oracle_reproduces_golden_fixture(oracle, name):
    z = np.load(GOLD / f"{name}.npz")
    W, H = int(z["W"]), int(z["H"])
    a = {k: z[k] for k in ("means", "quats", "scales", "opacities", "colors")}
    bg = z["backgrounds"] if z["backgrounds"].size else None
    r, al, m, grads = oracle_backward(oracle, a, z["viewmat"], z["K"], W, H, str(z["render_mode"]), str(z["rasterize_mode"]),
                                      bg, z["Gc"], z["Ga"])
    for key in ("radii", "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets", "last_ids"):
        assert np.array_equal(m[key], z[key]), key
    for key in ("means2d", "depths", "conics"):
        assert np.array_equal(m[key], z[key]), key             # fixed operation order: bit-stable
    np.testing.assert_allclose(r, z["render"], atol=1e-6)
    np.testing.assert_allclose(al, z["alpha"], atol=1e-6)
    for key, gname in [("means", "v_means"), ("quats", "v_quats"), ("scales", "v_scales"), ("opacities", "v_opacities"),
                       ("colors", "v_colors"), ("viewmat", "v_viewmat")]:
        ref = z[gname]
        assert np.abs(grads[key] - ref).max() <= 3e-5 * np.abs(ref).max() + 1e-6, key


@pytest.mark.parametrize("name", ["gsplat_1_4_0_classic", "gsplat_1_4_0_mtgs"])
def test_oracle_reproduces_gsplat_fixture(oracle, name):
    """THE PIN: outputs of the real gsplat 1.4.0 (tests/golden/make_gsplat_golden.py).  Skipped until the files exist --
    gsplat cannot be installed or run in the build container, so the oracle stays "parity unpinned" until someone runs that
    script on a CUDA machine and commits its two files."""
    path = GOLD / f"{name}.npz"
    if not path.exists():
        pytest.skip(f"{path.name} not generated yet (needs gsplat==1.4.0 + CUDA: tests/golden/make_gsplat_golden.py)")
    z = np.load(path)
    W, H = int(z["W"]), int(z["H"])
    a = {k: z[k] for k in ("means", "quats", "scales", "opacities", "colors")}
    bg = z["backgrounds"] if z["backgrounds"].size else None
    r, al, m, grads = oracle_backward(oracle, a, z["viewmat"], z["K"], W, H, str(z["render_mode"]), str(z["rasterize_mode"]),
                                      bg, z["Gc"], z["Ga"])
    vis = z["radii"] > 0
    assert np.array_equal(m["radii"], z["radii"]), "radii differ from gsplat"
    for key in ("means2d", "depths", "conics"):
        np.testing.assert_allclose(m[key][vis], z[key][vis], rtol=2e-5, atol=1e-6, err_msg=key)   # gsplat contracts FMAs
    for key in ("tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets"):
        assert np.array_equal(m[key], z[key]), key
    np.testing.assert_allclose(r, z["render"], atol=1e-4 * max(1.0, np.abs(z["render"]).max()))
    np.testing.assert_allclose(al, z["alpha"], atol=1e-4)
    for key, gname in [("means", "v_means"), ("quats", "v_quats"), ("scales", "v_scales"), ("opacities", "v_opacities"),
                       ("colors", "v_colors"), ("viewmat", "v_viewmat")]:
        ref = z[gname]
        assert np.abs(grads[key] - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-6, key
