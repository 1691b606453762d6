"""Shared scene builders for the tests (small scenes that exercise every branch)."""
import math

import numpy as np
import torch

from mtgs_amd.synthetic import make_camera, make_scene


def small_scene(N=300, W=100, H=70, seed=3, D=3, off_centre=True, sh_degree=None):
    """A compact scene in front of the camera: unnormalised quats, off-centre principal point,
    translated camera, some Gaussians behind the camera / outside the frustum."""
    sc = make_scene(N, seed=seed, extent=(4.0, 2.0, 4.0), sh_degree=sh_degree)
    sc["means"][:, 2] = sc["means"][:, 2].abs() + 1.0
    sc["means"][: N // 20, 2] *= -1.0          # behind the camera
    sc["means"][N // 20: N // 10, 0] *= 20.0   # far outside the frustum (clamped projection)
    sc["scales"] *= 3
    sc["quats"] = sc["quats"] * 1.7            # MTGS does not guarantee unit quaternions
    vm, K = make_camera(W, H)
    if off_centre:
        K[0, 0, 2] += 3.5
        K[0, 1, 2] -= 2.25
        vm[0, :3, 3] = torch.tensor([0.1, -0.2, 0.3])
    if D != 3 and sh_degree is None:
        g = torch.Generator().manual_seed(seed + 100)
        sc["colors"] = torch.rand(N, D, generator=g)
    return sc, vm, K


def to_np(d):
    return {k: v.numpy() for k, v in d.items()}


def assert_image_close(got, ref, critical, tol=1e-4, flip_bound=1.0 / 255.0, name="render", scale=None):
    """max-abs <= tol (x max(1, |ref|max)) on every well-conditioned pixel.  Pixels the oracle flags as
    threshold-critical (a Gaussian within 1e-4 relative of the alpha = 1/255 or T = 1e-4 decision, see
    orc_blend_fwd) may differ by one flipped decision: <= flip_bound * scale.  Critical pixels must be rare."""
    err = np.abs(got - ref)
    if scale is None:
        scale = max(1.0, float(np.abs(ref).max()))
    crit = np.broadcast_to(critical[..., None], err.shape)
    assert critical.mean() < 5e-3, f"{name}: too many threshold-critical pixels ({critical.mean():.2e})"
    ok = err[~crit]
    assert ok.size == 0 or ok.max() <= tol * scale, f"{name}: max err {ok.max():.3e} > {tol * scale:.1e}"
    bad = err[crit]
    assert bad.size == 0 or bad.max() <= (flip_bound * 1.5 + tol) * scale, f"{name}: critical-pixel err {bad.max():.3e}"
