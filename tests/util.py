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
