"""oracle/normals_oracle.py against the vectors produced by the reference's own helpers (tests/golden/normals_ref.npz,
made by tests/golden/make_normals_golden.py from /root/reference/.../utils.py::quat_to_rotmat composed as
mtgs_scene_graph.py:526-545)."""
from pathlib import Path

import numpy as np
import pytest

from oracle import normals_oracle as no

GOLD = np.load(Path(__file__).parent / "golden" / "normals_ref.npz")


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_normals_oracle_matches_reference_vectors(case):
    g = {k[2:]: GOLD[k] for k in GOLD.files if k.startswith(case + "_")}
    n = no.normals_fwd(g["quats"], g["scales"], g["means"], g["c2w"])
    assert np.allclose(n, g["normals"], rtol=0, atol=1e-13)
    gq = no.normals_bwd(g["quats"], g["scales"], g["means"], g["c2w"], g["G"])
    assert np.allclose(gq, g["g_quats"], rtol=1e-11, atol=1e-12)


def test_normals_properties():
    """Unit length, facing the camera, tie rule of argmin, and the gradient is orthogonal to the radial direction of a
    unit quaternion only through the explicit chain (finite differences)."""
    rng = np.random.default_rng(3)
    N = 200
    q = rng.normal(size=(N, 4)); q /= np.linalg.norm(q, axis=-1, keepdims=True)
    s = np.exp(rng.normal(size=(N, 3)))
    s[0] = 0.5                                   # all equal -> axis 0
    m = rng.normal(size=(N, 3)) * 5
    A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    c2w = np.concatenate([A, rng.normal(size=(3, 1))], 1)
    n = no.normals_fwd(q, s, m, c2w)
    assert np.allclose(np.linalg.norm(n, axis=-1), 1.0, atol=1e-12)
    world = n @ A.T                              # back to world space
    d = c2w[:, 3][None] - m
    assert ((world * d).sum(-1) >= -1e-12).all()
    w, x, y, z = q[0]
    col0 = np.array([1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)])
    assert np.allclose(np.abs(world[0]), np.abs(col0))
    G = rng.normal(size=(N, 3))
    gq = no.normals_bwd(q, s, m, c2w, G)
    eps = 1e-6
    for c in range(4):
        dq = np.zeros_like(q); dq[:, c] = eps
        fd = ((no.normals_fwd(q + dq, s, m, c2w) - no.normals_fwd(q - dq, s, m, c2w)) * G).sum(-1) / (2 * eps)
        assert np.allclose(fd, gq[:, c], rtol=1e-5, atol=1e-6)
