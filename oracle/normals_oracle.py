"""CPU oracle for the camera-space normals of the Gaussians (TEST INFRASTRUCTURE ONLY -- imported by tests/ alone).

Restates, in numpy float64 with an analytic backward, what MTGSSceneModel._get_gaussian_camera_space_normals computes
(/root/reference/mtgs/scene_model/mtgs_scene_graph.py:526-545; `quat_to_rotmat` = utils.py:14-41, wxyz, the quaternion is
NOT normalised there):

    k        = argmin(scales)                       (first of equal minima, torch.argmin)
    col      = column k of quat_to_rotmat(quats)
    n0       = col / max(|col|, 1e-12)              (F.normalize)
    s        = -1 if dot(n0, normalize(cam_pos - means)) < 0 else +1
    normals  = (s n0) @ camera_to_worlds[:3, :3]

PINNED: tests/golden/normals_ref.npz was produced by the reference's own quat_to_rotmat composed as in the method above
(tests/golden/make_normals_golden.py); tests/test_oracle_normals.py checks this file against it.
"""
from __future__ import annotations

import numpy as np


def _columns(q):
    """The three columns of quat_to_rotmat(q) and their derivatives with respect to (w, x, y, z): cols[k] is [N,3],
    dcols[k][c] is d cols[k] / d q_c, [N,3]."""
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    o = np.zeros_like(w)
    st = lambda *a: np.stack(a, -1)
    cols = [st(1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)),
            st(2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)),
            st(2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y))]
    dcols = [[st(o, 2 * z, -2 * y), st(o, 2 * y, 2 * z), st(-4 * y, 2 * x, -2 * w), st(-4 * z, 2 * w, 2 * x)],
             [st(-2 * z, o, 2 * x), st(2 * y, -4 * x, 2 * w), st(2 * x, o, 2 * z), st(-2 * w, -4 * z, 2 * y)],
             [st(2 * y, -2 * x, o), st(2 * z, -2 * w, -4 * x), st(2 * w, 2 * z, -4 * y), st(2 * x, 2 * y, o)]]
    return cols, dcols


def _forward(quats, scales, means, c2w):
    q = np.asarray(quats, dtype=np.float64)
    k = np.argmin(np.asarray(scales, dtype=np.float64), axis=-1)          # numpy: first occurrence, as torch.argmin
    cols, dcols = _columns(q)
    rows = np.arange(q.shape[0])
    col = np.stack(cols, 0)[k, rows]                                      # [N,3]
    nrm = np.maximum(np.linalg.norm(col, axis=-1, keepdims=True), 1e-12)
    n0 = col / nrm
    c2w = np.asarray(c2w, dtype=np.float64).reshape(3, 4)
    d = c2w[:, 3][None] - np.asarray(means, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        d = d / np.linalg.norm(d, axis=-1, keepdims=True)
        sign = np.where((n0 * d).sum(-1) < 0, -1.0, 1.0)[:, None]
    return q, k, dcols, rows, nrm, n0, sign, c2w[:, :3]


def normals_fwd(quats, scales, means, c2w):
    """normals[N,3] in camera space."""
    _, _, _, _, _, n0, sign, R = _forward(quats, scales, means, c2w)
    return (sign * n0) @ R


def normals_bwd(quats, scales, means, c2w, v_normals):
    """Gradient of sum(normals * v_normals) with respect to quats, [N,4] (scales / means / camera get none)."""
    q, k, dcols, rows, nrm, n0, sign, R = _forward(quats, scales, means, c2w)
    v = (np.asarray(v_normals, dtype=np.float64) @ R.T) * sign            # back through the camera rotation and the flip
    v_col = (v - n0 * (n0 * v).sum(-1, keepdims=True)) / nrm              # F.normalize
    g = np.zeros_like(q)
    for c in range(4):
        dc = np.stack([dcols[j][c] for j in range(3)], 0)[k, rows]        # d col / d q_c for the selected column
        g[:, c] = (v_col * dc).sum(-1)
    return g
