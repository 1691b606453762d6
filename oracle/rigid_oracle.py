"""TEST INFRASTRUCTURE ONLY (nothing under mtgs_amd/ imports this).

numpy float64 restatement of the rigid-node pose transform of MTGS
(/root/reference/mtgs/scene_model/gaussian_model/utils.py: quat_to_rotmat :14-41 -- no normalisation --, quat_mult :60-70;
composed as in rigid_node.py:205-216) with its analytic VJP.  PINNED: tests/test_oracle_rigid.py checks it against
tests/golden/rigid_ref.npz, which the reference functions themselves produced (tests/golden/make_rigid_golden.py)."""
import numpy as np


def quat_to_rotmat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def quat_mult(q1, q2):
    w1, x1, y1, z1 = q1
    w2, x2, y2, z2 = q2[..., 0], q2[..., 1], q2[..., 2], q2[..., 3]
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], axis=-1)


def forward(means, quats, q, t):
    qn = quats / np.linalg.norm(quats, axis=-1, keepdims=True)
    return means @ quat_to_rotmat(q).T + t, quat_mult(q, qn)


def backward(means, quats, q, t, Gm, Gq):
    """gradients of sum(global_means * Gm) + sum(global_quats * Gq) w.r.t. (means, quats, q, t)"""
    R = quat_to_rotmat(q)
    w, x, y, z = q
    g_means, g_t = Gm @ R, Gm.sum(0)
    vR = Gm.T @ means                                           # vR[i][j] = sum_n Gm[n,i] means[n,j]
    dR = np.zeros((4, 3, 3))                                    # d R / d (w, x, y, z)
    dR[0] = [[0, -2 * z, 2 * y], [2 * z, 0, -2 * x], [-2 * y, 2 * x, 0]]
    dR[1] = [[0, 2 * y, 2 * z], [2 * y, -4 * x, -2 * w], [2 * z, 2 * w, -4 * x]]
    dR[2] = [[-4 * y, 2 * x, 2 * w], [2 * x, 0, 2 * z], [-2 * w, 2 * z, -4 * y]]
    dR[3] = [[-4 * z, -2 * w, 2 * x], [2 * w, -4 * z, 2 * y], [2 * x, 2 * y, 0]]
    g_q = np.array([(dR[c] * vR).sum() for c in range(4)])
    nrm = np.linalg.norm(quats, axis=-1, keepdims=True)
    qn = quats / nrm
    vw, vx, vy, vz = Gq[:, 0], Gq[:, 1], Gq[:, 2], Gq[:, 3]
    w2, x2, y2, z2 = qn[:, 0], qn[:, 1], qn[:, 2], qn[:, 3]
    g_q += np.array([(vw * w2 + vx * x2 + vy * y2 + vz * z2).sum(), (-vw * x2 + vx * w2 - vy * z2 + vz * y2).sum(),
                     (-vw * y2 + vx * z2 + vy * w2 - vz * x2).sum(), (-vw * z2 - vx * y2 + vy * x2 + vz * w2).sum()])
    v_qn = np.stack([vw * w + vx * x + vy * y + vz * z, -vw * x + vx * w + vy * z - vz * y,
                     -vw * y - vx * z + vy * w + vz * x, -vw * z + vx * y - vy * x + vz * w], axis=-1)
    g_quats = (v_qn - (v_qn * qn).sum(-1, keepdims=True) * qn) / nrm
    return g_means, g_quats, g_q, g_t
