"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md section 3): numpy restatement of the deformation network of
MTGS's deformable nodes -- ConditionalDeformNetwork.forward (/root/reference/mtgs/scene_model/gaussian_model/utils.py:
315-333) over the frequency embedding of get_embedder (utils.py:235-283) as DeformableSubModel.get_deformation feeds
it (deformable_node.py:173-203).  PINNED by tests/golden/deform_ref.npz, which the reference's own module produced
(tests/golden/make_deform_golden.py)."""
import numpy as np


def embed(v: np.ndarray, n_freqs: int) -> np.ndarray:
    """[v, sin(v f_0), cos(v f_0), ... ] with f_i = 2^i (include_input, log sampling; utils.py:262-283)."""
    cols = [v]
    for i in range(n_freqs):
        f = 2.0 ** i
        cols += [np.sin(v * f), np.cos(v * f)]
    return np.concatenate(cols, axis=-1)


def deform_network(means, height, t, cond, weights, x_multires=10, t_multires=10, dtype=np.float64):
    """(delta_xyz, delta_quat, delta_scale); `weights`: the module's state dict as numpy arrays."""
    means = np.asarray(means, dtype=np.float32)
    x = (means / np.float32(height) * np.float32(2)).astype(dtype)          # deformable_node.py:181 (fp32 there)
    N = x.shape[0]
    t_emb = embed(np.full((N, 1), np.float32(t), dtype=dtype), t_multires)
    emb = np.concatenate([embed(x, x_multires), t_emb, np.repeat(np.asarray(cond, dtype=dtype).reshape(1, -1), N, 0)], -1)
    h, i = emb, 0
    while f"linear.{i}.weight" in weights:
        w, b = weights[f"linear.{i}.weight"].astype(dtype), weights[f"linear.{i}.bias"].astype(dtype)
        if i > 0 and w.shape[1] == h.shape[1] + emb.shape[1]:
            h = np.concatenate([emb, h], -1)                                    # utils.py:323-324
        h = np.maximum(h @ w.T + b, 0.0)
        i += 1
    head = lambda k: h @ weights[f"{k}.weight"].astype(dtype).T + weights[f"{k}.bias"].astype(dtype)
    return head("gaussian_warp"), head("gaussian_rotation"), head("gaussian_scaling")
