"""The numpy restatement of the deformation network (oracle/deform_oracle.py) against vectors produced by the reference's
own ConditionalDeformNetwork (tests/golden/deform_ref.npz <- tests/golden/make_deform_golden.py)."""
from pathlib import Path

import numpy as np

GOLD = Path(__file__).parent / "golden" / "deform_ref.npz"


def test_embedding_layout_and_network_match_the_reference_module():
    from oracle import deform_oracle as O
    g = np.load(GOLD)
    x = (g["means"] / g["height"] * np.float32(2)).astype(np.float64)
    assert np.abs(O.embed(x, 10) - g["x_emb"]).max() < 2e-4          # sin(512 x) of an fp32-rounded product vs fp64
    assert np.abs(O.embed(np.full((1, 1), g["t"], dtype=np.float64), 10) - g["t_emb"]).max() < 1e-4
    assert g["x_emb"].shape[1] == 63 and g["t_emb"].shape[1] == 21
    w = {k[2:]: g[k] for k in g.files if k.startswith("w.")}
    d_xyz, d_quat, d_scale = O.deform_network(g["means"], g["height"], g["t"], g["cond"], w)
    for got, ref in ((d_xyz, g["d_xyz"]), (d_quat, g["d_quat"]), (d_scale, g["d_scale"])):
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 5e-5 * max(1.0, np.abs(ref).max())
