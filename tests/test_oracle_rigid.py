"""oracle/rigid_oracle.py against the vectors the reference's own quat_to_rotmat / quat_mult produced
(tests/golden/rigid_ref.npz): the rigid-node row's oracle is pinned."""
from pathlib import Path

import numpy as np
import pytest

Z = np.load(Path(__file__).parent / "golden" / "rigid_ref.npz")


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_rigid_oracle_matches_reference_vectors(case):
    from oracle import rigid_oracle as ro
    f = lambda k: Z[f"{case}_{k}"]
    gm, gq = ro.forward(f("means"), f("quats"), f("q"), f("t"))
    assert np.abs(gm - f("global_means")).max() <= 1e-12 and np.abs(gq - f("global_quats")).max() <= 1e-12
    for got, name in zip(ro.backward(f("means"), f("quats"), f("q"), f("t"), f("Gm"), f("Gq")), ("g_means", "g_quats", "g_q", "g_t")):
        assert np.abs(got - f(name)).max() <= 1e-10 * max(1.0, np.abs(f(name)).max()), name
