"""oracle/ssim_oracle.py against the golden vectors the REFERENCE module produced (tests/golden/ssim_ref.npz,
tests/golden/make_ssim_golden.py imports /root/reference/mtgs/utils/ssim.py): this row's oracle is pinned."""
from pathlib import Path

import numpy as np
import pytest

Z = np.load(Path(__file__).parent / "golden" / "ssim_ref.npz")
CASES = sorted({k.split("_")[0] for k in Z.files})


@pytest.mark.parametrize("case", CASES)
def test_ssim_oracle_matches_reference_vectors(case):
    from oracle import ssim_oracle
    mask = Z[f"{case}_mask"]
    mask = None if mask.size == 0 else mask
    val, grad = ssim_oracle.masked_ssim(Z[f"{case}_gt"], Z[f"{case}_pred"], mask, with_grad=True)
    assert abs(val - float(Z[f"{case}_ssim_f64"])) <= 1e-12
    ref = Z[f"{case}_grad_f64"]
    assert np.abs(grad - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    # the reference in float32 agrees with its own float64 run to ~1e-6: the tolerance of the HIP kernel
    assert abs(float(Z[f"{case}_ssim_f32"]) - float(Z[f"{case}_ssim_f64"])) <= 2e-6


def test_window_is_the_reference_window():
    from oracle import ssim_oracle
    w = ssim_oracle.gauss_window()
    assert len(w) == 11 and abs(w.sum() - 1.0) < 1e-6 and np.allclose(w, w[::-1]) and w.argmax() == 5
