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


# ---- parity report: every comparison against the oracle appends its MEASURED errors here; tests/conftest.py writes the
# list to gpurun_out/parity_report.json at the end of the session (copied to profiles/ per round)
REPORT = []
# Threshold-critical pixels (some Gaussian of the pixel's list within 1e-4 relative of the alpha = 1/255 or T = 1e-4
# decision -- with several hundred candidates per pixel that is not rare): measured 2e-4 .. 4.5e-3 of the pixels over the
# scenes of the suite (4.5e-3 at 2M Gaussians / 1920x1080; profiles/r02_parity_report.json); the bound is about twice
# the largest measured rate.  How many of them actually differ from the oracle is reported too (`flipped`).
MAX_CRITICAL_RATE = 1e-2


def assert_image_close(got, ref, critical, tol=1e-4, flip_bound=1.0 / 255.0, name="render", scale=None, case=""):
    """max-abs <= tol (x max(1, |ref|max)) on every well-conditioned pixel.  Pixels the oracle flags as
    threshold-critical (see orc_blend_fwd) may differ by one flipped decision: <= flip_bound * scale.  Critical pixels
    must be rare: at most max(MAX_CRITICAL_RATE of the image, 2 pixels)."""
    err = np.abs(got - ref)
    if scale is None:
        scale = max(1.0, float(np.abs(ref).max()))
    crit = np.broadcast_to(critical[..., None], err.shape)
    n_crit, n_pix = int(critical.sum()), int(critical.size)
    ok = err[~crit]
    bad = err[crit]
    flipped = int((err.max(axis=-1) > tol * scale)[critical].sum()) if n_crit else 0
    REPORT.append({"kind": "image", "case": case, "name": name, "pixels": n_pix, "critical_pixels": n_crit,
                   "critical_pixels_over_tol": flipped, "max_abs_err": float(ok.max()) if ok.size else 0.0,
                   "scale": float(scale), "max_abs_err_critical": float(bad.max()) if bad.size else 0.0})
    assert n_crit <= max(MAX_CRITICAL_RATE * n_pix, 2), f"{name}: too many threshold-critical pixels ({n_crit} of {n_pix})"
    assert ok.size == 0 or ok.max() <= tol * scale, f"{name}: max err {ok.max():.3e} > {tol * scale:.1e}"
    assert bad.size == 0 or bad.max() <= (flip_bound * 1.5 + tol) * scale, f"{name}: critical-pixel err {bad.max():.3e}"


def grad_stats(got, ref, floor=1e-6):
    """Errors of a gradient tensor against the oracle's (fp64-summed) one.  Rows = Gaussians (the leading axes of a
    [.., N, k] tensor flattened; a 1-D tensor has one element per row).
      rel_to_max      max |got - ref| / max |ref|                     (the bound the round-1 tests used)
      row_rel_p999 / row_rel_max: per row r with |ref_r|_max > floor * max |ref|:  |got_r - ref_r|_max / |ref_r|_max
    A row whose gradient is 1e-4 of the largest one can be 100 % wrong under the first metric and pass; not under the
    second."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    scale = float(np.abs(ref).max())
    if ref.ndim == 1:
        g2, r2 = got[:, None], ref[:, None]
    elif ref.ndim == 2 and ref.shape[0] <= 4:      # a single small matrix (v_viewmat): one row
        g2, r2 = got.reshape(1, -1), ref.reshape(1, -1)
    else:
        if ref.ndim == 2:
            g2, r2 = got, ref
        elif ref.ndim == 3 and ref.shape[0] == 1:    # [1, N, k]
            g2, r2 = got[0], ref[0]
        else:                                        # [N, K, 3] -> [N, K*3]
            g2, r2 = got.reshape(got.shape[0], -1), ref.reshape(ref.shape[0], -1)
    row_ref = np.abs(r2).max(axis=1)
    row_err = np.abs(g2 - r2).max(axis=1)
    sel = row_ref > floor * scale
    rel = row_err[sel] / row_ref[sel]
    return {"rel_to_max": float(np.abs(got - ref).max() / scale) if scale > 0 else 0.0, "scale": scale,
            "rows_checked": int(sel.sum()), "row_rel_p999": float(np.quantile(rel, 0.999)) if rel.size else 0.0,
            "row_rel_p99": float(np.quantile(rel, 0.99)) if rel.size else 0.0,
            "row_rel_max": float(rel.max()) if rel.size else 0.0,
            "rows_over_1e-3": int((rel > 1e-3).sum())}


def assert_grad_close(name, got, ref, case="", rel_to_max=1e-3, row_rel_p999=1e-3):
    """Global bound (max error <= rel_to_max of the tensor's largest gradient; measured <= 6.7e-4 over the suite, 1e-6 ..
    2e-4 where no threshold decision flips) AND per-row bound: 99.9 % of the rows whose gradient exceeds 1e-6 of the
    largest are within row_rel_p999 relative (99 % when fewer than 5000 rows are checked: one flipped pixel of a small
    scene is already 0.4 % of its rows).  The device sums contributions with fp32 atomics in arbitrary order, the oracle
    in fp64: rows whose contributions cancel (random cotangents) and the few rows behind a flipped alpha / transmittance
    decision carry the difference; the measured values of every comparison go to the parity report."""
    if hasattr(got, "detach"):
        got = got.detach().cpu().numpy()
    st = grad_stats(got, ref)
    REPORT.append(dict({"kind": "gradient", "case": case, "name": name}, **st))
    assert st["rel_to_max"] <= rel_to_max, f"{name}: max err {st['rel_to_max']:.3e} of the largest gradient (> {rel_to_max:.0e})"
    key = "row_rel_p999" if st["rows_checked"] >= 5000 else "row_rel_p99"
    assert st[key] <= row_rel_p999, (f"{name}: {key} of the per-row relative error {st[key]:.3e} > {row_rel_p999:.0e} "
                                     f"(max {st['row_rel_max']:.3e}, {st['rows_over_1e-3']} of {st['rows_checked']} rows over 1e-3)")
    return st
