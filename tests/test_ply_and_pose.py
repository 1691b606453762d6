"""Host logic of the interchange row and of the rigid-node helpers: the PLY layout round
trip, and object_pose / interpolate_quats / idft_weights (pure torch, mirrors of the reference's functions) against
vectors the REFERENCE's own utils produced (tests/golden/make_pose_fourier_golden.py).  CPU."""
from pathlib import Path

import numpy as np
import torch

GOLD = Path(__file__).resolve().parent / "golden"


def test_ply_round_trip_and_layout(tmp_path):
    from mtgs_amd.ply import read_ply, write_ply
    g = torch.Generator().manual_seed(1)
    N, K = 137, 16
    p = {"means": torch.randn(N, 3, generator=g), "scales": torch.randn(N, 3, generator=g), "quats": torch.randn(N, 4, generator=g),
         "opacities": torch.randn(N, 1, generator=g), "features_dc": torch.randn(N, 3, generator=g),
         "features_rest": torch.randn(N, K - 1, 3, generator=g)}
    p["means"][5, 1] = float("nan")                      # dropped, as the reference's exporter drops non-finite rows
    n = write_ply(tmp_path / "a.ply", p)
    assert n == N - 1
    raw = (tmp_path / "a.ply").read_bytes()
    head = raw[: raw.index(b"end_header\n") + 11].decode()
    props = [ln.split()[-1] for ln in head.splitlines() if ln.startswith("property")]
    assert props[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] and props[9] == "f_rest_0"
    assert props[-8:] == ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"] and len(props) == 17 + 3 * (K - 1)
    assert "format binary_little_endian 1.0" in head and f"element vertex {N - 1}" in head
    assert len(raw) == len(head) + (N - 1) * len(props) * 4
    q = read_ply(tmp_path / "a.ply")
    keep = torch.ones(N, dtype=torch.bool); keep[5] = False
    for k in p:
        assert torch.equal(q[k], p[k][keep]), k
    # channel-major f_rest: the first K-1 columns are the RED coefficients
    row0 = np.frombuffer(raw[len(head):len(head) + len(props) * 4], dtype="<f4")
    assert np.array_equal(row0[9:9 + K - 1], p["features_rest"][0, :, 0].numpy())
    # degree-0 model (no f_rest) and an empty model
    write_ply(tmp_path / "b.ply", {k: v for k, v in p.items() if k != "features_rest"})
    assert read_ply(tmp_path / "b.ply")["features_rest"].shape == (N - 1, 0, 3)
    write_ply(tmp_path / "c.ply", {k: v[:0] for k, v in p.items()})
    assert read_ply(tmp_path / "c.ply")["means"].shape == (0, 3)


def test_interpolate_quats_and_pose_match_the_reference_vectors():
    from mtgs_amd.nodes import interpolate_quats, object_pose
    z = np.load(GOLD / "pose_fourier_ref.npz")
    q1, q2, t = (torch.from_numpy(z[k]) for k in ("slerp_q1", "slerp_q2", "slerp_t"))
    got = torch.stack([interpolate_quats(q1[i], q2[i], t[i]).squeeze(0) for i in range(q1.shape[0])])
    assert np.abs(got.numpy() - z["slerp_out"].reshape(-1, 4)).max() < 1e-12
    iq, it, ts = (torch.from_numpy(z[k]) for k in ("pose_iq", "pose_it", "pose_ts"))
    Gq, Gt = torch.from_numpy(z["pose_Gq"]), torch.from_numpy(z["pose_Gt"])
    for k, stamp in enumerate(torch.from_numpy(z["pose_stamps"])):
        A, B = iq.clone().requires_grad_(True), it.clone().requires_grad_(True)
        q, tr = object_pose(A, B, timestamp=stamp, frame_timestamps=ts)
        assert np.abs(q.detach().numpy() - z[f"pose{k}_q"]).max() < 1e-12 and np.abs(tr.detach().numpy() - z[f"pose{k}_t"]).max() < 1e-12
        ((q * Gq).sum() + (tr * Gt).sum()).backward()
        assert np.abs(A.grad.numpy() - z[f"pose{k}_g_iq"]).max() < 1e-10 and np.abs(B.grad.numpy() - z[f"pose{k}_g_it"]).max() < 1e-12
    # frame index given: the normalised row; an object that is not in the frame
    q, tr = object_pose(iq, it, frame_idx=3)
    assert torch.allclose(q, iq[3] / iq[3].norm()) and torch.equal(tr, it[3])
    mask = torch.ones(iq.shape[0], dtype=torch.bool); mask[3] = False
    assert object_pose(iq, it, frame_idx=3, in_frame_mask=mask) == (None, None)
    assert object_pose(iq, it, frame_idx=99) == (None, None)
    assert object_pose(iq, it, timestamp=0.5 * (ts[2] + ts[3]), frame_timestamps=ts, in_frame_mask=mask) == (None, None)


def test_idft_weights_match_the_reference_vectors():
    from mtgs_amd.nodes import idft_weights
    z = np.load(GOLD / "pose_fourier_ref.npz")
    for name in ("t5", "t8", "s6", "s1"):
        w = idft_weights(float(z[f"four_{name}_x"]), int(z[f"four_{name}_dim"]), bool(z[f"four_{name}_norm"]))
        assert np.abs(w.numpy() - z[f"four_{name}_w"]).max() < 2e-6, name
