"""mtgs_amd.checkpoint: key layout of MTGS checkpoints (custom_trainer.py:148-157, mtgs_scene_graph.py:1185-1203)."""
import pytest
import torch


def _state(T=2):
    g = torch.Generator().manual_seed(0)
    sd = {"_model.camera_optimizer.pose_adjustment": torch.zeros(4, 6)}
    def node(name, n, multicolor=False, dynamic=False):
        base = f"_model.gaussian_models.{name}."
        sd[base + "gauss_params.means"] = torch.randn(n, 3, generator=g) * 3 + torch.tensor([0.0, 0.0, 8.0])
        sd[base + "gauss_params.scales"] = torch.randn(n, 3, generator=g) * 0.3 - 2.0
        sd[base + "gauss_params.quats"] = torch.randn(n, 4, generator=g)
        sd[base + "gauss_params.opacities"] = torch.randn(n, 1, generator=g)
        sd[base + "gauss_params.features_dc"] = torch.randn(n, 3, generator=g)
        if multicolor:
            sd[base + "gauss_params.features_rest"] = torch.randn(n, T, 15, 3, generator=g) * 0.1
            sd[base + "gauss_params.features_adapters"] = torch.randn(n, T, 3, generator=g) * 0.1
        else:
            sd[base + "gauss_params.features_rest"] = torch.randn(n, 15, 3, generator=g) * 0.1
        if dynamic:
            sd[base + "instance_quats"] = torch.randn(5, 4, generator=g)
            sd[base + "instance_trans"] = torch.randn(5, 3, generator=g)
    node("background", 300, multicolor=True)
    node("road", 200)
    node("object_vehicle_12", 50, dynamic=True)
    node("object_deform_3", 20)
    sd["_model.gaussian_models.object_deform_3.deform_network.fc.weight"] = torch.zeros(4, 4)
    return sd


def test_load_nodes_from_checkpoint_file(tmp_path):
    from mtgs_amd import checkpoint as ck
    path = tmp_path / "step-000030000.ckpt"
    torch.save({"step": 30000, "pipeline": _state()}, path)
    nodes = ck.load_gaussian_nodes(str(path))
    assert list(nodes) == ["background", "road", "object_vehicle_12", "object_deform_3"]
    assert set(nodes["road"]) == {"means", "scales", "quats", "opacities", "features_dc", "features_rest"}
    assert nodes["background"]["features_rest"].shape == (300, 2, 15, 3)
    assert [ck.node_kind(nodes[n]) for n in nodes] == ["multicolor", "vanilla", "rigid", "dynamic"]
    assert "instance_trans" in nodes["object_vehicle_12"]
    with pytest.raises(ValueError):
        ck.load_gaussian_nodes({"pipeline": {"_model.foo": torch.zeros(1)}})


@pytest.mark.gpu
def test_collect_and_render_checkpoint(hip_lib):
    from mtgs_amd import checkpoint as ck, rasterization
    from mtgs_amd.synthetic import make_camera
    nodes = ck.load_gaussian_nodes({"pipeline": _state()})
    c2w = torch.eye(4)[None, :3]
    with pytest.raises(NotImplementedError, match="object_deform_3"):
        ck.collect_gaussians(nodes, c2w, 3, frame_idx=0)
    with pytest.raises(ValueError, match="frame_idx"):
        ck.collect_gaussians(nodes, c2w, 3, node_names=["object_vehicle_12"])
    gs = ck.collect_gaussians(nodes, c2w, 3, traversal_index=1, node_names=["background", "road", "object_vehicle_12"], frame_idx=2)
    assert gs["means"].shape == (550, 3) and gs["rgbs"].shape == (550, 3) and int(gs["model_id"].max()) == 2
    # the rigid node is posed with frame 2: R(q) m + t with the normalised per-frame quaternion (rigid_node.py:142, :205-209)
    v = nodes["object_vehicle_12"]
    q = (v["instance_quats"][2] / v["instance_quats"][2].norm()).double()
    w, x, y, z = q.tolist()
    R = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], dtype=torch.float64)
    expect = v["means"].double() @ R.T + v["instance_trans"][2].double()
    assert torch.allclose(gs["means"][500:].cpu().double(), expect, atol=1e-5)
    # the multi-colour node used traversal 1: features_dc + adapters[:, 1], features_rest[:, 1]
    from mtgs_amd.nodes import node_gaussians
    p = {k: v.cuda() for k, v in nodes["background"].items()}
    ref = node_gaussians(p["means"], p["scales"], p["quats"], p["opacities"],
                         (p["features_dc"] + p["features_adapters"][:, 1]).contiguous(),
                         p["features_rest"][:, 1].contiguous(), c2w.cuda(), 3, 3)
    assert torch.equal(gs["rgbs"][:300], ref["rgbs"])
    vm, K = make_camera(160, 120)
    render, alpha, info = rasterization(gs["means"], gs["quats"], gs["scales"], gs["opacities"], gs["rgbs"], vm.cuda(),
                                        K.cuda(), 160, 120, packed=False, render_mode="RGB+ED")
    assert render.shape == (1, 120, 160, 4) and torch.isfinite(render).all() and float(alpha.max()) > 0.1


@pytest.mark.gpu
def test_rigid_node_between_frames_and_fourier_colour(hip_lib):
    """A rigid node posed BETWEEN two frames (timestamp interpolation, rigid_node.py:145-166) and coloured by a Fourier
    series of the normalised timestamp (rigid_node.py:217-229): collect_gaussians against the reference composition in
    torch (interpolate_quats / IDFT restated in mtgs_amd.nodes and pinned by reference vectors in test_ply_and_pose.py)."""
    import numpy as np
    from pathlib import Path
    from mtgs_amd import checkpoint as ck
    from mtgs_amd.nodes import fourier_features_dc, interpolate_quats
    g = torch.Generator().manual_seed(4)
    n, F, frames = 70, 5, 6
    node = {"means": torch.randn(n, 3, generator=g), "scales": torch.randn(n, 3, generator=g) * 0.3 - 2, "quats": torch.randn(n, 4, generator=g),
            "opacities": torch.randn(n, 1, generator=g), "features_dc": torch.randn(n, F, 3, generator=g),
            "features_rest": torch.randn(n, 15, 3, generator=g) * 0.1, "instance_quats": torch.randn(frames, 4, generator=g),
            "instance_trans": torch.randn(frames, 3, generator=g) * 4}
    ts = torch.arange(frames, dtype=torch.float32) * 0.1
    c2w = torch.eye(4)[None, :3]
    gs = ck.collect_gaussians({"car": node}, c2w, 3, timestamp=0.23, frame_timestamps=ts,
                              fourier={"x": 0.23 / 0.5, "scale": 1.0, "space": "temporal"})
    t = (0.23 - 0.2) / 0.1
    q = interpolate_quats(node["instance_quats"][2].double(), node["instance_quats"][3].double(), t).squeeze(0)
    tr = torch.lerp(node["instance_trans"][2].double(), node["instance_trans"][3].double(), torch.tensor(t, dtype=torch.float64))
    w, x, y, z = q.tolist()
    R = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], dtype=torch.float64)
    assert torch.allclose(gs["means"].cpu().double(), node["means"].double() @ R.T + tr, atol=2e-5)
    # the Fourier kernel against the reference-generated vectors (value, gradient of the parameter and of the weights)
    zf = np.load(Path(__file__).resolve().parent / "golden" / "pose_fourier_ref.npz")
    for name in ("t5", "t8", "s6", "s1"):
        fdc = torch.from_numpy(zf[f"four_{name}_fdc"]).float().cuda().requires_grad_(True)
        from mtgs_amd.nodes import _FourierDC
        wv = torch.from_numpy(zf[f"four_{name}_w"]).float().cuda().requires_grad_(True)
        dc = _FourierDC.apply(fdc, wv)
        assert np.abs(dc.detach().cpu().numpy() - zf[f"four_{name}_dc"]).max() < 5e-6 * max(1.0, np.abs(zf[f"four_{name}_dc"]).max())
        (dc * torch.from_numpy(zf[f"four_{name}_G"]).float().cuda()).sum().backward()
        assert np.abs(fdc.grad.cpu().numpy() - zf[f"four_{name}_g_fdc"]).max() < 5e-6 * np.abs(zf[f"four_{name}_g_fdc"]).max()
        assert np.abs(wv.grad.cpu().numpy() - zf[f"four_{name}_g_w"]).max() < 2e-5 * max(1.0, np.abs(zf[f"four_{name}_g_w"]).max())
    x = float(zf["four_t5_x"])
    dc = fourier_features_dc(torch.from_numpy(zf["four_t5_fdc"]).float().cuda(), x, 1.0, "temporal")
    assert np.abs(dc.cpu().numpy() - zf["four_t5_dc"]).max() < 1e-5 * max(1.0, np.abs(zf["four_t5_dc"]).max())
